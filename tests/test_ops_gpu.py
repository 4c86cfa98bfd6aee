"""Per-kernel parity: every libvfx_hip entry point (called through the C ABI via ctypes)
against the fp32 torch-CPU statement of the same operator / the oracle.  Needs an MI355X."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from voicefixer_amd import ops, packing, _lib  # noqa: E402
from oracle import oracle  # noqa: E402  (checker only)
from conftest import GOLDEN  # noqa: E402

DEV = "cuda"


def _rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g) * scale


def _padded(t, lpad):
    """Device copy of (B,C,L) into a (B,C,lpad) buffer pre-filled with NaN (catches over-reads)."""
    B, Cn, L = t.shape
    buf = torch.full((B, Cn, lpad), float("nan"), device=DEV)
    buf[:, :, :L] = t.to(DEV)
    return buf


def _close(got, want, tol):
    got = got.cpu()
    assert got.shape == want.shape
    assert torch.isfinite(got).all()
    err = (got - want).abs().max().item()
    scale = want.abs().max().item() + 1e-12
    assert err <= tol * max(1.0, scale), "max err %g (scale %g)" % (err, scale)


def _ref_act(x, pre, slope, scale=None, shift=None):
    if pre == _lib.PRE_LRELU:
        return F.leaky_relu(x, slope)
    if pre == _lib.PRE_AFFINE_LRELU:
        shp = [1, -1] + [1] * (x.dim() - 2)
        return F.leaky_relu(x * scale.reshape(shp) + shift.reshape(shp), slope)
    return x


def _ref_post(y, post, slope):
    if post == _lib.POST_LRELU:
        return F.leaky_relu(y, slope)
    if post == _lib.POST_ELU:
        return F.elu(y)
    if post == _lib.POST_TANH:
        return torch.tanh(y)
    if post == _lib.POST_SIGMOID:
        return torch.sigmoid(y)
    if post == _lib.POST_LRELU_SNAKE:
        u = F.leaky_relu(y, slope)
        return u + torch.sin(u)
    return y


CONV1D_CASES = [
    # B, Cin, Cout, L, k, dil, pad, pre, post, res
    (2, 64, 64, 1000, 3, 1, 0, _lib.PRE_LRELU, _lib.POST_LRELU, False),
    (1, 64, 64, 3001, 3, 27, 0, _lib.PRE_LRELU, _lib.POST_NONE, True),
    (2, 128, 128, 700, 3, 243, 0, _lib.PRE_LRELU, _lib.POST_LRELU, False),
    (1, 256, 256, 1500, 3, 729, 0, _lib.PRE_LRELU, _lib.POST_LRELU, True),
    (1, 512, 512, 742, 3, 2187, 0, _lib.PRE_LRELU, _lib.POST_LRELU, False),   # dilation > L
    (1, 128, 512, 106, 3, 1, 0, _lib.PRE_NONE, _lib.POST_ELU, False),          # condnet.0
    (1, 512, 1024, 106, 7, 1, 1, _lib.PRE_NONE, _lib.POST_LRELU_SNAKE, False),  # pre conv, reflect
    (3, 64, 32, 333, 3, 9, 0, _lib.PRE_NONE, _lib.POST_SIGMOID, True),
    (2, 96, 384, 50, 1, 1, 0, _lib.PRE_NONE, _lib.POST_NONE, False),
]


@pytest.mark.parametrize("case", CONV1D_CASES)
def test_conv1d(case):
    B, Cin, Cout, L, k, dil, pad, pre, post, use_res = case
    x = _rand((B, Cin, L), 1)
    w = _rand((Cout, Cin, k), 2, (Cin * k) ** -0.5)
    bias = _rand((Cout,), 3, 0.1)
    res = _rand((B, Cout, L), 4) if use_res else None
    xin = _ref_act(x, pre, 0.01)
    p = (k - 1) // 2 * dil
    if pad == 1:
        ref = F.conv1d(F.pad(xin, (p, p), mode="reflect"), w, bias, dilation=dil)
    else:
        ref = F.conv1d(xin, w, bias, dilation=dil, padding=p)
    if use_res:
        ref = ref + res
    ref = _ref_post(ref, post, 0.2)
    lp = (L + 67) // 4 * 4
    xd = _padded(x, lp)
    yd = torch.full((B, Cout, lp), float("nan"), device=DEV)
    rd = _padded(res, lp) if use_res else None
    act = ops.Act(pre=pre, pre_slope=0.01, post=post, post_slope=0.2)
    before = _lib.lib().vfx_launch_count()
    ops.conv1d(xd, packing.pack_conv1d(w).to(DEV), bias.to(DEV), yd, L, k, dil, pad, act, rd)
    torch.cuda.synchronize()
    assert _lib.lib().vfx_launch_count() in (before + 1, before + 2)  # interior + boundary grids
    _close(yd[:, :, :L], ref, 2e-5)
    assert torch.isnan(yd[:, :, L:]).all()  # nothing written past L


def test_conv1d_guarded_input_all_interior():
    """With a guard band every tile (also the first/last) runs the interior kernel; the guard
    holds NaN to prove that whatever is read there is masked."""
    B, Cin, Cout, L, k, dil = 2, 64, 64, 1000, 3, 81
    x = _rand((B, Cin, L), 41)
    w = _rand((Cout, Cin, k), 42, (Cin * k) ** -0.5)
    bias = _rand((Cout,), 43, 0.1)
    ref = F.leaky_relu(F.conv1d(F.leaky_relu(x, 0.01), w, bias, dilation=dil, padding=dil), 0.01)
    xd = ops.guarded(B, Cin, L, dil + 264, DEV)
    xd._vfx_base.fill_(float("nan"))
    xd[:, :, :L] = x.to(DEV)
    yd = torch.full((B, Cout, 1000), float("nan"), device=DEV)
    act = ops.Act(pre=_lib.PRE_LRELU, pre_slope=0.01, post=_lib.POST_LRELU, post_slope=0.01)
    before = _lib.lib().vfx_launch_count()
    ops.conv1d(xd, packing.pack_conv1d(w).to(DEV), bias.to(DEV), yd, L, k, dil, 0, act)
    torch.cuda.synchronize()
    assert _lib.lib().vfx_launch_count() == before + 1  # one grid: no boundary launch
    _close(yd, ref, 2e-5)


X3_CASES = [
    # B, Cin, Cout, L, dil, res  (k = 3, guarded input: the bf16x3 kernel only runs all-interior launches)
    (2, 64, 64, 1000, 1, True),      # 64x128 tile, halo
    (1, 64, 64, 9000, 27, False),    # 64x256 tile, halo
    (1, 64, 64, 3001, 243, True),    # 64x128 tile, one segment per tap
    (2, 128, 128, 700, 9, False),    # 128x128 tile
    (1, 256, 256, 1500, 729, True),
    (1, 512, 512, 742, 2187, False),  # dilation > L
    (1, 128, 512, 106, 1, False),
]


@pytest.mark.parametrize("case", X3_CASES)
def test_conv1d_bf16x3(case):
    """Opt-in VFX_MATH_BF16X3: split-bf16 products (xh*wh + xh*wl + xl*wh), fp32 accumulation.  Per-product
    relative error <= 2^-16, so the result sits within 1e-4 of the fp32 operator (the fp32 kernel: 2e-5)."""
    B, Cin, Cout, L, dil, use_res = case
    x = _rand((B, Cin, L), 61)
    w = _rand((Cout, Cin, 3), 62, (Cin * 3) ** -0.5)
    bias = _rand((Cout,), 63, 0.1)
    res = _rand((B, Cout, L), 64) if use_res else None
    ref = F.conv1d(F.leaky_relu(x, 0.01), w, bias, dilation=dil, padding=dil)
    if use_res:
        ref = ref + res
    ref = F.leaky_relu(ref, 0.2)
    xd = ops.guarded(B, Cin, L, dil + 264, DEV)
    xd._vfx_base.fill_(float("nan"))
    xd[:, :, :L] = x.to(DEV)
    lp = (L + 3) // 4 * 4
    yd = torch.full((B, Cout, lp), float("nan"), device=DEV)
    rd = _padded(res, lp) if use_res else None
    wp = packing.pack_conv1d(w)
    w3 = packing.pack_x3(wp).to(DEV)
    act = ops.Act(pre=_lib.PRE_LRELU, pre_slope=0.01, post=_lib.POST_LRELU, post_slope=0.2)
    ops.conv1d(xd, wp.to(DEV), bias.to(DEV), yd, L, 3, dil, 0, act, rd, w3=w3)
    torch.cuda.synchronize()
    assert _lib.lib().vfx_last_conv_tile() % 100 == 16, "bf16x3 kernel did not run (fp32 fallback)"
    _close(yd[:, :, :L], ref, 1e-4)
    err = (yd[:, :, :L].cpu() - ref).pow(2).mean().sqrt().item() / ref.pow(2).mean().sqrt().item()
    assert err < 2e-5, err
    assert torch.isnan(yd[:, :, L:]).all()


@pytest.mark.parametrize("cfg", [(1, 1024, 512, 106, 7), (2, 256, 128, 531, 3), (1, 128, 64, 1000, 3)])
def test_convtr1d_bf16x3(cfg):
    B, Cin, Cout, Lin, s = cfg
    x = _rand((B, Cin, Lin), 65)
    w = _rand((Cin, Cout, 2 * s), 66, (2 * Cin) ** -0.5)
    bias = _rand((Cout,), 67, 0.1)
    ref = F.conv_transpose1d(x, w, bias, stride=s, padding=s // 2 + s % 2, output_padding=s % 2)
    xd = ops.guarded(B, Cin, Lin, 264, DEV)
    xd._vfx_base.fill_(float("nan"))
    xd[:, :, :Lin] = x.to(DEV)
    Lo = s * Lin
    yd = torch.full((B, Cout, (Lo + 7) // 4 * 4), float("nan"), device=DEV)
    wp = packing.pack_convtr1d(w)
    ops.convtr1d(xd, wp.to(DEV), bias.to(DEV), yd, Lin, s, w3=packing.pack_x3(wp).to(DEV))
    torch.cuda.synchronize()
    tiny = B * ((Lin + 127) // 128) * s * max(Cout // 128, 1) <= 96 and Cin * 2 >= 1024
    assert tiny or _lib.lib().vfx_last_conv_tile() % 100 == 16, "bf16x3 kernel did not run (fp32 fallback)"
    _close(yd[:, :, :Lo], ref, 1e-4)
    assert torch.isnan(yd[:, :, Lo:]).all()


def test_bf16x3_falls_back_to_fp32_outside_its_coverage():
    """Unguarded input (boundary tiles) -> the library runs the fp32 kernel; the result is the fp32 one."""
    B, Cin, Cout, L = 1, 64, 64, 500
    x = _rand((B, Cin, L), 68)
    w = _rand((Cout, Cin, 3), 69, (Cin * 3) ** -0.5)
    ref = F.conv1d(x, w, None, padding=1)
    xd = _padded(x, 504)
    yd = torch.full((B, Cout, 504), float("nan"), device=DEV)
    wp = packing.pack_conv1d(w)
    ops.conv1d(xd, wp.to(DEV), None, yd, L, 3, 1, 0, None, None, w3=packing.pack_x3(wp).to(DEV))
    torch.cuda.synchronize()
    assert _lib.lib().vfx_last_conv_tile() % 100 != 16
    _close(yd[:, :, :L], ref, 2e-5)


# ---- second-generation kernel: weights as L2-resident A-operand vectors (vfx_act.w_direct), fused ResStack layer ----
def _guarded_nan(t, guard):
    """Device copy of (B,C,L) into a guarded view whose guard band (and tail pad) holds NaN."""
    B, Cn, L = t.shape
    v = ops.guarded(B, Cn, L, guard, DEV)
    v._vfx_base.fill_(float("nan"))
    v[:, :, :L] = t.to(DEV)
    return v


CONVW_CASES = [
    # B, Cin, Cout, L, dil, pre, post, res   (k = 3; sized so that the launch has >= 384 workgroups -> convw_kernel)
    (4, 128, 128, 16000, 1, _lib.PRE_LRELU, _lib.POST_LRELU, False),
    (4, 128, 128, 16001, 3, _lib.PRE_LRELU, _lib.POST_NONE, True),
    (4, 256, 128, 15999, 9, _lib.PRE_LRELU, _lib.POST_LRELU, False),
    (4, 128, 128, 16002, 27, _lib.PRE_NONE, _lib.POST_NONE, True),
    (4, 128, 128, 16000, 81, _lib.PRE_LRELU, _lib.POST_LRELU_SNAKE, True),
    (4, 128, 256, 8000, 243, _lib.PRE_LRELU, _lib.POST_LRELU, False),
    (4, 128, 128, 16000, 729, _lib.PRE_LRELU, _lib.POST_NONE, True),
    (4, 128, 128, 16000, 2187, _lib.PRE_LRELU, _lib.POST_LRELU, False),
    (8, 64, 64, 16000, 1, _lib.PRE_LRELU, _lib.POST_LRELU, True),      # 64 x 256 tile
    (8, 64, 64, 16003, 27, _lib.PRE_LRELU, _lib.POST_NONE, False),
    (8, 128, 64, 16000, 81, _lib.PRE_LRELU, _lib.POST_NONE, True),
    (8, 64, 64, 16000, 2187, _lib.PRE_LRELU, _lib.POST_LRELU, True),
    (64, 32, 128, 1006, 1, _lib.PRE_NONE, _lib.POST_ELU, False),       # condnet-like: many short rows, Cin = 32 (one chunk)
]


@pytest.mark.parametrize("case", CONVW_CASES)
def test_conv1d_convw(case):
    B, Cin, Cout, L, dil, pre, post, use_res = case
    x = _rand((B, Cin, L), 61)
    w = _rand((Cout, Cin, 3), 62, (Cin * 3) ** -0.5)
    bias = _rand((Cout,), 63, 0.1)
    res = _rand((B, Cout, L), 64) if use_res else None
    ref = F.conv1d(_ref_act(x, pre, 0.01), w, bias, dilation=dil, padding=dil)
    if use_res:
        ref = ref + res
    ref = _ref_post(ref, post, 0.2)
    xd = _guarded_nan(x, dil + 264)
    lp = (L + 67) // 4 * 4
    yd = torch.full((B, Cout, lp), float("nan"), device=DEV)
    rd = _padded(res, lp) if use_res else None
    act = ops.Act(pre=pre, pre_slope=0.01, post=post, post_slope=0.2)
    wp = packing.pack_conv1d(w)
    before = _lib.lib().vfx_launch_count()
    ops.conv1d(xd, wp.to(DEV), bias.to(DEV), yd, L, 3, dil, 0, act, rd, wd=packing.pack_direct(wp).to(DEV))
    torch.cuda.synchronize()
    assert _lib.lib().vfx_launch_count() == before + 1
    assert _lib.lib().vfx_last_conv_tile() % 100 in (51, 52, 54), "launch did not run on convw_kernel"
    _close(yd[:, :, :L], ref, 2e-5)
    assert torch.isnan(yd[:, :, L:]).all()
    if use_res and post == _lib.POST_NONE:
        # in-place residual update (the engine's pattern for the unfused stages)
        rd2 = _padded(res, lp)
        ops.conv1d(xd, wp.to(DEV), bias.to(DEV), rd2, L, 3, dil, 0, act, rd2, wd=packing.pack_direct(wp).to(DEV))
        torch.cuda.synchronize()
        _close(rd2[:, :, :L], ref, 2e-5)


WINO_CASES = [
    # B, Cin, Cout, L, dil, pre, post, res   (k = 3; >= 512 workgroups of 128 channels x 64 pairs -> convwg_kernel)
    (8, 256, 256, 4100, 1, _lib.PRE_LRELU, _lib.POST_LRELU, False),
    (8, 256, 256, 4099, 1, _lib.PRE_NONE, _lib.POST_NONE, True),       # odd length: the last pair has one output
    (8, 256, 256, 4100, 3, _lib.PRE_LRELU, _lib.POST_LRELU, False),
    (8, 256, 128, 9001, 9, _lib.PRE_LRELU, _lib.POST_NONE, True),
    (8, 128, 256, 4100, 27, _lib.PRE_LRELU, _lib.POST_LRELU_SNAKE, True),
    (8, 256, 256, 4100, 81, _lib.PRE_LRELU, _lib.POST_LRELU, False),
    (8, 256, 256, 4100, 243, _lib.PRE_LRELU, _lib.POST_NONE, True),
    (8, 256, 256, 4100, 729, _lib.PRE_LRELU, _lib.POST_LRELU, False),
    (8, 256, 256, 4100, 2187, _lib.PRE_LRELU, _lib.POST_NONE, True),   # 2d > L: every pair has one output
    (4, 512, 512, 4200, 81, _lib.PRE_LRELU, _lib.POST_LRELU, False),
    (16, 32, 128, 4000, 1, _lib.PRE_NONE, _lib.POST_ELU, False),       # Cin = 32: two chunks
    (8, 64, 64, 17001, 1, _lib.PRE_LRELU, _lib.POST_NONE, True),       # Cout = 64: 64 channels x 128 pairs per workgroup
    (8, 64, 64, 18000, 27, _lib.PRE_LRELU, _lib.POST_LRELU, False),
    (8, 128, 64, 18000, 2187, _lib.PRE_LRELU, _lib.POST_NONE, True),
    (8, 64, 192, 8000, 81, _lib.PRE_LRELU, _lib.POST_LRELU, False),
]


@pytest.mark.parametrize("case", WINO_CASES)
def test_conv1d_winograd4(case):
    """convwg4_kernel (vfx_act.w_wino4): the k = 3 Conv1d as Winograd F(4,3) along the dilated axis -- six fp32 MFMA
    products per FOUR outputs instead of twelve.  Against torch's direct fp32 conv1d at the tolerance of the direct
    kernels (the transform constants, up to 8, cost about three times the direct sum's rounding: tools/winograd_error.py);
    no guard band is needed (taps outside the row are dropped by the buffer unit), nothing is written past L (NaN canaries),
    the in-place residual update works.  Dilation 1 runs the D1 instance: quads as aligned 16-byte vectors, with the quad
    that straddles the end of a row of odd length on single elements (L = 4099)."""
    B, Cin, Cout, L, dil, pre, post, use_res = case
    x = _rand((B, Cin, L), 261)
    w = _rand((Cout, Cin, 3), 262, (Cin * 3) ** -0.5)
    bias = _rand((Cout,), 263, 0.1)
    res = _rand((B, Cout, L), 264) if use_res else None
    ref = F.conv1d(_ref_act(x, pre, 0.01), w, bias, dilation=dil, padding=dil)
    if use_res:
        ref = ref + res
    ref = _ref_post(ref, post, 0.2)
    lp = (L + 67) // 4 * 4
    xd = torch.full((B, Cin, lp), float("nan"), device=DEV)     # no guard band; NaN right after the row
    xd[:, :, :L] = x.to(DEV)
    yd = torch.full((B, Cout, lp), float("nan"), device=DEV)
    rd = _padded(res, lp) if use_res else None
    act = ops.Act(pre=pre, pre_slope=0.01, post=post, post_slope=0.2)
    wp = packing.pack_conv1d(w)
    wg4 = packing.pack_wino4(wp).to(DEV)
    before = _lib.lib().vfx_launch_count()
    ops.conv1d(xd, wp.to(DEV), bias.to(DEV), yd, L, 3, dil, 0, act, rd, wg4=wg4)
    torch.cuda.synchronize()
    assert _lib.lib().vfx_launch_count() == before + 1
    assert _lib.lib().vfx_last_conv_tile() % 100 in (80, 81, 82), "launch did not run on convwg4_kernel"
    _close(yd[:, :, :L], ref, 2e-5)
    assert torch.isnan(yd[:, :, L:]).all()
    if use_res and post == _lib.POST_NONE:
        rd2 = _padded(res, lp)   # in-place residual update
        ops.conv1d(xd, wp.to(DEV), bias.to(DEV), rd2, L, 3, dil, 0, act, rd2, wg4=wg4)
        torch.cuda.synchronize()
        _close(rd2[:, :, :L], ref, 2e-5)


def test_conv1d_winograd4_ragged_rows():
    """Per-row lengths through the quad kernel: every row equals the same row convolved alone."""
    B, C, L = 8, 256, 4100
    lens = [4100, 4099, 2050, 2051, 3000, 54, 4047, 1]
    x = _rand((B, C, L), 271)
    w = _rand((C, C, 3), 272, (C * 3) ** -0.5)
    bias = _rand((C,), 273, 0.1)
    wp = packing.pack_conv1d(w)
    wg4 = packing.pack_wino4(wp).to(DEV)
    act = ops.Act(pre=_lib.PRE_LRELU, pre_slope=0.01)
    for dil in (27, 1, 729):
        xd = _guarded_nan(x, 8)
        yd = torch.full((B, C, L + 60), float("nan"), device=DEV)
        ops.with_rows(xd, torch.tensor(lens, dtype=torch.int32, device=DEV))
        ops.conv1d(xd, wp.to(DEV), bias.to(DEV), yd, L, 3, dil, 0, act, None, wg4=wg4)
        torch.cuda.synchronize()
        assert _lib.lib().vfx_last_conv_tile() % 100 in (80, 81, 82)
        for r, n in enumerate(lens):
            ref = F.conv1d(F.leaky_relu(x[r:r + 1, :, :n], 0.01), w, bias, dilation=dil, padding=dil)
            _close(yd[r:r + 1, :, :n], ref, 2e-5)
            assert torch.isnan(yd[r, :, n:]).all()


WINO_PERSIST_CASES = [
    # B, C, L, dil, second   -- the two launches of a ResStack layer (first: leaky ReLU before and after, no residual; second: dilation 1,
    # residual updated in place, no activation), long enough for the persistent kernel (>= 4 x 512 work items)
    (8, 128, 33001, 9, False),        # C = 128: one channel block, eight chunks
    (8, 128, 33001, 1, False),        # the dilation-1 first convolution (layer 0 of a stage): D1 instance, SPEC 1
    (8, 128, 33001, 1, True),         # odd length: the straddling quad on single elements
    (8, 128, 33000, 729, False),      # 4d blocks longer than a tile
    (8, 256, 16601, 27, False),       # two channel blocks per tile
    (8, 256, 16603, 1, True),
    (4, 512, 16600, 243, False),      # four channel blocks
    (4, 512, 16600, 1, True),
    (8, 64, 66001, 81, False),        # Cout = 64: 64 channels x 64 quads per workgroup, four chunks (first pair = all but the last pair)
    (8, 64, 66002, 1, True),
]


@pytest.mark.parametrize("case", WINO_PERSIST_CASES)
def test_conv1d_winograd4_persistent(case):
    """convwg4p_kernel (vfx_convwg4p.inc): the long launches of a ResStack layer on persistent workgroups whose tap prefetch, staging
    and A-vector prefetch run across tile boundaries.  Against torch's direct fp32 conv1d at the tolerance of convwg4_kernel (the
    arithmetic is the same); every workgroup walks >= 4 tiles of its list here, tiles of different batch items among them; nothing is
    written past L (NaN canaries), the in-place residual update of the second convolution works."""
    B, C, L, dil, second = case
    x = _rand((B, C, L), 281)
    w = _rand((C, C, 3), 282, (C * 3) ** -0.5)
    bias = _rand((C,), 283, 0.1)
    lp = (L + 67) // 4 * 4
    xd = torch.full((B, C, lp), float("nan"), device=DEV)
    xd[:, :, :L] = x.to(DEV)
    wp = packing.pack_conv1d(w)
    wg4 = packing.pack_wino4(wp).to(DEV)
    if second:
        res = _rand((B, C, L), 284)
        ref = F.conv1d(x, w, bias, padding=1) + res
        yd = _padded(res, lp)
        ops.conv1d(xd, wp.to(DEV), bias.to(DEV), yd, L, 3, 1, 0, ops.Act(), yd, wg4=wg4)
    else:
        ref = F.leaky_relu(F.conv1d(F.leaky_relu(x, 0.01), w, bias, dilation=dil, padding=dil), 0.01)
        yd = torch.full((B, C, lp), float("nan"), device=DEV)
        act = ops.Act(pre=_lib.PRE_LRELU, pre_slope=0.01, post=_lib.POST_LRELU, post_slope=0.01)
        ops.conv1d(xd, wp.to(DEV), bias.to(DEV), yd, L, 3, dil, 0, act, None, wg4=wg4)
    torch.cuda.synchronize()
    # C >= 128: convwg4x_kernel (round 6: 32 x 64 x 6 wave tile, deferred epilogue), C = 64: convwg4p_kernel<2,2>
    assert _lib.lib().vfx_last_conv_tile() % 100 == (82 if C >= 128 else 81), "launch did not run on convwg4x_kernel / convwg4p_kernel"
    _close(yd[:, :, :L], ref, 2e-5)
    assert torch.isnan(yd[:, :, L:]).all()


def test_conv1d_winograd4_persistent_ragged_rows():
    """Per-row lengths through the persistent kernel: tiles past a row's end are skipped and the pipeline restarts at the next valid
    tile of the workgroup's list; every row equals the same row convolved alone, nothing is written past a row's own end."""
    B, C, L = 8, 128, 33001
    lens = [33001, 32999, 16500, 1, 7, 33000, 20002, 4099]
    x = _rand((B, C, L), 291)
    w = _rand((C, C, 3), 292, (C * 3) ** -0.5)
    bias = _rand((C,), 293, 0.1)
    wp = packing.pack_conv1d(w)
    wg4 = packing.pack_wino4(wp).to(DEV)
    rows = torch.tensor(lens, dtype=torch.int32, device=DEV)
    lp = (L + 67) // 4 * 4
    for dil, second in ((9, False), (1, True), (1, False)):
        xd = torch.full((B, C, lp), float("nan"), device=DEV)
        xd[:, :, :L] = x.to(DEV)
        ops.with_rows(xd, rows)
        if second:
            res = _rand((B, C, L), 294)
            yd = _padded(res, lp)
            ops.conv1d(xd, wp.to(DEV), bias.to(DEV), yd, L, 3, 1, 0, ops.Act(), yd, wg4=wg4)
        else:
            yd = torch.full((B, C, lp), float("nan"), device=DEV)
            act = ops.Act(pre=_lib.PRE_LRELU, pre_slope=0.01, post=_lib.POST_LRELU, post_slope=0.01)
            ops.conv1d(xd, wp.to(DEV), bias.to(DEV), yd, L, 3, dil, 0, act, None, wg4=wg4)
        torch.cuda.synchronize()
        assert _lib.lib().vfx_last_conv_tile() % 100 == 82
        for r, n in enumerate(lens):
            if second:
                ref = F.conv1d(x[r:r + 1, :, :n], w, bias, padding=1) + res[r:r + 1, :, :n]
                _close(yd[r:r + 1, :, :n], ref, 2e-5)
                assert torch.equal(yd[r, :, n:L].cpu(), res[r, :, n:])          # the residual buffer past the row's end: untouched
            else:
                ref = F.leaky_relu(F.conv1d(F.leaky_relu(x[r:r + 1, :, :n], 0.01), w, bias, dilation=dil, padding=dil), 0.01)
                _close(yd[r:r + 1, :, :n], ref, 2e-5)
                assert torch.isnan(yd[r, :, n:]).all()


@pytest.mark.parametrize("cfg", [(8, 64, 128, 33003, 3, False), (8, 64, 128, 33002, 1, False), (8, 64, 128, 33001, 1, True),
                                 (8, 192, 256, 33002, 9, False), (8, 96, 128, 40001, 81, False), (8, 96, 128, 40002, 1, True)])
def test_conv1d_winograd4_reblocked_short_and_odd_k(cfg):
    """convwg4x_kernel's deferred epilogue off the happy path: Cin = 64 gives EIGHT 8-channel chunks per tile -- fewer than the nine
    (one to load + eight to use) the trickle of a finished tile's outputs needs, so the rest is drained un-overlapped in front of the
    next tile's epilogue; Cin = 192 / 80 put the slot-free second loop at an odd place.  Cin != Cout, odd lengths (the quad that
    straddles the row's end), NaN canaries past L, the in-place residual."""
    B, Cin, Cout, L, dil, second = cfg
    x = _rand((B, Cin, L), 301)
    w = _rand((Cout, Cin, 3), 302, (Cin * 3) ** -0.5)
    bias = _rand((Cout,), 303, 0.1)
    lp = (L + 67) // 4 * 4
    xd = torch.full((B, Cin, lp), float("nan"), device=DEV)
    xd[:, :, :L] = x.to(DEV)
    wp = packing.pack_conv1d(w)
    wg4 = packing.pack_wino4(wp).to(DEV)
    if second:
        res = _rand((B, Cout, L), 304)
        ref = F.conv1d(x, w, bias, padding=1) + res
        yd = _padded(res, lp)
        ops.conv1d(xd, wp.to(DEV), bias.to(DEV), yd, L, 3, 1, 0, ops.Act(), yd, wg4=wg4)
    else:
        ref = F.leaky_relu(F.conv1d(F.leaky_relu(x, 0.01), w, bias, dilation=dil, padding=dil), 0.01)
        yd = torch.full((B, Cout, lp), float("nan"), device=DEV)
        act = ops.Act(pre=_lib.PRE_LRELU, pre_slope=0.01, post=_lib.POST_LRELU, post_slope=0.01)
        ops.conv1d(xd, wp.to(DEV), bias.to(DEV), yd, L, 3, dil, 0, act, None, wg4=wg4)
    torch.cuda.synchronize()
    assert _lib.lib().vfx_last_conv_tile() % 100 == 82, "launch did not run on convwg4x_kernel"
    _close(yd[:, :, :L], ref, 2e-5)
    assert torch.isnan(yd[:, :, L:]).all()


def test_conv1d_winograd4_fallbacks():
    """Launches the Winograd kernel declines -- too few workgroups for its tiles, Cout = 96 -- run on the direct kernels
    with the same result."""
    for (b2, cin, cout, l2) in ((1, 256, 256, 900), (8, 64, 96, 4000)):
        x2 = _rand((b2, cin, l2), 174)
        w2 = _rand((cout, cin, 3), 175, (cin * 3) ** -0.5)
        wp2 = packing.pack_conv1d(w2)
        y2 = torch.full((b2, cout, l2 + 4), float("nan"), device=DEV)
        ops.conv1d(_guarded_nan(x2, 300), wp2.to(DEV), None, y2, l2, 3, 3, 0, None, None, wg4=packing.pack_wino4(wp2).to(DEV))
        torch.cuda.synchronize()
        assert _lib.lib().vfx_last_conv_tile() % 100 not in (80, 81, 82)
        _close(y2[:, :, :l2], F.conv1d(x2, w2, None, dilation=3, padding=3), 2e-5)


@pytest.mark.parametrize("cfg", [(4, 256, 128, 5000, 3), (2, 512, 256, 2000, 7), (8, 128, 64, 6000, 3)])
def test_convtr1d_convw(cfg):
    B, Cin, Cout, Lin, s = cfg
    x = _rand((B, Cin, Lin), 71)
    w = _rand((Cin, Cout, 2 * s), 72, (2 * Cin) ** -0.5)
    bias = _rand((Cout,), 73, 0.1)
    ref = F.conv_transpose1d(x + torch.sin(x) * 0, w, bias, stride=s, padding=s // 2 + s % 2, output_padding=s % 2)
    xd = _guarded_nan(x, 264)
    Lo = Lin * s
    yd = torch.full((B, Cout, (Lo + 67) // 4 * 4), float("nan"), device=DEV)
    wp = packing.pack_convtr1d(w)
    ops.convtr1d(xd, wp.to(DEV), bias.to(DEV), yd, Lin, s, None, wd=packing.pack_direct(wp).to(DEV))
    torch.cuda.synchronize()
    assert _lib.lib().vfx_last_conv_tile() % 100 in (51, 52, 54), "launch did not run on convw_kernel"
    _close(yd[:, :, :Lo], ref, 2e-5)
    assert torch.isnan(yd[:, :, Lo:]).all()


@pytest.mark.parametrize("cfg", [(4, 256, 128, 5000, 3), (2, 512, 256, 2001, 7), (8, 128, 64, 6002, 3), (2, 1024, 512, 1006, 7),
                                 (3, 64, 128, 4099, 7), (16, 32, 64, 6001, 2)])
def test_convtr1d_winograd32(cfg):
    """convtw_kernel (vfx_convtw.inc): the polyphase ConvTranspose1d as Winograd F(3,2) along the input axis -- four products per three
    outputs of a phase instead of six.  Against torch's direct fp32 conv_transpose1d at the direct kernels' tolerance (the transform
    constants are 1 and 1/2); input lengths that leave one or two positions in the last triple, both tile shapes (Cout % 128 == 0: 128
    channels x 32 triples, else 64 x 64), even and odd strides, NaN canaries in the input guard band and past the output's end."""
    B, Cin, Cout, Lin, s = cfg
    x = _rand((B, Cin, Lin), 311)
    w = _rand((Cin, Cout, 2 * s), 312, (2 * Cin) ** -0.5)
    bias = _rand((Cout,), 313, 0.1)
    ref = F.conv_transpose1d(x, w, bias, stride=s, padding=s // 2 + s % 2, output_padding=s % 2)
    xd = _guarded_nan(x, 264)
    Lo = Lin * s
    yd = torch.full((B, Cout, (Lo + 67) // 4 * 4), float("nan"), device=DEV)
    wp = packing.pack_convtr1d(w)
    ops.convtr1d(xd, wp.to(DEV), bias.to(DEV), yd, Lin, s, None, wd=packing.pack_direct(wp).to(DEV), wg4=packing.pack_wino32_tr(wp, s).to(DEV))
    torch.cuda.synchronize()
    assert _lib.lib().vfx_last_conv_tile() % 100 == 83, "launch did not run on convtw_kernel"
    _close(yd[:, :, :Lo], ref, 2e-5)
    assert torch.isnan(yd[:, :, Lo:]).all()
    # no bias, and a launch too small for it falls back to the direct kernels with the same result
    yd2 = torch.full((B, Cout, (Lo + 67) // 4 * 4), float("nan"), device=DEV)
    ops.convtr1d(xd, wp.to(DEV), None, yd2, Lin, s, None, wd=packing.pack_direct(wp).to(DEV), wg4=packing.pack_wino32_tr(wp, s).to(DEV))
    torch.cuda.synchronize()
    _close(yd2[:, :, :Lo], ref - bias[None, :, None], 2e-5)


def test_convtr1d_winograd32_ragged_rows():
    """Per-row input lengths through convtw_kernel: every row equals the same row up-sampled alone, nothing is written past s x its length."""
    B, Cin, Cout, Lin, s = 6, 256, 128, 7042, 7
    lens = [7042, 7041, 3521, 1, 2, 5000]
    x = _rand((B, Cin, Lin), 321)
    w = _rand((Cin, Cout, 2 * s), 322, (2 * Cin) ** -0.5)
    bias = _rand((Cout,), 323, 0.1)
    xd = _guarded_nan(x, 264)
    ops.with_rows(xd, torch.tensor(lens, dtype=torch.int32, device=DEV))
    Lo = Lin * s
    yd = torch.full((B, Cout, (Lo + 67) // 4 * 4), float("nan"), device=DEV)
    wp = packing.pack_convtr1d(w)
    ops.convtr1d(xd, wp.to(DEV), bias.to(DEV), yd, Lin, s, None, wd=packing.pack_direct(wp).to(DEV), wg4=packing.pack_wino32_tr(wp, s).to(DEV))
    torch.cuda.synchronize()
    assert _lib.lib().vfx_last_conv_tile() % 100 == 83
    for r, n in enumerate(lens):
        ref = F.conv_transpose1d(x[r:r + 1, :, :n], w, bias, stride=s, padding=s // 2 + s % 2, output_padding=s % 2)
        _close(yd[r:r + 1, :, :n * s], ref, 2e-5)
        assert torch.isnan(yd[r, :, n * s:]).all()


def test_convw_small_launches_stay_on_the_first_kernel():
    x = _rand((1, 128, 700), 81)
    w = _rand((128, 128, 3), 82, 0.05)
    xd = _guarded_nan(x, 300)
    yd = torch.empty((1, 128, 700), device=DEV)
    wp = packing.pack_conv1d(w)
    ops.conv1d(xd, wp.to(DEV), None, yd, 700, 3, 1, 0, None, wd=packing.pack_direct(wp).to(DEV))
    torch.cuda.synchronize()
    assert _lib.lib().vfx_last_conv_tile() % 100 < 50
    _close(yd, F.conv1d(x, w, padding=1), 2e-5)


RESBLOCK_CASES = [
    # B, C, L, dil, post
    (3, 64, 20000, 1, _lib.POST_NONE),
    (2, 64, 20001, 3, _lib.POST_NONE),
    (2, 64, 9999, 9, _lib.POST_LRELU),
    (2, 64, 20003, 27, _lib.POST_NONE),
    (2, 64, 20000, 81, _lib.POST_NONE),
    (2, 64, 20000, 243, _lib.POST_NONE),
    (2, 64, 20000, 729, _lib.POST_LRELU_SNAKE),
    (2, 64, 20000, 2187, _lib.POST_NONE),
    (1, 64, 300, 2187, _lib.POST_NONE),        # dilation > L, single tile
    (1, 64, 253, 1, _lib.POST_NONE),           # one tile minus one
    (1, 64, 254, 1, _lib.POST_NONE),           # exactly one tile of 254 outputs
    (1, 64, 255, 1, _lib.POST_NONE),
    (1, 64, 1, 1, _lib.POST_NONE),
    (1, 64, 251, 3, _lib.POST_NONE),           # F(4,3) second half: 252 outputs (63 quads) per tile
    (1, 64, 252, 1, _lib.POST_LRELU),
    (2, 64, 506, 1, _lib.POST_NONE),           # two tiles and half a quad
    (2, 64, 20002, 9, _lib.POST_LRELU_SNAKE),
    (1, 64, 247, 9, _lib.POST_NONE),           # both halves F(4,3), d = 9: 248 outputs (62 quads) per tile -- one tile minus one
    (1, 64, 248, 3, _lib.POST_NONE),           # exactly one tile
    (2, 64, 249, 9, _lib.POST_LRELU),          # one tile plus one output
    (1, 64, 212, 27, _lib.POST_NONE),          # d = 27: two blocks of 108 columns, 53 output quads
    (2, 64, 641, 27, _lib.POST_NONE),          # three tiles and one output
    (1, 64, 30, 27, _lib.POST_NONE),           # a row shorter than one block of 4d positions
    (2, 128, 9000, 1, _lib.POST_NONE),
    (2, 128, 9001, 3, _lib.POST_LRELU),
    (2, 128, 9000, 27, _lib.POST_NONE),
    (2, 128, 9002, 81, _lib.POST_NONE),
    (2, 128, 9000, 243, _lib.POST_NONE),
    (2, 128, 9000, 2187, _lib.POST_LRELU_SNAKE),
    (1, 128, 126, 9, _lib.POST_NONE),
    (1, 128, 127, 9, _lib.POST_NONE),
]


@pytest.mark.parametrize("case", RESBLOCK_CASES)
def test_resblock_fused(case):
    """vfx_resblock_f32 = one iteration of ResStack.forward (vocoder/model/modules.py:592-609):
    x + conv_k3_d1(lrelu(conv_k3_dil(lrelu(x)))) with the intermediate tile in LDS."""
    B, Cn, L, dil, post = case
    x = _rand((B, Cn, L), 91)
    w1 = _rand((Cn, Cn, 3), 92, (Cn * 3) ** -0.5)
    b1 = _rand((Cn,), 93, 0.1)
    w2 = _rand((Cn, Cn, 3), 94, (Cn * 3) ** -0.5)
    b2 = _rand((Cn,), 95, 0.1)
    mid = F.conv1d(F.leaky_relu(x, 0.01), w1, b1, dilation=dil, padding=dil)
    ref = x + F.conv1d(F.leaky_relu(mid, 0.01), w2, b2, padding=1)
    ref = _ref_post(ref, post, 0.2)
    xd = _guarded_nan(x, 2187 + 264)
    yd = ops.guarded(B, Cn, L, 2187 + 264, DEV)
    yd._vfx_base.fill_(float("nan"))
    w1d = packing.pack_direct(packing.pack_conv1d(w1)).to(DEV)
    w2d = packing.pack_direct(packing.pack_conv1d(w2)).to(DEV)
    before = _lib.lib().vfx_launch_count()
    ops.resblock(xd, yd, w1d, b1.to(DEV), w2d, b2.to(DEV), L, dil, 0.01, post, 0.2)
    torch.cuda.synchronize()
    assert _lib.lib().vfx_launch_count() == before + 1
    assert _lib.lib().vfx_last_conv_tile() % 100 in (61, 62, 64)
    _close(yd[:, :, :L], ref, 2e-5)
    base = yd._vfx_base
    g = yd._vfx_guard
    assert torch.isnan(base[:, :, :g]).all() and torch.isnan(base[:, :, g + L:]).all()  # nothing written outside [0, L)
    with pytest.raises(_lib.VfxError):
        ops.resblock(xd, xd, w1d, b1.to(DEV), w2d, b2.to(DEV), L, dil)      # in place is refused
    with pytest.raises(_lib.VfxError):
        ops.resblock(xd[:, :32], yd[:, :32], w1d, b1.to(DEV), w2d, b2.to(DEV), L, dil)   # C = 32 is not covered
    # the same layer with its dilation-1 half as Winograd F(2,3) on the LDS tile (vfx_resblock_f32 with w2_wino)
    yd2 = ops.guarded(B, Cn, L, 2187 + 264, DEV)
    yd2._vfx_base.fill_(float("nan"))
    ops.resblock(xd, yd2, w1d, b1.to(DEV), w2d, b2.to(DEV), L, dil, 0.01, post, 0.2,
                 w2g=packing.pack_wino(packing.pack_conv1d(w2)).to(DEV))
    torch.cuda.synchronize()
    assert _lib.lib().vfx_last_conv_tile() % 100 in (71, 72, 74)
    _close(yd2[:, :, :L], ref, 2e-5)
    base = yd2._vfx_base
    assert torch.isnan(base[:, :, :g]).all() and torch.isnan(base[:, :, g + L:]).all()
    if Cn != 64:
        return
    # ... and as Winograd F(4,3): wave = 32 channels x 32 output quads, residual and stores as 16-byte vectors, the quad
    # that straddles the end of a row of odd length as single elements (vfx_resblock_f32 with w2_wino4)
    yd3 = ops.guarded(B, Cn, L, 2187 + 264, DEV)
    yd3._vfx_base.fill_(float("nan"))
    ops.resblock(xd, yd3, w1d, b1.to(DEV), w2d, b2.to(DEV), L, dil, 0.01, post, 0.2,
                 w2g=packing.pack_wino(packing.pack_conv1d(w2)).to(DEV),
                 w2g4=packing.pack_wino4(packing.pack_conv1d(w2)).to(DEV))
    torch.cuda.synchronize()
    assert _lib.lib().vfx_last_conv_tile() % 100 in (91, 92, 94)
    _close(yd3[:, :, :L], ref, 2e-5)
    base = yd3._vfx_base
    assert torch.isnan(base[:, :, :g]).all() and torch.isnan(base[:, :, g + L:]).all()
    # ... and with the FIRST (dilated) convolution as F(4,3) too (vfx_resblock_f32 with w1_wino4: resblk4_kernel) for the dilations whose
    # blocks of 4d positions fit the 256-column tile (1, 3, 9, 27); wider dilations keep the form above
    yd4 = ops.guarded(B, Cn, L, 2187 + 264, DEV)
    yd4._vfx_base.fill_(float("nan"))
    ops.resblock(xd, yd4, w1d, b1.to(DEV), w2d, b2.to(DEV), L, dil, 0.01, post, 0.2,
                 w2g=packing.pack_wino(packing.pack_conv1d(w2)).to(DEV),
                 w2g4=packing.pack_wino4(packing.pack_conv1d(w2)).to(DEV),
                 w1g4=packing.pack_wino4(packing.pack_conv1d(w1)).to(DEV))
    torch.cuda.synchronize()
    assert _lib.lib().vfx_last_conv_tile() % 100 in ((96,) if dil <= 27 else (91, 92, 94)), _lib.lib().vfx_last_conv_tile()
    _close(yd4[:, :, :L], ref, 2e-5)
    base = yd4._vfx_base
    assert torch.isnan(base[:, :, :g]).all() and torch.isnan(base[:, :, g + L:]).all()
    # rows that are not 16-byte aligned fall back to the F(2,3) form
    if L > 8:
        yo = ops.guarded(B, Cn, L + 4, 2187 + 264, DEV)
        yv = yo[:, :, 1:1 + L]
        yv._vfx_guard, yv._vfx_base = yo._vfx_guard, yo._vfx_base
        ops.resblock(xd, yv, w1d, b1.to(DEV), w2d, b2.to(DEV), L, dil, 0.01, post, 0.2,
                     w2g=packing.pack_wino(packing.pack_conv1d(w2)).to(DEV),
                     w2g4=packing.pack_wino4(packing.pack_conv1d(w2)).to(DEV))
        torch.cuda.synchronize()
        assert _lib.lib().vfx_last_conv_tile() % 100 in (71, 72, 74)
        _close(yv, ref, 2e-5)



@pytest.mark.parametrize("dil", [1, 3, 9, 27])
def test_resblock_both_halves_winograd_ragged_rows(dil):
    """resblk4_kernel on a ragged batch: every row equals the layer applied to that row alone (zero padding of BOTH convolutions at
    the row's own end), nothing is written past a row's end."""
    B, Cn, Lmax = 4, 64, 3000
    lens = [3000, 2999, 1201, 500]
    x = _rand((B, Cn, Lmax), 191)
    w1 = _rand((Cn, Cn, 3), 192, (Cn * 3) ** -0.5)
    b1 = _rand((Cn,), 193, 0.1)
    w2 = _rand((Cn, Cn, 3), 194, (Cn * 3) ** -0.5)
    b2 = _rand((Cn,), 195, 0.1)
    rows = torch.tensor(lens, dtype=torch.int32, device=DEV)
    xd = ops.with_rows(_guarded_nan(x, 2187 + 264), rows)
    for b, n in enumerate(lens):                 # what lies behind a row's end inside the buffer must not matter
        xd[b, :, n:] = float("nan")
    yd = ops.with_rows(ops.guarded(B, Cn, Lmax, 2187 + 264, DEV), rows)
    yd._vfx_base.fill_(float("nan"))
    pk = lambda w: packing.pack_conv1d(w)
    ops.resblock(xd, yd, packing.pack_direct(pk(w1)).to(DEV), b1.to(DEV), packing.pack_direct(pk(w2)).to(DEV), b2.to(DEV), Lmax, dil,
                 0.01, _lib.POST_NONE, 0.0, w2g=packing.pack_wino(pk(w2)).to(DEV), w2g4=packing.pack_wino4(pk(w2)).to(DEV),
                 w1g4=packing.pack_wino4(pk(w1)).to(DEV))
    torch.cuda.synchronize()
    assert _lib.lib().vfx_last_conv_tile() % 100 == 96
    for b, n in enumerate(lens):
        xb = x[b:b + 1, :, :n]
        mid = F.conv1d(F.leaky_relu(xb, 0.01), w1, b1, dilation=dil, padding=dil)
        ref = xb + F.conv1d(F.leaky_relu(mid, 0.01), w2, b2, padding=1)
        _close(yd[b:b + 1, :, :n], ref, 2e-5)
        assert torch.isnan(yd[b, :, n:Lmax]).all()


def test_linear_transposed_output():
    """Linear as conv k=1 with a frame-major output view (B,T,out) -- used for GRU x-projections."""
    B, T, Cin, Cout = 2, 101, 512, 1536
    x = _rand((B, Cin, T), 5)
    w = _rand((Cout, Cin), 6, Cin ** -0.5)
    bias = _rand((Cout,), 7, 0.1)
    ref = F.linear(x.transpose(1, 2), w, bias)  # (B,T,Cout)
    xd = _padded(x, 104)
    out = torch.full((B, T, Cout), float("nan"), device=DEV)
    ops.conv1d(xd, packing.pack_linear(w).to(DEV), bias.to(DEV), out.transpose(1, 2), T, 1)
    torch.cuda.synchronize()
    _close(out, ref, 2e-5)


@pytest.mark.parametrize("cfg", [(1, 1024, 512, 106, 7), (2, 256, 128, 531, 3), (1, 128, 64, 1000, 3),
                                 (2, 64, 32, 37, 7)])
def test_convtr1d(cfg):
    B, Cin, Cout, Lin, s = cfg
    x = _rand((B, Cin, Lin), 8)
    w = _rand((Cin, Cout, 2 * s), 9, (2 * Cin) ** -0.5)
    bias = _rand((Cout,), 10, 0.1)
    ref = F.conv_transpose1d(x, w, bias, stride=s, padding=s // 2 + s % 2, output_padding=s % 2)
    assert ref.shape[-1] == s * Lin
    xd = _padded(x, (Lin + 7) // 4 * 4)
    Lo = s * Lin
    yd = torch.full((B, Cout, (Lo + 7) // 4 * 4), float("nan"), device=DEV)
    ops.convtr1d(xd, packing.pack_convtr1d(w).to(DEV), bias.to(DEV), yd, Lin, s)
    torch.cuda.synchronize()
    _close(yd[:, :, :Lo], ref, 2e-5)
    assert torch.isnan(yd[:, :, Lo:]).all()


def _to_pitch(t, lp):
    """(B,C,H,W=P-1) -> (B,C,H*P) with a NaN pad column (kernels must mask it on read)."""
    B, Cn, H, W = t.shape
    P = 1 << lp
    assert W == P - 1
    buf = torch.full((B, Cn, H, P), float("nan"))
    buf[..., :W] = t
    return buf.reshape(B, Cn, H * P)


def _from_pitch(t, H, lp):
    P = 1 << lp
    return t.reshape(t.shape[0], t.shape[1], H, P)


@pytest.mark.parametrize("cfg", [(2, 32, 32, 64, 7, True), (1, 2, 32, 64, 7, False), (2, 64, 64, 32, 6, True),
                                 (1, 384, 384, 4, 2, True), (1, 768, 384, 2, 1, False), (3, 128, 128, 16, 5, True)])
def test_conv2d_3x3_bn_lrelu_residual(cfg):
    B, Cin, Cout, H, lp, use_res = cfg
    P = 1 << lp
    x = _rand((B, Cin, H, P - 1), 11)
    w = _rand((Cout, Cin, 3, 3), 12, (Cin * 9) ** -0.5)
    scale = 0.8 + 0.4 * torch.rand(Cin, generator=torch.Generator().manual_seed(13))
    shift = _rand((Cin,), 14, 0.3)
    res = _rand((B, Cout, H, P - 1), 15) if use_res else None
    ref = F.conv2d(_ref_act(x, _lib.PRE_AFFINE_LRELU, 0.01, scale, shift), w, padding=1)
    if use_res:
        ref = ref + res
    xd = _to_pitch(x, lp).to(DEV)
    yd = torch.full((B, Cout, H * P), float("nan"), device=DEV)
    rd = _to_pitch(res, lp).to(DEV) if use_res else None
    if rd is not None:
        rd = torch.nan_to_num(rd, nan=0.0)  # residual pad column is a structural zero in real use
    act = ops.Act(pre=_lib.PRE_AFFINE_LRELU, pre_slope=0.01, scale=scale.to(DEV), shift=shift.to(DEV))
    ops.conv2d(xd, packing.pack_conv2d(w).to(DEV), None, yd, H, lp, 3, act, rd)
    torch.cuda.synchronize()
    got = _from_pitch(yd, H, lp)
    _close(got[..., : P - 1], ref, 2e-5)
    assert (got[..., P - 1] == 0).all()  # pad column written as zero


@pytest.mark.parametrize("cfg", [(4, 32, 32, 256, 7, True, True), (8, 8, 32, 256, 7, False, True), (12, 64, 64, 64, 6, True, False),
                                 (24, 128, 128, 64, 5, True, True), (32, 64, 128, 64, 5, False, True),
                                 (32, 384, 384, 64, 3, True, True)])
def test_conv2d_3x3_convw(cfg):
    """3x3 on a pitch map on convw_kernel (vfx_act.w_direct; three row segments per chunk): eval-BatchNorm pre-activation
    or none (the two convolutions of a ConvBlockRes), residual, pad column NaN on input and written as zero."""
    B, Cin, Cout, H, lp, affine, use_res = cfg
    P = 1 << lp
    x = _rand((B, Cin, H, P - 1), 111)
    w = _rand((Cout, Cin, 3, 3), 112, (Cin * 9) ** -0.5)
    scale = 0.8 + 0.4 * torch.rand(Cin, generator=torch.Generator().manual_seed(113))
    shift = _rand((Cin,), 114, 0.3)
    bias = _rand((Cout,), 116, 0.1)
    res = _rand((B, Cout, H, P - 1), 115) if use_res else None
    xin = _ref_act(x, _lib.PRE_AFFINE_LRELU, 0.01, scale, shift) if affine else x
    ref = F.conv2d(xin, w, bias, padding=1)
    if use_res:
        ref = ref + res
    ref = F.leaky_relu(ref, 0.01) if affine else ref
    G = P + 1 + 264
    xd = ops.guarded(B, Cin, H * P, G, DEV)
    xd._vfx_base.fill_(float("nan"))
    xd[:, :, :H * P] = _to_pitch(x, lp).to(DEV)
    yd = torch.full((B, Cout, H * P), float("nan"), device=DEV)
    rd = torch.nan_to_num(_to_pitch(res, lp).to(DEV), nan=0.0) if use_res else None
    act = (ops.Act(pre=_lib.PRE_AFFINE_LRELU, pre_slope=0.01, scale=scale.to(DEV), shift=shift.to(DEV),
                   post=_lib.POST_LRELU, post_slope=0.01) if affine else None)
    wp = packing.pack_conv2d(w)
    ops.conv2d(xd, wp.to(DEV), bias.to(DEV), yd, H, lp, 3, act, rd, wd=packing.pack_direct(wp).to(DEV))
    torch.cuda.synchronize()
    assert _lib.lib().vfx_last_conv_tile() % 100 == 59, "launch did not run on convw_kernel"
    got = _from_pitch(yd, H, lp)
    _close(got[..., : P - 1], ref, 2e-5)
    assert (got[..., P - 1] == 0).all()  # pad column written as zero
    if use_res:  # in-place residual (the engine's ConvBlockRes pattern: out aliases the residual)
        rd2 = rd.clone()
        ops.conv2d(xd, wp.to(DEV), bias.to(DEV), rd2, H, lp, 3, act, rd2, wd=packing.pack_direct(wp).to(DEV))
        torch.cuda.synchronize()
        _close(_from_pitch(rd2, H, lp)[..., : P - 1], ref, 2e-5)


@pytest.mark.parametrize("cfg", [(32, 64, 64, 128, 6, True, False), (32, 32, 64, 128, 6, False, True), (32, 128, 128, 64, 5, True, True),
                                 (32, 384, 384, 128, 3, True, True), (32, 256, 128, 130, 4, True, False),
                                 (48, 128, 64, 99, 5, True, True),
                                 # UNet level 0 (Cout = 32, pitch 128: the <1,4,8> instance), its 64 -> 32 decoder entry,
                                 # a map height that is not a multiple of 4, Cin = 16 (one chunk pair)
                                 (24, 32, 32, 64, 7, True, True), (24, 64, 32, 68, 7, True, False), (26, 16, 32, 63, 7, False, True),
                                 (8, 64, 64, 256, 6, True, True), (40, 64, 128, 256, 3, True, False)])
def test_conv2d_3x3_winograd4(cfg):
    """3x3 on a pitch map as Winograd F(4,3) along the map rows (vfx_act.w_wino4 = packing.pack_wino4_2d), 18 products per
    four outputs instead of 36, on convwg4s_kernel: the three kernel columns read ONE staged tile at column shifts -1 / 0 /
    +1, the pad columns of the map are the zero padding between rows (round 3).  NaN in every guard band and pad column of
    the input (the kernel must mask what it reads there), map heights that are not multiples of 4, Cout = 32 / 64 / 128 / 384
    blocks, the in-place residual; the output pad column is written as zero."""
    B, Cin, Cout, H, lp, affine, use_res = cfg
    P = 1 << lp
    x = _rand((B, Cin, H, P - 1), 311)
    w = _rand((Cout, Cin, 3, 3), 312, (Cin * 9) ** -0.5)
    scale = 0.8 + 0.4 * torch.rand(Cin, generator=torch.Generator().manual_seed(313))
    shift = _rand((Cin,), 314, 0.3)
    bias = _rand((Cout,), 316, 0.1)
    res = _rand((B, Cout, H, P - 1), 315) if use_res else None
    xin = _ref_act(x, _lib.PRE_AFFINE_LRELU, 0.01, scale, shift) if affine else x
    ref = F.conv2d(xin, w, bias, padding=1)
    if use_res:
        ref = ref + res
    ref = F.leaky_relu(ref, 0.01) if affine else ref
    G = P + 1 + 264
    xd = ops.guarded(B, Cin, H * P, G, DEV)
    xd._vfx_base.fill_(float("nan"))
    xd[:, :, :H * P] = _to_pitch(x, lp).to(DEV)
    yd = torch.full((B, Cout, H * P), float("nan"), device=DEV)
    rd = torch.nan_to_num(_to_pitch(res, lp).to(DEV), nan=0.0) if use_res else None
    act = (ops.Act(pre=_lib.PRE_AFFINE_LRELU, pre_slope=0.01, scale=scale.to(DEV), shift=shift.to(DEV),
                   post=_lib.POST_LRELU, post_slope=0.01) if affine else None)
    wp = packing.pack_conv2d(w)
    wg4 = packing.pack_wino4_2d(wp).to(DEV)
    before = _lib.lib().vfx_launch_count()
    ops.conv2d(xd, wp.to(DEV), bias.to(DEV), yd, H, lp, 3, act, rd, wg4=wg4)
    torch.cuda.synchronize()
    assert _lib.lib().vfx_launch_count() == before + 1
    assert _lib.lib().vfx_last_conv_tile() % 100 == 88, "launch did not run on convwg4s_kernel (3x3, shared staging)"
    got = _from_pitch(yd, H, lp)
    _close(got[..., : P - 1], ref, 2e-5)
    assert (got[..., P - 1] == 0).all()  # pad column written as zero
    if use_res:
        rd2 = rd.clone()
        ops.conv2d(xd, wp.to(DEV), bias.to(DEV), rd2, H, lp, 3, act, rd2, wg4=wg4)
        torch.cuda.synchronize()
        _close(_from_pitch(rd2, H, lp)[..., : P - 1], ref, 2e-5)


def test_conv2d_3x3_winograd4_ragged_rows():
    """Per-row map heights (ragged batches: row b of the batch is a map of rows[b] / P <= H rows): every row equals the
    convolution of that map alone (zero padding below ITS last row), nothing is written past it."""
    B, Cin, Cout, H, lp = 12, 64, 64, 128, 6
    P = 1 << lp
    heights = [128, 64, 127, 4, 65, 128, 1, 96, 33, 128, 2, 100]
    x = _rand((B, Cin, H, P - 1), 411)
    w = _rand((Cout, Cin, 3, 3), 412, (Cin * 9) ** -0.5)
    scale = 0.8 + 0.4 * torch.rand(Cin, generator=torch.Generator().manual_seed(413))
    shift = _rand((Cin,), 414, 0.3)
    xd = ops.guarded(B, Cin, H * P, P + 1 + 264, DEV)
    xd._vfx_base.fill_(float("nan"))
    xd[:, :, :H * P] = _to_pitch(x, lp).to(DEV)
    ops.with_rows(xd, torch.tensor([h * P for h in heights], dtype=torch.int32, device=DEV))
    yd = torch.full((B, Cout, H * P), float("nan"), device=DEV)
    act = ops.Act(pre=_lib.PRE_AFFINE_LRELU, pre_slope=0.01, scale=scale.to(DEV), shift=shift.to(DEV))
    wp = packing.pack_conv2d(w)
    ops.conv2d(xd, wp.to(DEV), None, yd, H, lp, 3, act, None, wg4=packing.pack_wino4_2d(wp).to(DEV))
    torch.cuda.synchronize()
    assert _lib.lib().vfx_last_conv_tile() % 100 == 88
    got = _from_pitch(yd, H, lp)
    for b, h in enumerate(heights):
        ref = F.conv2d(_ref_act(x[b:b + 1, :, :h], _lib.PRE_AFFINE_LRELU, 0.01, scale, shift), w, padding=1)
        _close(got[b:b + 1, :, :h, : P - 1], ref, 2e-5)
        assert torch.isnan(got[b, :, h:]).all()


@pytest.mark.parametrize("cfg", [(2, 32, 32, 64, 7, True, True), (1, 64, 64, 48, 6, True, False),
                                 (2, 128, 128, 16, 5, False, True), (1, 384, 384, 8, 3, True, True),
                                 (1, 768, 384, 8, 3, False, False), (1, 64, 128, 32, 5, True, False)])
def test_conv2d_3x3_bf16x3(cfg):
    """3x3 on a pitch map with VFX_MATH_BF16X3: runs as 3 kernel rows x the 3-tap case (Cout 32 / 64 / 128k
    tiles), with the fused eval-BatchNorm + leaky-ReLU pre-activation, NaN pad columns and NaN guards."""
    B, Cin, Cout, H, lp, use_res, affine = cfg
    P = 1 << lp
    x = _rand((B, Cin, H, P - 1), 71)
    w = _rand((Cout, Cin, 3, 3), 72, (Cin * 9) ** -0.5)
    scale = 0.8 + 0.4 * torch.rand(Cin, generator=torch.Generator().manual_seed(73))
    shift = _rand((Cin,), 74, 0.3)
    res = _rand((B, Cout, H, P - 1), 75) if use_res else None
    xin = _ref_act(x, _lib.PRE_AFFINE_LRELU, 0.01, scale, shift) if affine else x
    ref = F.conv2d(xin, w, padding=1)
    if use_res:
        ref = ref + res
    ref = F.leaky_relu(ref, 0.01)
    xd = ops.guarded(B, Cin, H * P, P + 1 + 264, DEV)
    xd._vfx_base.fill_(float("nan"))
    xp = _to_pitch(x, lp)
    if not affine:
        xp = torch.nan_to_num(xp, nan=0.0)  # without a pre-activation the pad column is a structural zero
    xd[:, :, :H * P] = xp.to(DEV)
    yd = torch.full((B, Cout, H * P), float("nan"), device=DEV)
    rd = torch.nan_to_num(_to_pitch(res, lp), nan=0.0).to(DEV) if use_res else None
    act = ops.Act(pre=_lib.PRE_AFFINE_LRELU if affine else _lib.PRE_NONE, pre_slope=0.01,
                  scale=scale.to(DEV) if affine else None, shift=shift.to(DEV) if affine else None,
                  post=_lib.POST_LRELU, post_slope=0.01)
    wp = packing.pack_conv2d(w)
    ops.conv2d(xd, wp.to(DEV), None, yd, H, lp, 3, act, rd, w3=packing.pack_x3(wp).to(DEV))
    torch.cuda.synchronize()
    # (launches of <= 96 workgroups with K >= 1024 deliberately fall back to the fp32 split-K path)
    tiny = (B * H * P + 127) // 128 * max(Cout // 128, 1) <= 96 and Cin * 9 >= 1024
    assert tiny or _lib.lib().vfx_last_conv_tile() % 100 == 16, "bf16x3 kernel did not run (fp32 fallback)"
    got = _from_pitch(yd, H, lp)
    _close(got[..., : P - 1], ref, 1e-4)
    assert (got[..., P - 1] == 0).all()  # pad column written as zero


def test_conv2d_1x1_bf16x3():
    B, Cin, Cout, H, lp = 2, 64, 32, 16, 6
    P = 1 << lp
    x = _rand((B, Cin, H, P - 1), 76)
    w = _rand((Cout, Cin, 1, 1), 77, Cin ** -0.5)
    bias = _rand((Cout,), 78, 0.1)
    ref = F.conv2d(x, w, bias)
    xd = ops.guarded(B, Cin, H * P, 264, DEV)
    xd[:, :, :H * P] = torch.nan_to_num(_to_pitch(x, lp), nan=0.0).to(DEV)
    yd = torch.full((B, Cout, H * P), float("nan"), device=DEV)
    wp = packing.pack_conv2d(w)
    ops.conv2d(xd, wp.to(DEV), bias.to(DEV), yd, H, lp, 1, None, None, w3=packing.pack_x3(wp).to(DEV))
    torch.cuda.synchronize()
    assert _lib.lib().vfx_last_conv_tile() % 100 == 16
    got = _from_pitch(yd, H, lp)
    _close(got[..., : P - 1], ref, 1e-4)
    assert (got[..., P - 1] == 0).all()


def test_conv2d_1x1_bias_residual():
    B, Cin, Cout, H, lp = 2, 64, 32, 16, 6
    P = 1 << lp
    x = _rand((B, Cin, H, P - 1), 16)
    w = _rand((Cout, Cin, 1, 1), 17, Cin ** -0.5)
    bias = _rand((Cout,), 18, 0.1)
    res = _rand((B, Cout, H, P - 1), 19)
    ref = F.conv2d(x, w, bias) + res
    xd = torch.nan_to_num(_to_pitch(x, lp), nan=0.0).to(DEV)
    rd = torch.nan_to_num(_to_pitch(res, lp), nan=0.0).to(DEV)
    yd = torch.full((B, Cout, H * P), float("nan"), device=DEV)
    ops.conv2d(xd, packing.pack_conv2d(w).to(DEV), bias.to(DEV), yd, H, lp, 1, None, rd)
    torch.cuda.synchronize()
    got = _from_pitch(yd, H, lp)
    _close(got[..., : P - 1], ref, 2e-5)
    assert (got[..., P - 1] == 0).all()


@pytest.mark.parametrize("cfg", [(2, 64, 32, 8, 3), (1, 384, 384, 2, 1), (1, 128, 64, 16, 5)])
def test_convtr2d(cfg):
    B, Cin, Cout, h, lp = cfg
    Pi = 1 << lp
    x = _rand((B, Cin, h, Pi - 1), 20)
    w = _rand((Cin, Cout, 3, 3), 21, (Cin * 2.25) ** -0.5)
    scale = 0.8 + 0.4 * torch.rand(Cin, generator=torch.Generator().manual_seed(22))
    shift = _rand((Cin,), 23, 0.3)
    ref = F.conv_transpose2d(_ref_act(x, _lib.PRE_AFFINE_LRELU, 0.0, scale, shift), w, stride=2)[:, :, :-1, :]
    assert ref.shape[2:] == (2 * h, 2 * Pi - 1)
    xd = _to_pitch(x, lp).to(DEV)
    # write into the first Cout channels of a concat buffer with 2*Cout channels
    cat = torch.full((B, 2 * Cout, 2 * h * 2 * Pi), float("nan"), device=DEV)
    act = ops.Act(pre=_lib.PRE_AFFINE_LRELU, pre_slope=0.0, scale=scale.to(DEV), shift=shift.to(DEV))
    ops.convtr2d_3x3s2(xd, packing.pack_convtr2d(w).to(DEV), cat[:, :Cout], h, lp, act)
    torch.cuda.synchronize()
    got = _from_pitch(cat[:, :Cout], 2 * h, lp + 1)
    _close(got[..., : 2 * Pi - 1], ref, 2e-5)
    assert (got[..., 2 * Pi - 1] == 0).all()
    assert torch.isnan(cat[:, Cout:]).all()


def test_conv_cout1_k7_reflect_tanh():
    B, Cin, L = 2, 64, 4410
    x = _rand((B, Cin, L), 24)
    w = _rand((1, Cin, 7), 25, (Cin * 7) ** -0.5)
    bias = _rand((1,), 26, 0.1)
    ref = torch.tanh(F.conv1d(F.pad(x, (3, 3), mode="reflect"), w, bias))
    xd = _padded(x, L + 6)
    yd = torch.full((B, 1, L + 2), float("nan"), device=DEV)
    ops.conv1d_cout1(xd, packing.pack_cout1(w).to(DEV), bias.to(DEV), yd, L, 7, _lib.PAD_REFLECT, _lib.POST_TANH)
    torch.cuda.synchronize()
    _close(yd[:, :, :L], ref, 1e-5)


@pytest.mark.parametrize("L,lpad,ypad", [(4410, 4416, 4412), (4412, 4412, 4412), (1025, 1028, 1028), (9, 12, 12), (4410, 4413, 4411)])
@pytest.mark.parametrize("pad_mode", ["reflect", "zero"])
def test_conv_cout1_k7_four_outputs_per_thread(L, lpad, ypad, pad_mode):
    """Round 4: the k = 7 instance gives a thread four consecutive outputs and loads its window as three aligned 16-byte
    vectors (1.33 -> 0.72 ms at batch 32); rows whose strides are not multiples of 4 (last case) keep the one-output-per-thread
    kernel.  Both against torch on lengths that are / are not multiples of 4, a row shorter than one window, reflect and zero
    padding, with NaN behind every row (over-reads) and behind every output row (over-writes)."""
    B, Cin = 3, 64
    x = _rand((B, Cin, L), 124)
    w = _rand((1, Cin, 7), 125, (Cin * 7) ** -0.5)
    bias = _rand((1,), 126, 0.1)
    xp = F.pad(x, (3, 3), mode="reflect") if pad_mode == "reflect" else F.pad(x, (3, 3))
    ref = torch.tanh(F.conv1d(xp, w, bias))
    xd = _padded(x, lpad)
    yd = torch.full((B, 1, ypad), float("nan"), device=DEV)
    ops.conv1d_cout1(xd, packing.pack_cout1(w).to(DEV), bias.to(DEV), yd, L, 7,
                     _lib.PAD_REFLECT if pad_mode == "reflect" else _lib.PAD_ZERO, _lib.POST_TANH)
    torch.cuda.synchronize()
    _close(yd[:, :, :L], ref, 1e-5)
    assert torch.isnan(yd[:, :, L:]).all()


def test_conv_cout1_k7_ragged_rows():
    """Per-row lengths (ragged batches): every row reflects at its own end and nothing is written past it."""
    B, Cin, Lmax = 4, 64, 3000
    lens = [3000, 2999, 1537, 1026]
    x = _rand((B, Cin, Lmax), 127)
    w = _rand((1, Cin, 7), 128, (Cin * 7) ** -0.5)
    bias = _rand((1,), 129, 0.1)
    xd = ops.with_rows(_padded(x, Lmax), torch.tensor(lens, dtype=torch.int32, device=DEV))
    yd = torch.full((B, 1, Lmax), float("nan"), device=DEV)
    ops.conv1d_cout1(xd, packing.pack_cout1(w).to(DEV), bias.to(DEV), yd, Lmax, 7, _lib.PAD_REFLECT, _lib.POST_TANH)
    torch.cuda.synchronize()
    for b, n in enumerate(lens):
        ref = torch.tanh(F.conv1d(F.pad(x[b:b + 1, :, :n], (3, 3), mode="reflect"), w, bias))
        _close(yd[b:b + 1, :, :n], ref, 1e-5)
        assert torch.isnan(yd[b, :, n:]).all()


def test_conv_cout1_1x1_masked():
    B, Cin, H, lp = 2, 32, 8, 7
    P = 1 << lp
    x = _rand((B, Cin, H, P - 1), 27)
    w = _rand((1, Cin, 1, 1), 28, Cin ** -0.5)
    bias = _rand((1,), 29, 0.1)
    ref = F.conv2d(x, w, bias)
    xd = torch.nan_to_num(_to_pitch(x, lp), nan=0.0).to(DEV)
    yd = torch.full((B, 1, H * P), float("nan"), device=DEV)
    ops.conv1d_cout1(xd, packing.pack_cout1(w).to(DEV), bias.to(DEV), yd, H * P, 1, 0, 0, lp)
    torch.cuda.synchronize()
    got = _from_pitch(yd, H, lp)
    _close(got[..., : P - 1], ref, 1e-5)
    assert (got[..., P - 1] == 0).all()


def test_avgpool():
    B, Cn, H, lp = 2, 5, 8, 4
    P = 1 << lp
    x = _rand((B, Cn, H, P - 1), 30)
    ref = F.avg_pool2d(x, 2)
    xd = torch.nan_to_num(_to_pitch(x, lp), nan=0.0).to(DEV)
    yd = torch.full((B, Cn, (H // 2) * (P // 2)), float("nan"), device=DEV)
    ops.avgpool2x2(xd, yd, H, lp)
    torch.cuda.synchronize()
    got = _from_pitch(yd, H // 2, lp - 1)
    _close(got[..., : P // 2 - 1], ref, 1e-6)
    assert (got[..., P // 2 - 1] == 0).all()


@pytest.mark.parametrize("n", [1025, 15523, 44100])
def test_stft_mel(n):
    wav = _rand((2, n), 31, 0.2)
    ref = oracle.wav_to_mel(wav)[:, 0]  # (B,T,128)
    T = 1 + n // 441
    wd = torch.zeros((2, n + 5), device=DEV)
    wd[:, :n] = wav.to(DEV)
    mel = torch.full((2, T, 128), float("nan"), device=DEV)
    ops.stft_mel(wd, mel, n)
    torch.cuda.synchronize()
    got = mel.cpu()
    assert torch.isfinite(got).all()
    rel = (got - ref).norm() / ref.norm()
    assert rel < 1e-5, rel.item()
    assert (got - ref).abs().max() <= 1e-4 * ref.abs().max()


def test_device_filterbank_tables_are_bit_exact():
    """north_star: "bit-exact on mel bin indexing" -- asserted on what the DEVICE holds: the banded tables are read
    back from HBM (vfx_frontend_readback) and compared with the reference-derived golden support / hashes."""
    import hashlib
    from conftest import GOLDEN
    from voicefixer_amd import frontend_tables as ft
    ops.frontend_init()
    g = np.load(os.path.join(GOLDEN, "filterbank.npz"))
    lo, hi, off, coef = ops.frontend_readback(0)
    assert np.array_equal(lo, g["lo"]) and np.array_equal(hi, g["hi"])
    assert hashlib.sha256(ft.dense(lo, hi, off, coef).tobytes()).hexdigest() == str(g["sha256_abs"])
    assert coef.shape[0] == int((hi - lo + 1).sum()) and np.array_equal(off, np.concatenate([[0], np.cumsum(hi - lo + 1)[:-1]]))
    # the slaney table of Vocoder.oracle: equal to the product's host table, which tests/test_librosa_like.py pins
    w = torch.zeros((1, 4096), device=DEV)
    w[0, 100] = 1.0
    ops.oracle_mel(w, 4096)
    lo1, hi1, off1, coef1 = ops.frontend_readback(1)
    tlo, thi, toff, tcoef = ft.oracle_tables()
    assert np.array_equal(lo1, tlo) and np.array_equal(hi1, thi) and np.array_equal(off1, toff)
    assert np.array_equal(coef1, tcoef)


def test_gru_bidir():
    B, T, H = 3, 37, 256
    x = _rand((B, T, 512), 32)
    params = {}
    g = torch.Generator().manual_seed(33)
    for suf in ("", "_reverse"):
        params["w_ih" + suf] = (torch.rand((768, 512), generator=g) * 2 - 1) / 16
        params["w_hh" + suf] = (torch.rand((768, 256), generator=g) * 2 - 1) / 16
        params["b_ih" + suf] = (torch.rand((768,), generator=g) * 2 - 1) / 16
        params["b_hh" + suf] = (torch.rand((768,), generator=g) * 2 - 1) / 16
    outs, gis = [], []
    for suf, rev in (("", False), ("_reverse", True)):
        outs.append(oracle._gru_dir(x, params["w_ih" + suf], params["w_hh" + suf], params["b_ih" + suf],
                                    params["b_hh" + suf], rev))
        gis.append(x @ params["w_ih" + suf].t() + params["b_ih" + suf])
    ref = torch.cat(outs, -1)  # (B,T,512)
    gi = torch.cat(gis, -1).contiguous().to(DEV)  # (B,T,1536)
    whh_t = packing.pack_gru_whh(params["w_hh"], params["w_hh_reverse"], *ops.gru_layout()).to(DEV)
    bhh = torch.stack([params["b_hh"], params["b_hh_reverse"]]).to(DEV)
    out = torch.full((B, 512, 40), float("nan"), device=DEV)
    ops.gru_bidir(gi, whh_t, bhh, out, T)
    torch.cuda.synchronize()
    _close(out[:, :, :T].transpose(1, 2), ref, 1e-5)


def test_gru_bidir_two_cu():
    B, T = 5, 61
    x = _rand((B, T, 512), 60)
    g = torch.Generator().manual_seed(61)
    P = {}
    for suf in ("", "_reverse"):
        P["w_ih" + suf] = (torch.rand((768, 512), generator=g) * 2 - 1) / 16
        P["w_hh" + suf] = (torch.rand((768, 256), generator=g) * 2 - 1) / 16
        P["b_ih" + suf] = (torch.rand((768,), generator=g) * 2 - 1) / 16
        P["b_hh" + suf] = (torch.rand((768,), generator=g) * 2 - 1) / 16
    outs, gis = [], []
    for suf, rev in (("", False), ("_reverse", True)):
        outs.append(oracle._gru_dir(x, P["w_ih" + suf], P["w_hh" + suf], P["b_ih" + suf], P["b_hh" + suf], rev))
        gis.append(x @ P["w_ih" + suf].t() + P["b_ih" + suf])
    ref = torch.cat(outs, -1)
    gi = torch.cat(gis, -1).contiguous().to(DEV)
    whh_t = torch.stack([P["w_hh"].t().contiguous(), P["w_hh_reverse"].t().contiguous()]).to(DEV)
    bhh = torch.stack([P["b_hh"], P["b_hh_reverse"]]).to(DEV)
    out = torch.full((B, 512, 64), float("nan"), device=DEV)
    err = torch.zeros(1, dtype=torch.int32, device=DEV)
    keep = ops.gru_bidir2(gi, whh_t, bhh, out, T, err)
    torch.cuda.synchronize()
    assert int(err.item()) == 0, "partner workgroup never answered"
    _close(out[:, :, :T].transpose(1, 2), ref, 1e-5)
    del keep


def test_mel_to_cond():
    g = torch.Generator().manual_seed(34)
    for T in (101, 24):
        mel = 10 ** (torch.rand((2, 1, T, 128), generator=g) * 6 - 3)
        mel[0, 0, :2, :7] = 0.0
        ref = oracle.mel_to_cond(mel)
        Tc = T + T % 2 + 4
        cond = torch.full((2, 128, Tc + 3), float("nan"), device=DEV)
        ops.mel_to_cond(mel[:, 0].contiguous().to(DEV), cond, T)
        torch.cuda.synchronize()
        _close(cond[:, :, :Tc], ref, 2e-5)


def test_unet_input_output_and_post():
    B, T, Tp = 2, 36, 64
    g = torch.Generator().manual_seed(35)
    mel = 10 ** (torch.rand((B, 1, T, 128), generator=g) * 4 - 2)
    mask = torch.rand((B, 1, T, 128), generator=g)
    x = oracle.to_log(mask * mel)
    u_ref = torch.cat([oracle.to_log(mel), x], dim=1)
    mel_d = mel[:, 0].contiguous().to(DEV)
    mask_cm = torch.zeros((B, 128, 40), device=DEV)
    mask_cm[:, :, :T] = mask[:, 0].transpose(1, 2).to(DEV)
    u = torch.full((B, 4, Tp * 128), float("nan"), device=DEV)  # 2 real + 2 filler channels
    ops.unet_input(mel_d, mask_cm, u, T, Tp)
    torch.cuda.synchronize()
    got = u.cpu().reshape(B, 4, Tp, 128)
    _close(got[:, :2, :T, :127], u_ref[..., :127], 1e-6)
    assert (got[:, :2, T:, :] == 0).all() and (got[:, :2, :, 127] == 0).all() and (got[:, 2:] == 0).all()

    uo = _rand((B, 1, Tp, 128), 36)
    uo[..., 127] = 0
    logmel = torch.empty((B, T, 128), device=DEV)
    den = torch.empty((B, T, 128), device=DEV)
    ops.unet_output(uo.reshape(B, 1, Tp * 128).to(DEV), u, mel_d, mask_cm, logmel, den, T, Tp)
    torch.cuda.synchronize()
    ref_lm = uo[:, 0, :T] + x[:, 0]
    _close(logmel, ref_lm, 1e-5)
    _close(den, oracle.from_log(ref_lm), 1e-5)

    # peak rule + centre trim
    Ly, N = 441 * 40, 15523
    y = _rand((B, Ly), 37, 0.2)
    y[1] *= 8.0  # second utterance exceeds 1.0 -> normalised by its own peak only
    yd = y.to(DEV)
    out = torch.empty((B, N), device=DEV)
    ws = torch.zeros(B, dtype=torch.int32, device=DEV)
    ops.post(yd, Ly, out, N, ws)
    torch.cuda.synchronize()
    for b in range(B):
        e = y[b][None, None]
        pk = e.abs().max()
        if pk > 1:
            e = e / pk
        _close(out[b], oracle.trim_center(e, N)[0, 0], 1e-6)


def test_bad_arguments_fail_loudly():
    x = torch.zeros((1, 64, 100), device=DEV)
    w = torch.zeros((3, 64, 48), device=DEV)  # Cout = 48 not a multiple of 32
    with pytest.raises(_lib.VfxError):
        ops.conv1d(x, w, None, torch.zeros((1, 48, 100), device=DEV), 100, 3)
    with pytest.raises(_lib.VfxError):
        ops.conv1d(torch.zeros((1, 64, 100)), w, None, x, 100, 3)  # CPU tensor


@pytest.mark.parametrize("n", [22050, 132300])
def test_hf_cut_mode1_prefilter(n):
    """vfx_hf_cut_f32 vs the numpy restatement of remove_higher_frequency (librosa semantics)."""
    g = torch.Generator().manual_seed(50)
    t = np.arange(n) / 44100.0
    w0 = (0.03 * torch.randn(n, generator=g).numpy() + 0.3 * np.sin(2 * np.pi * 440 * t)).astype(np.float32)
    w1 = (0.2 * torch.randn(n, generator=g).numpy()).astype(np.float32)
    wav = torch.from_numpy(np.stack([w0, w1])).to(DEV)
    out, cut = ops.hf_cut(wav, n, 0.95)
    torch.cuda.synchronize()
    assert tuple(out.shape) == (2, 512 * (n // 512))
    for b, w in enumerate((w0, w1)):
        # the cumulative energy of these inputs crosses the 95 % threshold with a margin of >= 1e-4 of the total
        # on both sides (float32 rounding: ~1e-7), so the cut-off bin is not a matter of summation order: the
        # device must find the oracle's bin and the waveforms are ALWAYS compared
        ref, rcut = oracle.remove_higher_frequency(w)
        assert int(cut[b]) == rcut
        assert np.abs(out[b].cpu().numpy() - ref).max() < 2e-5


def test_conv1d_randomised_geometry_sweep():
    """Seeded sweep over odd geometries (lengths around tile multiples, tiny lengths, every dilation of the
    ResStack, guarded and unguarded inputs, residual in place, both arithmetics): exercises the exact-width
    tiles, the per-slot-count instances, the accumulator-initialised bias/residual and the fp32 fallback of
    bf16x3 launches against the torch operator."""
    rng = np.random.default_rng(2024)
    dils = [1, 3, 9, 27, 81, 243, 729, 2187]
    lens = [5, 63, 124, 126, 127, 128, 129, 250, 252, 253, 254, 255, 257, 1000, 2521, 4099]
    for case in range(36):
        B = int(rng.integers(1, 4))
        Cin = int(rng.choice([32, 64, 96, 128, 256]))
        Cout = int(rng.choice([32, 64, 128, 256]))
        L = int(rng.choice(lens))
        dil = int(rng.choice(dils))
        k = int(rng.choice([1, 3, 3, 3]))
        guarded = bool(rng.integers(0, 2))
        use_res = bool(rng.integers(0, 2))
        inplace = use_res and Cin == Cout and bool(rng.integers(0, 2))
        x3 = guarded and Cin % 32 == 0 and bool(rng.integers(0, 2))
        pre = int(rng.choice([_lib.PRE_NONE, _lib.PRE_LRELU]))
        post = int(rng.choice([_lib.POST_NONE, _lib.POST_LRELU]))
        x = _rand((B, Cin, L), 500 + case)
        w = _rand((Cout, Cin, k), 600 + case, (Cin * k) ** -0.5)
        bias = _rand((Cout,), 700 + case, 0.1)
        res = _rand((B, Cout, L), 800 + case) if use_res else None
        p = (k - 1) // 2 * dil
        ref = F.conv1d(_ref_act(x, pre, 0.01), w, bias, dilation=dil, padding=p)
        if use_res:
            ref = ref + res
        ref = _ref_post(ref, post, 0.2)
        lp = (L + 3) // 4 * 4
        if guarded:
            xd = ops.guarded(B, Cin, L, p + 264, DEV)
            xd._vfx_base.fill_(float("nan"))
            xd[:, :, :L] = x.to(DEV)
        else:
            xd = _padded(x, lp)
        if inplace:
            yd = _padded(res, lp)
            rd = yd
        else:
            yd = torch.full((B, Cout, lp), float("nan"), device=DEV)
            rd = _padded(res, lp) if use_res else None
        act = ops.Act(pre=pre, pre_slope=0.01, post=post, post_slope=0.2)
        wp = packing.pack_conv1d(w)
        w3 = packing.pack_x3(wp).to(DEV) if x3 else None
        ops.conv1d(xd, wp.to(DEV), bias.to(DEV), yd, L, k, dil, 0, act, rd, w3=w3)
        torch.cuda.synchronize()
        tol = 1e-4 if x3 else 2e-5
        got = yd[:, :, :L].cpu()
        err = (got - ref).abs().max().item()
        assert torch.isfinite(got).all() and err <= tol * max(1.0, ref.abs().max().item()), \
            "case %d: B=%d Cin=%d Cout=%d L=%d k=%d dil=%d guarded=%s res=%s inplace=%s x3=%s err=%g" % (
                case, B, Cin, Cout, L, k, dil, guarded, use_res, inplace, x3, err)
        if not inplace:
            assert torch.isnan(yd[:, :, L:]).all(), "case %d wrote past L" % case


def test_hf_cut_against_the_reference_executed_fixture():
    """vfx_hf_cut_f32 vs VoiceFixer.remove_higher_frequency EXECUTED from the reference's code (base.py:87-104; fixture
    tests/golden/mode1_speech_ref.npz, oracle/make_golden.py through ref_shim) on 0.75 s of the reference's utterance."""
    g = np.load(os.path.join(GOLDEN, "mode1_speech_ref.npz"))
    wav = torch.from_numpy(g["wav"])[None].to(DEV)
    out, cut = ops.hf_cut(wav, wav.shape[1], 0.95)
    torch.cuda.synchronize()
    assert tuple(out.shape) == (1,) + g["hf_cut"].shape
    assert np.abs(out[0].cpu().numpy() - g["hf_cut"]).max() < 2e-5


def _wino_adversarial_cases():
    import importlib.util
    spec = importlib.util.spec_from_file_location("winograd_error", os.path.join(os.path.dirname(GOLDEN), "..", "tools", "winograd_error.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_winograd_kernel_on_adversarial_operand_statistics():
    """convwg4_kernel (its dilation-1 instance) against a FLOAT64 convolution on the operand statistics of tools/winograd_error.py
    --sweep (profiles/r03_winograd_error_sweep.txt): log-normal weight-norm row gains (sigma 1, 2), Student-t(2) entries,
    smooth and second-difference filters, per-channel log-normal activation scales, a DC offset 30x the spread, slowly
    varying inputs, 1e4 x outliers.  Bounds: the fp32 rounding the CPU restatement of the same arithmetic shows, x3 --
    rms error / rms y per case, the worst output row's, and (where no outlier dominates a 6-tap window: F(4,3) mixes the
    inputs of a quad, so its error is relative to the window's largest operand) max |error| / sum |w||x|."""
    mod = _wino_adversarial_cases()
    c, n, B = 256, 4096, 8
    gen = torch.Generator().manual_seed(0)
    # (rms_rel, worst_row_rms_rel, max_over_mag) bounds = 3x the sweep's F(4,3) column
    for label, w, x in mod.cases(c, n, gen):
        w, x = w.float(), x.float()
        xin = x[:, : n + 2]                                    # y_dev[q] = sum_k w_k xin[q + k - 1]  (zero padding, dilation 1)
        L = xin.shape[1]
        ref = F.conv1d(xin.double()[None], w.double(), padding=1)[0]
        mag = F.conv1d(xin.double().abs()[None], w.double().abs(), padding=1)[0]
        lp = (L + 67) // 4 * 4
        xd = torch.zeros((B, c, lp), device=DEV)
        xd[:, :, :L] = xin.to(DEV)[None]
        yd = torch.empty((B, c, lp), device=DEV)
        wp = packing.pack_conv1d(w)
        ops.conv1d(xd, wp.to(DEV), torch.zeros(c, device=DEV), yd, L, 3, 1, 0, None, None, wg4=packing.pack_wino4(wp).to(DEV))
        torch.cuda.synchronize()
        assert _lib.lib().vfx_last_conv_tile() % 100 in (80, 81, 82), "launch did not run on the Winograd kernel"
        y = yd[3, :, :L].cpu().double()
        assert torch.equal(yd[0, :, :L], yd[7, :, :L])         # identical rows: identical bits
        err = y - ref
        rms_rel = float(err.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())
        row_rel = float((err.pow(2).mean(1).sqrt() / ref.pow(2).mean(1).sqrt()).max())
        over_mag = float((err.abs() / mag.clamp_min(1e-300)).max())
        ill = "second-difference filters on slowly varying" in label     # |y| << sum |w||x|: every algorithm loses digits here
        assert rms_rel < (3e-5 if ill else 3e-6), (label, rms_rel)
        assert row_rel < (4e-5 if ill else 2e-5), (label, row_rel)
        if "outliers" not in label:
            assert over_mag < 3e-5, (label, over_mag)


@pytest.mark.parametrize("cfg", [(2, 256, 147294, 1), (2, 256, 147294, 27), (2, 128, 441882, 1)])
def test_conv1d_winograd4_long_rows(cfg):
    """Rows as long as a 30 s segment produces them (147 294 / 441 882 positions: > 64 tiles, so the XCD-aware tile order
    and the channel-block fold are active; lengths that are not multiples of 4, so the dilation-1 instance meets its
    straddling quad), batch 2, with and without the in-place residual.  The short cases of test_conv1d_winograd4 run
    in linear tile order; a round-3 epilogue variant that was wrong only on long rows of a batch > 1 passed all of them."""
    B, C, L, dil = cfg
    x = _rand((B, C, L), 801)
    w = _rand((C, C, 3), 802, (C * 3) ** -0.5)
    bias = _rand((C,), 803, 0.1)
    res = _rand((B, C, L), 804)
    ref0 = F.conv1d(F.leaky_relu(x, 0.01), w, bias, dilation=dil, padding=dil)
    lp = (L + 3) // 4 * 4
    xd = torch.full((B, C, lp + 64), float("nan"), device=DEV)
    xd[:, :, :L] = x.to(DEV)
    wp = packing.pack_conv1d(w)
    wg4 = packing.pack_wino4(wp).to(DEV)
    act = ops.Act(pre=_lib.PRE_LRELU, pre_slope=0.01)
    for use_res in (False, True):
        yd = torch.full((B, C, lp + 64), float("nan"), device=DEV)
        rd = None
        if use_res:
            yd[:, :, :L] = res.to(DEV)
            rd = yd[:, :, :lp]                      # in place
        ops.conv1d(xd[:, :, :lp], wp.to(DEV), bias.to(DEV), yd[:, :, :lp], L, 3, dil, 0, act, rd, wg4=wg4)
        torch.cuda.synchronize()
        assert _lib.lib().vfx_last_conv_tile() % 100 in (80, 81, 82)
        _close(yd[:, :, :L], ref0 + res if use_res else ref0, 2e-5)
        assert torch.isnan(yd[:, :, L:]).all()      # nothing written past the rows
