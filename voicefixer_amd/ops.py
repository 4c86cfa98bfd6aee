"""Thin Python launchers for the libvfx_hip C ABI.

torch is used here only as the owner of device memory and of the current HIP stream:
every function takes CUDA(=HIP) tensors, fills ``vfx_tensor`` descriptors from their
strides and calls the C entry point on ``torch.cuda.current_stream()``.  No torch
arithmetic happens on this path.
"""
import ctypes as C

import torch

from . import _lib
from ._lib import (vfx_tensor, vfx_act, check, PRE_NONE, PRE_LRELU, PRE_AFFINE_LRELU, POST_NONE,
                   POST_LRELU, POST_ELU, POST_TANH, POST_SIGMOID, POST_LRELU_SNAKE, PAD_ZERO,
                   PAD_REFLECT, MATH_F32, MATH_BF16X3)


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _lib.VfxError("libvfx_hip operates on device tensors only (got a CPU tensor)")


def tdesc(t):
    """vfx_tensor for a (B, C, L) tensor view (any strides; element units).  Views created by
    ``guarded()`` carry a ``_vfx_guard`` attribute: readable slack on both sides of every row."""
    assert t.dim() == 3 and t.dtype == torch.float32
    rows = getattr(t, "_vfx_rows", None)   # ragged batches: device int32 (B,) valid length per batch item
    return vfx_tensor(t.data_ptr(), t.stride(0), t.stride(1), t.stride(2), getattr(t, "_vfx_guard", 0),
                      rows.data_ptr() if rows is not None else None)


def with_rows(t, rows):
    """Tag a (B, C, L) view with per-batch-item valid lengths (device int32 (B,)); None removes the tag."""
    if rows is not None:
        assert rows.dtype == torch.int32 and rows.is_cuda and rows.numel() == t.shape[0]
    t._vfx_rows = rows
    return t


def guarded(B, Cn, L, guard, device):
    """(B, C, Lp) view (Lp = L rounded up to 4) with ``guard`` readable elements before and after every
    row; the conv kernels mask whatever they read there, so the slack is never initialised."""
    guard = (guard + 3) // 4 * 4
    Lp = (L + 3) // 4 * 4
    buf = torch.empty((B, Cn, guard + Lp + guard), device=device)
    v = buf[:, :, guard:guard + Lp]
    v._vfx_guard = guard
    v._vfx_base = buf  # keep the allocation alive
    return v


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


# Optional per-launch profiling of the MFMA conv family (bench.py roofline bookkeeping):
# when PROFILE is a list, every conv-family launch is bracketed by HIP events on the launch
# stream and (tile id, algorithmic MACs, start event, end event) is appended.  EVENT_POOL, if set,
# is a list of pre-created timing events that are consumed instead of creating new ones (creating
# an event costs ~10 us of host time, which a launch-bound batch-1 run would otherwise show).
PROFILE = None
EVENT_POOL = None


def _event():
    if EVENT_POOL:
        return EVENT_POOL.pop()
    return torch.cuda.Event(enable_timing=True)


def _prof_begin():
    if PROFILE is None:
        return None
    e = _event()
    e.record()
    return e


def _prof_end(e0, macs):
    if e0 is None:
        return
    e1 = _event()
    e1.record()
    PROFILE.append((_lib.lib().vfx_last_conv_tile(), macs, e0, e1))


class Act:
    """Bundles the fused pre/post activation of a conv launch (keeps tensors alive)."""

    def __init__(self, pre=PRE_NONE, pre_slope=0.0, scale=None, shift=None, post=POST_NONE,
                 post_slope=0.0):
        self.scale, self.shift = scale, shift
        self.c = vfx_act(pre, float(pre_slope), scale.data_ptr() if scale is not None else None,
                         shift.data_ptr() if shift is not None else None, post, float(post_slope))


_NOACT = None
_ACTVARIANTS = {}


def _act(a, w3=None, wd=None, wg4=None):
    """vfx_act* for a launch.  ``w3`` (packing.pack_x3 planes on the device) opts the launch into VFX_MATH_BF16X3 (the
    library falls back to fp32 for geometries its bf16x3 kernel does not cover); ``wd`` (packing.pack_direct on the
    device) offers the fp32 launch the convw_kernel weight layout (vfx_act.w_direct; the library decides); ``wg4``
    (packing.pack_wino4 / pack_wino4_2d on the device) offers a k = 3 / 3x3 launch the Winograd F(4,3) kernels
    (vfx_act.w_wino4)."""
    global _NOACT
    if a is None:
        if _NOACT is None:
            _NOACT = Act()
        a = _NOACT
    if w3 is None and wd is None and wg4 is None:
        return C.byref(a.c)
    key = (id(a), w3.data_ptr() if w3 is not None else 0, wd.data_ptr() if wd is not None else 0,
           wg4.data_ptr() if wg4 is not None else 0)
    ent = _ACTVARIANTS.get(key)
    if ent is None:
        c = vfx_act(a.c.pre_act, a.c.pre_slope, a.c.pre_scale, a.c.pre_shift, a.c.post_act, a.c.post_slope,
                    MATH_BF16X3 if w3 is not None else MATH_F32, w3.data_ptr() if w3 is not None else None,
                    wd.data_ptr() if wd is not None else None, wg4.data_ptr() if wg4 is not None else None)
        ent = _ACTVARIANTS[key] = (c, a, w3, wd, wg4)  # keep the owners alive with the struct
    return C.byref(ent[0])


def conv1d(x, w, bias, y, L, k, dilation=1, pad_mode=PAD_ZERO, act=None, res=None, cin=None, w3=None, wd=None, wg4=None):
    """x (B,Cin,>=L) -> y (B,Cout,>=L) views; w packed [k][CinPad][Cout].  Optional weight layouts the library may use
    instead (it decides per launch, see include/vfx_hip.h: vfx_act): w3 = bf16x3 planes (opts the launch into that
    arithmetic), wd = packing.pack_direct, wg4 = the Winograd F(4,3) transform (k = 3 only)."""
    _need_cuda(x, w, y, res, bias)
    B = x.shape[0]
    cin = x.shape[1] if cin is None else cin
    cout = w.shape[2]
    xd, yd = tdesc(x), tdesc(y)
    rd = tdesc(res) if res is not None else None
    e0 = _prof_begin()
    rc = _lib.lib().vfx_conv1d_f32(C.byref(xd), _ptr(w), _ptr(bias), C.byref(rd) if rd is not None else None,
                                   C.byref(yd), B, cin, cout, L, k, dilation, pad_mode, _act(act, w3, wd, wg4), _stream())
    check(rc, "vfx_conv1d_f32")
    _prof_end(e0, B * L * cin * cout * k)


def resblock(x, y, w1d, b1, w2d, b2, L, dilation, slope=0.01, post=POST_NONE, post_slope=0.0, w2g=None, w2g4=None, w1g4=None):
    """One fused ResStack layer (vfx_resblock_f32): x (B,C,>=L) guarded view -> y (B,C,>=L), y must not alias x.
    ``w2g`` (packing.pack_wino of the second convolution, optional): its dilation-1 half runs as Winograd F(2,3);
    ``w2g4`` (packing.pack_wino4, optional, C = 64): as F(4,3) when the rows of x and y are 16-byte aligned;
    ``w1g4`` (packing.pack_wino4 of the FIRST convolution, optional, C = 64): with w2g4 and a dilation <= 27 BOTH halves run as
    F(4,3) (resblk4_kernel)."""
    _need_cuda(x, y, w1d, b1, w2d, b2, w2g, w2g4, w1g4)
    B, Cn = x.shape[0], x.shape[1]
    xd, yd = tdesc(x), tdesc(y)
    e0 = _prof_begin()
    wts = _lib.vfx_resblock_w(_ptr(w1d), _ptr(b1), _ptr(w2d), _ptr(b2), _ptr(w2g), _ptr(w2g4), _ptr(w1g4))
    rc = _lib.lib().vfx_resblock_f32(C.byref(xd), C.byref(yd), C.byref(wts), B, Cn, L, dilation, float(slope), post,
                                     float(post_slope), _stream())
    check(rc, "vfx_resblock_f32")
    _prof_end(e0, 2 * B * L * Cn * Cn * 3)


def convtr1d(x, w, bias, y, Lin, stride, act=None, w3=None, wd=None, wg4=None):
    """``wg4`` (packing.pack_wino32_tr on the device) offers the launch the Winograd F(3,2) kernel (convtw_kernel)."""
    _need_cuda(x, w, y, bias)
    B, cin = x.shape[0], x.shape[1]
    cout = w.shape[2]
    xd, yd = tdesc(x), tdesc(y)
    e0 = _prof_begin()
    rc = _lib.lib().vfx_convtr1d_f32(C.byref(xd), _ptr(w), _ptr(bias), C.byref(yd), B, cin, cout, Lin, stride,
                                     _act(act, w3, wd, wg4), _stream())
    check(rc, "vfx_convtr1d_f32")
    _prof_end(e0, B * Lin * cin * cout * 2 * stride)


def conv2d(x, w, bias, y, H, pitch_log2, ksize, act=None, res=None, cin=None, w3=None, wd=None, wg4=None):
    """x (B,Cin,H*P) pitch map -> y (B,Cout,H*P)."""
    _need_cuda(x, w, y, res, bias)
    B = x.shape[0]
    cin = x.shape[1] if cin is None else cin
    cout = w.shape[2]
    xd, yd = tdesc(x), tdesc(y)
    rd = tdesc(res) if res is not None else None
    e0 = _prof_begin()
    rc = _lib.lib().vfx_conv2d_f32(C.byref(xd), _ptr(w), _ptr(bias), C.byref(rd) if rd is not None else None,
                                   C.byref(yd), B, cin, cout, H, pitch_log2, ksize, _act(act, w3, wd, wg4), _stream())
    check(rc, "vfx_conv2d_f32")
    _prof_end(e0, B * H * ((1 << pitch_log2) - 1) * cin * cout * ksize * ksize)


def convtr2d_3x3s2(x, w, y, h, in_pitch_log2, act=None, w3=None):
    _need_cuda(x, w, y)
    B, cin = x.shape[0], x.shape[1]
    cout = w.shape[2]
    xd, yd = tdesc(x), tdesc(y)
    e0 = _prof_begin()
    rc = _lib.lib().vfx_convtr2d_3x3s2_f32(C.byref(xd), _ptr(w), C.byref(yd), B, cin, cout, h, in_pitch_log2,
                                           _act(act, w3), _stream())
    check(rc, "vfx_convtr2d_3x3s2_f32")
    _prof_end(e0, B * h * ((1 << in_pitch_log2) - 1) * cin * cout * 9)


def conv1d_cout1(x, w, bias, y, L, k, pad_mode=PAD_ZERO, post=POST_NONE, out_mask_log2=0):
    _need_cuda(x, w, y, bias)
    B, cin = x.shape[0], x.shape[1]
    xd, yd = tdesc(x), tdesc(y)
    rc = _lib.lib().vfx_conv1d_cout1_f32(C.byref(xd), _ptr(w), _ptr(bias), C.byref(yd), B, cin, L, k, pad_mode,
                                         post, out_mask_log2, _stream())
    check(rc, "vfx_conv1d_cout1_f32")


def avgpool2x2(x, y, H, pitch_log2):
    _need_cuda(x, y)
    B, Cn = x.shape[0], x.shape[1]
    xd, yd = tdesc(x), tdesc(y)
    check(_lib.lib().vfx_avgpool2x2_f32(C.byref(xd), C.byref(yd), B, Cn, H, pitch_log2, _stream()),
          "vfx_avgpool2x2_f32")


_frontend_ready = set()   # device indices whose tables are uploaded (the library keeps one copy per device)


def frontend_init():
    """Upload window / twiddles / banded HTK filterbank (voicefixer/tools/mel_scale.py:147-238
    restated in float32 torch with the same op order, so the support set is bit-identical)."""
    dev = torch.cuda.current_device()
    if dev in _frontend_ready:
        return
    from .frontend_tables import tables
    win, tw, lo, hi, off, coef = tables()
    check(_lib.lib().vfx_frontend_init(win.ctypes.data, tw.ctypes.data, lo.ctypes.data, hi.ctypes.data,
                                       off.ctypes.data, coef.ctypes.data, int(coef.shape[0])),
          "vfx_frontend_init")
    _frontend_ready.add(dev)


def frontend_readback(which=0):
    """Test hook: the banded filterbank as it sits in device memory (0: HTK / restorer, 1: slaney / Vocoder.oracle)."""
    import numpy as np
    lo, hi, off = (np.zeros(128, np.int32) for _ in range(3))
    coef = np.zeros(8192, np.float32)
    nnz = C.c_int(0)
    check(_lib.lib().vfx_frontend_readback(which, lo.ctypes.data, hi.ctypes.data, off.ctypes.data, coef.ctypes.data,
                                           int(coef.shape[0]), C.byref(nnz)), "vfx_frontend_readback")
    return lo, hi, off, coef[:nnz.value].copy()


def stft_mel(wav, mel, N):
    """wav (B, >=N) device float32 -> mel (B, T, 128)."""
    _need_cuda(wav, mel)
    frontend_init()
    assert wav.stride(1) == 1 and mel.is_contiguous()
    e0 = _prof_begin()
    check(_lib.lib().vfx_stft_mel_f32(_ptr(wav), wav.stride(0), wav.shape[0], N, _ptr(mel), _stream()),
          "vfx_stft_mel_f32")
    if e0 is not None:  # bench.py: HBM roofline of the front-end, algorithmic bytes 4*N + 512*T per utterance
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        PROFILE.append((-1, wav.shape[0] * (4 * N + 512 * (1 + N // 441)), e0, e1))


def stft_mel_rows(wav, mel, n_rows, T):
    """wav (B, >= max n) device float32, n_rows device int32 (B,) with 1 + n // 441 == T for every row -> mel (B, T, 128)."""
    _need_cuda(wav, mel, n_rows)
    frontend_init()
    assert wav.stride(1) == 1 and mel.is_contiguous() and n_rows.dtype == torch.int32
    check(_lib.lib().vfx_stft_mel_rows_f32(_ptr(wav), wav.stride(0), wav.shape[0], _ptr(n_rows), T, _ptr(mel), _stream()),
          "vfx_stft_mel_rows_f32")


_oracle_ready = set()


def oracle_mel(wav, N):
    """Vocoder.oracle front-end on the device: wav (B, >=N) -> slaney mel (B, T, 128) of wav/max|wav|."""
    _need_cuda(wav)
    frontend_init()
    if torch.cuda.current_device() not in _oracle_ready:
        from .frontend_tables import oracle_tables
        lo, hi, off, coef = oracle_tables()
        check(_lib.lib().vfx_frontend_init_oracle(lo.ctypes.data, hi.ctypes.data, off.ctypes.data, coef.ctypes.data,
                                                  int(coef.shape[0])), "vfx_frontend_init_oracle")
        _oracle_ready.add(torch.cuda.current_device())
    B = wav.shape[0]
    T = 1 + N // 441
    peak = torch.empty((B,), dtype=torch.int32, device=wav.device)
    check(_lib.lib().vfx_peak_f32(_ptr(wav), wav.stride(0), N, B, _ptr(peak), _stream()), "vfx_peak_f32")
    mel = torch.empty((B, T, 128), device=wav.device)
    check(_lib.lib().vfx_stft_mel_oracle_f32(_ptr(wav), wav.stride(0), B, N, _ptr(peak), _ptr(mel), _stream()),
          "vfx_stft_mel_oracle_f32")
    return mel, T


def mel_to_cond_plain(mel, cond, T):
    """dB / normalise / clip / tail-pad WITHOUT the mel-weight division (Vocoder.oracle path)."""
    _need_cuda(mel, cond)
    cd = tdesc(cond)
    check(_lib.lib().vfx_mel_to_cond_ex_f32(_ptr(mel), C.byref(cd), mel.shape[0], T, 0, _stream()),
          "vfx_mel_to_cond_ex_f32")


def hf_cut(wav, N, ratio=0.95):
    """mode-1 pre-filter (remove_higher_frequency): wav (B, >=N) device -> ((B, 512*(N//512)), cut-off bins)."""
    _need_cuda(wav)
    frontend_init()
    B = wav.shape[0]
    nbytes = _lib.lib().vfx_hf_workspace_bytes(B, N)
    ws = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=wav.device)
    out = torch.empty((B, 512 * (N // 512)), device=wav.device)
    cut = torch.empty((B,), dtype=torch.int32, device=wav.device)
    check(_lib.lib().vfx_hf_cut_f32(_ptr(wav), wav.stride(0), B, N, _ptr(out), out.stride(0), float(ratio), _ptr(ws),
                                    nbytes, _ptr(cut), _stream()), "vfx_hf_cut_f32")
    return out, cut


def tm_to_cm(src, dst, T, Cn):
    """src (B,T,C) contiguous -> dst (B,C,>=T) view."""
    _need_cuda(src, dst)
    assert src.is_contiguous() and dst.stride(2) == 1
    check(_lib.lib().vfx_tm_to_cm_f32(_ptr(src), _ptr(dst), src.shape[0], T, Cn, dst.stride(0), dst.stride(1),
                                      _stream()), "vfx_tm_to_cm_f32")


def unet_input(mel, mask, unet_in, T, Tp):
    """unet_in: (B, nch >= 2, >= Tp*128) view; channels >= 2 are written as zero."""
    _need_cuda(mel, mask, unet_in)
    md, ud = tdesc(mask), tdesc(unet_in)
    check(_lib.lib().vfx_unet_input_f32(_ptr(mel), C.byref(md), C.byref(ud), unet_in.shape[1], mel.shape[0], T, Tp,
                                        _stream()), "vfx_unet_input_f32")


def unet_output(unet_out, unet_in, mel, mask, logmel, denoised, T, Tp):
    _need_cuda(unet_out, unet_in, mel, mask, logmel, denoised)
    md, od, ud = tdesc(mask), tdesc(unet_out), tdesc(unet_in)
    check(_lib.lib().vfx_unet_output_f32(C.byref(od), C.byref(ud), _ptr(mel), C.byref(md), _ptr(logmel),
                                         _ptr(denoised), mel.shape[0], T, Tp, _stream()), "vfx_unet_output_f32")


def gru_layout():
    a, b, c = C.c_int(), C.c_int(), C.c_int()
    _lib.lib().vfx_gru_layout(C.byref(a), C.byref(b), C.byref(c))
    return a.value, b.value, c.value


def gru_bidir(gi, whh_t, bhh, out, T):
    """whh_t: packed by packing.pack_gru_whh(..., *gru_layout())."""
    _need_cuda(gi, whh_t, bhh, out)
    od = tdesc(out)
    check(_lib.lib().vfx_gru_bidir_f32(_ptr(gi), _ptr(whh_t), _ptr(bhh), C.byref(od), gi.shape[0], T, _stream()),
          "vfx_gru_bidir_f32")


GRU2_MAX_B = 60  # C ABI limit: 4 workgroups per utterance, one per CU, all resident (engine.Pipeline.set_streams sizes launches per stream count)


def gru_bidir2(gi, whh_t, bhh, out, T, err_flag):
    """Two-CU-per-sequence GRU (vfx_gru_bidir2_f32).  whh_t: plain (2,256,768); err_flag: device int32[1]."""
    _need_cuda(gi, whh_t, bhh, out, err_flag)
    B = gi.shape[0]
    assert B <= GRU2_MAX_B
    nbytes = B * 2 * 2 * 2 * 384 * 8
    mbox = torch.empty((nbytes // 4,), dtype=torch.int32, device=gi.device)
    od = tdesc(out)
    check(_lib.lib().vfx_gru_bidir2_f32(_ptr(gi), _ptr(whh_t), _ptr(bhh), C.byref(od), B, T, _ptr(mbox), nbytes,
                                        _ptr(err_flag), _stream()), "vfx_gru_bidir2_f32")
    return mbox  # keep alive until the stream has consumed it (caller holds the reference)


def mel_to_cond(mel, cond, T, t_rows=None):
    """mel (B, T, 128) -> cond (B, 128, >= T'); t_rows (device int32 or None): frames of every row (ragged batches)."""
    _need_cuda(mel, cond, t_rows)
    assert mel.is_contiguous()
    cd = tdesc(cond)
    check(_lib.lib().vfx_mel_to_cond_rows_f32(_ptr(mel), C.byref(cd), mel.shape[0], T, _ptr(t_rows), 1, _stream()),
          "vfx_mel_to_cond_rows_f32")


def post_rows(y, Ly, out, n_rows, n_max, peak_ws, ly_rows=None):
    """y (B, >=Ly) -> out (B, >= n_max): per-utterance peak rule + centre trim to n_rows[b] samples (device int32);
    ly_rows (device int32 or None): vocoder samples of every row."""
    _need_cuda(y, out, peak_ws, n_rows, ly_rows)
    check(_lib.lib().vfx_post_rows_f32(_ptr(y), y.stride(0), Ly, _ptr(ly_rows), _ptr(out), out.stride(0), _ptr(n_rows),
                                       n_max, y.shape[0], _ptr(peak_ws), _stream()), "vfx_post_rows_f32")


def post(y, Ly, out, N, peak_ws):
    """y (B, >=Ly) -> out (B, N): per-utterance peak rule + centre trim."""
    _need_cuda(y, out, peak_ws)
    check(_lib.lib().vfx_post_f32(_ptr(y), y.stride(0), Ly, _ptr(out), out.stride(0), N, y.shape[0],
                                  _ptr(peak_ws), _stream()), "vfx_post_f32")
