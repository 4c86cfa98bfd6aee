"""Host-side weight packing for libvfx_hip (runs once at model load, CPU tensors).

Every conv-family kernel wants its weights as ``[slab][CinPad][Cout]`` float32 with Cout
contiguous (the MFMA A-operand is read K-major), one slab per kernel tap, CinPad = Cin
rounded up to 8 with zero fill (include/vfx_hip.h).
"""
import torch


def _pad_cin(w):  # w: [slab, Cin, Cout]
    cin = w.shape[1]
    cpad = (cin + 7) // 8 * 8
    if cpad != cin:
        w = torch.cat([w, w.new_zeros(w.shape[0], cpad - cin, w.shape[2])], dim=1)
    return w.contiguous().float()


def pack_conv1d(w):
    """torch Conv1d weight (Cout, Cin, k) -> [k][CinPad][Cout]."""
    return _pad_cin(w.permute(2, 1, 0))


def pack_convtr1d(w):
    """torch ConvTranspose1d weight (Cin, Cout, k) -> [k][CinPad][Cout]."""
    return _pad_cin(w.permute(2, 0, 1))


def pack_conv2d(w):
    """torch Conv2d weight (Cout, Cin, kh, kw) -> [kh*kw][CinPad][Cout], slab = ky*kw+kx."""
    co, ci, kh, kw = w.shape
    return _pad_cin(w.permute(2, 3, 1, 0).reshape(kh * kw, ci, co))


def pack_convtr2d(w):
    """torch ConvTranspose2d weight (Cin, Cout, 3, 3) -> [9][CinPad][Cout]."""
    ci, co, kh, kw = w.shape
    return _pad_cin(w.permute(2, 3, 0, 1).reshape(kh * kw, ci, co))


def pack_linear(w):
    """torch Linear weight (out, in) -> [1][inPad][out]."""
    return _pad_cin(w.t()[None])


def pack_direct(w_packed):
    """[slab][CinPad][Cout] -> [slab][CinPad/8][Cout][8] for convw_kernel (vfx_act.w_direct, vfx_resblock_f32): within
    each group of 8 input channels the order is (0,2,4,6,1,3,5,7), so that the 16-byte vector at [..][m][4*hi:4*hi+4]
    is the MFMA A operand of row m for the four consecutive k-steps k = 2*kk + hi of v_mfma_f32_32x32x2_f32."""
    s, cp, co = w_packed.shape
    assert cp % 8 == 0
    w = w_packed.reshape(s, cp // 8, 8, co)[:, :, [0, 2, 4, 6, 1, 3, 5, 7], :]
    return w.permute(0, 1, 3, 2).contiguous().float()


def pack_wino(w_packed):
    """[3][CinPad][Cout] (tap slabs of a k = 3 Conv1d) -> [4][CinPad/8][Cout][8]: the Winograd F(2,3) weight
    transform U0 = w0, U1 = (w0 + w1 + w2)/2, U2 = (w0 - w1 + w2)/2, U3 = w2 (sums in float64, rounded once) in the
    A-operand layout of pack_direct: the second half of the fused ResStack layer (vfx_resblock_f32: w2_wino)."""
    assert w_packed.shape[0] == 3
    g = w_packed.double()
    u = torch.stack([g[0], (g[0] + g[1] + g[2]) * 0.5, (g[0] - g[1] + g[2]) * 0.5, g[2]])
    return pack_direct(u.float())


def pack_wino4(w_packed):
    """[3][CinPad][Cout] -> [6][CinPad/8][Cout][8]: the Winograd F(4,3) weight transform U = G w,
    G = [[1/4,0,0],[-1/6,-1/6,-1/6],[-1/6,1/6,-1/6],[1/24,1/12,1/6],[1/24,-1/12,1/6],[0,0,1]] (float64, rounded once),
    in the A-operand layout of pack_direct, for convwg4_kernel (vfx_act.w_wino4)."""
    assert w_packed.shape[0] == 3
    g0, g1, g2 = w_packed.double()
    u = torch.stack([g0 / 4, -(g0 + g1 + g2) / 6, -(g0 - g1 + g2) / 6, g0 / 24 + g1 / 12 + g2 / 6,
                     g0 / 24 - g1 / 12 + g2 / 6, g2])
    return pack_direct(u.float())


def pack_wino32_tr(w_packed, stride):
    """[2 s][CinPad][Cout] (tap slabs of a polyphase ConvTranspose1d, kernel 2 s, stride s: out[q s + r - pad] = w[r + s] x[q - 1] +
    w[r] x[q]) -> [4 s][CinPad/8][Cout][8], slab 4 r + plane: for every output phase r the Winograd F(3,2) weight transform along the
    INPUT axis, U0 = g0, U1 = (g0 + g1) / 2, U2 = (g0 - g1) / 2, U3 = g1 with g0 = w[r + s], g1 = w[r] (float64, rounded once), in the
    A-operand layout of pack_direct: convtw_kernel (vfx_act.w_wino4 of vfx_convtr1d_f32)."""
    assert w_packed.shape[0] == 2 * stride
    g = w_packed.double()
    planes = []
    for r in range(stride):
        g0, g1 = g[r + stride], g[r]
        planes += [g0, (g0 + g1) * 0.5, (g0 - g1) * 0.5, g1]
    return pack_direct(torch.stack(planes).float())


def _f43(g0, g1, g2):
    return [g0 / 4, -(g0 + g1 + g2) / 6, -(g0 - g1 + g2) / 6, g0 / 24 + g1 / 12 + g2 / 6, g0 / 24 - g1 / 12 + g2 / 6, g2]


def pack_wino4_2d(w_packed):
    """[9][CinPad][Cout] (3x3 slabs ky*3 + kx) -> [18][CinPad/8][Cout][8], slab kx*6 + plane: the Winograd F(4,3) weight
    transform along the kernel's ROW axis for every kernel column (convwg4_kernel with NKX = 3, vfx_act.w_wino4 of
    vfx_conv2d_f32)."""
    assert w_packed.shape[0] == 9
    g = w_packed.double().reshape(3, 3, w_packed.shape[1], w_packed.shape[2])   # [ky][kx]
    planes = []
    for kx in range(3):
        planes += _f43(g[0, kx], g[1, kx], g[2, kx])
    return pack_direct(torch.stack(planes).float())


def pack_cout1(w):
    """Conv weight (1, Cin, k[, 1]) -> [Cin][k]."""
    return w.reshape(w.shape[1], -1).contiguous().float()


def pack_gru_whh(w_hh_fwd, w_hh_bwd, kreg, klds, kstr):
    """W_hh (768, 256) per direction -> the packed recurrent-weight buffer of vfx_gru_bidir_f32.

    Per direction, with k = half*128 + ... the reduction index and n = gate*256 + unit the row:
      R: [2 halves][kreg][768]            k = half*128 + kk            (lives in VGPRs)
      L: [2 halves][klds][768]            k = half*128 + kreg + kk     (lives in LDS)
      S: [2 halves][kstr/4][3][256][4]    k = half*128 + kreg + klds + 4*q + e  (streamed, float4)
    """
    assert kreg + klds + kstr == 128 and kstr % 4 == 0
    out = []
    for w in (w_hh_fwd, w_hh_bwd):
        wt = w.float().t().contiguous()  # (256 k, 768 n)
        wt = wt.reshape(2, 128, 768)
        r = wt[:, :kreg].reshape(-1)
        l = wt[:, kreg:kreg + klds].reshape(-1)
        st = wt[:, kreg + klds:].reshape(2, kstr // 4, 4, 3, 256)  # half, q, e, gate, unit
        st = st.permute(0, 1, 3, 4, 2).reshape(-1)                   # half, q, gate, unit, e
        out.append(torch.cat([r, l, st]))
    return torch.cat(out).contiguous()


def pack_x3(w_packed):
    """[slab][CinPad][Cout] float32 -> the bf16 (hi, lo) planes of VFX_MATH_BF16X3 (include/vfx_hip.h):
    ``[slab][Cin/16][plane][k-half][Cout][8]`` bfloat16 with hi = bf16(w) and lo = bf16(w - hi), both
    round-to-nearest-even (what v_cvt_pk_bf16_f32 does to the activations on the device).
    Cin is zero-padded to a multiple of 16; returns None when Cout is not a multiple of 32."""
    s, cin, cout = w_packed.shape
    if cout % 32 != 0:
        return None
    c16 = (cin + 15) // 16 * 16
    w = w_packed.float()
    if c16 != cin:
        w = torch.cat([w, w.new_zeros(s, c16 - cin, cout)], dim=1)
    hi = w.to(torch.bfloat16)
    lo = (w - hi.float()).to(torch.bfloat16)
    planes = torch.stack([hi, lo], dim=1)                       # [slab][plane][Cin][Cout]
    planes = planes.reshape(s, 2, c16 // 16, 2, 8, cout)        # [slab][plane][chunk][k-half][k][Cout]
    return planes.permute(0, 2, 1, 3, 5, 4).contiguous()        # [slab][chunk][plane][k-half][Cout][k]
