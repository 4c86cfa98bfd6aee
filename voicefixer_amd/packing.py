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


def pack_cout1(w):
    """Conv weight (1, Cin, k[, 1]) -> [Cin][k]."""
    return w.reshape(w.shape[1], -1).contiguous().float()
