"""ctypes binding of libvfx_hip.so (the C ABI declared in include/vfx_hip.h).

The product path has NO CPU fallback: if the shared library is missing or a call fails,
an exception is raised (``VfxError``).  ``build()`` compiles the library in-tree with hipcc
for gfx950 (cross-compiles without a GPU).
"""
import ctypes as C
import os
import subprocess

from ._dev import dev_env

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = dev_env("VFX_LIB", os.path.join(_HERE, "libvfx_hip.so"))  # VFX_LIB: development override (with VFX_DEV=1 only)
CSRC = os.path.join(_HERE, "csrc")


class VfxError(RuntimeError):
    pass


class vfx_tensor(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("bstride", C.c_int64), ("cstride", C.c_int64),
                ("lstride", C.c_int64), ("guard", C.c_int64), ("rows", C.c_void_p)]


class vfx_act(C.Structure):
    _fields_ = [("pre_act", C.c_int), ("pre_slope", C.c_float), ("pre_scale", C.c_void_p),
                ("pre_shift", C.c_void_p), ("post_act", C.c_int), ("post_slope", C.c_float),
                ("math", C.c_int), ("w_x3", C.c_void_p), ("w_direct", C.c_void_p), ("w_wino4", C.c_void_p)]


class vfx_resblock_w(C.Structure):
    _fields_ = [("w1_direct", C.c_void_p), ("bias1", C.c_void_p), ("w2_direct", C.c_void_p), ("bias2", C.c_void_p),
                ("w2_wino", C.c_void_p), ("w2_wino4", C.c_void_p), ("w1_wino4", C.c_void_p)]


PRE_NONE, PRE_LRELU, PRE_AFFINE_LRELU = 0, 1, 2
POST_NONE, POST_LRELU, POST_ELU, POST_TANH, POST_SIGMOID, POST_LRELU_SNAKE = 0, 1, 2, 3, 4, 5
PAD_ZERO, PAD_REFLECT = 0, 1
MATH_F32, MATH_BF16X3 = 0, 1

_T = C.POINTER(vfx_tensor)
_A = C.POINTER(vfx_act)
_P = C.c_void_p
_I = C.c_int

# name -> (restype, argtypes); must list every symbol declared in include/vfx_hip.h
SIGNATURES = {
    "vfx_version": (_I, []),
    "vfx_build_id": (C.c_char_p, []),
    "vfx_launch_count": (C.c_uint64, []),
    "vfx_last_conv_tile": (_I, []),
    "vfx_conv1d_f32": (_I, [_T, _P, _P, _T, _T, _I, _I, _I, _I, _I, _I, _I, _A, _P]),
    "vfx_resblock_f32": (_I, [_T, _T, C.POINTER(vfx_resblock_w), _I, _I, _I, _I, C.c_float, _I, C.c_float, _P]),
    "vfx_convtr1d_f32": (_I, [_T, _P, _P, _T, _I, _I, _I, _I, _I, _A, _P]),
    "vfx_conv2d_f32": (_I, [_T, _P, _P, _T, _T, _I, _I, _I, _I, _I, _I, _A, _P]),
    "vfx_convtr2d_3x3s2_f32": (_I, [_T, _P, _T, _I, _I, _I, _I, _I, _A, _P]),
    "vfx_conv1d_cout1_f32": (_I, [_T, _P, _P, _T, _I, _I, _I, _I, _I, _I, _I, _P]),
    "vfx_avgpool2x2_f32": (_I, [_T, _T, _I, _I, _I, _I, _P]),
    "vfx_frontend_init": (_I, [_P, _P, _P, _P, _P, _P, _I]),
    "vfx_frontend_readback": (_I, [_I, _P, _P, _P, _P, _I, C.POINTER(_I)]),
    "vfx_stft_mel_f32": (_I, [_P, C.c_int64, _I, _I, _P, _P]),
    "vfx_stft_mel_rows_f32": (_I, [_P, C.c_int64, _I, _P, _I, _P, _P]),
    "vfx_post_rows_f32": (_I, [_P, C.c_int64, _I, _P, _P, C.c_int64, _P, _I, _I, _P, _P]),
    "vfx_mel_to_cond_rows_f32": (_I, [_P, _T, _I, _I, _P, _I, _P]),
    "vfx_frontend_init_oracle": (_I, [_P, _P, _P, _P, _I]),
    "vfx_peak_f32": (_I, [_P, C.c_int64, _I, _I, _P, _P]),
    "vfx_stft_mel_oracle_f32": (_I, [_P, C.c_int64, _I, _I, _P, _P, _P]),
    "vfx_mel_to_cond_ex_f32": (_I, [_P, _T, _I, _I, _I, _P]),
    "vfx_hf_workspace_bytes": (C.c_size_t, [_I, _I]),
    "vfx_hf_cut_f32": (_I, [_P, C.c_int64, _I, _I, _P, C.c_int64, C.c_float, _P, C.c_size_t, _P, _P]),
    "vfx_tm_to_cm_f32": (_I, [_P, _P, _I, _I, _I, C.c_int64, C.c_int64, _P]),
    "vfx_unet_input_f32": (_I, [_P, _T, _T, _I, _I, _I, _I, _P]),
    "vfx_unet_output_f32": (_I, [_T, _T, _P, _T, _P, _P, _I, _I, _I, _P]),
    "vfx_gru_bidir_f32": (_I, [_P, _P, _P, _T, _I, _I, _P]),
    "vfx_gru_bidir2_f32": (_I, [_P, _P, _P, _T, _I, _I, _P, C.c_size_t, _P, _P]),
    "vfx_gru_layout": (None, [C.POINTER(_I), C.POINTER(_I), C.POINTER(_I)]),
    "vfx_mel_to_cond_f32": (_I, [_P, _T, _I, _I, _P]),
    "vfx_post_f32": (_I, [_P, C.c_int64, _I, _P, C.c_int64, _I, _I, _P, _P]),
}

_lib = None


def build(verbose=False):
    """Compile libvfx_hip.so in-tree (``make -C voicefixer_amd/csrc``)."""
    r = subprocess.run(["make", "-C", CSRC, "-j8"], capture_output=True, text=True)
    if verbose or r.returncode != 0:
        print(r.stdout[-4000:])
        print(r.stderr[-4000:])
    if r.returncode != 0:
        raise VfxError("building libvfx_hip.so failed (hipcc --offload-arch=gfx950)")
    return LIB_PATH


def lib():
    """Load (once) and return the ctypes handle; raises VfxError if the .so is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise VfxError(
            "libvfx_hip.so not found at %s -- run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(there is no CPU fallback on the product path)" % LIB_PATH)
    h = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(h, name)  # AttributeError here == header/library mismatch
        fn.restype = res
        fn.argtypes = args
    _lib = h
    return h


def check(rc, what):
    if rc != 0:
        raise VfxError("%s failed with code %d" % (what, rc))
