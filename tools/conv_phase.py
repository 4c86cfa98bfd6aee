#!/usr/bin/env python
"""Per-phase cycle breakdown of conv_taps_kernel (needs a -DVFX_ABL=8 build passed via VFX_DEV=1 VFX_LIB)."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from voicefixer_amd import ops, packing, _lib
sys.argv = [sys.argv[0]] + sys.argv[1:]
import importlib.util
spec = importlib.util.spec_from_file_location("cb", os.path.join(os.path.dirname(__file__), "conv_bench.py"))
cb = importlib.util.module_from_spec(spec); spec.loader.exec_module(cb)
h = _lib.lib()
h.vfx_debug_read.restype = C.c_int
buf = (C.c_ulonglong * 10)()
B = 8
X3 = "--x3" in sys.argv
for name in [a for a in sys.argv[1:] if not a.startswith("--")]:
    kind, cin, cout, L, k, dil = cb.SHAPES[name]
    g = torch.Generator().manual_seed(1)
    Lp = (L + 3) // 4 * 4
    x = ops.guarded(B, cin, L, dil + 264, "cuda"); x.normal_(); y = torch.empty((B, cout, Lp), device="cuda")
    wp = packing.pack_conv1d(torch.randn((cout, cin, k), generator=g) * (cin * k) ** -0.5)
    w = wp.cuda()
    w3 = packing.pack_x3(wp).cuda() if X3 else None
    bias = torch.zeros(cout, device="cuda")
    act = ops.Act(pre=_lib.PRE_LRELU, pre_slope=0.01, post=_lib.POST_LRELU, post_slope=0.01)
    ops.conv1d(x, w, bias, y, L, k, dil, 0, act, w3=w3); torch.cuda.synchronize()
    h.vfx_debug_read(buf, 1)
    ops.conv1d(x, w, bias, y, L, k, dil, 0, act, w3=w3); torch.cuda.synchronize()
    h.vfx_debug_read(buf, 1)
    n, steps = buf[6], buf[7]
    per = lambda v: v / max(steps, 1)
    print("%-10s WGs=%d chunks/WG=%.0f | per chunk: write %.0f  load %.0f  mfma %.0f  barrier %.0f | per WG: prologue %.0f  loop %.0f  epilogue %.0f (s_memtime units)" %
          (name, n, steps / max(n, 1), per(buf[0]), per(buf[1]), per(buf[2]), per(buf[3]), buf[8] / max(n, 1), buf[4] / max(n, 1), buf[5] / max(n, 1)))
