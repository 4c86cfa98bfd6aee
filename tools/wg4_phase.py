#!/usr/bin/env python
"""Per-phase breakdown of a workgroup's life in convwg4_kernel (development; needs a -DVFX_ABL=8 build passed via VFX_DEV=1 VFX_LIB):
    make -C voicefixer_amd/csrc abl && VFX_DEV=1 VFX_LIB=voicefixer_amd/libvfx_hip_abl.so python tools/wg4_phase.py res2_d27 res1_d1 ...
Wave 0 of every workgroup adds its s_memtime deltas (prologue up to the first barrier, K loop, of which at barriers, epilogue);
the tool prints the per-workgroup means in microseconds (ticks calibrated by the launch's own wall time) next to the pipe time the
workgroup's MFMAs need (768 MFMAs x 64 cycles for Cin = 256 ...)."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from voicefixer_amd import ops, packing, _lib
import importlib.util
spec = importlib.util.spec_from_file_location("cb", os.path.join(os.path.dirname(__file__), "conv_bench.py"))
cb = importlib.util.module_from_spec(spec); spec.loader.exec_module(cb)
h = _lib.lib()
h.vfx_debug_read.restype = C.c_int
buf = (C.c_ulonglong * 10)()
bx = (C.c_ulonglong * 6)()
B = 32
for name in [a for a in sys.argv[1:] if not a.startswith("--")]:
    kind, cin, cout, L, k, dil = cb.SHAPES[name]
    g = torch.Generator().manual_seed(1)
    Lp = (L + 3) // 4 * 4
    x = torch.randn((B, cin, Lp), device="cuda"); y = torch.empty((B, cout, Lp), device="cuda")
    w = torch.randn((cout, cin, k), generator=g) * (cin * k) ** -0.5
    wp = packing.pack_conv1d(w)
    wd = packing.pack_direct(wp).cuda()
    wg4 = packing.pack_wino4(wp).cuda()
    bias = torch.zeros(cout, device="cuda")
    act = ops.Act(pre=_lib.PRE_LRELU, pre_slope=0.01)
    res = y if dil == 1 else None
    for rep in range(3):
        ops.conv1d(x, wp.cuda(), bias, y, L, k, dil, 0, act, res=res, wd=wd, wg4=wg4); torch.cuda.synchronize()
        h.vfx_debug_read(buf, 1); h.vfx_debug_read_x(bx, 1)
        # reset leaves [0] = 0: atomicMin needs a large start value -> write it through a second read-modify (tool-side: ignore [0] of rep 0)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); ops.conv1d(x, wp.cuda(), bias, y, L, k, dil, 0, act, res=res, wd=wd, wg4=wg4); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    h.vfx_debug_read(buf, 1); h.vfx_debug_read_x(bx, 1)
    n, chunks = max(buf[6], 1), buf[7]
    mfma_cycles = (chunks / n) * 48 * 64
    print("%-10s %.3f ms  tile=%d WGs=%d chunks/WG=%.0f | per WG (s_memtime ticks): prologue %.0f [issue %.0f = set-up %.0f + residual loads %.0f + tap/A loads %.0f, x0 in LDS %.0f, start values %.0f, barrier %.0f]  loop %.0f (at barriers %.0f)  epilogue %.0f  total %.0f | "
          "MFMA pipe cycles needed per wave %.0f" % (name, ms, h.vfx_last_conv_tile(), n, chunks / n, buf[8] / n, buf[0] / n, bx[0] / n, bx[1] / n, bx[2] / n, buf[1] / n, buf[2] / n, buf[9] / n, buf[4] / n, buf[3] / n, buf[5] / n,
                                                     (buf[8] + buf[4] + buf[5]) / n, mfma_cycles))
