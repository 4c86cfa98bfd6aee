#!/usr/bin/env python
"""Per-CU timeline of convwg4_kernel workgroups (development; needs the -DVFX_ABL=8 build):
    make -C voicefixer_amd/csrc abl && VFX_DEV=1 VFX_LIB=voicefixer_amd/libvfx_hip_abl.so python tools/wg4_timeline.py res3_d1 res2_d27 ...
Wave 0 of every workgroup records {HW_ID, XCC_ID, s_memtime at entry / K-loop start / K-loop end / exit}.  The tool groups the
records by compute unit (XCC, SE, SH, CU from HW_ID) and answers the question the round-4 review asked: are the workgroups that
share a CU IN PHASE (prologues and epilogues at the same time, nothing hides them) or spread?  Per launch it prints
  * the share of the CU's busy time with 0 / 1 / 2 / 3+ resident workgroups inside their K loop,
  * the share of all prologue + epilogue time that overlaps a co-resident workgroup's K loop (hidden) -- in-phase lock gives ~0,
    a uniform spread gives the K-loop share of a lifetime,
  * the start-offset histogram of co-resident pairs in units of a workgroup lifetime (0 = in phase, 0.5 = anti-phase).
--stack issues the launches as the ResStack does (dilation 1 = second convolution: residual in place, no activations)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from voicefixer_amd import ops, packing, _lib  # noqa: E402
import importlib.util  # noqa: E402

spec = importlib.util.spec_from_file_location("cb", os.path.join(os.path.dirname(__file__), "conv_bench.py"))
cb = importlib.util.module_from_spec(spec)
spec.loader.exec_module(cb)


def analyse(rec, name, ms):
    hw = rec[:, 0]
    xcc = (hw >> 32) & 0xF
    lo = hw & 0xFFFFFFFF
    cu = (lo >> 8) & 0xF
    sh = (lo >> 12) & 0x1
    se = (lo >> 13) & 0x7
    slot = lo & 0xF
    key = (xcc << 12) | (se << 8) | (sh << 4) | cu
    t0, t1, t2, t3 = (rec[:, i].astype(np.int64) for i in (2, 3, 4, 5))
    life = np.median(t3 - t0)
    kfrac = np.median((t2 - t1) / np.maximum(t3 - t0, 1))
    tot_pe = hid_pe = 0.0
    span_sum = res_sum = 0.0                       # per CU: first entry .. last exit, and the sum of the workgroups' resident times
    gaps = []                                      # exit of a workgroup -> entry of the next one that takes a free slot of the CU
    occ = np.zeros(5)
    offs = []
    ncu = 0
    for k in np.unique(key):
        m = key == k
        if m.sum() < 8:
            continue
        ncu += 1
        a0, a1, a2, a3 = t0[m], t1[m], t2[m], t3[m]
        o = np.argsort(a0)
        a0, a1, a2, a3 = a0[o], a1[o], a2[o], a3[o]
        span_sum += float(a3.max() - a0.min())
        res_sum += float((a3 - a0).sum())
        # slot hand-over: every entry after the first residency round takes the slot of the earliest not yet re-used exit
        ends = np.sort(a3)
        nslot = int(((a0 < a3.min())).sum())      # workgroups resident before the first exit = slots of this CU
        for i in range(nslot, len(a0)):
            gaps.append(float(a0[i] - ends[i - nslot]))
        # event sweep: +1 at K-loop start, -1 at K-loop end -> time with n workgroups in their K loop
        ev = np.concatenate([np.stack([a1, np.ones_like(a1)], 1), np.stack([a2, -np.ones_like(a2)], 1)])
        ev = ev[np.argsort(ev[:, 0], kind="stable")]
        n = 0
        for i in range(len(ev) - 1):
            n += ev[i, 1]
            occ[min(int(n), 4)] += ev[i + 1, 0] - ev[i, 0]
        # prologue / epilogue intervals of each workgroup against the K loops of the others on this CU
        for i in range(len(a0)):
            for (s, e) in ((a0[i], a1[i]), (a2[i], a3[i])):
                tot_pe += e - s
                lo_ = np.maximum(a1, s)
                hi_ = np.minimum(a2, e)
                ov = np.clip(hi_ - lo_, 0, None)
                ov[i] = 0
                # union is approximated by the largest single overlap plus the rest clipped to the interval length
                hid_pe += min(float(e - s), float(ov.sum()))
            # start offset to the workgroup resident when this one started
            live = np.where((a0 < a0[i]) & (a3 > a0[i]))[0]
            for j in live:
                offs.append(((a0[i] - a0[j]) / max(a3[j] - a0[j], 1)) % 1.0)
    occ = occ / max(occ.sum(), 1)
    h, _ = np.histogram(np.array(offs), bins=10, range=(0, 1))
    h = h / max(h.sum(), 1)
    g = np.array(gaps) if gaps else np.zeros(1)
    tick_ghz = span_sum / max(ncu, 1) / (ms * 1e6)
    print("%-10s slots per CU %.2f busy (sum of residencies / span / slots); ticks per ns %.3f; exit -> next entry on the freed slot: median %.0f, "
          "mean %.0f, 90th pct %.0f ticks (%.2f us median)" % (name, res_sum / max(span_sum, 1), tick_ghz, np.median(g), g.mean(), np.percentile(g, 90),
                                                             np.median(g) / max(tick_ghz, 1e-9) / 1e3))
    print("%-10s %.3f ms  records %d on %d CUs  lifetime %.0f ticks, K loop %.0f %% of it | CU time with n workgroups in their K loop: "
          "0: %.1f %%  1: %.1f %%  2: %.1f %%  3+: %.1f %% | prologue+epilogue time under a co-resident K loop: %.0f %% | "
          "start offsets of co-resident workgroups (tenths of a lifetime): %s | wave slots seen: %s"
          % (name, ms, len(rec), ncu, life, 100 * kfrac, 100 * occ[0], 100 * occ[1], 100 * occ[2], 100 * (occ[3] + occ[4]),
             100 * hid_pe / max(tot_pe, 1), " ".join("%.2f" % v for v in h), sorted(set(int(v) for v in slot))))


def main():
    h = _lib.lib()
    h.vfx_debug_trace.restype = C.c_int
    h.vfx_debug_trace.argtypes = [C.c_void_p, C.c_uint]
    h.vfx_debug_trace_count.restype = C.c_uint
    stack = "--stack" in sys.argv
    B = 32
    cap = 1 << 17
    buf = torch.zeros((cap, 6), dtype=torch.int64, device="cuda")
    for name in [a for a in sys.argv[1:] if not a.startswith("--")]:
        kind, cin, cout, L, k, dil = cb.SHAPES[name]
        g = torch.Generator().manual_seed(1)
        Lp = (L + 3) // 4 * 4
        x = torch.randn((B, cin, Lp), device="cuda")
        y = torch.randn((B, cout, Lp), device="cuda")
        w = torch.randn((cout, cin, k), generator=g) * (cin * k) ** -0.5
        wp = packing.pack_conv1d(w)
        wd = packing.pack_direct(wp).cuda()
        wg4 = packing.pack_wino4(wp).cuda()
        bias = torch.zeros(cout, device="cuda")
        second = stack and dil == 1
        act = ops.Act() if second else ops.Act(pre=_lib.PRE_LRELU, pre_slope=0.01, post=_lib.POST_LRELU, post_slope=0.01)
        res = y if (second or (not stack and dil == 1)) else None
        wpc = wp.cuda()
        for _ in range(2):
            ops.conv1d(x, wpc, bias, y, L, k, dil, 0, act, res=res, wd=wd, wg4=wg4)
        torch.cuda.synchronize()
        buf.zero_()
        torch.cuda.synchronize()
        h.vfx_debug_trace(C.c_void_p(buf.data_ptr()), cap)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.conv1d(x, wpc, bias, y, L, k, dil, 0, act, res=res, wd=wd, wg4=wg4)
        e1.record()
        torch.cuda.synchronize()
        h.vfx_debug_trace(None, 0)
        rec = buf.cpu().numpy().astype(np.uint64)
        rec = rec[rec[:, 1] != 0]                   # (records are indexed by the linear workgroup id; column 1 = id + 1)
        analyse(rec, name + ("(stack)" if stack else ""), e0.elapsed_time(e1))


if __name__ == "__main__":
    main()
