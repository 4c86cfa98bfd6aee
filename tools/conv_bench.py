#!/usr/bin/env python
"""Micro-benchmark of single conv-family launches (development tool, GPU only).

    python tools/conv_bench.py res4 res1 ...      # named shapes, B from --batch
"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from voicefixer_amd import ops, packing, _lib  # noqa: E402

SHAPES = {
    # name: (kind, Cin, Cout, L, k, dil)
    "res4_d1": ("c1", 64, 64, 443646, 3, 1),
    "res4_d27": ("c1", 64, 64, 443646, 3, 27),
    "res4_d2187": ("c1", 64, 64, 443646, 3, 2187),
    "res3_d1": ("c1", 128, 128, 147882, 3, 1),
    "res3_d9": ("c1", 128, 128, 147882, 3, 9),
    "res2_d1": ("c1", 256, 256, 49294, 3, 1),
    "res2_d243": ("c1", 256, 256, 49294, 3, 243),
    "res2_d27": ("c1", 256, 256, 49294, 3, 27),
    "res1_d1": ("c1", 512, 512, 7042, 3, 1),
    "res1_d729": ("c1", 512, 512, 7042, 3, 729),
    "pre_k7": ("c1r", 512, 1024, 1006, 7, 1),
    "cond": ("c1", 512, 512, 1006, 3, 1),
    "up1": ("t1", 1024, 512, 1006, 7, 0),
    "up2": ("t1", 512, 256, 7042, 7, 0),
    "up3": ("t1", 256, 128, 49294, 3, 0),
    "up4": ("t1", 128, 64, 147882, 3, 0),
    "unet1": ("c2", 32, 32, 1024, 7, 0),     # H=1024, lp=7
    "unet2": ("c2", 64, 64, 512, 6, 0),
    "unet3": ("c2", 128, 128, 256, 5, 0),
    "unet4": ("c2", 256, 256, 128, 4, 0),
    "unet5": ("c2", 384, 384, 64, 3, 0),
    "unet5q": ("c2", 96, 384, 64, 3, 0),      # unet5 with a quarter of K (split-K what-if, run at 4x batch)
    "unet4q": ("c2", 64, 256, 128, 4, 0),
    "unet6": ("c2", 384, 384, 32, 2, 0),
    "unet6q": ("c2", 96, 384, 32, 2, 0),
    "gru_proj": ("c1", 512, 1536, 1001, 1, 1),
    "post_k7": ("co1", 64, 1, 443646, 7, 1),   # ReflectionPad1d(3) + Conv1d(64, 1, 7) + Tanh: HBM-bound (reports GB/s in the TFLOP/s column / 1000)
    # what-if shapes (not in the model): long K at a fixed output tile count, to separate per-tile overhead from the loop
    "k3072": ("c1", 1024, 256, 49294, 3, 1),
    "k6144": ("c1", 2048, 256, 49294, 3, 1),
    "k384": ("c1", 128, 256, 49294, 3, 1),
    "k192": ("c1", 64, 256, 49294, 3, 1),
    # fused ResStack layers (vfx_resblock_f32): (kind, C, C, L, 3, dil); FLOPs = both convolutions
    "rb4_d1": ("rb", 64, 64, 443646, 3, 1),
    "rb4_d9": ("rb", 64, 64, 443646, 3, 9),
    "rb4_d27": ("rb", 64, 64, 443646, 3, 27),
    "rb4_d81": ("rb", 64, 64, 443646, 3, 81),
    "rb4_d243": ("rb", 64, 64, 443646, 3, 243),
    "rb4_d2187": ("rb", 64, 64, 443646, 3, 2187),
    "rb3_d1": ("rb", 128, 128, 147882, 3, 1),
    "rb3_d27": ("rb", 128, 128, 147882, 3, 27),
    "rb3_d81": ("rb", 128, 128, 147882, 3, 81),
    "rb3_d243": ("rb", 128, 128, 147882, 3, 243),
    "rb3_d2187": ("rb", 128, 128, 147882, 3, 2187),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("names", nargs="*", default=["res4_d1", "res3_d1", "res2_d1", "res1_d1"])
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--x3", action="store_true", help="VFX_MATH_BF16X3 (guarded inputs, 1-D shapes only)")
    ap.add_argument("--wd", action="store_true", help="offer the convw_kernel weight layout (vfx_act.w_direct)")
    ap.add_argument("--wg", action="store_true", help="fused layers (rb*): the second half as Winograd F(2,3); "
                    "TFLOP/s stay the DIRECT algorithm's 2*MACs / time")
    ap.add_argument("--stack", action="store_true", help="1-D k = 3 shapes as the ResStack issues them: dilation 1 = the SECOND convolution of a "
                    "layer (no pre-activation, no post-activation, residual updated in place), otherwise the first (lrelu before and after)")
    ap.add_argument("--wg4", action="store_true", help="offer the Winograd F(4,3) weights (vfx_act.w_wino4; k = 3 1-D shapes; fused C = 64 layers: their second half)")
    ap.add_argument("--clocks", action="store_true", help="sample shader clock / socket power (bench.ClockSampler) while the timed launches run")
    args = ap.parse_args()
    dev = "cuda"
    B = args.batch
    for name in args.names:
        kind, cin, cout, L, k, dil = SHAPES[name]
        g = torch.Generator().manual_seed(1)
        act = ops.Act(pre=_lib.PRE_LRELU, pre_slope=0.01, post=_lib.POST_LRELU, post_slope=0.01)
        if kind in ("c1", "c1r"):
            Lp = (L + 3) // 4 * 4
            x = ops.guarded(B, cin, L, dil + 264, dev)
            x.normal_()
            y = torch.empty((B, cout, Lp), device=dev)
            wp = packing.pack_conv1d(torch.randn((cout, cin, k), generator=g) * (cin * k) ** -0.5)
            w = wp.to(dev)
            w3 = packing.pack_x3(wp).to(dev) if args.x3 else None
            bias = torch.zeros(cout, device=dev)
            pad = 1 if kind == "c1r" else 0
            wd = packing.pack_direct(wp).to(dev) if args.wd else None
            wg4 = packing.pack_wino4(wp).to(dev) if (args.wg4 and k == 3) else None
            if args.stack and k == 3 and dil == 1 and kind == "c1":
                a2 = ops.Act()
                fn = lambda: ops.conv1d(x, w, bias, y, L, k, dil, pad, a2, res=y, w3=w3, wd=wd, wg4=wg4)
            else:
                fn = lambda: ops.conv1d(x, w, bias, y, L, k, dil, pad, act, w3=w3, wd=wd, wg4=wg4)
            macs = B * L * cin * cout * k
        elif kind == "co1":
            Lp = (L + 3) // 4 * 4
            x = ops.guarded(B, cin, L, 264, dev)
            x.normal_()
            y = torch.empty((B, 1, Lp), device=dev)
            wc = torch.randn((cin, k), generator=g).to(dev).contiguous()
            bias = torch.zeros(1, device=dev)
            fn = lambda: ops.conv1d_cout1(x, wc, bias, y, L, k, _lib.PAD_REFLECT, _lib.POST_TANH)
            macs = B * L * (cin + 1) * 4 / 2 * 1e3     # bytes / 2 * 1000: the printed "TFLOP/s" figure is GB/s / 1000 ... x 1000 = GB/s
        elif kind == "rb":
            x = ops.guarded(B, cin, L, 2187 + 264, dev)
            x.normal_()
            y = ops.guarded(B, cin, L, 2187 + 264, dev)
            w1d = packing.pack_direct(packing.pack_conv1d(torch.randn((cout, cin, 3), generator=g) * (cin * 3) ** -0.5)).to(dev)
            w2d = packing.pack_direct(packing.pack_conv1d(torch.randn((cout, cin, 3), generator=g) * (cin * 3) ** -0.5)).to(dev)
            bias = torch.zeros(cout, device=dev)
            w2p = packing.pack_conv1d(torch.randn((cout, cin, 3), generator=g) * (cin * 3) ** -0.5)
            w2g = packing.pack_wino(w2p).to(dev) if args.wg else None
            w2g4 = packing.pack_wino4(w2p).to(dev) if (args.wg4 and cin == 64) else None
            w1g4 = packing.pack_wino4(w2p).to(dev) if (args.wg4 and cin == 64) else None     # (any weights do for timing)
            fn = lambda: ops.resblock(x, y, w1d, bias, w2d, bias, L, dil, w2g=w2g, w2g4=w2g4, w1g4=w1g4)
            macs = 2 * B * L * cin * cout * 3
        elif kind == "t1":
            s = k
            Lp = (L + 3) // 4 * 4
            x = ops.guarded(B, cin, L, 264, dev)
            x.normal_()
            y = torch.empty((B, cout, (L * s + 3) // 4 * 4), device=dev)
            wp = packing.pack_convtr1d(torch.randn((cin, cout, 2 * s), generator=g) * (2 * cin) ** -0.5)
            w = wp.to(dev)
            w3 = packing.pack_x3(wp).to(dev) if args.x3 else None
            bias = torch.zeros(cout, device=dev)
            wd = packing.pack_direct(wp).to(dev) if args.wd else None
            wt = packing.pack_wino32_tr(wp, s).to(dev) if args.wg4 else None      # (--wg4: offer the Winograd F(3,2) planes, convtw_kernel)
            fn = lambda: ops.convtr1d(x, w, bias, y, L, s, w3=w3, wd=wd, wg4=wt)
            macs = B * L * cin * cout * 2 * s
        else:
            H, lp = L, k
            P = 1 << lp
            x = torch.randn((B, cin, H * P), device=dev)
            y = torch.empty((B, cout, H * P), device=dev)
            x = ops.guarded(B, cin, H * P, P + 1 + 264, dev)
            x.normal_()
            wp = packing.pack_conv2d(torch.randn((cout, cin, 3, 3), generator=g) * (cin * 9) ** -0.5)
            w = wp.to(dev)
            wd = packing.pack_direct(wp).to(dev) if args.wd else None
            sc = torch.ones(cin, device=dev)
            sh = torch.zeros(cin, device=dev)
            a2 = ops.Act(pre=_lib.PRE_AFFINE_LRELU, pre_slope=0.01, scale=sc, shift=sh)
            wg42 = packing.pack_wino4_2d(wp).to(dev) if args.wg4 else None
            fn = lambda: ops.conv2d(x, w, None, y, H, lp, 3, a2, wd=wd, wg4=wg42)
            macs = B * H * (P - 1) * cin * cout * 9
        fn()
        torch.cuda.synchronize()
        tile = _lib.lib().vfx_last_conv_tile()
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        clk = ""
        if args.clocks:
            import bench
            sampler = bench.ClockSampler(torch.cuda.current_device(), period=0.02)
            with sampler:
                t0 = time.perf_counter()
                e0.record()
                for _ in range(args.iters):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
            sm = sampler.summary(t0 + 0.4 * (t1 - t0), t1)
            if sm.get("sclk_mhz"):
                clk = "  sclk %4.0f MHz  %4.0f W (n=%d)" % (sm["sclk_mhz"]["mean"], (sm["socket_power_w"] or {"mean": 0})["mean"], sm["sclk_mhz"]["n"])
        else:
            e0.record()
            for _ in range(args.iters):
                fn()
            e1.record()
            torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / args.iters
        print("%-12s B=%d tile=%d  %8.3f ms  %7.2f TFLOP/s%s" % (name, B, tile, ms, 2 * macs / ms / 1e9, clk), flush=True)


if __name__ == "__main__":
    main()
