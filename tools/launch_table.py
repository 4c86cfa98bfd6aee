#!/usr/bin/env python
"""Every conv-family launch of one batch through the path: operator, shape, the kernel family that ran it, milliseconds
(development; GPU).   python tools/launch_table.py [--batch 32] [--seconds 10] [--taps-only]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from voicefixer_amd import engine, ops, weights, _lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--seconds", type=float, default=10.0)
    ap.add_argument("--taps-only", action="store_true", help="only the launches that ran on the first-generation conv_taps_kernel")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    n = int(round(args.seconds * 44100))
    pipe = engine.Pipeline(weights.seeded_vocoder_state(1234), weights.seeded_restorer_state(4321), dev)
    wav = bench.synth_batch(args.batch, n, 1000, dev)
    pipe.restore(wav, n)
    torch.cuda.synchronize()
    notes = []
    names = ["conv1d", "resblock", "convtr1d", "conv2d", "convtr2d_3x3s2"]
    orig = {k: getattr(ops, k) for k in names}

    def wrap(name, fn):
        def inner(*a, **kw):
            before = len(ops.PROFILE)
            out = fn(*a, **kw)
            x, w = a[0], a[1]
            desc = {"conv1d": lambda: "Cin %d Cout %d L %d k %d d %d%s" % (x.shape[1] if kw.get("cin") is None else kw["cin"], w.shape[2], a[4], a[5],
                                                                          kw.get("dilation", a[6] if len(a) > 6 else 1), " res" if kw.get("res") is not None else ""),
                    "resblock": lambda: "C %d L %d d %d" % (x.shape[1], a[6], a[7]),
                    "convtr1d": lambda: "Cin %d Cout %d Lin %d s %d" % (x.shape[1], w.shape[2], a[4], a[5]),
                    "conv2d": lambda: "Cin %d Cout %d H %d P %d k %d%s" % (x.shape[1] if kw.get("cin") is None else kw["cin"], w.shape[2], a[4], 1 << a[5], a[6],
                                                                         " res" if kw.get("res") is not None else ""),
                    "convtr2d_3x3s2": lambda: "Cin %d Cout %d h %d P %d" % (x.shape[1], w.shape[2], a[3], 1 << a[4])}[name]()
            for _ in range(len(ops.PROFILE) - before):
                notes.append((name, desc))
            return out
        return inner

    for k in names:
        setattr(ops, k, wrap(k, orig[k]))
    ops.PROFILE = []
    pipe.restore(wav, n)
    torch.cuda.synchronize()
    prof, ops.PROFILE = ops.PROFILE, None
    conv = [p for p in prof if p[0] != -1]
    assert len(conv) == len(notes), (len(conv), len(notes))
    tot = 0.0
    for (tile, macs, e0, e1), (name, desc) in zip(conv, notes):
        code = tile % 100
        fam = {51: "convw", 52: "convw", 54: "convw", 59: "convw 3x3", 61: "fused", 62: "fused", 64: "fused", 71: "fused+F23", 72: "fused+F23", 74: "fused+F23",
               91: "fused+F43", 92: "fused+F43", 94: "fused+F43", 96: "fused F43+F43", 80: "convwg4", 81: "convwg4p", 82: "convwg4x", 83: "convtw", 88: "convwg4s", 16: "x3"}.get(code, "conv_taps KC=%d" % code)
        ms = e0.elapsed_time(e1)
        if args.taps_only and not fam.startswith("conv_taps"):
            continue
        tot += ms
        print("%-15s %-44s %-18s tile %3dx%-3d %7.3f ms %7.1f TFLOP/s" % (name, desc, fam, tile // 100000, tile // 100 % 1000, ms, 2e-9 * macs / ms))
    print("total %.3f ms" % tot)


if __name__ == "__main__":
    main()
