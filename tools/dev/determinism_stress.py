#!/usr/bin/env python
"""Race hunt (development, GPU): the same batch through the whole path many times, alternating over two HIP streams with a DIFFERENT batch in
flight on the other stream, every result compared bit for bit with the first.  The persistent kernels with deferred epilogues (convwg4x_kernel)
hand tiles over through LDS and trickle stores under the next tile's K loop: a missing barrier or an early overwrite would show up here as a
run that differs.   python tools/dev/determinism_stress.py [--reps 40] [--batch 32] [--seconds 10]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from voicefixer_amd import engine, weights  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=40)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--seconds", type=float, default=10.0)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    n = int(round(args.seconds * 44100))
    pipe = engine.Pipeline(weights.seeded_vocoder_state(1234), weights.seeded_restorer_state(4321), dev)
    a = bench.synth_batch(args.batch, n, 11, dev)
    others = [bench.synth_batch(args.batch, n, 100 + k, dev) for k in range(3)]
    s0, s1 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    pipe.set_streams(2)
    want = pipe.restore(a, n).clone()
    torch.cuda.synchronize()
    bad = 0
    for i in range(args.reps):
        with torch.cuda.stream(s1 if i % 2 else s0):
            got = pipe.restore(a, n)
        with torch.cuda.stream(s0 if i % 2 else s1):
            pipe.restore(others[i % 3], n)
        torch.cuda.synchronize()
        pipe.check()
        if not torch.equal(got, want):
            bad += 1
            print("rep %d differs: max |diff| %.3e" % (i, float((got - want).abs().max())), flush=True)
    # ragged rows as well (tiles past a row's end are skipped, the pipelines restart)
    lens = [n - 441 * (3 * r) - 7 * r for r in range(args.batch)]
    want_r = pipe.restore_rows(a, lens).clone()
    for i in range(max(4, args.reps // 4)):
        with torch.cuda.stream(s1 if i % 2 else s0):
            got = pipe.restore_rows(a, lens)
        with torch.cuda.stream(s0 if i % 2 else s1):
            pipe.restore(others[i % 3], n)
        torch.cuda.synchronize()
        pipe.check()
        if not torch.equal(got, want_r):
            bad += 1
            print("ragged rep %d differs: max |diff| %.3e" % (i, float((got - want_r).abs().max())), flush=True)
    print("determinism_stress: %d repetitions + %d ragged, batch %d x %.0f s, two streams: %s" % (
        args.reps, max(4, args.reps // 4), args.batch, args.seconds, "ALL BIT-IDENTICAL" if bad == 0 else "%d DIFFER" % bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
