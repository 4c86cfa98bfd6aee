#!/usr/bin/env python
"""Folder-style inference on RAGGED lengths (every utterance its own length): aggregate real-time factor of
VoiceFixer.restore_batch (ragged batches with per-row lengths) as a function of the number of HIP streams and of the
ragged ratio (0.999 = one batch per frame count, the round-1 behaviour), host to host."""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from voicefixer_amd import VoiceFixer, weights  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=32)
    ap.add_argument("--math", default="f32")
    ap.add_argument("--streams", default="1,2,4", help="comma-separated stream counts to time")
    ap.add_argument("--ratios", default="0.75", help="comma-separated ragged ratios to time")
    ap.add_argument("--batch", type=int, default=32)
    args = ap.parse_args()
    rng = np.random.default_rng(0)
    lens = rng.integers(5 * 44100, 10 * 44100, size=args.n)
    wavs = [(0.1 * rng.standard_normal(int(n))).astype(np.float32) for n in lens]
    total = float(sum(lens)) / 44100.0
    vf = VoiceFixer.from_state(weights.seeded_vocoder_state(1234), weights.seeded_restorer_state(4321))
    vf.set_math(args.math)
    vf.restore_batch(wavs[:4], streams=2)  # warm-up
    ref = None
    from voicefixer_amd.api import plan_batches
    for ratio in [float(v) for v in args.ratios.split(",")]:
        plan = plan_batches(sorted(int(n) for n in lens), args.batch, ratio)
        print("%d utterances, ragged ratio %.3f: %d batches" % (args.n, ratio, len(plan)))
        for st in [int(v) for v in args.streams.split(",")]:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            outs = vf.restore_batch(wavs, streams=st, batch_size=args.batch, ragged_ratio=ratio)
            dt = time.perf_counter() - t0
            if ref is None:
                ref = outs
            err = max(float(np.abs(a - b).max()) for a, b in zip(ref, outs))
            print("  streams=%d: %d ragged utterances (%.0f s of audio) in %.3f s = %.0fx real time (max |diff| vs first run %.1e)"
                  % (st, args.n, total, dt, total / dt, err))


if __name__ == "__main__":
    main()
