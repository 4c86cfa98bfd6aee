#!/usr/bin/env python
"""BASELINE config 5: one 30-minute 44.1 kHz input, host to host (H2D, restore, D2H, concatenation).

  reference chunking  : VoiceFixer.restore_inmem  (60 hard-cut 30 s segments, batched 8 at a time)
  overlap-add streaming: VoiceFixer.restore_stream (30 s chunks every 29 s, 1 s linear cross-fade) with the latency
                         until the first finished stretch of output
"""
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
    ap.add_argument("--minutes", type=float, default=30.0)
    ap.add_argument("--math", default="f32")
    ap.add_argument("--segment-batch", type=int, default=8)
    args = ap.parse_args()
    n = int(args.minutes * 60 * 44100)
    rng = np.random.default_rng(0)
    t = np.arange(n, dtype=np.float32) / 44100.0
    wav = (0.05 * rng.standard_normal(n).astype(np.float32) + 0.3 * np.sin(2 * np.pi * 200.0 * t)).astype(np.float32)
    vf = VoiceFixer.from_state(weights.seeded_vocoder_state(1234), weights.seeded_restorer_state(4321))
    vf.set_math(args.math)
    vf.segment_batch = args.segment_batch
    vf.restore_inmem(wav[: 44100 * 31], cuda=True)  # warm-up (tables, allocator)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = vf.restore_inmem(wav, cuda=True)
    dt = time.perf_counter() - t0
    print("restore_inmem  (hard-cut 30 s segments): %.2f s wall = %.0fx real time, out %s" % (dt, n / 44100 / dt, out.shape))
    first = []
    t0 = time.perf_counter()
    out2 = vf.restore_stream(wav, 30.0, 1.0, batch_size=args.segment_batch,
                             on_chunk=lambda a, y: first.append(time.perf_counter() - t0) if not first else None)
    dt2 = time.perf_counter() - t0
    print("restore_stream (1 s overlap-add)       : %.2f s wall = %.0fx real time, first output after %.3f s, out %s"
          % (dt2, n / 44100 / dt2, first[0], out2.shape))
    t0 = time.perf_counter()
    first1 = []
    vf.restore_stream(wav[: 44100 * 120], 30.0, 1.0, batch_size=1,
                      on_chunk=lambda a, y: first1.append(time.perf_counter() - t0) if not first1 else None)
    print("restore_stream batch_size=1: first 29 s of output after %.3f s" % first1[0])


if __name__ == "__main__":
    main()
