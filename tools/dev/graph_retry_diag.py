#!/usr/bin/env python
"""Reproducer (development, GPU): HIP-graph replays around ONE eager pass of the same shape on the same stream.

With the peak workspace and the GRU mailboxes zeroed by hipMemsetAsync (memset NODES in the captured graph; library builds up to
3373d8f871350c85) the replays after the eager pass came back divided by ~3 -- the peak rule read a stale workspace -- on every trial;
with the zeroing as a kernel node (vfx_zero_u32, vfx_misc.hip) all five figures per trial are equal.  Select an older build with
VFX_DEV=1 VFX_LIB=path/to/libvfx_hip.so to see the difference.  The regression test is
tests/test_api_gpu.py::test_graph_replays_survive_an_eager_pass_in_between."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from voicefixer_amd import engine, weights  # noqa: E402

gg = np.load(os.path.join(ROOT, "tests", "golden", "restore_noise_T36.npz"))
rms = lambda a: float(np.sqrt(np.mean((np.asarray(a.cpu().numpy(), np.float64) - gg["restored"]) ** 2)))
x = torch.from_numpy(gg["wav"])[None].cuda()
n = x.shape[1]
keep = []
for trial in range(3):
    pipe = engine.Pipeline(weights.seeded_vocoder_state(1234), weights.seeded_restorer_state(4321), "cuda:0")
    pipe.restore(x, n)
    torch.cuda.synchronize()
    pipe.enable_graphs(max_shapes=2, max_batch=1)
    a = (rms(pipe.restore(x, n)), rms(pipe.restore(x, n)))
    g, pipe._graphs = pipe._graphs, None          # an eager pass while the captured graph stays alive
    r = rms(pipe.restore(x, n))
    pipe._graphs = g
    b = (rms(pipe.restore(x, n)), rms(pipe.restore(x, n)))
    print("trial %d: replays %.2e %.2e | eager %.2e | replays after %.2e %.2e   (RMS distance from the golden waveform)" % (trial, a[0], a[1], r, b[0], b[1]),
          flush=True)
    keep.append(g)
