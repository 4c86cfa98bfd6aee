#!/usr/bin/env python
"""GRU recurrent kernel timing (development tool, GPU only)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from voicefixer_amd import ops, packing
B, T = int(sys.argv[1]) if len(sys.argv) > 1 else 32, 1001
g = torch.Generator().manual_seed(0)
gi = torch.randn((B, T, 1536), generator=g).cuda()
w = packing.pack_gru_whh(torch.randn((768, 256), generator=g) / 16, torch.randn((768, 256), generator=g) / 16,
                         *ops.gru_layout()).cuda()
bhh = torch.zeros((2, 768)).cuda()
out = torch.empty((B, 512, 1004)).cuda()
ops.gru_bidir(gi, w, bhh, out, T)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(3):
    ops.gru_bidir(gi, w, bhh, out, T)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 3
print("gru B=%d T=%d layout=%s: %.3f ms/launch, %.2f us/step" % (B, T, ops.gru_layout(), ms, ms * 1e3 / T))

if B <= ops.GRU2_MAX_B:
    wt = torch.randn((2, 256, 768), generator=g).cuda() / 16
    err = torch.zeros(1, dtype=torch.int32).cuda()
    keep = ops.gru_bidir2(gi, wt, bhh, out, T, err)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(3):
        keep = ops.gru_bidir2(gi, wt, bhh, out, T, err)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    print("gru2 (two CUs per sequence) B=%d: %.3f ms/launch, %.2f us/step, err=%d" % (B, ms, ms * 1e3 / T, int(err.item())))
