import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from voicefixer_amd import VoiceFixer, weights
rng = np.random.default_rng(0)
lens = rng.integers(5 * 44100, 10 * 44100, size=12)
wavs = [(0.1 * rng.standard_normal(int(n))).astype(np.float32) for n in lens]
vf = VoiceFixer.from_state(weights.seeded_vocoder_state(1234), weights.seeded_restorer_state(4321))
vf.set_math("bf16x3")
pipe = vf._get_pipe()
import voicefixer_amd.api as api
def rb(streams):
    saved = vf.math; vf.math = "f32"   # bypass the single-stream guard for this experiment
    try: return vf.restore_batch(wavs, streams=streams)
    finally: vf.math = saved
rb(2)
ref = rb(1)
tot = 0
for rep in range(4):
    out = rb(4)
    bad = sum(int(not np.array_equal(a, b)) for a, b in zip(ref, out)); tot += bad
    print("bf16x3, 4 streams, rep %d: %d of %d utterances differ from the 1-stream run" % (rep, bad, len(ref)))
print("total", tot)
