python -m pytest tests/test_api_gpu.py -x -q -m gpu -k "long_input or segmentation" 2>&1 | tail -3
python - <<'PY'
import time, torch, numpy as np, sys
sys.path.insert(0, '.')
import voicefixer_amd
from voicefixer_amd import weights
vf = voicefixer_amd.VoiceFixer.from_state(weights.seeded_vocoder_state(1234), weights.seeded_restorer_state(4321))
n = 30*60*44100
wav = (0.1*torch.randn(n)).numpy()
for sb in (8, 15):
    vf.segment_batch = sb
    vf.restore_inmem(wav[:44100*90], cuda=True)
    torch.cuda.synchronize(); t=time.perf_counter()
    out = vf.restore_inmem(wav, cuda=True)
    dt=time.perf_counter()-t
    print("30 min input, segment_batch=%d: %.2f s wall (host to host) -> %.0f x RT" % (sb, dt, 1800/dt), torch.cuda.max_memory_allocated()/2**30, "GiB peak")
PY
