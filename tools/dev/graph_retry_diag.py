"""diagnostic: the body of test_gru_retry_with_graphs_enabled_runs_eager as a plain script (fresh process)"""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import voicefixer_amd
from voicefixer_amd import weights, _lib
vf = voicefixer_amd.VoiceFixer.from_state(weights.seeded_vocoder_state(1234), weights.seeded_restorer_state(4321))
gg = np.load(os.path.join(ROOT, "tests", "golden", "restore_noise_T36.npz"))
rms = lambda a: float(np.sqrt(np.mean((np.asarray(a, np.float64) - gg["restored"]) ** 2)))
pipe = vf._get_pipe()
pipe.enable_graphs(max_shapes=2, max_batch=1)
step = lambda s: print(s, flush=True)
try:
    vf.restore_inmem(gg["wav"], cuda=True); step("captured %d" % len(pipe._graphs))
    before = _lib.lib().vfx_launch_count()
    vf.restore_inmem(gg["wav"], cuda=True)
    replay_launches = _lib.lib().vfx_launch_count() - before; step("replay launches %d" % replay_launches)
    retries = getattr(pipe, "gru_retries", 0)
    pipe.restorer.gru_err.fill_(1)
    before = _lib.lib().vfx_launch_count()
    out = vf.restore_inmem(gg["wav"], cuda=True)
    step("retry: retries %d single %s rms %.2e launches %d" % (pipe.gru_retries, pipe.restorer.gru_single, rms(out), _lib.lib().vfx_launch_count() - before))
    again = vf.restore_inmem(gg["wav"], cuda=True); step("again rms %.2e graphs %d" % (rms(again), len(pipe._graphs)))
    pipe.restorer.gru_single = True
    try:
        n_graphs = len(pipe._graphs)
        pipe.restore(torch.from_numpy(gg["wav"][None, :12000]).cuda(), 12000)
        step("gru_single new shape: graphs %d -> %d" % (n_graphs, len(pipe._graphs)))
    finally:
        pipe.restorer.gru_single = False
finally:
    step("disable ...")
    pipe.disable_graphs()
    step("disabled ok")
step("flag %d" % int(pipe.restorer.gru_err.item()))
