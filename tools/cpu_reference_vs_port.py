#!/usr/bin/env python
"""Time the REFERENCE'S OWN modules (imported from /root/reference through oracle/ref_shim.py) and
the oracle port (oracle/oracle.py) side by side: same host cores, same thread count, same seeded weights, same input.

bench.py's ``cpu_baseline`` has ``kind: "port"`` because /root/reference cannot travel to the GPU box; this tool is the
measured equivalence behind that label: run it in the build container,
    python tools/cpu_reference_vs_port.py --seconds 10 --reps 3 > profiles/r04_cpu_reference_vs_port.json
Test infrastructure: nothing here is on the product path."""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=10.0)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--threads", type=int, default=os.cpu_count())
    args = ap.parse_args()
    from oracle import oracle, ref_shim
    from voicefixer_amd import weights
    import bench
    assert ref_shim.reference_available(), "/root/reference is not here (build container only)"
    vsd, rsd = weights.seeded_vocoder_state(1234), weights.seeded_restorer_state(4321)
    ref_vf = ref_shim.build_reference_models(tempfile.mkdtemp(prefix="vfx_home_"), vsd,
                                             {"generator." + k: v for k, v in rsd.items()})
    n = int(round(args.seconds * 44100))
    wav = bench.synth_batch(1, n, 1000, "cpu")[0].numpy()
    torch.set_num_threads(args.threads)

    def timed(fn):
        ts, y = [], None
        for _ in range(args.reps + 1):          # the first repetition is the warm-up
            t0 = time.perf_counter()
            with torch.no_grad():
                y = fn()
            ts.append(time.perf_counter() - t0)
        return sorted(ts[1:]), y

    t_ref, y_ref = timed(lambda: ref_vf.restore_inmem(wav, cuda=False, mode=0))    # voicefixer/base.py:107-139, B = 1
    t_port, y_port = timed(lambda: oracle.restore_inmem(wav, vsd, rsd))
    med = lambda ts: ts[len(ts) // 2]
    out = {
        "what": "reference's own modules (via oracle/ref_shim.py) vs the oracle port, one %.0f s utterance, mode 0, B = 1, "
                "seeded weights, %d threads, median of %d repetitions after one warm-up" % (args.seconds, args.threads, args.reps),
        "cpu_model": bench.cpu_model(), "os_cpu_count": os.cpu_count(), "threads": args.threads, "torch": torch.__version__,
        "reference": {"seconds": [round(t, 3) for t in t_ref], "x_real_time": round(args.seconds / med(t_ref), 3)},
        "port": {"seconds": [round(t, 3) for t in t_port], "x_real_time": round(args.seconds / med(t_port), 3)},
        "port_over_reference_time": round(med(t_port) / med(t_ref), 4),
        "rms_port_vs_reference": float(np.sqrt(np.mean((y_ref - y_port) ** 2))),
        "note": "the reference computes the STFT as a 2 x 1025-channel conv1d with the DFT basis (torchlibrosa), re-normalises "
                "the weight-normed vocoder weights on every forward and evaluates the dead UpsampleNet.skip_conv; the port "
                "uses torch.stft and folded weights and skips the dead branch (SURVEY.md Appendix C) -- the port is the FASTER "
                "of the two, i.e. the more demanding baseline",
    }
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
