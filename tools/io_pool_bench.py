#!/usr/bin/env python
"""Host side of the folder driver in isolation (no GPU): what restore_folder's worker pool decodes and encodes per second
when the files are FLAC -- audio_io.load_wav / save_wave on N threads, with the C frame codec (libvfx_audio.so) and with
the Python one (VFX_FLAC_NATIVE=0).  The device restores ~1390 s of audio per second; the pool has to decode AND encode
that much to stay out of the way.

    python tools/io_pool_bench.py [files] [seconds per file]
"""
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from concurrent.futures import ThreadPoolExecutor
    from voicefixer_amd import audio_io, flac
    nfiles = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    secs = float(sys.argv[2]) if len(sys.argv) > 2 else 10.0
    rng = np.random.default_rng(0)
    d = tempfile.mkdtemp()
    paths = []
    n = int(44100 * secs)
    for i in range(nfiles):
        x = (0.3 * np.sin(np.arange(n) * (0.01 + 0.001 * i)) + 0.02 * rng.standard_normal(n)).astype(np.float32)
        p = os.path.join(d, "f%03d.flac" % i)
        audio_io.save_wave(x[None], p)
        paths.append(p)
    size = sum(os.path.getsize(p) for p in paths) / nfiles
    print("%d FLAC files of %.0f s (%.0f kB each, %.2f of PCM16), codec: %s, %d cores"
          % (nfiles, secs, size / 1e3, size / (2.0 * n), "libvfx_audio.so (C)" if flac.native() else "flac.py (Python)",
             len(os.sched_getaffinity(0))), flush=True)
    for threads in (1, 2, 4, 8):
        with ThreadPoolExecutor(threads) as pool:
            list(pool.map(audio_io.load_wav, paths[:threads]))      # warm-up: threads started
            td, te = [], []
            for _ in range(3):
                t0 = time.perf_counter()
                xs = list(pool.map(audio_io.load_wav, paths))
                td.append(time.perf_counter() - t0)
                t0 = time.perf_counter()
                list(pool.map(lambda a: audio_io.save_wave(a[0][None], a[1]), zip(xs, [p + ".out.flac" for p in paths])))
                te.append(time.perf_counter() - t0)
        print("  %d thread(s): decode %6.0f x real time, encode %6.0f x real time (best of 3)"
              % (threads, nfiles * secs / min(td), nfiles * secs / min(te)), flush=True)
    # other sample rates: the pool also resamples (48 kHz input, the rate of the corpus VoiceFixer was trained on)
    x48 = (0.3 * np.sin(np.arange(int(48000 * secs)) * 0.01)).astype(np.float32)
    audio_io.resample_hq(x48, 48000, 44100)
    for threads in (1, 8):
        with ThreadPoolExecutor(threads) as pool:
            list(pool.map(lambda _: audio_io.resample_hq(x48, 48000, 44100), range(threads)))
            ts = []
            for _ in range(3):
                t0 = time.perf_counter()
                list(pool.map(lambda _: audio_io.resample_hq(x48, 48000, 44100), range(4 * threads)))
                ts.append(time.perf_counter() - t0)
        print("  %d thread(s): resample 48 kHz -> 44.1 kHz %6.0f x real time (%s)"
              % (threads, 4 * threads * secs / min(ts), "vfx_resample_poly_f32" if flac.native() else "scipy upfirdn"), flush=True)
    if flac.native() and not (os.environ.get("VFX_DEV") == "1" and os.environ.get("VFX_FLAC_NATIVE") == "0"):
        env = dict(os.environ, VFX_DEV="1", VFX_FLAC_NATIVE="0")
        subprocess.run([sys.executable, os.path.abspath(__file__), str(min(nfiles, 16)), str(secs)], env=env)


if __name__ == "__main__":
    main()
