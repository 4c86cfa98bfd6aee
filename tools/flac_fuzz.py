#!/usr/bin/env python
"""Corrupted-stream fuzzing of the C FLAC frame decoder (voicefixer_amd/csrc_host/vfx_flac.c) -- bit flips, byte
overwrites and truncations of the reference's fixtures and of a 24-bit stereo stream of our own; every decode must
either succeed or raise FlacError (never crash).  With ``--sanitize`` the codec is first rebuilt with
-fsanitize=address,undefined into a scratch directory and the run repeats under it.

    python tools/flac_fuzz.py [--sanitize] [iterations] [seed]
"""
import glob
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.environ.get("VFX_FUZZ_PACKAGE_ROOT", ROOT))     # (--sanitize: a scratch copy holding the sanitized .so)


def run(iters, seed):
    from voicefixer_amd import flac
    assert flac.native() is not None, "libvfx_audio.so is not built"
    rng = np.random.default_rng(seed)
    files = [open(p, "rb").read() for p in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "ref_utterance", "*.flac")))[:2]]
    x = rng.integers(-2 ** 22, 2 ** 22, (9000, 2)).astype(np.int64)
    files.append(flac.encode((x * np.hanning(9000)[:, None]).astype(np.int64), 44100, 24))
    ok = err = 0
    for it in range(iters):
        d = bytearray(files[it % len(files)])
        for _ in range(int(rng.integers(1, 6))):
            mode = int(rng.integers(0, 3))
            pos = int(rng.integers(38, len(d)))
            if mode == 0:
                d[pos] ^= 1 << int(rng.integers(0, 8))
            elif mode == 1:
                d[pos] = int(rng.integers(0, 256))
            else:
                d = d[:pos]
        for verify in (True, False):
            try:
                flac.decode(bytes(d), verify=verify, use_native=True)
                ok += 1
            except flac.FlacError:
                err += 1
    print("seed %d: %d corrupted streams decoded without complaint (CRC checks off, or the damage was in padding), "
          "%d rejected with FlacError, no crash" % (seed, ok, err))


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    iters = int(args[0]) if args else 4000
    seed = int(args[1]) if len(args) > 1 else 1
    if "--sanitize" in sys.argv:
        import shutil
        import tempfile
        tmp = tempfile.mkdtemp()
        pkg = os.path.join(tmp, "voicefixer_amd")
        os.makedirs(pkg)
        for f in ("flac.py",):
            shutil.copy(os.path.join(ROOT, "voicefixer_amd", f), pkg)
        open(os.path.join(pkg, "__init__.py"), "w").close()
        subprocess.check_call(["gcc", "-O1", "-g", "-std=c11", "-fPIC", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                               "-shared", os.path.join(ROOT, "voicefixer_amd", "csrc_host", "vfx_flac.c"),
                               os.path.join(ROOT, "voicefixer_amd", "csrc_host", "vfx_resample.c"), "-lm", "-o",
                               os.path.join(pkg, "libvfx_audio.so")])
        asan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
        env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0", VFX_FUZZ_PACKAGE_ROOT=tmp)
        print("under -fsanitize=address,undefined:", flush=True)
        sys.exit(subprocess.call([sys.executable, os.path.abspath(__file__), str(iters), str(seed)], env=env))
    run(iters, seed)


if __name__ == "__main__":
    main()
