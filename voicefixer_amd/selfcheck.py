"""``python -m voicefixer_amd --selfcheck IN.wav`` / ``python tools/selfcheck.py``: one command that tells a user with REAL
checkpoints whether the fast arithmetic of this build is safe on THEIR weights and input.

The reference is evaluated in fp32 direct sums (ATen); this build evaluates the k = 3 / 3x3 convolutions as Winograd F(4,3)
(half the products, rounding relative to the largest operand of a quad's window: DESIGN.md 3.0b) and offers split-bf16
products as an opt-in.  Parity tests pin both on seeded weights and golden vectors; trained checkpoints are not in the image
(the reference's own end-to-end check, test/test.py:27-95, needs them and its FLAC goldens).  This check needs neither a CPU
reference nor goldens: it runs the SAME input through the path three times on the device --

    default   Winograd F(4,3) where the kernels take it (what ``restore`` runs)
    direct    every convolution as the direct fp32 sum (engine.set_winograd(False))
    bf16x3    the opt-in split-bf16 products (set_math("bf16x3"))

-- and reports, stage by stage (mel, denoiser mask, UNet output, log-mel, vocoder conditioning, condnet, the four up-stages,
waveform), the largest difference relative to the stage's own peak, plus the waveform RMS difference.  Exit status 1 when a
stage of default-vs-direct or bf16x3-vs-direct differs by more than ``--tol`` (1e-4; the north-star bound on the waveform is
1e-3 RMS).  Nothing here touches the CPU oracle: it is a product-side diagnostic."""
import argparse
import sys

import numpy as np
import torch

STAGES = ("mel", "mask", "unet_out", "logmel", "denoised", "cond", "condnet", "up1", "up2", "up3", "up4", "wav")


def run_variants(pipe, wav, n, variants=("default", "direct", "bf16x3")):
    """wav: device (B, >= n).  Returns {variant: {stage: float64 CPU tensor}}."""
    from . import engine
    out = {}
    math0 = pipe.math
    try:
        for v in variants:
            engine.set_winograd(v != "direct")
            pipe.set_math("bf16x3" if v == "bf16x3" else "f32")
            rep = pipe.stage_report(wav, n)
            torch.cuda.synchronize()
            pipe.check()
            out[v] = {k: rep[k].detach().double().cpu() for k in STAGES}
    finally:
        engine.set_winograd(True)
        pipe.set_math(math0)
    return out


def compare(a, b):
    """Per stage: max |a - b| / max |b| (the stage's own peak); for the waveform also the RMS difference."""
    rows = {}
    for k in STAGES:
        peak = float(b[k].abs().max())
        rows[k] = float((a[k] - b[k]).abs().max()) / max(peak, 1e-30)
    rows["wav_rms"] = float(torch.sqrt(torch.mean((a["wav"] - b["wav"]) ** 2)))
    return rows


def report(results, tol, out=sys.stdout):
    base = results["direct"]
    bad = []
    print("%-10s %s" % ("stage", "  ".join("%-22s" % ("%s vs direct" % v) for v in results if v != "direct")), file=out)
    table = {v: compare(results[v], base) for v in results if v != "direct"}
    for k in STAGES + ("wav_rms",):
        cells = []
        for v, t in table.items():
            flag = "" if t[k] <= tol else "  <-- above %.0e" % tol
            if flag:
                bad.append((v, k, t[k]))
            cells.append("%-22s" % ("%.3e%s" % (t[k], flag)))
        print("%-10s %s" % (k, "  ".join(cells)), file=out)
    return bad, table


def main(argv=None):
    ap = argparse.ArgumentParser(prog="voicefixer_amd --selfcheck", description=__doc__.split("\n\n")[0])
    ap.add_argument("infile", nargs="?", default="", help="a 44.1 kHz (or resampled) WAV / FLAC file; omitted: a synthetic speech-like input")
    ap.add_argument("--seconds", type=float, default=10.0, help="how much of the input is used (from its start)")
    ap.add_argument("--batch", type=int, default=1, help="replicate the input this many times (the batch-32 launch geometry: --batch 32)")
    ap.add_argument("--tol", type=float, default=1e-4)
    ap.add_argument("--seeded", action="store_true", help="seeded random weights instead of the checkpoints under ~/.cache/voicefixer")
    ap.add_argument("--no-bf16x3", action="store_true")
    args = ap.parse_args(argv)
    from . import api, audio_io, weights
    if args.seeded:
        vf = api.VoiceFixer.from_state(weights.seeded_vocoder_state(1234), weights.seeded_restorer_state(4321))
    else:
        vf = api.VoiceFixer()
    pipe = vf._get_pipe()
    n_max = int(round(args.seconds * 44100))
    if args.infile:
        w = audio_io.load_wav(args.infile, 44100)[:n_max]
    else:
        t = np.arange(n_max) / 44100.0
        rng = np.random.default_rng(0)
        w = (0.05 * rng.standard_normal(n_max) + 0.2 * np.sin(2 * np.pi * 180.0 * t) * (1 + 0.5 * np.sin(2 * np.pi * 3.0 * t))).astype(np.float32)
    n = len(w)
    if n < 1025:
        raise SystemExit("selfcheck needs more than 1024 samples")
    wav = torch.from_numpy(np.ascontiguousarray(w))[None].repeat(max(1, args.batch), 1).to(pipe.device)
    variants = ("default", "direct") + (() if args.no_bf16x3 else ("bf16x3",))
    res = run_variants(pipe, wav, n, variants)
    print("selfcheck: %s, %d x %.2f s, weights: %s" % (args.infile or "synthetic input", wav.shape[0], n / 44100.0,
                                                      "seeded" if args.seeded else "checkpoints"))
    bad, _ = report(res, args.tol)
    if bad:
        print("selfcheck FAILED: %s" % ", ".join("%s/%s %.2e" % b for b in bad))
        print("(voicefixer_amd.engine.set_winograd(False) runs every convolution as the direct sum)")
        return 1
    print("selfcheck ok: every stage within %.0e of the direct fp32 sums" % args.tol)
    return 0


if __name__ == "__main__":
    sys.exit(main())
