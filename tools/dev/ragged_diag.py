#!/usr/bin/env python
"""Development check: rows of a ragged batch vs the same utterances alone, stage by stage (mel, denoised mel, vocoder
output, final), at folder-bench sizes (5-10 s)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from voicefixer_amd import VoiceFixer, weights, ops  # noqa: E402
from voicefixer_amd.engine import RaggedRows  # noqa: E402


def rms(a, b):
    return float(torch.sqrt(torch.mean((a - b) ** 2)) / (torch.sqrt(torch.mean(b ** 2)) + 1e-12))


def main():
    n_utt = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    lo, hi = (float(sys.argv[2]), float(sys.argv[3])) if len(sys.argv) > 3 else (5.0, 10.0)
    rng = np.random.default_rng(0)
    lens = sorted(int(v) for v in rng.integers(int(lo * 44100), int(hi * 44100), size=n_utt))
    vf = VoiceFixer.from_state(weights.seeded_vocoder_state(1234), weights.seeded_restorer_state(4321))
    pipe = vf._get_pipe()
    wav = torch.zeros((n_utt, max(lens)))
    for r, n in enumerate(lens):
        wav[r, :n] = torch.from_numpy(0.1 * rng.standard_normal(n).astype(np.float32))
    wav = wav.cuda()
    rg = RaggedRows(lens, wav.device)
    T = rg.T_max
    mel = torch.empty((n_utt, T, 128), device="cuda")
    ops.stft_mel_rows(wav, mel, rg.n, T)
    dbg = {}
    _, den = pipe.restorer.forward(mel, T, debug=dbg, ragged=rg)
    y, Ly = pipe.vocoder.forward(den, T, ragged=rg)
    from voicefixer_amd.engine import _rows, G_TILE
    Tc = T + T % 2 + 4
    cond = _rows(n_utt, 128, Tc, G_TILE, wav.device, rg.voc[1])
    ops.mel_to_cond(den, cond, T, rg.T)
    st = {}
    yy, _ = pipe.vocoder.forward_cond(cond, Tc, stages=st, ragged=rg)
    st["wav"] = yy
    out = pipe.restore_rows(wav, lens)
    for r, n in enumerate(lens):
        w1 = wav[r:r + 1, :n].contiguous()
        m1, T1 = pipe.wav_to_mel(w1, n)
        d1 = {}
        _, den1 = pipe.restorer.forward(m1, T1, debug=d1)
        y1, Ly1 = pipe.vocoder.forward(den1, T1)
        o1 = pipe.restore(w1, n)
        Tc1 = T1 + T1 % 2 + 4
        c1 = _rows(1, 128, Tc1, G_TILE, wav.device)
        ops.mel_to_cond(den[r:r + 1, :T1].contiguous(), c1, T1)
        s1 = {}
        yy1, _ = pipe.vocoder.forward_cond(c1, Tc1, stages=s1)
        s1["wav"] = yy1
        msg = ["cond %.1e" % rms(cond[r:r + 1, :, :Tc1], c1[:, :, :Tc1])]
        for k in s1:
            L1 = Tc1 * {"condnet": 1, "up1": 7, "up2": 49, "up3": 147, "up4": 441, "wav": 441}[k]
            d = (st[k][r:r + 1, :, :L1] - s1[k][:, :, :L1]).abs().amax(dim=1)[0]
            bad = torch.nonzero(d > 1e-3 * float(s1[k][:, :, :L1].abs().max()))
            msg.append("%s %.1e (first bad col %s of %d, n bad %d)" % (k, rms(st[k][r:r + 1, :, :L1], s1[k][:, :, :L1]),
                                                             int(bad[0]) if len(bad) else "-", L1, len(bad)))
        print("   ", "; ".join(msg))
        # vocoder alone on the ragged batch's own denoised mel (isolates the vocoder)
        y1b, _ = pipe.vocoder.forward(den[r:r + 1, :T1].contiguous(), T1)
        print("row %d n=%d T=%d: mel %.2e mask %.2e unet %.2e den %.2e voc %.2e (voc on same den %.2e) final %.2e maxabs %.2e"
              % (r, n, T1, rms(mel[r:r + 1, :T1], m1), rms(dbg["mask"][r:r + 1, :, :T1], d1["mask"]),
                 rms(dbg["unet_out"][r:r + 1, :T1], d1["unet_out"]), rms(den[r:r + 1, :T1], den1),
                 rms(y[r:r + 1, :, :Ly1], y1[:, :, :Ly1]), rms(y[r:r + 1, :, :Ly1], y1b[:, :, :Ly1]),
                 rms(out[r:r + 1, :n], o1), float((out[r:r + 1, :n] - o1).abs().max())))


if __name__ == "__main__":
    with torch.no_grad():
        main()
