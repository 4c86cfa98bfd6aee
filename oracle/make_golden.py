"""Generate tests/golden/*.npz by running the REFERENCE'S OWN modules (via ref_shim) in the
build container.  TEST INFRASTRUCTURE ONLY.

    python oracle/make_golden.py            # rewrites tests/golden/

Weights are not stored: they are regenerated from a seed by
``voicefixer_amd.weights.seeded_*_state`` (torch CPU generator, same image on both sides),
and this script first asserts that the reference accepts those state dicts key-for-key.
Stored per fixture: the input, the reference's stage outputs and final waveform (float32).

Fixtures
  restore_noise_T36.npz   restore_inmem on 15 523 samples of seeded noise+sine (T=36 frames)
  restore_speech_T51.npz  restore_inmem on the first 0.5 s of the reference's PCM16 test
                          utterance test/utterance/original/original.wav (22 050 samples)
  vocoder_T101.npz        Vocoder.forward on a (1,1,101,128) mel: stage-1 dilations
                          (243, 729, 2187) exceed L1 = 7*106 = 742
  vocoder_B2_T24.npz      Vocoder.forward, batch 2, T even (pad_tail = 4)
  filterbank.npz          mel fb support (lo, hi per mel bin) + sha256 of the float32 bytes
  mel_weight_table.npz    the 128-entry Config.mel_weight_torch table the reference holds
                          (vocoder/config.py:161-290) and the constants (a, b) of its analytic fit
                          (config.py:300-316): the reciprocal slaney area normalisation of librosa.filters.mel,
                          used to pin oracle/librosa_like.mel_basis (python oracle/make_golden.py --constants)
"""
import hashlib
import os
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import ref_shim  # noqa: E402
from voicefixer_amd import weights  # noqa: E402

VOC_SEED = 1234
RES_SEED = 4321
OUT = os.path.join(ROOT, "tests", "golden")


def synth_wave(n, seed):
    g = torch.Generator().manual_seed(seed)
    wav = (0.1 * torch.randn(n, generator=g)).numpy().astype(np.float32)
    t = np.arange(n, dtype=np.float64) / 44100.0
    wav += (0.2 * np.sin(2 * np.pi * 220.0 * t) + 0.1 * np.sin(2 * np.pi * 1330.0 * t)).astype(np.float32)
    return wav


def write_reference_constants():
    """Reference-held numeric tables (no model run): read straight from the imported reference Config."""
    os.makedirs(OUT, exist_ok=True)
    home = tempfile.mkdtemp(prefix="vfx_home_")
    ref_shim.prepare_home(home)
    ref_shim.import_reference(home)
    from voicefixer.vocoder.config import Config
    import inspect
    table = Config.mel_weight_torch.numpy().astype(np.float64)
    sig = inspect.signature(Config.get_mel_weight_torch)
    a, b = float(sig.parameters["a"].default), float(sig.parameters["b"].default)
    assert table.shape == (128,)
    np.savez_compressed(os.path.join(OUT, "mel_weight_table.npz"), table=table, a=a, b=b)
    print("mel_weight_table", table[0], table[-1], "fit a, b =", a, b)

    # the restorer's HTK filterbank exactly as restorer/model.py:203 builds it
    from voicefixer.tools.mel_scale import MelScale
    fb = MelScale(n_mels=128, sample_rate=44100, n_stft=2048 // 2 + 1).fb
    assert tuple(fb.shape) == (1025, 128) and fb.dtype == torch.float32
    nz = fb > 0
    lo = np.array([int(torch.nonzero(nz[:, m])[0]) for m in range(128)], dtype=np.int32)
    hi = np.array([int(torch.nonzero(nz[:, m])[-1]) for m in range(128)], dtype=np.int32)
    raw = fb.numpy().astype(np.float32)
    sha = hashlib.sha256(raw.tobytes()).hexdigest()
    # the same matrix with the sign of zeros dropped (fb[0, 0] is -0.0 in the reference: (-1 * 0) / f_diff; a banded
    # table cannot and need not represent the sign of a zero OUTSIDE a band)
    sha_abs = hashlib.sha256(np.abs(raw).tobytes()).hexdigest()
    np.savez_compressed(os.path.join(OUT, "filterbank.npz"), lo=lo, hi=hi, nnz=int(nz.sum()), sha256=sha,
                        sha256_abs=sha_abs, negative_zeros=int(np.signbit(raw).sum()))
    print("fb nnz", int(nz.sum()), "sha256", sha, "sha256(|fb|)", sha_abs, "negative zeros", int(np.signbit(raw).sum()))


def main():
    if "--constants" in sys.argv[1:]:
        write_reference_constants()
        return
    write_reference_constants()
    os.makedirs(OUT, exist_ok=True)
    home = tempfile.mkdtemp(prefix="vfx_home_")
    vsd = weights.seeded_vocoder_state(VOC_SEED)
    rsd = weights.seeded_restorer_state(RES_SEED)
    vf = ref_shim.build_reference_models(home, vsd, {"generator." + k: v for k, v in rsd.items()})
    # the reference must have taken our weights verbatim
    ref_r = vf._model.generator.state_dict()
    ref_v = vf._model.vocoder.model.state_dict()
    assert list(ref_r.keys()) == list(rsd.keys()) and list(ref_v.keys()) == list(vsd.keys())
    assert all(torch.equal(ref_r[k], rsd[k]) for k in rsd)
    assert all(torch.equal(ref_v[k], vsd[k]) for k in vsd)

    def run_restore(wav, name):
        with torch.no_grad():
            sp, mel = vf._pre(vf._model, wav, False)
            out = vf._model(sp, mel)
            den = (10 ** torch.clip(out["mel"], max=5))
            voc = vf._model.vocoder(den, cuda=False)
            final = vf.restore_inmem(wav, cuda=False, mode=0)
        np.savez_compressed(
            os.path.join(OUT, name),
            wav=wav, mel=mel.numpy(), mask=(out["clean"] / mel).numpy(),
            logmel=out["mel"].numpy(), unet_out=out["unet_out"].numpy(),
            voc_wav=voc.numpy(), restored=final,
            voc_seed=VOC_SEED, res_seed=RES_SEED)
        print(name, "T =", mel.shape[2], "out", final.shape, "rms", float(np.sqrt((final ** 2).mean())))

    run_restore(synth_wave(15523, 7), "restore_noise_T36.npz")

    from scipy.io import wavfile
    sr, pcm = wavfile.read(os.path.join(ref_shim.REFERENCE_ROOT, "test/utterance/original/original.wav"))
    assert sr == 44100 and pcm.dtype == np.int16
    speech = (pcm[44100:44100 + 22050].astype(np.float32) / 32768.0)
    run_restore(speech, "restore_speech_T51.npz")

    def run_vocoder(mel, name):
        with torch.no_grad():
            out = vf._model.vocoder(mel, cuda=False)
        np.savez_compressed(os.path.join(OUT, name), mel=mel.numpy(), wav=out.numpy(), voc_seed=VOC_SEED)
        print(name, tuple(mel.shape), "->", tuple(out.shape))

    g = torch.Generator().manual_seed(11)
    # log-uniform magnitudes over ~6 decades so every branch of the dB/clip front-end is hit
    mel = 10 ** (torch.rand((1, 1, 101, 128), generator=g) * 6 - 3)
    mel[0, 0, :3, :5] = 0.0  # exact zeros -> min_level clamp
    run_vocoder(mel, "vocoder_T101.npz")
    mel = 10 ** (torch.rand((2, 1, 24, 128), generator=g) * 5 - 2)
    run_vocoder(mel, "vocoder_B2_T24.npz")


if __name__ == "__main__":
    main()
