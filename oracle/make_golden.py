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
  mode1_speech_ref.npz    the reference's OWN VoiceFixer.remove_higher_frequency (base.py:87-104) and restore_inmem(mode=1)
                          on 0.75 s of the reference's test utterance, executed through ref_shim with librosa's three
                          transforms bound to oracle/librosa_like: ``hf_cut`` (the pre-filtered waveform), ``restored``
  vocoder_oracle_ref.npz  the reference's OWN Vocoder.oracle (vocoder/base.py:58-77) on 1 s of its fixture
                          test/utterance/original/p360_001_mic1.flac: PCM16 input, the conditioning its numpy front-end
                          builds (``cond``), and the int16 frames it hands to soundfile.write (``out_pcm``)
  ref_utterance/          the reference's test utterances and FLAC goldens (test/utterance/original/*.flac,
                          target/{oracle,output_mode_0,output_mode_1}.flac) copied verbatim: inputs / targets of the
                          real-checkpoint harness (tests/test_real_checkpoints.py = test/test.py:27-95) and, together with
                          ``flac_original_pcm.npz`` (the PCM of original.wav), known answers for voicefixer_amd/flac.py
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

    # ---- the librosa legs, executed from the reference's own code (ref_shim binds librosa.stft / istft / filters.mel
    # to oracle/librosa_like; everything else on these lines is the reference's numpy / torch code, unmodified)
    seg = (pcm[30000:30000 + 33075].astype(np.float32) / 32768.0)          # 0.75 s of the reference's utterance
    with torch.no_grad():
        hf = vf.remove_higher_frequency(seg)                                # base.py:87-104
        restored1 = vf.restore_inmem(seg, cuda=False, mode=1)               # base.py:107-139 with mode == 1
    assert hf.shape == (512 * (len(seg) // 512),) and restored1.shape == (1, hf.shape[0])
    np.savez_compressed(os.path.join(OUT, "mode1_speech_ref.npz"), wav=seg, hf_cut=hf.astype(np.float32),
                        restored=restored1, voc_seed=VOC_SEED, res_seed=RES_SEED)
    print("mode1_speech_ref", seg.shape, "->", restored1.shape)

    from voicefixer_amd import flac
    import soundfile as sf_stub                                             # (the ref_shim stub: records sf.write calls)
    rate, p360, bps = flac.read(os.path.join(ref_shim.REFERENCE_ROOT, "test/utterance/original/p360_001_mic1.flac"))
    assert rate == 44100 and bps == 16 and p360.shape[1] == 1
    pcm_in = p360[20000:20000 + 44100, 0].astype(np.int16)
    tmp_in = os.path.join(home, "oracle_in.wav")
    wavfile.write(tmp_in, 44100, pcm_in)
    sf_stub.written.clear()
    voc = vf._model.vocoder
    voc.oracle(fpath=tmp_in, out_path=os.path.join(home, "oracle_out.wav"), cuda=False)   # vocoder/base.py:58-77
    (_, frames, rate_out), = sf_stub.written
    assert rate_out == 44100 and frames.dtype == np.int16
    # the conditioning tensor its numpy front-end built (same calls, same order as vocoder/base.py:61-71)
    from voicefixer.vocoder.model.util import linear_to_mel, normalize, amp_to_db, pre
    from voicefixer.tools.wav import read_wave
    from voicefixer.vocoder.config import Config
    import librosa as librosa_stub
    w = read_wave(tmp_in, sample_rate=44100)[..., 0]
    w = w / np.max(np.abs(w))
    st = np.abs(librosa_stub.stft(w, hop_length=Config.hop_length, win_length=Config.win_size, n_fft=Config.n_fft))
    cond = pre(np.transpose(normalize(amp_to_db(np.abs(linear_to_mel(st))) - 20), (1, 0)))
    np.savez_compressed(os.path.join(OUT, "vocoder_oracle_ref.npz"), pcm_in=pcm_in, cond=cond.numpy(),
                        out_pcm=frames.reshape(-1), voc_seed=VOC_SEED)
    print("vocoder_oracle_ref", pcm_in.shape, "cond", tuple(cond.shape), "->", frames.shape)

    # ---- the reference's FLAC fixtures, verbatim (data, not source), + the PCM known answer for the decoder
    import shutil
    dst = os.path.join(OUT, "ref_utterance")
    os.makedirs(dst, exist_ok=True)
    for rel in ("original/original.flac", "original/p360_001_mic1.flac", "target/oracle.flac",
                "target/output_mode_0.flac", "target/output_mode_1.flac"):
        shutil.copyfile(os.path.join(ref_shim.REFERENCE_ROOT, "test/utterance", rel),
                        os.path.join(dst, rel.replace("/", "_")))
    np.savez_compressed(os.path.join(OUT, "flac_original_pcm.npz"), pcm=pcm)   # == original.wav, PCM16


if __name__ == "__main__":
    main()
