"""The CPU oracle against the golden vectors produced by the reference's own modules
(oracle/make_golden.py).  Runs without /root/reference and without a GPU."""
import hashlib
import os

import numpy as np
import torch

from conftest import GOLDEN
from oracle import oracle

TOL_WAV = 2e-5      # abs, waveform in [-1, 1]; observed ~6e-6 max (fp32 summation order)
TOL_LOGMEL = 1e-4   # abs, log10-mel values O(1)


def _load(name):
    return np.load(os.path.join(GOLDEN, name))


def test_filterbank_known_answer():
    g = _load("filterbank.npz")
    fb = oracle.mel_filterbank()
    assert fb.shape == (1025, 128)
    assert hashlib.sha256(fb.numpy().astype(np.float32).tobytes()).hexdigest() == str(g["sha256"])
    nz = fb > 0
    assert int(nz.sum()) == int(g["nnz"]) == 2018
    lo = np.array([int(torch.nonzero(nz[:, m])[0]) for m in range(128)])
    hi = np.array([int(torch.nonzero(nz[:, m])[-1]) for m in range(128)])
    assert np.array_equal(lo, g["lo"]) and np.array_equal(hi, g["hi"])   # bit-exact bin indexing
    assert lo[0] == 1 and hi[-1] == 1023 and (hi - lo + 1).max() == 55


def _check_restore(name, seeded_states):
    g = _load(name)
    vsd, rsd = seeded_states
    wav = torch.from_numpy(g["wav"])
    with torch.no_grad():
        mel = oracle.wav_to_mel(wav[None])
        assert mel.shape == g["mel"].shape
        rel = np.linalg.norm(mel.numpy() - g["mel"]) / np.linalg.norm(g["mel"])
        assert rel < 5e-6, rel
        o = oracle.restorer_forward(torch.from_numpy(g["mel"]), rsd, return_all=True)
        assert np.abs(o["mask"].numpy() - g["mask"]).max() < 1e-5
        assert np.abs(o["mel"].numpy() - g["logmel"]).max() < TOL_LOGMEL
        assert np.all(o["unet_out"].numpy()[..., 127] == 0.0)  # Appendix C: bin 127 passthrough
        voc = oracle.vocoder_forward(oracle.from_log(torch.from_numpy(g["logmel"])), vsd)
        assert np.abs(voc.numpy() - g["voc_wav"]).max() < TOL_WAV
        out = oracle.restore_inmem(g["wav"], vsd, rsd)
    assert out.shape == g["restored"].shape == (1, g["wav"].shape[0])
    assert np.abs(out - g["restored"]).max() < 5e-5
    assert np.sqrt(np.mean((out - g["restored"]) ** 2)) < 1e-5


def test_restore_noise_T36(seeded_states):
    _check_restore("restore_noise_T36.npz", seeded_states)


def test_restore_speech_T51(seeded_states):
    _check_restore("restore_speech_T51.npz", seeded_states)


def test_vocoder_T101_dilation_exceeds_length(seeded_states):
    g = _load("vocoder_T101.npz")
    with torch.no_grad():
        out = oracle.vocoder_forward(torch.from_numpy(g["mel"]), seeded_states[0])
    assert out.shape == g["wav"].shape == (1, 1, 441 * (101 + 1 + 4))
    assert np.abs(out.numpy() - g["wav"]).max() < TOL_WAV


def test_vocoder_batch2_even_T(seeded_states):
    g = _load("vocoder_B2_T24.npz")
    with torch.no_grad():
        out = oracle.vocoder_forward(torch.from_numpy(g["mel"]), seeded_states[0])
    assert out.shape == (2, 1, 441 * 28)
    assert np.abs(out.numpy() - g["wav"]).max() < TOL_WAV


def test_legacy_key_style_loads_identically(seeded_states):
    from voicefixer_amd import weights
    legacy = weights.seeded_vocoder_state(1234, legacy=True)
    assert any(k.endswith(".weight_g") for k in legacy)
    g = _load("vocoder_B2_T24.npz")
    with torch.no_grad():
        out = oracle.vocoder_forward(torch.from_numpy(g["mel"][:1]), legacy)
    assert np.abs(out.numpy() - g["wav"][:1]).max() < TOL_WAV


def test_trim_lengths():
    # SURVEY.md A.7 / fixture headers: 96 076 -> 97 902 vocoder samples, trimmed back to N
    for n in (96076, 441000, 132300, 50000, 15523):
        T = 1 + n // 441
        Tp = T + T % 2 + 4
        est = torch.zeros(1, 1, 441 * Tp)
        assert oracle.trim_center(est, n).shape[-1] == n
    assert 441 * (218 + 0 + 4) == 97902


def test_int16_truncation():
    x = np.array([[0.5, -0.5, 0.99999, -1.0, 1e-5]], dtype=np.float32)
    q = oracle.to_int16(x)
    assert q.dtype == np.int16 and list(q[0]) == [16384, -16384, 32767, -32768, 0]


def test_mode1_against_the_references_own_remove_higher_frequency(seeded_states):
    """mode1_speech_ref.npz = VoiceFixer.remove_higher_frequency / restore_inmem(mode=1) EXECUTED from the reference's
    code (base.py:87-104,121-122) through ref_shim; the oracle's restatement must reproduce both."""
    g = _load("mode1_speech_ref.npz")
    filt, cut = oracle.remove_higher_frequency(g["wav"])
    assert filt.shape == g["hf_cut"].shape == (512 * (len(g["wav"]) // 512),) and 0 < cut < 1025
    assert np.abs(filt - g["hf_cut"]).max() < 1e-6
    with torch.no_grad():
        out = oracle.restore_inmem(filt, *seeded_states)
    assert out.shape == g["restored"].shape
    assert np.sqrt(np.mean((out - g["restored"]) ** 2)) < 1e-5


def test_vocoder_oracle_against_the_references_own_front_end(seeded_states):
    """vocoder_oracle_ref.npz = Vocoder.oracle EXECUTED from the reference's code (vocoder/base.py:58-77): the
    conditioning its numpy front-end builds and the int16 frames it hands to soundfile.write."""
    from oracle import librosa_like
    g = _load("vocoder_oracle_ref.npz")
    wav = g["pcm_in"].astype(np.float32) / 32768.0
    cond = librosa_like.wav_to_cond(wav)
    T = 1 + len(wav) // 441
    assert tuple(cond.shape) == g["cond"].shape == (1, 128, T + T % 2 + 4)      # (pre() appends the tail frames)
    assert np.abs(cond.numpy() - g["cond"]).max() < 1e-5
    with torch.no_grad():
        y = oracle.vocoder_generator(cond, seeded_states[0])
    pcm = oracle.to_int16((y[0] * 2 ** 15).numpy())[0]
    assert pcm.shape == g["out_pcm"].shape
    assert np.abs(pcm.astype(np.int32) - g["out_pcm"].astype(np.int32)).max() <= 1   # truncation of ~1e-6 differences
