"""Live check of the oracle (and of the seeded-checkpoint manifests) against the
reference's own modules.  Only runs where /root/reference exists (build container)."""
import tempfile

import numpy as np
import pytest
import torch

from oracle import oracle, ref_shim

pytestmark = pytest.mark.skipif(not ref_shim.reference_available(),
                                reason="/root/reference not present")


@pytest.fixture(scope="module")
def ref_vf(seeded_states):
    vsd, rsd = seeded_states
    home = tempfile.mkdtemp(prefix="vfx_home_")
    return ref_shim.build_reference_models(home, vsd, {"generator." + k: v for k, v in rsd.items()})


def test_manifests_match_reference(ref_vf, seeded_states):
    vsd, rsd = seeded_states
    ref_r = ref_vf._model.generator.state_dict()
    ref_v = ref_vf._model.vocoder.model.state_dict()
    assert list(ref_r.keys()) == list(rsd.keys())
    assert list(ref_v.keys()) == list(vsd.keys())
    for k in rsd:
        assert torch.equal(ref_r[k], rsd[k]), k
    for k in vsd:
        assert torch.equal(ref_v[k], vsd[k]), k


@pytest.mark.parametrize("n", [4410 + 17, 44100 // 2 + 123])
def test_restore_inmem_matches_reference(ref_vf, seeded_states, n):
    vsd, rsd = seeded_states
    g = torch.Generator().manual_seed(n)
    wav = (0.15 * torch.randn(n, generator=g)).numpy().astype(np.float32)
    with torch.no_grad():
        ref = ref_vf.restore_inmem(wav, cuda=False, mode=0)
        out = oracle.restore_inmem(wav, vsd, rsd)
    assert ref.shape == out.shape == (1, n)
    assert np.sqrt(np.mean((ref - out) ** 2)) < 1e-5


def test_stft_fft_vs_conv_dft(ref_vf):
    """oracle uses an FFT, the reference (torchlibrosa) a conv-DFT: same magnitudes."""
    g = torch.Generator().manual_seed(3)
    wav = torch.randn(2, 9000, generator=g) * 0.3
    with torch.no_grad():
        sp, _, _ = ref_vf._model.f_helper.wav_to_spectrogram_phase(wav[:, None, :])
        mine = oracle.stft_mag(wav)
    assert sp.shape == (2, 1, 21, 1025)
    rel = (sp[:, 0] - mine).norm() / mine.norm()
    assert rel < 5e-6


def test_vocoder_forward_legacy_ckpt(seeded_states):
    """The reference loads legacy weight_g/weight_v checkpoints through torch's compat
    hook; both key styles must give the same generator."""
    from voicefixer_amd import weights
    legacy = weights.seeded_vocoder_state(1234, legacy=True)
    home = tempfile.mkdtemp(prefix="vfx_home_")
    vf = ref_shim.build_reference_models(
        home, legacy, {"generator." + k: v for k, v in seeded_states[1].items()})
    g = torch.Generator().manual_seed(5)
    mel = 10 ** (torch.rand((1, 1, 12, 128), generator=g) * 4 - 2)
    with torch.no_grad():
        ref = vf._model.vocoder(mel, cuda=False)
        out = oracle.vocoder_forward(mel, legacy)
    assert (ref - out).abs().max() < 2e-5
