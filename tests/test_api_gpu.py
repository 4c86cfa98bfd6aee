"""Drop-in API on the MI355X: same calls a user of the reference makes (test/test.py:45-102),
checked against the CPU oracle / goldens."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import voicefixer_amd  # noqa: E402
from voicefixer_amd import audio_io, _lib  # noqa: E402
from conftest import GOLDEN  # noqa: E402
from oracle import oracle  # noqa: E402  (checker only)


@pytest.fixture(scope="module")
def vf(seeded_states):
    return voicefixer_amd.VoiceFixer.from_state(*seeded_states)


def _rms(a, b):
    return float(np.sqrt(np.mean((np.asarray(a, np.float64) - np.asarray(b, np.float64)) ** 2)))


def test_restore_inmem_matches_golden(vf):
    g = np.load(os.path.join(GOLDEN, "restore_speech_T51.npz"))
    for cuda in (False, True):
        out = vf.restore_inmem(g["wav"], cuda=cuda, mode=0)
        assert isinstance(out, np.ndarray) and out.dtype == np.float32 and out.shape == g["restored"].shape
        assert _rms(out, g["restored"]) < 2e-5


def test_restore_file_roundtrip(vf, tmp_path):
    g = np.load(os.path.join(GOLDEN, "restore_noise_T36.npz"))
    fin, fout = str(tmp_path / "in.wav"), str(tmp_path / "out.wav")
    audio_io.save_wave(g["wav"][None], fin)
    vf.restore(input=fin, output=fout, cuda=True, mode=0)
    got = audio_io.load_wav(fout)
    # the file path quantises input AND output to PCM16: compare with the oracle on the quantised input
    with torch.no_grad():
        ref = oracle.restore_inmem(audio_io.load_wav(fin), *_states(vf))
    want = oracle.to_int16(ref)[0].astype(np.float32) / 32768.0
    assert got.shape == want.shape
    assert np.abs(got - want).max() <= 1.5 / 32768.0  # at most one PCM step from truncation of ~1e-6 differences


def _states(vf):
    return vf._vocoder._state, vf._restorer_state


def test_segmentation_31s_matches_oracle(vf):
    """> 30 s input: two hard-cut segments (1 323 000 + 44 100 samples), concatenated."""
    n = 31 * 44100
    g = torch.Generator().manual_seed(3)
    t = torch.arange(n, dtype=torch.float64) / 44100.0
    wav = (0.05 * torch.randn(n, generator=g) + 0.2 * torch.sin(2 * np.pi * 150.0 * t).float()).float().numpy()
    out = vf.restore_inmem(wav, cuda=True)
    with torch.no_grad():
        ref = oracle.restore_inmem(wav, *_states(vf))
    assert out.shape == ref.shape == (1, n)
    assert _rms(out, ref) < 2e-5


def test_long_input_segments_batched_equals_sequential(vf):
    """61 s input = two full 30 s segments + a 1 s tail.  Restoring the full segments as one batch
    must equal restoring them one by one (the reference's order), hard cuts at the same samples."""
    n = 61 * 44100
    g = torch.Generator().manual_seed(4)
    wav = (0.1 * torch.randn(n, generator=g)).numpy()
    vf.segment_batch = 8
    a = vf.restore_inmem(wav, cuda=True)
    vf.segment_batch = 1
    b = vf.restore_inmem(wav, cuda=True)
    vf.segment_batch = 8
    assert a.shape == b.shape == (1, n)
    assert _rms(a, b) < 2e-5


def test_vocoder_forward_and_plugin_hook(vf, seeded_states):
    voc = voicefixer_amd.Vocoder.from_state(seeded_states[0])
    g = np.load(os.path.join(GOLDEN, "vocoder_B2_T24.npz"))
    out = voc.forward(torch.from_numpy(g["mel"]), cuda=False)
    assert out.device.type == "cpu" and tuple(out.shape) == g["wav"].shape
    assert _rms(out.numpy(), g["wav"]) < 2e-5
    assert voc.forward(torch.from_numpy(g["mel"]), cuda=True).is_cuda
    # our vocoder as a foreign ``your_vocoder_func`` (README.md:232-254): identical to the built-in route
    gg = np.load(os.path.join(GOLDEN, "restore_noise_T36.npz"))
    a = vf.restore_inmem(gg["wav"], cuda=True)
    b = vf.restore_inmem(gg["wav"], cuda=True, your_vocoder_func=voc)
    assert _rms(a, b) < 1e-6
    # a foreign vocoder returning host tensors of a different length is trimmed like the reference does
    def fake(mel):
        assert tuple(mel.shape[:2]) == (1, 1) and mel.shape[-1] == 128
        return torch.zeros(1, 1, 441 * (mel.shape[2] + 6)) + 0.25
    c = vf.restore_inmem(gg["wav"], cuda=True, your_vocoder_func=fake)
    assert c.shape == (1, gg["wav"].shape[0]) and np.all(c == 0.25)


def test_set_math_api(vf, seeded_states):
    """The opt-in arithmetic through the public classes: same calls, results within the fp32 parity bar."""
    g = np.load(os.path.join(GOLDEN, "restore_speech_T51.npz"))
    vf.set_math("bf16x3")
    try:
        out = vf.restore_inmem(g["wav"], cuda=True, mode=0)
    finally:
        vf.set_math("f32")
    assert _rms(out, g["restored"]) < 2e-5
    with pytest.raises(ValueError):
        vf.set_math("fp8")
    voc = voicefixer_amd.Vocoder.from_state(seeded_states[0])
    voc.set_math("bf16x3")
    gv = np.load(os.path.join(GOLDEN, "vocoder_T101.npz"))
    wav = voc.forward(torch.from_numpy(gv["mel"]), cuda=False)
    assert _rms(wav.numpy(), gv["wav"]) < 2e-5


def test_restore_batch_bucketing(vf):
    g = torch.Generator().manual_seed(8)
    wavs = [(0.1 * torch.randn(n, generator=g)).numpy() for n in (20000, 30000, 20000, 25000)]
    outs = vf.restore_batch(wavs, batch_size=2)
    for w, o in zip(wavs, outs):
        assert o.shape == (1, len(w))
        single = vf.restore_inmem(w, cuda=True)
        assert _rms(o, single) < 2e-5


def test_vocoder_oracle_file(seeded_states, tmp_path):
    voc = voicefixer_amd.Vocoder.from_state(seeded_states[0])
    g = torch.Generator().manual_seed(9)
    fin, fout = str(tmp_path / "in.wav"), str(tmp_path / "o.wav")
    audio_io.save_wave((0.3 * torch.randn(1, 20000, generator=g)).numpy(), fin)
    voc.oracle(fin, fout, cuda=True)
    out = audio_io.load_wav(fout)
    T = 1 + 20000 // 441
    assert out.shape == (441 * (T + T % 2 + 4),)
    # checker: the numpy restatement of the librosa front-end (oracle_frontend, host) + the CPU oracle generator
    from voicefixer_amd import oracle_frontend
    with torch.no_grad():
        cond_ref = oracle_frontend.wav_to_cond(audio_io.load_wav(fin))
        ref = oracle.vocoder_generator(cond_ref, seeded_states[0])
    want = oracle.to_int16((ref[0] * 2 ** 15).numpy())[0].astype(np.float32) / 32768.0
    assert np.abs(out - want).max() <= 2.5 / 32768.0
    # and the device front-end alone against the host restatement
    from voicefixer_amd import ops, engine
    w = torch.from_numpy(audio_io.load_wav(fin))[None].cuda()
    mel, T2 = ops.oracle_mel(w, w.shape[1])
    cond = ops.guarded(1, 128, T2 + T2 % 2 + 4, engine.G_TILE, "cuda")
    ops.mel_to_cond_plain(mel, cond, T2)
    torch.cuda.synchronize()
    assert (cond[:, :, : cond_ref.shape[-1]].cpu() - cond_ref).abs().max() < 2e-3  # dB domain: 20*log10 amplifies ulps near the 1e-5 floor


def test_mode1_matches_oracle(vf):
    """mode 1 = high-frequency cut (device) + the mode-0 path on the shortened segment."""
    g = np.load(os.path.join(GOLDEN, "restore_speech_T51.npz"))
    wav = g["wav"]
    out = vf.restore_inmem(wav, cuda=True, mode=1)
    filt, _ = oracle.remove_higher_frequency(wav)
    with torch.no_grad():
        ref = oracle.restore_inmem(filt, *_states(vf))
    assert out.shape == ref.shape == (1, 512 * (len(wav) // 512))
    assert _rms(out, ref) < 1e-4


def test_restore_folder_matches_per_file_restore(vf, tmp_path):
    """Folder driver (voicefixer/__main__.py:176-212 semantics): every *.wav of the input folder appears
    under the same name in the output folder and equals what restore() writes for that file alone
    (two files share a length and are batched; one has another length; one is 22.05 kHz stereo)."""
    from scipy.io import wavfile
    rng = np.random.default_rng(5)
    ind, outd, single = tmp_path / "in", tmp_path / "out", tmp_path / "single"
    ind.mkdir(); single.mkdir()
    for name, n in (("a.wav", 30000), ("b.wav", 30000), ("c.wav", 41000)):
        audio_io.save_wave((0.2 * rng.standard_normal(n)).astype(np.float32)[None], str(ind / name))
    st = (0.2 * rng.standard_normal((16000, 2)) * 32767).astype(np.int16)
    wavfile.write(str(ind / "d.wav"), 22050, st)
    (ind / "notes.txt").write_text("ignored")
    files = vf.restore_folder(str(ind), str(outd), batch_size=4, io_threads=2)
    assert files == ["a.wav", "b.wav", "c.wav", "d.wav"]
    assert sorted(os.listdir(outd)) == files
    for f in files:
        vf.restore(input=str(ind / f), output=str(single / f), cuda=True, mode=0)
        sr1, x1 = wavfile.read(str(outd / f))
        sr2, x2 = wavfile.read(str(single / f))
        assert sr1 == sr2 == 44100 and x1.shape == x2.shape and x1.dtype == np.int16
        # batch-vs-single launches may pick different K-chunk depths (summation order): <= 1 LSB of PCM16
        assert np.max(np.abs(x1.astype(np.int32) - x2.astype(np.int32))) <= 1


def test_restore_stream_overlap_add(vf):
    """Overlap-add streaming (BASELINE config 5; a capability beyond the reference): outside the overlaps the
    output equals the independent restoration of that chunk, inside them it is the linear cross-fade of the two
    chunks, the callback delivers every sample exactly once and in order."""
    from voicefixer_amd.api import plan_stream_chunks
    rng = np.random.default_rng(9)
    n = 44100 * 5 + 1234
    wav = (0.2 * rng.standard_normal(n)).astype(np.float32)
    cs, os_ = 2.0, 0.25
    chunk, ov = 88200, 11025
    got_chunks = []
    out = vf.restore_stream(wav, chunk_seconds=cs, overlap_seconds=os_, batch_size=2,
                            on_chunk=lambda a, y: got_chunks.append((a, y)))
    assert out.shape == (1, n) and out.dtype == np.float32
    plan = plan_stream_chunks(n, chunk, ov)
    assert len(plan) == 3
    singles = [vf.restore_inmem(wav[a:a + l], cuda=True, mode=0) for a, l in plan]
    w = (np.arange(ov, dtype=np.float32) / ov)[None]
    for k, (a, l) in enumerate(plan):
        lo = a + (ov if k > 0 else 0)
        hi = a + l - (ov if k + 1 < len(plan) else 0)
        assert _rms(out[:, lo:hi], singles[k][:, lo - a:hi - a]) < 2e-5
        if k > 0:
            want = singles[k - 1][:, a - plan[k - 1][0]:a - plan[k - 1][0] + ov] * (1 - w) + singles[k][:, :ov] * w
            assert _rms(out[:, a:a + ov], want) < 2e-5
    pos = 0
    for a, y in got_chunks:
        assert a == pos
        assert np.array_equal(y, out[:, a:a + y.shape[1]])
        pos += y.shape[1]
    assert pos == n


def test_restore_batch_multistream_bit_reproducible(vf):
    """Ragged folder on 4 HIP streams (the restore_batch default) reproduces the single-stream results bit for bit
    in both arithmetics (several B = 1 utterances run concurrently on the chip)."""
    rng = np.random.default_rng(12)
    wavs = [(0.1 * rng.standard_normal(int(n))).astype(np.float32) for n in rng.integers(2 * 44100, 4 * 44100, size=8)]
    assert vf.math == "f32"
    for math in ("f32", "bf16x3"):   # (bf16x3: guards the -fno-slp-vectorize build)
        vf.set_math(math)
        try:
            one = vf.restore_batch(wavs, streams=1)
            for _ in range(2):
                four = vf.restore_batch(wavs, streams=4)
                assert all(np.array_equal(a, b) for a, b in zip(one, four)), math
        finally:
            vf.set_math("f32")
