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
    # checker: oracle/librosa_like.py (pinned against scipy + reference-held tables, tests/test_librosa_like.py)
    # + the CPU oracle generator
    from oracle import librosa_like
    with torch.no_grad():
        cond_ref = librosa_like.wav_to_cond(audio_io.load_wav(fin))
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


def test_vocoder_oracle_baseline_config0(seeded_states, tmp_path):
    """BASELINE configs[0]: Vocoder.oracle on a 2 s utterance of the fixture's length, N = 96 076 ->
    T = 218, T' = 222 -> 97 902 samples (the length of the reference's test/utterance/target/oracle.flac);
    a STEREO file: the reference vocodes the FIRST channel (vocoder/base.py:61), not a down-mix."""
    voc = voicefixer_amd.Vocoder.from_state(seeded_states[0])
    n = 96076
    g = torch.Generator().manual_seed(96)
    t = np.arange(n) / 44100.0
    left = (0.05 * torch.randn(n, generator=g).numpy() + 0.3 * np.sin(2 * np.pi * 180 * t) * np.sin(2 * np.pi * 1.5 * t))
    right = 0.5 * torch.randn(n, generator=g).numpy()
    fin, fout = str(tmp_path / "stereo.wav"), str(tmp_path / "o.wav")
    from scipy.io import wavfile
    wavfile.write(fin, 44100, (np.stack([left, right], 1) * 32767 / 1.6).astype(np.int16))
    voc.oracle(fin, fout, cuda=True)
    out = audio_io.load_wav(fout)
    assert out.shape == (97902,)
    from oracle import librosa_like
    first = audio_io.load_wav(fin, mono=False)[0]
    with torch.no_grad():
        ref = oracle.vocoder_generator(librosa_like.wav_to_cond(first), seeded_states[0])
    want = oracle.to_int16((ref[0] * 2 ** 15).numpy())[0].astype(np.float32) / 32768.0
    assert np.abs(out - want).max() <= 2.5 / 32768.0
    assert _rms(out, want) < 1e-3                        # north-star bound (PCM16 quantisation included)


def test_default_constructors_load_checkpoint_files(seeded_states, tmp_path, monkeypatch):
    """VoiceFixer() / Vocoder(44100) through the DEFAULT constructors (voicefixer/base.py:15-30,
    vocoder/base.py:24-32): checkpoints under ~/.cache/voicefixer, both weight-norm key styles, the
    {"generator": ...} wrapper, a full restorer.model.VoiceFixer state dict with foreign keys to filter
    (f_helper.*, mel.fb) and a ``vocoder.model.*`` override inside vf.ckpt -- bit-identical to from_state."""
    from voicefixer_amd import weights
    vsd, rsd = seeded_states
    g = np.load(os.path.join(GOLDEN, "restore_noise_T36.npz"))
    want = voicefixer_amd.VoiceFixer.from_state(vsd, rsd).restore_inmem(g["wav"], cuda=True)

    def write_home(home, voc_state, ana_state):
        a = os.path.join(home, ".cache/voicefixer/analysis_module/checkpoints")
        v = os.path.join(home, ".cache/voicefixer/synthesis_module/44100")
        os.makedirs(a)
        os.makedirs(v)
        torch.save({"generator": voc_state}, os.path.join(v, "model.ckpt-1490000_trimed.pt"))
        torch.save(ana_state, os.path.join(a, "vf.ckpt"))

    full = {"generator." + k: v for k, v in rsd.items()}
    full["mel.fb"] = torch.zeros(1025, 128)
    full["f_helper.stft.conv_real.weight"] = torch.zeros(1025, 1, 2048)
    # (1) current key style, flat vf.ckpt
    h1 = str(tmp_path / "h1")
    write_home(h1, vsd, full)
    monkeypatch.setenv("HOME", h1)
    a = voicefixer_amd.VoiceFixer().restore_inmem(g["wav"], cuda=True)
    assert np.array_equal(a, want)
    voc = voicefixer_amd.Vocoder(44100)
    gv = np.load(os.path.join(GOLDEN, "vocoder_B2_T24.npz"))
    assert np.array_equal(voc.forward(torch.from_numpy(gv["mel"])).numpy(),
                          voicefixer_amd.Vocoder.from_state(vsd).forward(torch.from_numpy(gv["mel"])).numpy())
    # (2) legacy weight_g / weight_v keys in the vocoder file, Lightning-style {"state_dict": ...} vf.ckpt
    h2 = str(tmp_path / "h2")
    write_home(h2, weights.seeded_vocoder_state(1234, legacy=True), {"state_dict": full, "epoch": 3})
    monkeypatch.setenv("HOME", h2)
    b = voicefixer_amd.VoiceFixer().restore_inmem(g["wav"], cuda=True)
    assert np.array_equal(b, want)
    # (3) vf.ckpt carries vocoder.model.* keys: they override the synthesis-module file --
    # a WRONG vocoder file plus the right weights inside vf.ckpt must still give the right answer
    h3 = str(tmp_path / "h3")
    over = dict(full)
    over.update({"vocoder.model." + k: v for k, v in vsd.items()})
    write_home(h3, weights.seeded_vocoder_state(999), over)
    monkeypatch.setenv("HOME", h3)
    c = voicefixer_amd.VoiceFixer().restore_inmem(g["wav"], cuda=True)
    assert np.array_equal(c, want)
    # ... and without the override the wrong file really is used
    h4 = str(tmp_path / "h4")
    write_home(h4, weights.seeded_vocoder_state(999), full)
    monkeypatch.setenv("HOME", h4)
    d = voicefixer_amd.VoiceFixer().restore_inmem(g["wav"], cuda=True)
    assert not np.array_equal(d, want)
    # missing files keep the reference's error strings
    monkeypatch.setenv("HOME", str(tmp_path / "empty"))
    with pytest.raises(RuntimeError, match="Error 0"):
        voicefixer_amd.VoiceFixer()
    with pytest.raises(RuntimeError, match="Error 1"):
        voicefixer_amd.Vocoder(44100)


def test_plugin_vocoder_shorter_than_segment(vf):
    """_trim_center's other branch (base.py:71-76): an estimate SHORTER than the segment is returned as it is."""
    gg = np.load(os.path.join(GOLDEN, "restore_noise_T36.npz"))
    n = gg["wav"].shape[0]

    def short(mel):
        return torch.full((1, 1, n - 1000), 0.5)
    c = vf.restore_inmem(gg["wav"], cuda=True, your_vocoder_func=short)
    assert c.shape == (1, n - 1000) and np.all(c == 0.5)

    def loud(mel):  # peak rule (base.py:131-133) on a short estimate
        return torch.full((1, 1, n - 7), -2.0)
    c = vf.restore_inmem(gg["wav"], cuda=True, your_vocoder_func=loud)
    assert c.shape == (1, n - 7) and np.all(c == -1.0)


def test_gru_error_flag_is_read_on_the_product_path(vf):
    """The two-CU GRU's device flag (partner workgroup missed the bounded spin) is read on every API that returns host
    data.  Since round 3 a raised flag no longer discards the call: the call is issued again with the recurrences on the
    one-workgroup kernel (vfx_gru_bidir_f32, nothing to miss) and the caller gets the right waveform."""
    gg = np.load(os.path.join(GOLDEN, "restore_noise_T36.npz"))
    pipe = vf._get_pipe()
    vf.restore_inmem(gg["wav"], cuda=True)          # allocates the flag
    assert int(pipe.restorer.gru_err.item()) == 0
    retries = getattr(pipe, "gru_retries", 0)
    pipe.restorer._force_gru_miss = 1                # the next two-CU launch reports a missed hand-off
    out = vf.restore_inmem(gg["wav"], cuda=True)
    assert pipe.gru_retries == retries + 1 and int(pipe.restorer.gru_err.item()) == 0 and not pipe.restorer.gru_single
    assert _rms(out, gg["restored"]) < 2e-5          # the re-run's answer, against the reference-generated golden
    pipe.restorer._force_gru_miss = 1
    outs = vf.restore_batch([gg["wav"], gg["wav"][:9000]])
    assert pipe.gru_retries == retries + 2 and int(pipe.restorer.gru_err.item()) == 0
    assert _rms(outs[0], gg["restored"]) < 2e-5 and outs[1].shape == (1, 9000)
    # a flag that is ALREADY set when a call starts is an earlier, unchecked launch's: the raw check still raises
    pipe.restorer.gru_err.fill_(1)
    with pytest.raises(_lib.VfxError, match="gru"):
        pipe.check()
    assert int(pipe.restorer.gru_err.item()) == 0
    out = vf.restore_inmem(gg["wav"], cuda=True)     # healthy again, two-CU kernel back
    assert _rms(out, gg["restored"]) < 2e-5


def test_one_workgroup_gru_path_equals_the_two_cu_path(vf):
    """The fallback itself: the whole path with ``gru_single`` against the default, ragged rows included."""
    pipe = vf._get_pipe()
    g = torch.Generator().manual_seed(41)
    wav = (0.1 * torch.randn(3, 30000, generator=g)).cuda()
    a = pipe.restore(wav, 30000).clone()
    ra = pipe.restore_rows(wav, [30000, 21000, 25555]).clone()
    pipe.restorer.gru_single = True
    try:
        b = pipe.restore(wav, 30000).clone()
        rb = pipe.restore_rows(wav, [30000, 21000, 25555]).clone()
    finally:
        pipe.restorer.gru_single = False
    pipe.check()
    assert float((a - b).abs().max()) < 2e-5 and float((ra - rb).abs().max()) < 2e-5


def test_four_streams_of_batch32_keep_the_gru_resident(vf):
    """restore_batch's default: 4 streams; 8 buckets of 32 equal-length utterances -> four concurrent launch
    sequences.  The GRU launches are sized per stream count (Pipeline.set_streams) so that every two-CU pair is
    resident: the flag stays 0 and the results equal the single-stream ones bit for bit."""
    g = torch.Generator().manual_seed(77)
    lens = [4410 + 441 * k for k in range(8)]
    wavs = [(0.1 * torch.randn(n, generator=g)).numpy() for n in lens for _ in range(32)]
    pipe = vf._get_pipe()
    a = vf.restore_batch(wavs, batch_size=32, streams=4)
    assert int(pipe.restorer.gru_err.item()) == 0
    assert pipe.restorer.gru_group == 60             # back to the single-stream launch size
    b = vf.restore_batch(wavs, batch_size=32, streams=1)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))


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


def test_folder_job_survives_bad_files_on_the_device(vf, tmp_path):
    """Round-4 review item 2 on the REAL device stage (tests/test_dist_cpu.py runs the same folder on two gloo ranks with a stub):
    a header cut off before its data chunk, a 500-sample file, a file whose decoder raises and a file shorter than its header
    says sit between good files -- every good file is written, the short-data one at its real length and equal to restoring
    those samples alone, the three bad ones are listed with a reason, nothing half-written is left, and a second run with
    skip_existing touches nothing."""
    import importlib.util
    import warnings
    from scipy.io import wavfile
    spec = importlib.util.spec_from_file_location("tdc", os.path.join(os.path.dirname(os.path.abspath(__file__)), "test_dist_cpu.py"))
    tdc = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tdc)
    ind, outd, single = str(tmp_path / "in"), str(tmp_path / "out"), tmp_path / "single"
    single.mkdir()
    lens = tdc._make_ragged_folder(ind, 7, seed=11)
    bad, trunc = tdc._add_bad_files(ind)
    good = dict(lens, **trunc)
    st = {}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")          # (scipy warns about the truncated data chunk)
        names = vf.restore_folder(ind, outd, batch_size=4, io_threads=2, stats=st)
        assert sorted(names) == sorted(good) == sorted(os.listdir(outd))        # (no .part-* leftovers either)
        assert sorted(n for n, _ in st["failed"]) == sorted(bad) and all(why for _, why in st["failed"])
        assert st["truncated"] == [("truncated.wav", 4000, 2900)]
        for name, n in good.items():
            sr, y = wavfile.read(os.path.join(outd, name))
            assert sr == 44100 and y.shape == (n,) and y.dtype == np.int16, name
        x = audio_io.load_wav(os.path.join(ind, "truncated.wav"))
        assert x.shape == (2900,)
        audio_io.save_wave(vf.restore_inmem(x, cuda=True), str(single / "t.wav"))
        a = wavfile.read(os.path.join(outd, "truncated.wav"))[1].astype(np.int32)
        b = wavfile.read(str(single / "t.wav"))[1].astype(np.int32)
        assert np.max(np.abs(a - b)) <= 1
        before = {f: os.path.getmtime(os.path.join(outd, f)) for f in os.listdir(outd)}
        st2 = {}
        assert vf.restore_folder(ind, outd, batch_size=4, io_threads=2, stats=st2, skip_existing=True) == []
        assert sorted(st2["skipped"]) == sorted(good) and sorted(n for n, _ in st2["failed"]) == sorted(bad)
        assert before == {f: os.path.getmtime(os.path.join(outd, f)) for f in os.listdir(outd)}


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


def test_rccl_scatter_restore_gather_world1(vf, tmp_path):
    """The multi-GPU job (BASELINE configs[3]) on the REAL backend: torch.distributed "nccl" (= RCCL) with
    world_size 1 on this box's single GPU -- scatter_utterances / Pipeline.restore / gather_utterances run
    through the RCCL code path (broadcast of the job size, device tensors) and return what a plain batched
    restore returns.  (world_size 2 runs on gloo in tests/test_dist_cpu.py; the 8-GPU run is the driver's.)"""
    import torch.distributed as dist
    from voicefixer_amd import dist as vdist
    pipe = vf._get_pipe()
    n = 13230
    g = torch.Generator().manual_seed(21)
    wavs = (0.1 * torch.randn(5, n, generator=g)).cuda()
    dist.init_process_group(backend="nccl", init_method="file://" + str(tmp_path / "rdzv"), world_size=1, rank=0,
                            device_id=torch.device("cuda", torch.cuda.current_device()))
    try:
        assert dist.get_backend() == "nccl"
        timing = {}
        out = vdist.restore_sharded(lambda w: pipe.restore(w, n), wavs, n, wavs.device, batch_size=2, timing=timing)
        torch.cuda.synchronize()
    finally:
        dist.destroy_process_group()
    assert tuple(out.shape) == (5, n) and set(timing) == {"scatter_s", "compute_s", "gather_s"}
    want = torch.cat([pipe.restore(wavs[i:i + 2], n) for i in range(0, 5, 2)], 0)
    assert torch.equal(out, want)
    pipe.check()


def test_hip_graph_replay_is_bit_identical(vf):
    """Pipeline.enable_graphs: the ~300 launches of one (B, N) shape captured once and replayed -- same bits as the
    eager path, for a repeated shape, a second shape, and after eviction (max_shapes = 1)."""
    pipe = vf._get_pipe()
    g = torch.Generator().manual_seed(31)
    a = (0.1 * torch.randn(2, 12345, generator=g)).cuda()
    b = (0.1 * torch.randn(1, 7000, generator=g)).cuda()
    ea, eb = pipe.restore(a, 12345).clone(), pipe.restore(b, 7000).clone()
    before = _lib.lib().vfx_launch_count()
    pipe.enable_graphs(max_shapes=1, max_batch=2)
    try:
        r1 = pipe.restore(a, 12345)                  # capture (+ two eager warm-up passes) and first replay
        n_capture = _lib.lib().vfx_launch_count() - before
        r2 = pipe.restore(a * 1.0, 12345)            # pure replay: no new launches issued by the library
        assert _lib.lib().vfx_launch_count() - before == n_capture
        r3 = pipe.restore(b, 7000)                   # second shape evicts the first
        r4 = pipe.restore(a, 12345)                  # re-captured
        torch.cuda.synchronize()
        assert torch.equal(r1, ea) and torch.equal(r2, ea) and torch.equal(r3, eb) and torch.equal(r4, ea)
        big = (0.1 * torch.randn(3, 7000, generator=g)).cuda()   # above max_batch: eager
        assert pipe.restore(big, 7000).shape == (3, 7000)
    finally:
        pipe.disable_graphs()
    pipe.check()


def test_restore_batch_ragged_rows(vf):
    """Utterances of DIFFERENT lengths run as ONE launch sequence with per-row lengths in every length-dependent
    kernel (STFT reflect, GRU reverse start, conv zero/reflect padding, UNet map height, peak rule, centre trim):
    every row must equal the restoration of that utterance alone (fp32 tile-shape noise only) and the CPU oracle."""
    g = torch.Generator().manual_seed(41)
    T = 37
    lens = [441 * (T - 1), 441 * (T - 1) + 1, 441 * (T - 1) + 220, 441 * T - 1, 441 * (T - 1) + 220]   # all T = 37
    lens += [441 * T, 441 * T + 5]                                                                   # T = 38
    lens += [441 * 44 + 17, 441 * 47]                                                                # T = 45, 48
    wavs = [(0.1 * torch.randn(n, generator=g)).numpy() for n in lens]
    before = _lib.lib().vfx_launch_count()
    outs = vf.restore_batch(wavs, batch_size=16, streams=2)
    launches = _lib.lib().vfx_launch_count() - before
    single = [vf.restore_inmem(w, cuda=True) for w in wavs]
    per_utt = (_lib.lib().vfx_launch_count() - before - launches) / len(wavs)
    assert launches < 1.5 * per_utt                      # one ragged batch, not nine
    for w, o, s1 in zip(wavs, outs, single):
        assert o.shape == (1, len(w)) and _rms(o, s1) < 2e-5      # (batch 9 and batch 1 pick different tile shapes)
    with torch.no_grad():
        ref = oracle.restore_inmem(wavs[2], *_states(vf))
    assert _rms(outs[2], ref) < 2e-5
    with pytest.raises(_lib.VfxError):
        vf._get_pipe().restore_rows(torch.zeros((2, 441 * 40), device="cuda"), [441 * 36, 1000])   # < 1025 samples


def test_ragged_batches_never_read_uninitialised_memory(vf, monkeypatch):
    """Round-4 review: the mode-1 ragged path builds its cut rows in a torch.empty buffer, and the staging / workspace tensors of the
    whole path are torch.empty too -- correct only if no kernel ever reads past a row's own end (vectorised tail loads included).
    With EVERY floating-point torch.empty of the process returning NaN-filled memory -- pinned staging rows, the cut buffer, every
    activation buffer and workspace of the engine -- ragged batches of mode 0 and mode 1 must come back bit-identical."""
    g = torch.Generator().manual_seed(77)
    lens = [441 * 36 + 3, 441 * 36 + 221, 441 * 44 + 17, 441 * 47, 441 * 40 + 1]
    wavs = [(0.1 * torch.randn(n, generator=g)).numpy() for n in lens]
    want0 = vf.restore_batch(wavs, batch_size=8, streams=1)
    want1 = vf.restore_batch(wavs, batch_size=8, streams=1, mode=1)
    real_empty = torch.empty

    def nan_empty(*a, **k):
        t = real_empty(*a, **k)
        if t.is_floating_point() and t.numel():
            t.fill_(float("nan"))
        return t

    monkeypatch.setattr(torch, "empty", nan_empty)
    got0 = vf.restore_batch(wavs, batch_size=8, streams=1)
    got1 = vf.restore_batch(wavs, batch_size=8, streams=1, mode=1)
    monkeypatch.undo()
    for w, a, b in zip(want0 + want1, got0 + got1, lens + lens):
        assert np.isfinite(a).all()
        assert np.array_equal(w, a)


def test_ragged_rows_cross_unet_and_tile_boundaries(vf):
    """Row lengths on both sides of the ResUNet's 64-frame padding (T = 64 | 65 -> maps of 64 | 128 rows), of an odd /
    even frame count (vocoder tail T%2) and of the 256-column conv tiles; a row of 0.6 x the longest.  Each row vs
    the utterance alone and vs the oracle; then the two exactness properties that do not depend on tile shapes:
    (1) equal-length rows pushed through the ragged kernels == the plain batch bit for bit;
    (2) a row does not change by a single bit when the OTHER rows of the batch change content and length."""
    pipe = vf._get_pipe()
    g = torch.Generator().manual_seed(43)
    lens = [441 * 63 + 3, 441 * 64 + 100, 441 * 100, 441 * 39 + 440, 441 * 104 + 7]   # T = 64, 65, 101, 40, 105
    n_max = max(lens)
    wav = torch.zeros((len(lens), n_max))
    for r, n in enumerate(lens):
        wav[r, :n] = 0.1 * torch.randn(n, generator=g)
    wav[3, lens[3]:] = 7.0      # whatever lies past a row's end in the input buffer must not matter
    wav = wav.cuda()
    out = pipe.restore_rows(wav, lens)
    assert out.shape == (len(lens), n_max)
    worst = 0.0
    for r, n in enumerate(lens):
        alone = pipe.restore(wav[r:r + 1, :n].contiguous(), n)[0]
        e = _rms(out[r, :n].cpu().numpy(), alone.cpu().numpy())
        worst = max(worst, e)
        assert e < 2e-5, (r, n, e)
        assert float(out[r, n:].abs().max()) == 0.0 if n < n_max else True
    with torch.no_grad():
        ref = oracle.restore_inmem(wav[3, :lens[3]].cpu().numpy(), *_states(vf))
    assert _rms(out[3:4, :lens[3]].cpu().numpy(), ref) < 2e-5
    # (1) the ragged kernels on rows of full length == the plain batched kernels
    eq = wav[:, :lens[0]].contiguous()
    assert torch.equal(pipe.restore_rows(eq, [lens[0]] * len(lens), force_ragged=True), pipe.restore(eq, lens[0]))
    # (2) rows 0 and 4 are unchanged when rows 1..3 become other utterances of other lengths
    lens2 = [lens[0], 441 * 80 + 9, 441 * 30, 441 * 99 + 1, lens[4]]
    wav2 = wav.clone()
    for r in (1, 2, 3):
        wav2[r] = 0.0
        wav2[r, :lens2[r]] = 0.1 * torch.randn(lens2[r], generator=g).cuda()
    out2 = pipe.restore_rows(wav2, lens2)
    assert torch.equal(out2[0], out[0]) and torch.equal(out2[4], out[4])
    assert not torch.equal(out2[2], out[2])
    pipe.check()
    print("ragged rows vs alone: worst rms error %.3g" % worst)


def test_ragged_rows_at_folder_sizes(vf):
    """Rows of 3-7 s: the launches are large enough for the second-generation kernels (convw, fused ResStack layer),
    a shorter row ends inside tiles that are INTERIOR for the longest row (the reflect-padded vocoder input conv must
    mirror there, not zero), and the wide dilations (3^7) reach across row ends."""
    pipe = vf._get_pipe()
    g = torch.Generator().manual_seed(47)
    lens = [441 * 300 + 11, 441 * 520 + 200, 441 * 710 + 5, 441 * 389]
    wav = torch.zeros((len(lens), max(lens)))
    for r, n in enumerate(lens):
        wav[r, :n] = 0.1 * torch.randn(n, generator=g)
    wav = wav.cuda()
    out = pipe.restore_rows(wav, lens)
    for r, n in enumerate(lens):
        alone = pipe.restore(wav[r:r + 1, :n].contiguous(), n)[0]
        e = _rms(out[r, :n].cpu().numpy(), alone.cpu().numpy())
        assert e < 2e-5, (r, n, e)
    pipe.check()


def test_model_handle_forward_is_the_restorer(vf):
    """``vf._model(sp, mel)`` (restorer/model.py:102-120): log-mel of the restored spectrogram, [B, 1, T, 128], equal to what
    the pipeline computes between the STFT and the vocoder; ``sp`` is ignored as in the reference."""
    g = torch.Generator().manual_seed(5)
    wav = (0.1 * torch.randn(2, 22050, generator=g)).cuda()
    pipe = vf._get_pipe()
    mel, T = pipe.wav_to_mel(wav, 22050)
    logmel, den = pipe.restorer.forward(mel, T)
    out = vf._model(None, mel[:, None].cpu())
    assert set(out) >= {"mel", "clean", "noisy", "unet_out"} and out["mel"].shape == (2, 1, T, 128)
    assert not out["mel"].is_cuda                                       # follows the handle's placeholder device
    assert torch.equal(out["mel"][:, 0], logmel.cpu())
    assert torch.allclose(10 ** torch.clamp(out["mel"][:, 0], max=5), den.cpu(), rtol=1e-5)   # from_log (pytorch_util.py:24-27)
    vf._model.to("cuda")
    assert vf._model(None, mel[:, None])["mel"].is_cuda
    vf._model.to("cpu")


# ---------------------------------------------------------------------------------------------------------------------
# round 3: fixtures EXECUTED from the reference's own code for the librosa legs, CLI, mode 1 everywhere, FLAC files,
# graph replay next to the multi-stream batch driver
def test_mode1_matches_the_reference_executed_fixture(vf):
    """mode1_speech_ref.npz: VoiceFixer.restore_inmem(mode=1) run from the reference's code (base.py:87-104,107-139
    through oracle/ref_shim with librosa's transforms bound to oracle/librosa_like) -- not the oracle's restatement."""
    g = np.load(os.path.join(GOLDEN, "mode1_speech_ref.npz"))
    out = vf.restore_inmem(g["wav"], cuda=True, mode=1)
    assert out.shape == g["restored"].shape == (1, 512 * (len(g["wav"]) // 512))
    assert _rms(out, g["restored"]) < 2e-5


def test_vocoder_oracle_matches_the_reference_executed_fixture(seeded_states, tmp_path):
    """vocoder_oracle_ref.npz: Vocoder.oracle run from the reference's code (vocoder/base.py:58-77) on 1 s of its own
    fixture p360_001_mic1.flac; the int16 frames it handed to soundfile.write are the known answer -- through a WAV and
    through a FLAC output file."""
    from scipy.io import wavfile
    g = np.load(os.path.join(GOLDEN, "vocoder_oracle_ref.npz"))
    voc = voicefixer_amd.Vocoder.from_state(seeded_states[0])
    fin = str(tmp_path / "in.wav")
    wavfile.write(fin, 44100, g["pcm_in"])
    for ext in (".wav", ".flac"):
        fout = str(tmp_path / ("o" + ext))
        voc.oracle(fin, fout, cuda=True)
        got = np.round(audio_io.load_wav(fout) * 32768.0).astype(np.int32)
        assert got.shape == g["out_pcm"].shape
        # int16 truncation turns a ~1e-6 float difference into at most one step (two where the device front-end's
        # dB-domain conditioning differs by an ulp near a clip edge)
        assert np.abs(got - g["out_pcm"].astype(np.int32)).max() <= 2
        assert np.mean(np.abs(got - g["out_pcm"].astype(np.int32))) < 0.2


def test_restore_reads_and_writes_flac(vf, tmp_path):
    """test/test.py:45-75 feeds restore() a .flac and asks for a .flac: both ends now exist (voicefixer_amd/flac.py)."""
    src = os.path.join(GOLDEN, "ref_utterance", "original_original.flac")
    fout_flac, fout_wav = str(tmp_path / "o.flac"), str(tmp_path / "o.wav")
    vf.restore(input=src, output=fout_flac, cuda=True, mode=0)
    vf.restore(input=src, output=fout_wav, cuda=True, mode=0)
    a, b = audio_io.load_wav(fout_flac), audio_io.load_wav(fout_wav)
    assert a.shape == (132300,) and np.array_equal(a, b)          # the same PCM16 through either container
    with torch.no_grad():
        ref = oracle.restore_inmem(audio_io.load_wav(src)[:66150], *_states(vf))   # first 1.5 s against the oracle
    out = vf.restore_inmem(audio_io.load_wav(src)[:66150], cuda=True)
    assert _rms(out, ref) < 2e-5


def _seeded_home(tmp_path, seeded_states, monkeypatch):
    vsd, rsd = seeded_states
    home = str(tmp_path / "home")
    a = os.path.join(home, ".cache/voicefixer/analysis_module/checkpoints")
    v = os.path.join(home, ".cache/voicefixer/synthesis_module/44100")
    os.makedirs(a)
    os.makedirs(v)
    torch.save({"generator": vsd}, os.path.join(v, "model.ckpt-1490000_trimed.pt"))
    torch.save({"generator." + k: t for k, t in rsd.items()}, os.path.join(a, "vf.ckpt"))
    monkeypatch.setenv("HOME", home)


def test_cli_file_and_folder_equal_per_file_restore(vf, seeded_states, tmp_path, monkeypatch, capsys):
    """``python -m voicefixer_amd`` (voicefixer/__main__.py:69-215): -i/-o on one file, -ifdr/-ofdr on a folder, --mode 1
    and --mode all; every output equals what VoiceFixer.restore() writes for that file and mode alone."""
    from scipy.io import wavfile
    from voicefixer_amd import __main__ as cli
    _seeded_home(tmp_path, seeded_states, monkeypatch)
    rng = np.random.default_rng(77)
    ind, single = tmp_path / "in", tmp_path / "single"
    ind.mkdir(); single.mkdir()
    t = np.arange(40000) / 44100.0
    for name, n in (("a.wav", 30000), ("b.wav", 36000), ("c.wav", 40000)):
        x = 0.05 * rng.standard_normal(n) + 0.3 * np.sin(2 * np.pi * 300 * t[:n])
        audio_io.save_wave(x.astype(np.float32)[None], str(ind / name))

    def same(f1, f2):
        x1, x2 = wavfile.read(f1)[1], wavfile.read(f2)[1]
        assert x1.shape == x2.shape and np.max(np.abs(x1.astype(np.int32) - x2.astype(np.int32))) <= 1

    # one file, mode 0, FLAC output
    assert cli.main(["-i", str(ind / "a.wav"), "-o", str(tmp_path / "o" / "a.flac"), "--silent"]) == 0
    vf.restore(input=str(ind / "a.wav"), output=str(single / "a0.wav"), cuda=True, mode=0)
    assert np.max(np.abs(audio_io.load_wav(str(tmp_path / "o" / "a.flac")) - audio_io.load_wav(str(single / "a0.wav")))) <= 1.01 / 32768
    # folder, mode 1
    assert cli.main(["-ifdr", str(ind), "-ofdr", str(tmp_path / "f1"), "--mode", "1", "--silent", "--batch-size", "2"]) == 0
    assert sorted(os.listdir(tmp_path / "f1")) == ["a.wav", "b.wav", "c.wav"]
    for f in ("a.wav", "b.wav", "c.wav"):
        vf.restore(input=str(ind / f), output=str(single / ("m1_" + f)), cuda=True, mode=1)
        same(str(tmp_path / "f1" / f), str(single / ("m1_" + f)))
        assert wavfile.read(str(tmp_path / "f1" / f))[1].shape[0] == 512 * (wavfile.read(str(ind / f))[1].shape[0] // 512)
    # folder, all built modes: <name>-mode<k>.wav (__main__.py:13-18)
    assert cli.main(["-ifdr", str(ind), "-ofdr", str(tmp_path / "fa"), "--mode", "all"]) == 0
    assert "mode 2 is not built" in capsys.readouterr().out
    assert sorted(os.listdir(tmp_path / "fa")) == sorted("%s-mode%d.wav" % (b, m) for b in "abc" for m in (0, 1))
    vf.restore(input=str(ind / "b.wav"), output=str(single / "b0.wav"), cuda=True, mode=0)
    same(str(tmp_path / "fa" / "b-mode0.wav"), str(single / "b0.wav"))
    same(str(tmp_path / "fa" / "b-mode1.wav"), str(single / "m1_b.wav"))


def test_restore_stream_mode1(vf):
    """mode 1 in the overlap-add streaming driver: chunks are multiples of 512 samples, every chunk is pre-filtered on its
    own (base.py:121-122 applies the cut per segment), the output loses only the last chunk's sub-512 tail."""
    from voicefixer_amd.api import plan_stream_chunks
    rng = np.random.default_rng(19)
    n = 44100 * 4 + 999
    t = np.arange(n) / 44100.0
    wav = (0.05 * rng.standard_normal(n) + 0.3 * np.sin(2 * np.pi * 250 * t)).astype(np.float32)
    chunk, ov = 88200 - 88200 % 512, 11025
    out = vf.restore_stream(wav, chunk_seconds=2.0, overlap_seconds=0.25, batch_size=2, mode=1)
    plan = plan_stream_chunks(n, chunk, ov, 1535)
    a_last, l_last = plan[-1]
    assert out.shape == (1, a_last + 512 * (l_last // 512))
    singles = [vf.restore_inmem(wav[a:a + l], cuda=True, mode=1) for a, l in plan]
    for k, (a, l) in enumerate(plan):
        lo = a + (ov if k > 0 else 0)
        hi = a + singles[k].shape[1] - (ov if k + 1 < len(plan) else 0)
        assert _rms(out[:, lo:hi], singles[k][:, lo - a:hi - a]) < 2e-5


def test_graph_replay_is_bypassed_while_several_streams_issue_batches(vf):
    """after enable_graphs(), restore_batch's multi-segment buckets went through Pipeline.restore on side
    streams and shared ONE static input / output pair per (1, 30 s) shape.  With more than one stream active the replay
    path is now bypassed; results equal the plain run."""
    pipe = vf._get_pipe()
    rng = np.random.default_rng(23)
    n = 44100 * 30 + 5000                                  # two segments per file: the "samples" bucket kind
    wavs = [(0.1 * rng.standard_normal(n)).astype(np.float32) for _ in range(3)]
    wavs[1] = wavs[1][:n - 7]                              # a different length -> its own bucket, on the other stream
    want = vf.restore_batch(wavs, batch_size=1, streams=2)
    before = _lib.lib().vfx_launch_count()
    pipe.enable_graphs(max_shapes=2, max_batch=1)
    try:
        got = vf.restore_batch(wavs, batch_size=1, streams=2)
        assert _lib.lib().vfx_launch_count() - before > 6 * 250     # eager launches, not replays
        one = pipe.restore(torch.from_numpy(wavs[0][None, :44100]).cuda(), 44100)   # single stream again: graphs serve
        assert one.shape == (1, 44100) and len(pipe._graphs) == 1
    finally:
        pipe.disable_graphs()
    for a, b in zip(want, got):
        assert np.array_equal(a, b)


def test_restore_batch_recovers_after_a_failing_batch(vf):
    """a batch that raises (a file too short for the reflect-padded STFT) must leave the pipeline as it
    found it: single-stream GRU launch size, no pending device flag, the next call works."""
    pipe = vf._get_pipe()
    rng = np.random.default_rng(29)
    good = [(0.1 * rng.standard_normal(20000)).astype(np.float32) for _ in range(3)]
    with pytest.raises(_lib.VfxError):
        vf.restore_batch(good + [np.zeros(500, np.float32)], batch_size=2, streams=2)
    assert pipe.restorer.gru_group == min(60, 256 // 4) and getattr(pipe, "_n_streams", 1) == 1
    outs = vf.restore_batch(good, batch_size=2, streams=2)
    assert all(o.shape == (1, 20000) and np.isfinite(o).all() for o in outs)


def test_folder_with_a_24bit_wav(vf, tmp_path):
    """one 24-bit PCM file (scipy's mmap reader refuses it) must not abort the folder."""
    import struct
    ind, outd = tmp_path / "in", tmp_path / "out"
    ind.mkdir()
    rng = np.random.default_rng(31)
    audio_io.save_wave((0.2 * rng.standard_normal(25000)).astype(np.float32)[None], str(ind / "a.wav"))
    vals = (0.2 * rng.standard_normal(26000) * (1 << 23)).astype(np.int64).clip(-(1 << 23), (1 << 23) - 1)
    raw = b"".join(int(v).to_bytes(3, "little", signed=True) for v in vals)
    fmt = struct.pack("<HHIIHH", 1, 1, 44100, 44100 * 3, 3, 24)
    with open(ind / "b.wav", "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", 4 + 8 + len(fmt) + 8 + len(raw)) + b"WAVE" + b"fmt " + struct.pack("<I", len(fmt)) + fmt)
        f.write(b"data" + struct.pack("<I", len(raw)) + raw)
    assert vf.restore_folder(str(ind), str(outd), batch_size=4, io_threads=2) == ["a.wav", "b.wav"]
    assert audio_io.wav_length(str(outd / "b.wav")) == 26000


def test_gru_retry_with_graphs_enabled_runs_eager(vf):
    """with enable_graphs() the re-run after a missed GRU hand-off used to REPLAY the captured graph (two-CU
    kernel baked in), and a shape first seen during a re-run was captured with the one-workgroup kernel for good.  Now
    the replay path is bypassed while ``gru_single`` is set: the re-run is eager, nothing is captured in that state."""
    gg = np.load(os.path.join(GOLDEN, "restore_noise_T36.npz"))
    pipe = vf._get_pipe()
    pipe.enable_graphs(max_shapes=2, max_batch=1)
    try:
        # (a) the shape is captured; a call that ends with the flag raised must re-run EAGERLY (a whole pass of launches),
        # not replay the graph with the two-CU kernel baked in
        vf.restore_inmem(gg["wav"], cuda=True)
        assert len(pipe._graphs) == 1
        before = _lib.lib().vfx_launch_count()
        vf.restore_inmem(gg["wav"], cuda=True)
        replay_launches = _lib.lib().vfx_launch_count() - before
        retries = getattr(pipe, "gru_retries", 0)
        pipe.restorer.gru_err.fill_(1)             # what a missed hand-off inside the replayed graph leaves behind
        before = _lib.lib().vfx_launch_count()
        out = vf.restore_inmem(gg["wav"], cuda=True)
        assert pipe.gru_retries == retries + 1 and not pipe.restorer.gru_single
        assert _rms(out, gg["restored"]) < 2e-5
        assert _lib.lib().vfx_launch_count() - before >= replay_launches + 250
        # (b) the entry is untouched and serves the next call
        assert len(pipe._graphs) == 1
        again = vf.restore_inmem(gg["wav"], cuda=True)
        assert pipe.gru_retries == retries + 1 and _rms(again, gg["restored"]) < 2e-5
        # (c) a shape FIRST seen during a re-run is not captured
        pipe.restorer.gru_single = True
        try:
            n_graphs = len(pipe._graphs)
            pipe.restore(torch.from_numpy(gg["wav"][None, :12000]).cuda(), 12000)
            assert len(pipe._graphs) == n_graphs
        finally:
            pipe.restorer.gru_single = False
    finally:
        pipe.disable_graphs()
    assert int(pipe.restorer.gru_err.item()) == 0


def test_restore_batches_generator_order_and_gru_miss(vf):
    """The streaming device stage: results come back in submission order with the submitted tags, whatever stream a batch
    ran on; a missed GRU hand-off anywhere among the batches in flight re-issues all of them on the one-workgroup
    kernel; an abandoned generator leaves the pipeline in its single-stream state."""
    pipe = vf._get_pipe()
    rng = np.random.default_rng(37)
    lens = [[20000, 21000], [30000, 25000, 28000], [16000], [22050, 22050]]
    items, want = [], []
    for k, ls in enumerate(lens):
        host = torch.zeros((len(ls), max(ls)), dtype=torch.float32, pin_memory=True)
        for r, n in enumerate(ls):
            host[r, :n] = torch.from_numpy((0.1 * rng.standard_normal(n)).astype(np.float32))
        items.append(("tag%d" % k, "ragged", host, ls))
        want.append([vf.restore_inmem(host[r, :n].numpy(), cuda=True) for r, n in enumerate(ls)])

    def check(results):
        assert [t for t, _, _ in results] == ["tag%d" % k for k in range(len(lens))]
        for (tag, out_host, lens_out), ls, w in zip(results, lens, want):
            assert list(lens_out) == ls and out_host.is_pinned()
            for r, n in enumerate(ls):
                assert _rms(out_host[r, :n].numpy(), w[r][0]) < 2e-5

    check(list(vf.restore_batches(iter(items), streams=2)))
    retries = getattr(pipe, "gru_retries", 0)
    pipe.restorer._force_gru_miss = 1
    check(list(vf.restore_batches(iter(items), streams=2)))
    assert pipe.gru_retries == retries + 1 and int(pipe.restorer.gru_err.item()) == 0 and not pipe.restorer.gru_single
    gen = vf.restore_batches(iter(items), streams=2)
    next(gen)
    gen.close()
    assert getattr(pipe, "_n_streams", 1) == 1 and int(pipe.restorer.gru_err.item()) == 0
    check(list(vf.restore_batches(iter(items), streams=1)))


def test_restore_folder_two_ranks_in_one_process_equal_the_unsharded_folder(vf, tmp_path):
    """rank= / world= on the real device path: the two halves dist.deal_files deals are disjoint, cover the folder, and every
    file equals (to one PCM16 step: other batch shapes) what the unsharded folder run writes; counters add up."""
    from scipy.io import wavfile
    rng = np.random.default_rng(43)
    ind = tmp_path / "in"
    ind.mkdir()
    for k in range(7):
        n = int(rng.integers(12000, 40000))
        audio_io.save_wave((0.2 * rng.standard_normal(n)).astype(np.float32)[None], str(ind / ("f%d.wav" % k)))
    whole = vf.restore_folder(str(ind), str(tmp_path / "whole"), batch_size=3, io_threads=2)
    s0, s1 = {}, {}
    a = vf.restore_folder(str(ind), str(tmp_path / "sharded"), batch_size=3, io_threads=2, rank=0, world=2, stats=s0)
    b = vf.restore_folder(str(ind), str(tmp_path / "sharded"), batch_size=3, io_threads=2, rank=1, world=2, stats=s1)
    assert not (set(a) & set(b)) and sorted(a + b) == whole == sorted(os.listdir(tmp_path / "sharded"))
    assert s0["files"] + s1["files"] == 7 and abs(s0["audio_s"] - s1["audio_s"]) <= 40000 / 44100.0
    assert s0["decode_worker_s"] > 0 and s0["encode_worker_s"] > 0 and s0["wall_s"] > 0
    for f in whole:
        _, x1 = wavfile.read(str(tmp_path / "whole" / f))
        _, x2 = wavfile.read(str(tmp_path / "sharded" / f))
        assert x1.shape == x2.shape and np.max(np.abs(x1.astype(np.int32) - x2.astype(np.int32))) <= 1


def test_cli_folder_under_the_launcher_on_rccl(vf, seeded_states, tmp_path, monkeypatch):
    """``python -m voicefixer_amd -ifdr .. --gpus N``: (a) started plain with --gpus 4 on this one-GPU box it clamps (loudly) to the
    visible devices and runs; (b) started as torch.distributed.run starts it on every rank -- RANK / WORLD_SIZE in the
    environment, backend nccl = RCCL -- it initialises the group, restores the files dist.deal_files deals the rank, exchanges the
    per-rank counters with ONE all-gather on the device and prints the job summary.  World size 1 is all one GPU allows; world size
    2 runs on gloo in tests/test_dist_cpu.py."""
    import subprocess
    import sys
    from scipy.io import wavfile
    _seeded_home(tmp_path, seeded_states, monkeypatch)
    rng = np.random.default_rng(83)
    ind = tmp_path / "in"
    ind.mkdir()
    for k in range(5):
        n = int(rng.integers(15000, 40000))
        audio_io.save_wave((0.2 * rng.standard_normal(n)).astype(np.float32)[None], str(ind / ("u%d.wav" % k)))
    vf.restore_folder(str(ind), str(tmp_path / "want"), batch_size=2)
    env = dict(os.environ, PYTHONPATH=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # (HOME: the seeded checkpoints)
    root = env["PYTHONPATH"]
    # (a) plain start, more GPUs asked for than the box has
    r = subprocess.run([sys.executable, "-m", "voicefixer_amd", "-ifdr", str(ind), "-ofdr", str(tmp_path / "plain"), "--gpus", "4",
                        "--batch-size", "2"], env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    if torch.cuda.device_count() < 4:
        assert "--gpus 4 requested but only" in r.stderr
    # (b) as a rank of the launcher, RCCL
    port = 29000 + os.getpid() % 2000
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), "-m", "voicefixer_amd", "-ifdr", str(ind), "-ofdr", str(tmp_path / "ranked"),
                        "--gpus", "1", "--batch-size", "2"], env=dict(env, HSA_ENABLE_IPC_MODE_LEGACY="0"), cwd=root,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "rank 0: 5 files" in r.stdout and "whole job: 5 files" in r.stdout and "on 1 GPU(s)" in r.stdout
    # (c) TWO ranks on the real device path: both ranks share this box's one GPU (LOCAL_RANK % device_count), so the counter exchange
    # runs on gloo (RCCL refuses two ranks on one device) -- everything else is the product path: each rank restores the files
    # dist.deal_files deals it with the real kernels, the union equals the unsharded folder
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port + 1), "-m", "voicefixer_amd", "-ifdr", str(ind), "-ofdr", str(tmp_path / "two"),
                        "--gpus", "2", "--batch-size", "2", "--dist-backend", "gloo"], env=env, cwd=root, capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "rank 0:" in r.stdout and "rank 1:" in r.stdout and "whole job: 5 files" in r.stdout and "on 2 GPU(s)" in r.stdout
    counts = [int(m) for m in __import__("re").findall(r"rank \d: (\d+) files", r.stdout)]
    assert sorted(counts) in ([2, 3],) and sum(counts) == 5
    for out in ("plain", "ranked", "two"):
        assert sorted(os.listdir(tmp_path / out)) == sorted(os.listdir(tmp_path / "want"))
        for f in os.listdir(tmp_path / "want"):
            x1, x2 = wavfile.read(str(tmp_path / out / f))[1], wavfile.read(str(tmp_path / "want" / f))[1]
            assert x1.shape == x2.shape and np.max(np.abs(x1.astype(np.int32) - x2.astype(np.int32))) <= 1


def test_bench_oversubscribed_folder_job_two_ranks_on_one_device(tmp_path):
    """``bench.py --gpus 2 --oversubscribe --synth-folder 4``: both launcher ranks run on this box's one GPU (rank r on device r %
    visible), the counters travel over gloo, every file of the synthetic folder is written exactly once and the JSON line says what
    ran -- the rehearsal of the N-rank folder job that profiles/r05_folder_8ranks_on_one_device.json records at N = 8."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--oversubscribe", "--synth-folder", "4", "--batch", "4",
                        "--folder-streams", "1", "--steps", "2"], cwd=root, capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, PYTHONPATH=root))
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    j = json.loads(lines[0])
    if torch.cuda.device_count() >= 2:
        assert j["n_gpus"] == 2 and j["ranks"] == 2
        return
    # n_gpus counts DISTINCT devices (a harness parsing n_gpus must not read a rehearsal as an N-GPU result)
    assert j["n_gpus"] == 1 and j["ranks"] == 2 and "oversubscribed" in j["config"]
    assert j["requested_gpus"] == 2 and j["visible_devices"] == 1 and j["rccl"]["backend"] == "gloo"
    assert [pr["files"] for pr in j["per_rank"]] == [4, 4] and all(pr["device"] == "cuda:0" for pr in j["per_rank"])
    assert j["value"] > 10 and abs(j["per_rank"][0]["audio_s"] - j["per_rank"][1]["audio_s"]) < 1e-6
    # every rank reports the core slice it was pinned to (dist.pin_rank_cpus runs before the process group exists)
    cpus = [pr["cpu"] for pr in j["per_rank"]]
    assert all(c["n_cores"] >= 1 for c in cpus)
    if all(c["pinned"] for c in cpus):
        assert cpus[0]["cores"] != cpus[1]["cores"] and all(c["torch_threads"] == min(c["n_physical"], 16) for c in cpus)


def test_bench_oversubscribed_weak_scaling_line_carries_the_one_rank_leg():
    """``bench.py --gpus 2 --oversubscribe`` (no folder): the weak-scaling line the driver runs at N > 1, rehearsed with two launcher ranks on
    this box's one GPU (gloo for the collectives): rank 0 first runs the K steps alone (``one_rank_leg``), the line carries the self-measured
    efficiency beside ``value`` and counts distinct devices."""
    import json
    import subprocess
    import sys
    if torch.cuda.device_count() >= 2:
        pytest.skip("the rehearsal is for one-GPU boxes (test_bench_two_devices_... covers the real thing)")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--oversubscribe", "--batch", "4", "--steps", "2",
                        "--warmup", "1", "--no-bf16x3"], cwd=root, capture_output=True, text=True, timeout=900, env=dict(os.environ, PYTHONPATH=root))
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 1 and j["ranks"] == 2 and j["rccl"]["backend"] == "gloo" and len(j["per_rank"]) == 2
    assert j["one_rank_leg"]["value"] > 0 and 0.2 < j["scaling_efficiency_self_measured"] < 1.2
    assert j["value"] > 10 and "roofline" in j and "clocks" in j


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two HIP devices (the first real N > 1 run: lights up on a multi-GPU lease)")
def test_bench_two_devices_on_rccl_folder_and_weak_scaling_lines():
    """On a node with >= 2 GPUs: ``bench.py --gpus 2 --synth-folder 32`` (BASELINE configs[3] disk to disk, RCCL for the counters) and
    ``bench.py --gpus 2`` (the weak-scaling line the driver runs) -- one rank per device, backend nccl, every file written once,
    and the second line carries the one-rank leg + the self-measured efficiency."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def line(argv):
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + argv, cwd=root, capture_output=True, text=True, timeout=1800,
                           env=dict(os.environ, PYTHONPATH=root))
        assert r.returncode == 0, r.stderr[-3000:]
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1, r.stdout[-2000:]
        return json.loads(lines[0])

    j = line(["--gpus", "2", "--synth-folder", "32", "--batch", "16", "--steps", "2"])
    assert j["n_gpus"] == 2 and j["ranks"] == 2 and j["rccl"] == {"backend": "nccl", "world_size": 2}
    assert sorted(pr["device"] for pr in j["per_rank"]) == ["cuda:0", "cuda:1"] and [pr["files"] for pr in j["per_rank"]] == [32, 32]
    assert all(pr["cpu"]["pinned"] for pr in j["per_rank"]) or os.cpu_count() < 4
    j = line(["--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "8", "--no-bf16x3"])
    assert j["n_gpus"] == 2 and j["rccl"]["backend"] == "nccl" and sorted(pr["device"] for pr in j["per_rank"]) == ["cuda:0", "cuda:1"]
    assert j["one_rank_leg"]["value"] > 0 and 0.5 < j["scaling_efficiency_self_measured"] < 1.2


def test_graph_replays_survive_an_eager_pass_in_between(seeded_states):
    """Round 4 regression: capture, two replays, ONE eager pass of the same shape on the same stream, replays again -- with the peak
    workspace and the GRU mailboxes zeroed by hipMemsetAsync (= memset nodes in the captured graph) every replay after the eager
    pass returned the waveform scaled by the peak rule's stale workspace.  The zeroing is a kernel now; the sequence must be
    bit-exact, on a fresh pipeline (the allocator state matters) and for both GRU kernels in the eager pass."""
    from voicefixer_amd import engine
    gg = np.load(os.path.join(GOLDEN, "restore_noise_T36.npz"))
    x = torch.from_numpy(gg["wav"])[None].cuda()
    n = x.shape[1]
    for single in (False, True):
        pipe = engine.Pipeline(seeded_states[0], seeded_states[1], "cuda:0")
        want = pipe.restore(x, n).clone()
        pipe.enable_graphs(max_shapes=2, max_batch=1)
        try:
            a, b = pipe.restore(x, n).clone(), pipe.restore(x, n).clone()
            assert len(pipe._graphs) == 1 and torch.equal(a, want) and torch.equal(b, want)
            graphs, pipe._graphs = pipe._graphs, None          # an eager pass while the captured graph stays alive
            pipe.restorer.gru_single = single
            try:
                e = pipe.restore(x, n).clone()
            finally:
                pipe.restorer.gru_single = False
                pipe._graphs = graphs
            assert float((e - want).abs().max()) < 2e-5
            c, d = pipe.restore(x, n).clone(), pipe.restore(x, n).clone()
            assert torch.equal(c, want) and torch.equal(d, want), (float((c - want).abs().max()), float((d - want).abs().max()))
            pipe.check()
        finally:
            pipe.disable_graphs()
