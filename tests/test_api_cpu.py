"""Host-side behaviour of the drop-in API that needs no GPU: error contract, the C ABI
library exporting every declared symbol, packing layouts, I/O quantisation."""
import os
import re

import numpy as np
import pytest
import torch

import voicefixer_amd
from voicefixer_amd import _lib, api, audio_io, packing, weights

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_exported():
    """Every function declared in include/vfx_hip.h is exported by libvfx_hip.so and bound."""
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    hdr = open(os.path.join(ROOT, "include", "vfx_hip.h")).read()
    names = set(re.findall(r"\b(vfx_[a-z0-9_]+)\s*\(", hdr))
    names -= {"vfx_tensor", "vfx_act"}
    assert len(names) >= 18
    h = _lib.lib()
    for n in sorted(names):
        assert hasattr(h, n), n
        assert n in _lib.SIGNATURES, "%s not bound in _lib.SIGNATURES" % n
    assert h.vfx_version() >= 100


def test_release_library_exports_the_c_abi_only_and_reads_no_environment():
    """The dynamic symbol table of libvfx_hip.so is exactly the set of functions include/vfx_hip.h declares (csrc/vfx.map: no
    C++-mangled helper leaks out), and the release build holds none of the development switches: the VFX_* environment names that
    select kernels (VFX_DEV_ENV, csrc/vfx_common.h) are compiled in by `make dev` only."""
    import subprocess
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    rel = os.path.join(ROOT, "voicefixer_amd", "libvfx_hip.so")      # (not VFX_LIB: the release library is what ships)
    hdr = open(os.path.join(ROOT, "include", "vfx_hip.h")).read()
    names = set(re.findall(r"\b(vfx_[a-z0-9_]+)\s*\(", hdr)) - {"vfx_tensor", "vfx_act", "vfx_resblock_w"}
    out = subprocess.run(["nm", "-D", "--defined-only", rel], capture_output=True, text=True, check=True).stdout
    exported = {ln.split()[-1] for ln in out.splitlines() if ln.split()}
    assert exported == names, (sorted(exported - names), sorted(names - exported))
    blob = open(rel, "rb").read()
    switches = set(re.findall(rb"VFX_[A-Z0-9_]{3,}", blob))
    assert not switches, sorted(switches)
    # ... and the sources read the environment through VFX_DEV_ENV only
    csrc = os.path.join(ROOT, "voicefixer_amd", "csrc")
    for f in os.listdir(csrc):
        if f.endswith((".hip", ".inc")):
            assert "getenv(" not in open(os.path.join(csrc, f)).read(), f


def test_python_package_reads_the_environment_through_the_dev_gate_only(monkeypatch):
    """voicefixer_amd/*.py consult no VFX_* environment variable unless VFX_DEV=1 (voicefixer_amd/_dev.py, the one gate): the
    sources touch os.environ only for the launcher's rendezvous variables (RANK / WORLD_SIZE / ..., __main__.py, dist.exec_ranks),
    and a development switch set WITHOUT the gate changes nothing."""
    pkg = os.path.join(ROOT, "voicefixer_amd")
    for f in sorted(os.listdir(pkg)):
        if not f.endswith(".py") or f == "_dev.py":
            continue
        src = open(os.path.join(pkg, f)).read()
        for m in re.finditer(r"environ[^\n]*", src):
            assert "VFX_" not in m.group(0), (f, m.group(0))
        assert "getenv" not in src, f
    from voicefixer_amd import _dev
    monkeypatch.delenv("VFX_DEV", raising=False)
    monkeypatch.setenv("VFX_UNFUSE_WIDE", "0")
    monkeypatch.setenv("VFX_LIB", "/nonexistent/libvfx_hip.so")
    assert _dev.dev_env("VFX_UNFUSE_WIDE", "1") == "1" and _dev.dev_env("VFX_LIB", "x") == "x"
    monkeypatch.setenv("VFX_DEV", "1")
    assert _dev.dev_env("VFX_UNFUSE_WIDE", "1") == "0"


def test_missing_checkpoints_raise_like_reference(tmp_path, monkeypatch):
    monkeypatch.setenv("HOME", str(tmp_path))
    with pytest.raises(RuntimeError, match="Error 0"):
        voicefixer_amd.VoiceFixer()
    with pytest.raises(RuntimeError, match="Error 1"):
        voicefixer_amd.Vocoder(44100)
    with pytest.raises(RuntimeError, match="44100"):
        voicefixer_amd.Vocoder(16000)


def test_no_cpu_fallback(seeded_states):
    """Without a HIP device the product path raises instead of computing on the CPU."""
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    vf = voicefixer_amd.VoiceFixer.from_state(*seeded_states)
    with pytest.raises(RuntimeError, match="no HIP device"):
        vf.restore_inmem(np.zeros(4410, np.float32), cuda=False)
    with pytest.raises(RuntimeError, match="no HIP device"):
        voicefixer_amd.Vocoder.from_state(seeded_states[0]).forward(torch.ones(1, 1, 8, 128))


def test_modes(seeded_states):
    vf = voicefixer_amd.VoiceFixer.from_state(*seeded_states)
    with pytest.raises(NotImplementedError):
        vf.restore_inmem(np.zeros(4410, np.float32), mode=2)
    with pytest.raises(ValueError):
        vf.restore_inmem(np.zeros(4410, np.float32), mode=7)


def test_product_path_does_not_import_oracle():
    """voicefixer_amd must never reach into oracle/ (test infrastructure)."""
    pkg = os.path.join(ROOT, "voicefixer_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), fn
            assert "ref_shim" not in src, fn


def test_packing_layouts():
    w = torch.arange(2 * 3 * 5, dtype=torch.float32).reshape(2, 3, 5)  # (Cout=2, Cin=3, k=5)
    p = packing.pack_conv1d(w)
    assert p.shape == (5, 8, 2) and torch.equal(p[4, 2], w[:, 2, 4]) and (p[:, 3:] == 0).all()
    wt = torch.randn(3, 2, 6)  # ConvTranspose1d (Cin, Cout, k)
    p = packing.pack_convtr1d(wt)
    assert p.shape == (6, 8, 2) and torch.equal(p[5, 1], wt[1, :, 5])
    w2 = torch.randn(4, 3, 3, 3)
    p = packing.pack_conv2d(w2)
    assert p.shape == (9, 8, 4) and torch.equal(p[2 * 3 + 1, 2], w2[:, 2, 2, 1])
    l = torch.randn(6, 10)
    p = packing.pack_linear(l)
    assert p.shape == (1, 16, 6) and torch.equal(p[0, 9], l[:, 9])
    whh = torch.randn(768, 256), torch.randn(768, 256)
    pk = packing.pack_gru_whh(whh[0], whh[1], 40, 24, 64)
    assert pk.numel() == 2 * 768 * 256
    # streamed region: [half][q][gate][unit][e] with k = half*128 + 64 + 4q + e
    base = 2 * 40 * 768 + 2 * 24 * 768
    half, q, gate, unit, e = 1, 3, 2, 17, 1
    idx = base + ((((half * 16 + q) * 3 + gate) * 256 + unit) * 4 + e)
    assert pk[idx] == whh[0][gate * 256 + unit, half * 128 + 64 + 4 * q + e]


def test_weight_norm_fold_matches_torch():
    conv = torch.nn.utils.parametrizations.weight_norm(torch.nn.Conv1d(4, 6, 3))
    g = conv.parametrizations.weight.original0.detach()
    v = conv.parametrizations.weight.original1.detach()
    assert torch.allclose(weights.fold_weight_norm(g, v), conv.weight.detach(), atol=1e-7)
    tr = torch.nn.utils.parametrizations.weight_norm(torch.nn.ConvTranspose1d(4, 6, 4))
    g = tr.parametrizations.weight.original0.detach()
    v = tr.parametrizations.weight.original1.detach()
    assert g.shape == (4, 1, 1)  # dim 0 of a transposed conv is C_in
    assert torch.allclose(weights.fold_weight_norm(g, v), tr.weight.detach(), atol=1e-7)


def test_int16_truncation_and_wav_roundtrip(tmp_path):
    x = np.array([[0.5, -0.5, 0.99999, -1.0, 1e-5, 0.25]], dtype=np.float32)
    assert list(audio_io.to_int16(x)[0]) == [16384, -16384, 32767, -32768, 0, 8192]
    f = str(tmp_path / "a.wav")
    audio_io.save_wave(x, f)
    y = audio_io.load_wav(f)
    assert y.shape == (6,) and np.allclose(y, np.array([16384, -16384, 32767, -32768, 0, 8192]) / 32768.0)
    with pytest.raises(RuntimeError):
        audio_io.load_wav(str(tmp_path / "a.ogg"))        # (FLAC is read and written since round 3: tests/test_flac.py)


def test_stream_chunk_plan():
    """Overlap-add planner: chunks every (chunk - overlap) samples, cover [0, n) exactly, short tails merged."""
    from voicefixer_amd.api import plan_stream_chunks
    chunk, ov = 44100 * 30, 44100
    plan = plan_stream_chunks(44100 * 60 * 30, chunk, ov)          # 30 minutes
    assert plan[0] == (0, chunk) and all(l == chunk for _, l in plan[:-1])
    assert all(b[0] - a[0] == chunk - ov for a, b in zip(plan, plan[1:]))
    assert plan[-1][0] + plan[-1][1] == 44100 * 60 * 30 and len(plan) == 63
    assert plan_stream_chunks(1000000, chunk, ov) == [(0, 1000000)]          # shorter than one chunk
    # a tail of <= overlap + 1024 samples is merged into the previous chunk
    n = chunk + 100
    assert plan_stream_chunks(n, chunk, ov) == [(0, n)]
    n = 2 * chunk - ov + 5000
    p = plan_stream_chunks(n, chunk, ov)
    assert p[-1][0] + p[-1][1] == n and p[-1][1] > ov + 1024
    with pytest.raises(ValueError):
        plan_stream_chunks(10 ** 6, 2000, 1500)
    # mode 1: the pre-filter shortens a chunk to 512 * (len // 512) samples, so a tail of overlap + 1025 ..
    # overlap + 1535 would arrive at the STFT with 1024 samples and raise; restore_stream asks for min_tail = 1535
    small_ov = 100
    n = 2 * chunk - small_ov + 1300                      # tail chunk = small_ov + 1300 samples
    assert len(plan_stream_chunks(n, chunk, small_ov)) == 3 and len(plan_stream_chunks(n, chunk, small_ov, 1535)) == 2
    p = plan_stream_chunks(n + 300, chunk, small_ov, 1535)  # tail = small_ov + 1600: kept, 512 * (1700 // 512) = 1536 > 1024
    assert len(p) == 3 and 512 * (p[-1][1] // 512) > 1024


def test_pack_direct_layout():
    """packing.pack_direct: [slab][CinPad][Cout] -> [slab][CinPad/8][Cout][8] with the 8 channels of a group in the
    order (0,2,4,6,1,3,5,7): the 16-byte vector at [..][m][4*hi : 4*hi+4] is lane (m, hi)'s MFMA A operand of the
    four consecutive k-steps k = 2*kk + hi (include/vfx_hip.h, vfx_act.w_direct)."""
    from voicefixer_amd import packing
    g = torch.Generator().manual_seed(5)
    w = torch.randn((16, 20, 3), generator=g)          # Cout, Cin (padded to 24), k
    wp = packing.pack_conv1d(w)
    wd = packing.pack_direct(wp)
    assert tuple(wp.shape) == (3, 24, 16) and tuple(wd.shape) == (3, 3, 16, 8) and wd.is_contiguous()
    for s in range(3):
        for c8 in range(3):
            for hi in range(2):
                for kk in range(4):
                    assert torch.equal(wd[s, c8, :, 4 * hi + kk], wp[s, 8 * c8 + 2 * kk + hi, :])
    assert torch.equal(wd[:, 2, :, [2, 3, 6, 7]], torch.zeros(3, 16, 4))   # channels 20..23 are zero fill


def test_wav_length_from_header(tmp_path):
    """restore_folder sorts / windows the folder by the lengths in the WAV headers before decoding anything."""
    from scipy.io import wavfile
    rng = np.random.default_rng(0)
    for sr, n in ((44100, 1000), (22050, 1001), (48000, 777), (16000, 1234)):
        p = str(tmp_path / ("f%d.wav" % sr))
        wavfile.write(p, sr, (rng.standard_normal((n, 2)) * 1000).astype(np.int16))
        assert audio_io.wav_length(p) == len(audio_io.load_wav(p))


def test_resample_hq_meets_the_soxr_hq_recipe():
    """Other sample rates (voicefixer/base.py:47-49: librosa.load(sr=44100), soxr_hq by default): tones in the pass
    band (up to 0.913 of the lower Nyquist frequency) come out at the new rate to 1e-6, tones above that Nyquist frequency
    are rejected by more than 120 dB, and the length is the ceil(n * ratio) the header-based planner predicts."""
    for sr_in in (48000, 16000, 96000):
        n = sr_in
        t = np.arange(n) / sr_in
        f_nyq = min(sr_in, audio_io.SR) / 2
        for f in (440.0, 0.9 * f_nyq):
            y = audio_io.resample_hq(np.sin(2 * np.pi * f * t), sr_in, audio_io.SR)
            assert len(y) == -(-n * audio_io.SR // sr_in) and y.dtype == np.float32
            ref = np.sin(2 * np.pi * f * np.arange(len(y)) / audio_io.SR)
            assert np.abs(y[3000:-3000] - ref[3000:-3000]).max() < 1e-6, (sr_in, f)
        if sr_in > audio_io.SR:
            y = audio_io.resample_hq(np.sin(2 * np.pi * (f_nyq + 60.0) * t), sr_in, audio_io.SR)
            assert np.sqrt(np.mean(y[3000:-3000].astype(np.float64) ** 2)) < 1e-6 / np.sqrt(2)     # -120 dB re the tone


def test_plan_batches_ragged_runs():
    """restore_batch's batch plan: ascending lengths -> runs of <= batch_size utterances whose shortest member has at
    least ragged_ratio of the frames of the longest; multi-segment files and plugin vocoders fall back to exact
    lengths; every position appears exactly once and in order."""
    from voicefixer_amd.api import plan_batches, SEG_LENGTH
    lens = sorted([441 * 100, 441 * 100 + 5, 441 * 120, 441 * 134, 441 * 200, 441 * 201, SEG_LENGTH, SEG_LENGTH + 1,
                   SEG_LENGTH + 1, 3 * SEG_LENGTH, 900])
    plan = plan_batches(lens, batch_size=4, ragged_ratio=0.75)
    assert [p for _, grp in plan for p in grp] == list(range(len(lens)))
    kinds = [(k, [lens[p] for p in grp]) for k, grp in plan]
    assert kinds[0] == ("samples", [900])                                   # too short: alone (raises downstream)
    assert kinds[1] == ("ragged", [441 * 100, 441 * 100 + 5, 441 * 120])    # 134 frames: 101 < 0.75 * 135
    assert kinds[2] == ("ragged", [441 * 134])                              # 135 < 0.75 * 201
    assert kinds[3] == ("ragged", [441 * 200, 441 * 201])                   # the 30 s file is far longer
    assert kinds[4] == ("ragged", [SEG_LENGTH])
    assert kinds[5] == ("samples", [SEG_LENGTH + 1, SEG_LENGTH + 1]) and kinds[6] == ("samples", [3 * SEG_LENGTH])
    for k, grp in plan_batches(lens, 4, 0.75, ragged=False):
        assert k == "samples" and len({lens[p] for p in grp}) == 1
    big = plan_batches([441 * 300 + i for i in range(70)], batch_size=32)
    assert [len(grp) for _, grp in big] == [32, 32, 6]
    assert plan_batches([], 8) == []


def _unpack_direct(wd):
    """[slab][Cin/8][Cout][8 channels in the order 0,2,4,6,1,3,5,7] -> [slab][Cin][Cout] (inverse of packing.pack_direct)."""
    s, g, co, _ = wd.shape
    inv = [0, 4, 1, 5, 2, 6, 3, 7]       # position of channel c within its group
    return wd.permute(0, 1, 3, 2)[:, :, inv, :].reshape(s, g * 8, co)


def test_pack_wino_is_the_f23_weight_transform():
    """packing.pack_wino (vfx_resblock_f32: w2_wino): with V = (d0-d2, d1+d2, d2-d1, d1-d3) of the inputs
    x[q-d], x[q], x[q+d], x[q+2d] and m_k = U_k V_k, the pair (m0+m1+m2, m1-m2-m3) must be the direct k = 3 convolution
    at q and q + d -- evaluated here in float64 on the CPU from the PACKED weights."""
    import torch.nn.functional as F
    from voicefixer_amd import packing
    g = torch.Generator().manual_seed(5)
    cin, cout, L, d = 16, 32, 40, 3
    w = torch.randn((cout, cin, 3), generator=g)
    x = torch.randn((1, cin, L), generator=g)
    ref = F.conv1d(x.double(), w.double(), dilation=d, padding=d)[0]
    U = _unpack_direct(packing.pack_wino(packing.pack_conv1d(w))).double()            # [4][Cin][Cout]
    xp = F.pad(x.double()[0], (d, 2 * d))
    for q in (0, 1, 7, L - d - 1):
        d0, d1, d2, d3 = (xp[:, q + k * d] for k in range(4))                        # x[q-d], x[q], x[q+d], x[q+2d]
        V = [d0 - d2, d1 + d2, d2 - d1, d1 - d3]
        m = [U[k].t() @ V[k] for k in range(4)]
        assert torch.allclose(m[0] + m[1] + m[2], ref[:, q], atol=1e-5)
        assert torch.allclose(m[1] - m[2] - m[3], ref[:, q + d], atol=1e-5)


def test_pack_wino4_is_the_f43_weight_transform():
    """packing.pack_wino4 (vfx_act.w_wino4): with V = B^T d of the six inputs x[q-d] .. x[q+4d] and m_k = U_k V_k, A^T m
    must be the direct k = 3 convolution at q, q+d, q+2d, q+3d -- float64 evaluation from the PACKED weights; and the
    accumulator start values of convwg4_kernel (m2 = m4 = 0) must reproduce bias + residual on all four outputs."""
    import torch.nn.functional as F
    from voicefixer_amd import packing
    g = torch.Generator().manual_seed(6)
    cin, cout, L, d = 16, 32, 60, 3
    w = torch.randn((cout, cin, 3), generator=g)
    x = torch.randn((1, cin, L), generator=g)
    ref = F.conv1d(x.double(), w.double(), dilation=d, padding=d)[0]
    U = _unpack_direct(packing.pack_wino4(packing.pack_conv1d(w))).double()           # [6][Cin][Cout]
    BT = torch.tensor([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0],
                       [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], dtype=torch.float64)
    AT = torch.tensor([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], dtype=torch.float64)
    xp = F.pad(x.double()[0], (d, 4 * d))
    for q in (0, 2, 11, L - 3 * d - 1):
        dd = torch.stack([xp[:, q + j * d] for j in range(6)])                        # x[q-d] .. x[q+4d], [6][Cin]
        V = BT @ dd
        m = torch.stack([U[k].t() @ V[k] for k in range(6)])                          # [6][Cout]
        y = AT @ m
        for i in range(4):
            assert torch.allclose(y[i], ref[:, q + i * d], atol=1e-5)
    t = torch.randn(4, dtype=torch.float64)
    m3 = 0.5 * (t[2] - t[1])
    m1 = 2 * t[1] - t[2]
    start = torch.tensor([t[0] - m1 - m3, m1, 0, m3, 0, t[3] - m1 - 8 * m3])
    assert torch.allclose(AT @ start, t, atol=1e-12)


def test_ragged_rows_length_vectors():
    """engine.RaggedRows: the per-utterance lengths of every stage follow the reference's own shape rules
    (T = 1 + n // 441 frames, the UNet pads T to a multiple of 64 and halves it per level, the vocoder
    appends T % 2 + 4 frames and upsamples by 7, 7, 3, 3)."""
    from voicefixer_amd.engine import RaggedRows
    lens = [1025, 44100, 441 * 64 - 1, 441 * 64, 441000, 1323000]
    rg = RaggedRows(lens, "cpu")
    assert rg.B == len(lens) and rg.n_max == 1323000 and rg.T_max == 3001
    for b, n in enumerate(lens):
        T = 1 + n // 441
        Tp = -(-T // 64) * 64
        Tc = T + T % 2 + 4
        assert int(rg.n[b]) == n and int(rg.T[b]) == T
        for k in range(7):
            assert int(rg.unet[k][b]) == (Tp >> k) * (128 >> k)       # rows x pitch of a level-k map
        for mult in (1, 7, 49, 147, 441):
            assert int(rg.voc[mult][b]) == Tc * mult
    assert int(rg.voc[441][4]) == 443646                             # the 10 s utterance of SURVEY.md section 8
    assert rg.T.dtype == torch.int32


def test_winograd_index_spaces_cover_every_position_once():
    """The pair / quad index spaces of convwg_kernel / convwg4_kernel (vfx_convwg.inc: position q = 2d*blk + r resp.
    4d*blk + i*d + r): restated here and checked to cover every output position of a row exactly once, for the dilations
    of the ResStacks and the pitches of the UNet, with the block counts the launchers use."""
    for L in (1, 2, 7, 54, 4099, 7042):
        for d in (1, 3, 27, 128, 729, 2187):
            for nout in (2, 4):
                nblk = -(-L // (nout * d))
                seen = [0] * L
                for P in range(nblk * d):
                    blk, r = divmod(P, d)
                    q = nout * d * blk + r
                    for i in range(nout):
                        if q + i * d < L:
                            seen[q + i * d] += 1
                assert min(seen) == 1 and max(seen) == 1, (L, d, nout)


def test_model_attribute_is_a_module_handle():
    """``VoiceFixer._model`` (base.py:13): reference callers move it between devices and look at its parameters
    (test/streamlit.py:40-42) and reach the vocoder through it (base.py:127).  Here it is a handle: ``.to()`` / ``.eval()``
    work, a forward in train mode is the unsupported mode 2, the weights stay in the engine."""
    from voicefixer_amd import VoiceFixer, weights
    vf = VoiceFixer.from_state(weights.seeded_vocoder_state(1), weights.seeded_restorer_state(2))
    assert isinstance(vf._model, torch.nn.Module) and not vf._model.training
    assert list(vf._model.parameters())[0].is_cuda is False
    vf._model = vf._model.to("cpu")
    assert vf._model.eval() is vf._model and vf._model.vocoder is vf._vocoder
    # nn.Module.train() recurses into children: toggling the OWNER's mode must not raise (it did before round 3); what
    # is unsupported is a forward pass in train mode (the reference's mode 2), and that is what raises
    assert vf.train() is vf and vf._model.training
    with pytest.raises(NotImplementedError):
        vf._model(None, torch.zeros(1, 1, 4, 128))
    assert vf.eval() is vf and not vf._model.training


def test_restore_folder_host_pipeline_with_a_stub_device(tmp_path):
    """The folder driver's host side alone (header lengths -> length-sorted windows -> decode / resample / down-mix in
    the pool -> batches -> encode in the pool), with the device stage replaced by the identity: mixed WAV / FLAC inputs
    at three sample rates, mono and stereo, come out under their own names, in their own container, at 44.1 kHz with
    the lengths the header-only planner predicted."""
    from scipy.io import wavfile
    from voicefixer_amd import flac
    from voicefixer_amd.api import VoiceFixer
    rng = np.random.default_rng(0)
    ind, outd = tmp_path / "in", tmp_path / "out"
    ind.mkdir()
    spec = {"a.wav": (44100, 5000, 1), "b.wav": (48000, 7001, 2), "c.flac": (16000, 3000, 1), "d.flac": (44100, 4100, 2),
            "e.txt": None}
    for name, sp in spec.items():
        if sp is None:
            (ind / name).write_text("not audio")
            continue
        sr, n, ch = sp
        x = (3000 * np.sin(np.arange(n)[:, None] * 0.03 * (1 + np.arange(ch))) + rng.integers(-50, 50, (n, ch))).astype(np.int16)
        if name.endswith(".wav"):
            wavfile.write(str(ind / name), sr, x if ch > 1 else x[:, 0])
        else:
            flac.write(str(ind / name), x, sr, 16)
    seen = []

    class Stub(VoiceFixer):
        def __init__(self):        # no checkpoints, no device: only the host logic of the class is under test
            pass

        def restore_batches(self, batches, your_vocoder_func=None, streams=2, mode=0):     # the device stage
            for tag, kind, host, lens in batches:
                seen.append(list(lens))
                yield tag, host.clone(), list(lens)

    vf = Stub()
    assert vf.restore_folder(str(ind), str(outd), batch_size=2) == ["a.wav", "b.wav"]          # the reference's filter
    names = vf.restore_folder(str(ind), str(outd), batch_size=2, extensions=(".wav", ".flac"), name_suffix="-mode0")
    assert names == ["a-mode0.wav", "b-mode0.wav", "c-mode0.flac", "d-mode0.flac"]
    want = {"a": 5000, "b": -(-7001 * 44100 // 48000), "c": -(-3000 * 44100 // 16000), "d": 4100}
    assert [n for b in seen[-2:] for n in b] == sorted(want.values())     # ascending lengths (what the ragged planner wants), batches of 2
    for nm in names:
        p = str(outd / nm)
        assert audio_io.wav_length(p) == want[nm[0]]
        y = audio_io.load_wav(p)
        assert y.shape == (want[nm[0]],) and np.abs(y).max() > 0.05
    a_in = audio_io.load_wav(str(ind / "a.wav"))
    assert np.abs(audio_io.load_wav(str(outd / "a-mode0.wav")) - a_in).max() <= 1.0 / 32768     # identity through PCM16


def test_selfcheck_compare_and_report_on_made_up_stages(capsys):
    """voicefixer_amd/selfcheck.py, host logic only: per-stage figures are relative to the DIRECT variant's own peak, the waveform
    also gets an RMS line, and a stage above the tolerance is reported (and named) as a failure."""
    import io
    from voicefixer_amd import selfcheck
    g = torch.Generator().manual_seed(0)
    base = {k: torch.randn(2, 5, 7, generator=g, dtype=torch.float64) for k in selfcheck.STAGES}
    same = {k: v.clone() for k, v in base.items()}
    off = {k: v.clone() for k, v in base.items()}
    off["up3"][0, 0, 0] += 1e-3 * float(base["up3"].abs().max())
    rows = selfcheck.compare(off, base)
    assert abs(rows["up3"] - 1e-3) < 1e-9 and rows["up2"] == 0.0 and rows["wav_rms"] == 0.0
    buf = io.StringIO()
    bad, table = selfcheck.report({"default": same, "direct": base, "bf16x3": off}, 1e-4, out=buf)
    assert [(v, k) for v, k, _ in bad] == [("bf16x3", "up3")] and set(table) == {"default", "bf16x3"}
    assert "above 1e-04" in buf.getvalue() and buf.getvalue().count("\n") == len(selfcheck.STAGES) + 2
    bad, _ = selfcheck.report({"default": same, "direct": base}, 1e-4, out=io.StringIO())
    assert bad == []
