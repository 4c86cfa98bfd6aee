"""voicefixer_amd/flac.py (the FLAC reader / writer that stands in for libsndfile) and the header-only length queries
of audio_io.  Known answers: the reference's own fixtures (test/utterance/original/original.flac is the FLAC twin of
original.wav; oracle/make_golden.py copied the .flac files and stored the .wav's PCM), libFLAC's per-frame CRC-8 /
CRC-16 and the MD5 of the decoded audio that every one of those files carries."""
import os
import struct

import numpy as np
import pytest

from conftest import GOLDEN
from voicefixer_amd import audio_io, flac

REF = os.path.join(GOLDEN, "ref_utterance")


def test_decode_equals_the_wav_twin_bit_for_bit():
    sr, pcm, bps = flac.read(os.path.join(REF, "original_original.flac"))
    want = np.load(os.path.join(GOLDEN, "flac_original_pcm.npz"))["pcm"]
    assert (sr, bps) == (44100, 16) and pcm.shape == (132300, 1)
    assert np.array_equal(pcm[:, 0], want)


@pytest.mark.parametrize("name, n", [("original_p360_001_mic1.flac", 96076), ("target_oracle.flac", 97902),
                                     ("target_output_mode_0.flac", 132300), ("target_output_mode_1.flac", 132096)])
def test_reference_fixtures_decode_with_crc_and_md5_verified(name, n):
    # lengths: SURVEY.md 8(c)(ii) -- 96 076 -> 97 902 (oracle), 132 300 -> 132 300 (mode 0) / 132 096 (mode 1)
    path = os.path.join(REF, name)
    assert flac.info(path) == (44100, 1, 16, n)
    sr, pcm, bps = flac.read(path)          # raises on any CRC-8 / CRC-16 / MD5 mismatch
    assert pcm.shape == (n, 1) and np.abs(pcm).max() > 1000
    assert audio_io.wav_length(path) == n
    x = audio_io.load_wav(path)
    assert x.dtype == np.float32 and x.shape == (n,) and np.array_equal(x, pcm[:, 0].astype(np.float32) / 32768.0)


def test_corruption_is_detected():
    data = bytearray(open(os.path.join(REF, "target_oracle.flac"), "rb").read())
    data[len(data) // 2] ^= 0x10
    with pytest.raises(flac.FlacError):
        flac.decode(bytes(data))
    with pytest.raises(flac.FlacError):
        flac.decode(b"RIFF" + bytes(100))


@pytest.mark.parametrize("n, nch, bps", [(0, 1, 16), (1, 1, 16), (4096, 1, 16), (4097, 2, 24), (10000, 2, 16), (30000, 1, 8)])
def test_encode_decode_roundtrip(n, nch, bps):
    rng = np.random.default_rng(n + nch)
    t = np.arange(n)
    x = ((1 << (bps - 3)) * np.sin(t * 0.01)[:, None] * np.ones((1, nch)) + rng.integers(-40, 40, (n, nch))).astype(np.int64)
    if n > 10:
        x[5] = (1 << (bps - 1)) - 1         # full-scale extremes survive
        x[6] = -(1 << (bps - 1))
    blob = flac.encode(x if nch > 1 else x[:, 0], 44100, bps)
    sr, y, b = flac.decode(blob)            # (verifies the CRCs and the MD5 the encoder wrote)
    assert (sr, b) == (44100, bps) and np.array_equal(x, y)
    if n >= 4096 and bps == 16:
        assert len(blob) < n * nch * 2      # the fixed predictor + Rice coding really compresses


def test_white_noise_falls_back_to_verbatim_subframes():
    x = np.random.default_rng(1).integers(-32768, 32767, 9000)
    assert np.array_equal(flac.decode(flac.encode(x, 44100))[1][:, 0], x)


def test_reencoding_a_reference_file_is_lossless():
    sr, pcm, _ = flac.read(os.path.join(REF, "target_output_mode_0.flac"))
    assert np.array_equal(flac.decode(flac.encode(pcm, sr))[1], pcm)


def test_save_wave_picks_the_container_from_the_extension(tmp_path):
    x = (0.25 * np.sin(np.arange(5000) * 0.02)).astype(np.float32)[None]
    for ext in (".wav", ".flac"):
        f = str(tmp_path / ("o" + ext))
        audio_io.save_wave(x, f)
        y = audio_io.load_wav(f)
        assert np.array_equal(y, audio_io.to_int16(x)[0].astype(np.float32) / 32768.0)   # wav.py:27-34 truncation
    with pytest.raises(RuntimeError):
        audio_io.save_wave(x, str(tmp_path / "o.mp3"))
    with pytest.raises(RuntimeError):
        audio_io.load_wav(str(tmp_path / "o.mp3"))


def _write_pcm24(path, sr, frames):
    """Minimal 24-bit PCM WAV (the format scipy's memory-mapped reader refuses)."""
    raw = b"".join(int(v).to_bytes(3, "little", signed=True) for v in frames)
    fmt = struct.pack("<HHIIHH", 1, 1, sr, sr * 3, 3, 24)
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", 4 + 8 + len(fmt) + 8 + 4 + 8 + len(raw)) + b"WAVE")
        f.write(b"fmt " + struct.pack("<I", len(fmt)) + fmt)
        f.write(b"LIST" + struct.pack("<I", 4) + b"INFO")          # an extra chunk in front of the data
        f.write(b"data" + struct.pack("<I", len(raw)) + raw)


def test_wav_length_from_header_any_bit_depth(tmp_path):
    from scipy.io import wavfile
    f16, f24, f22 = (str(tmp_path / n) for n in ("a16.wav", "a24.wav", "a22k.wav"))
    wavfile.write(f16, 44100, np.zeros((1234, 2), np.int16))
    assert audio_io.wav_length(f16) == 1234
    vals = (np.sin(np.arange(777) * 0.05) * 8e6).astype(np.int64)
    _write_pcm24(f24, 44100, vals)
    assert audio_io.wav_length(f24) == 777                        # (mmap=True raised ValueError here)
    x = audio_io.load_wav(f24)
    assert x.shape == (777,) and np.allclose(x, vals / float(1 << 23), atol=1e-6)
    wavfile.write(f22, 22050, np.zeros(1000, np.int16))
    assert audio_io.wav_length(f22) == len(audio_io.load_wav(f22)) == 2000


# ---- the C frame codec (voicefixer_amd/csrc_host/vfx_flac.c, include/vfx_audio.h) against the Python specification ----
@pytest.fixture(scope="module")
def native_codec():
    flac.build_native()                       # one C file, about a second
    h = flac.native()
    if h is None:
        pytest.skip("libvfx_audio.so switched off (VFX_FLAC_NATIVE=0)")
    return h


def test_native_library_exports_the_header(native_codec):
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = open(os.path.join(root, "include", "vfx_audio.h")).read()
    names = set(re.findall(r"\b(vfx_[a-z0-9_]+)\s*\(", hdr))
    assert names == {"vfx_audio_version", "vfx_flac_decode_frames", "vfx_flac_encode_frames", "vfx_resample_poly_f32"}
    for n in names:
        getattr(native_codec, n)              # AttributeError == header / library mismatch


@pytest.mark.parametrize("name", ["original_original.flac", "original_p360_001_mic1.flac", "target_oracle.flac",
                                  "target_output_mode_0.flac", "target_output_mode_1.flac"])
def test_native_decoder_equals_the_python_decoder_on_the_reference_fixtures(native_codec, name):
    data = open(os.path.join(REF, name), "rb").read()      # libFLAC streams: LPC subframes, partitioned Rice codes
    a = flac.decode(data, use_native=False)
    b = flac.decode(data, use_native=True)
    assert a[0] == b[0] and a[2] == b[2] and a[1].dtype == b[1].dtype and np.array_equal(a[1], b[1])
    bad = bytearray(data)
    bad[len(bad) // 2] ^= 0x10
    with pytest.raises(flac.FlacError):
        flac.decode(bytes(bad), use_native=True)            # CRC-16 (or a broken code) is caught in C as well
    with pytest.raises(flac.FlacError):
        flac.decode(data[:len(data) // 2], use_native=True)  # truncated stream


@pytest.mark.parametrize("n, nch, bps", [(0, 1, 16), (1, 1, 16), (2, 2, 16), (3, 1, 16), (4095, 1, 16), (4096, 2, 16),
                                         (4097, 1, 24), (20000, 2, 24), (9000, 1, 8), (12345, 3, 12), (5000, 2, 20)])
def test_native_encoder_writes_the_python_encoder_s_bytes(native_codec, n, nch, bps):
    rng = np.random.default_rng(n + nch + bps)
    t = np.arange(n)[:, None]
    amp = (1 << (bps - 1)) - 1
    x = 0.4 * amp * np.sin(2 * np.pi * 220.0 * (1 + np.arange(nch)) * t / 44100.0) + 0.01 * amp * rng.standard_normal((n, nch))
    x[n // 2:n // 2 + 700] = amp * rng.uniform(-1, 1, x[n // 2:n // 2 + 700].shape)     # a stretch of noise: VERBATIM frames
    pcm = np.clip(np.round(x), -amp - 1, amp).astype(np.int64)
    e_py = flac.encode(pcm, 44100, bps, use_native=False)
    e_c = flac.encode(pcm, 44100, bps, use_native=True)
    assert e_py == e_c
    for use in (False, True):
        sr, back, b = flac.decode(e_c, use_native=use)
        assert (sr, b) == (44100, bps) and np.array_equal(back, pcm.reshape(n, nch))


class _BitWriter:
    def __init__(self):
        self.bits = []

    def put(self, v, n):
        self.bits += [(int(v) >> (n - 1 - i)) & 1 for i in range(n)]

    def signed(self, v, n):
        self.put(int(v) & ((1 << n) - 1), n)

    def rice(self, v, k):
        u = 2 * v if v >= 0 else -2 * v - 1
        self.put(0, u >> k)
        self.put(1, 1)
        self.put(u & ((1 << k) - 1), k)

    def align(self):
        self.bits += [0] * (-len(self.bits) % 8)

    def bytes(self):
        assert len(self.bits) % 8 == 0
        return np.packbits(np.array(self.bits, np.uint8)).tobytes()


def _lpc_residual(x, coefs, shift):
    order = len(coefs)
    res = []
    for i in range(order, len(x)):
        pred = sum(int(c) * int(x[i - 1 - j]) for j, c in enumerate(coefs)) >> shift
        res.append(int(x[i]) - pred)
    return res


def _handmade_stream(sub_specs, cassign, chans, bps, blocksize):
    """One-frame stream with hand-written subframes.  sub_specs[c] = ("constant" | "verbatim" | ("fixed", order, porder,
    escape_partition) | ("lpc", coefs, shift, precision), wasted_bits); chans[c] = what subframe c carries."""
    w = _BitWriter()
    w.put(0x3FFE, 14); w.put(0, 2)
    w.put(7, 4); w.put(0, 4)                      # 16-bit block size follows; sample rate from STREAMINFO
    w.put(cassign, 4); w.put({8: 1, 12: 2, 16: 4, 20: 5, 24: 6}[bps], 3); w.put(0, 1)
    w.put(0, 8)                                   # frame number 0
    w.put(blocksize - 1, 16)
    w.put(flac._crc8(w.bytes()), 8)
    side = {8: (1,), 9: (0,), 10: (1,)}.get(cassign, ())
    for c, ((kind, wasted), x) in enumerate(zip(sub_specs, chans)):
        sb = bps + (1 if c in side else 0) - wasted
        x = [int(v) >> wasted for v in x]
        w.put(0, 1)
        typ = {"constant": 0, "verbatim": 1}.get(kind) if isinstance(kind, str) else \
            (8 + kind[1] if kind[0] == "fixed" else 31 + len(kind[1]))
        w.put(typ, 6)
        if wasted:
            w.put(1, 1); w.put(0, wasted - 1); w.put(1, 1)
        else:
            w.put(0, 1)
        if kind == "constant":
            w.signed(x[0], sb)
        elif kind == "verbatim":
            for v in x:
                w.signed(v, sb)
        else:
            if kind[0] == "fixed":
                order, porder, esc_part = kind[1], kind[2], kind[3]
                coefs, shift = flac._FIXED[order], 0
            else:
                coefs, shift, prec = kind[1], kind[2], kind[3]
                order, porder, esc_part = len(coefs), 1, -1
            for v in x[:order]:
                w.signed(v, sb)
            if kind[0] == "lpc":
                w.put(prec - 1, 4); w.signed(shift, 5)
                for cf in coefs:
                    w.signed(cf, prec)
            res = _lpc_residual(x, coefs, shift)
            w.put(1, 2); w.put(porder, 4)             # Rice method 1: 5-bit parameters, escape code 31
            o = 0
            for part in range(1 << porder):
                cnt = (blocksize >> porder) - (order if part == 0 else 0)
                seg = res[o:o + cnt]
                o += cnt
                if part == esc_part:
                    nb = max(int(abs(v)).bit_length() for v in seg) + 1
                    w.put(31, 5); w.put(nb, 5)
                    for v in seg:
                        w.signed(v, nb)
                else:
                    k = 3 + part
                    w.put(k, 5)
                    for v in seg:
                        w.rice(v, k)
    w.align()
    frame = w.bytes()
    frame += struct.pack(">H", flac._crc16(frame))
    nch = len(chans)
    v = (44100 << 44) | ((nch - 1) << 41) | ((bps - 1) << 36) | blocksize
    si = struct.pack(">HH", blocksize, blocksize) + len(frame).to_bytes(3, "big") * 2 + v.to_bytes(8, "big") + bytes(16)
    return b"fLaC" + bytes([0x80]) + len(si).to_bytes(3, "big") + si + frame


@pytest.mark.parametrize("cassign", [1, 8, 9, 10])
def test_both_decoders_on_handmade_subframes(native_codec, cassign):
    """What neither the reference's (mono, libFLAC) fixtures nor our own encoder produce: CONSTANT subframes, FIXED
    orders other than 2, partitioned residuals with an escape partition, LPC with a shift, wasted bits, and the three
    stereo decorrelation modes -- written bit by bit here, decoded by the Python and the C decoder."""
    rng = np.random.default_rng(cassign)
    bs, bps = 64, 16
    left = (3000 * np.sin(np.arange(bs) / 5.0) + rng.integers(-40, 40, bs)).astype(np.int64)
    right = (left * 0.9 + rng.integers(-300, 300, bs)).astype(np.int64)
    left4 = (left >> 2) << 2                                   # two wasted bits
    cases = [
        ([("verbatim", 0), (("fixed", 4, 1, -1), 0)], left, right),
        ([(("fixed", 1, 2, 2), 0), (("lpc", (1200, -400, 37), 10, 12), 0)], left, right),
        ([(("fixed", 3, 0, -1), 2), (("fixed", 0, 1, 0), 0)], left4, right),
        ([("constant", 0), (("lpc", (2047, -1024), 11, 12), 0)], np.full(bs, -1234), right),
    ]
    for specs, a, b in cases:
        if cassign == 1:
            chans = want = [a, b]
        elif cassign == 8:
            chans, want = [a, a - b], [a, b]
        elif cassign == 9:
            chans, want = [a - b, b], [a, b]
        else:
            mid, sd = (a + b) >> 1, a - b
            chans, want = [mid, sd], [a, b]
            if specs[0][1]:                                    # wasted bits belong to the CODED channel: keep mid divisible
                a2 = ((mid >> specs[0][1]) << specs[0][1]) * 2 + (sd & 1) + sd
                a = a2 >> 1
                b = a - sd
                mid = (a + b) >> 1
                chans, want = [mid, sd], [a, b]
        if specs[0][0] == "constant" and cassign != 1:
            continue                                           # (a constant mid / side channel would change `want`)
        if specs[0][1] and cassign in (8, 9):
            chans[0] = (np.asarray(chans[0]) >> specs[0][1]) << specs[0][1]
            want = [chans[0], chans[0] - chans[1]] if cassign == 8 else [chans[0] + chans[1], chans[1]]
        data = _handmade_stream(specs, cassign, chans, bps, bs)
        for use in (False, True):
            sr, pcm, b_ = flac.decode(data, use_native=use)
            assert (sr, b_) == (44100, bps)
            assert np.array_equal(pcm[:, 0], want[0]) and np.array_equal(pcm[:, 1], want[1]), (cassign, specs, use)


def test_native_codec_runs_without_the_interpreter_lock(native_codec):
    """The point of the C codec: restore_folder's worker threads decode / encode files in parallel, because ctypes drops
    the interpreter lock around the call.  Shown without a stopwatch: a pure-Python thread keeps counting WHILE one
    long native decode is in flight (it could not execute a single bytecode if the lock were held)."""
    import threading
    rng = np.random.default_rng(3)
    n = 44100 * 240
    pcm = (8000 * np.sin(np.arange(n) / 30.0) + rng.integers(-200, 200, n)).astype(np.int64)
    data = flac.encode(pcm, 44100, 16, use_native=True)
    state = {"count": 0, "stop": False, "during": None}

    def spin():
        while not state["stop"]:
            state["count"] += 1

    th = threading.Thread(target=spin)
    th.start()
    try:
        import ctypes
        out = np.empty((n, 1), np.int32)
        done, err = ctypes.c_ulonglong(0), ctypes.c_ulonglong(0)
        before = state["count"]
        rc = native_codec.vfx_flac_decode_frames(data, len(data), 42, 1, 16, out.ctypes.data_as(ctypes.POINTER(ctypes.c_int)),
                                                 n, ctypes.byref(done), 1, ctypes.byref(err))      # ~0.15 s in C
        state["during"] = state["count"] - before
    finally:
        state["stop"] = True
        th.join()
    assert rc == 0 and done.value == n and np.array_equal(out[:, 0], pcm)
    assert state["during"] > 100, state["during"]            # (holding the lock it would be 0; free-running: millions)


def test_native_decoder_survives_corrupted_streams(native_codec):
    """Bit flips, overwritten bytes and truncations: the C decoder either decodes or reports an error (FlacError) --
    tools/flac_fuzz.py runs the same under -fsanitize=address,undefined (profiles/r03_flac_codec_fuzz.txt)."""
    rng = np.random.default_rng(5)
    src = open(os.path.join(REF, "target_oracle.flac"), "rb").read()
    rejected = 0
    for _ in range(300):
        d = bytearray(src)
        for _ in range(int(rng.integers(1, 5))):
            pos = int(rng.integers(38, len(d)))
            mode = int(rng.integers(0, 3))
            if mode == 0:
                d[pos] ^= 1 << int(rng.integers(0, 8))
            elif mode == 1:
                d[pos] = int(rng.integers(0, 256))
            else:
                d = d[:pos]
        try:
            flac.decode(bytes(d), use_native=True)
        except flac.FlacError:
            rejected += 1
    assert rejected > 250          # (with the CRCs and the MD5 on, only damage past the last frame goes unnoticed)


@pytest.mark.parametrize("sr_in", [48000, 16000, 22050, 96000, 8000, 24000, 11025])
def test_native_resampler_equals_the_scipy_path(native_codec, sr_in):
    """vfx_resample_poly_f32 (one float32 dot product per output sample) against scipy's upfirdn in float64 with the
    same filter: same length, same alignment, float32 rounding apart -- on noise, on very short inputs and on stereo."""
    rng = np.random.default_rng(sr_in)
    for shape in ((1,), (7,), (1000,), (2 * sr_in + 17,), (2, 3001)):
        x = rng.standard_normal(shape).astype(np.float32)
        a = audio_io.resample_hq(x, sr_in, 44100, use_native=False)
        b = audio_io.resample_hq(x, sr_in, 44100, use_native=True)
        assert a.shape == b.shape == shape[:-1] + (-(-shape[-1] * 44100 // sr_in),) and b.dtype == np.float32
        assert np.abs(a - b).max() < 2e-6, (shape, np.abs(a - b).max())
    assert audio_io.resample_hq(np.zeros(0, np.float32), sr_in, 44100).shape == (0,)


def _patched_total(data, total):
    """The same stream with STREAMINFO's 36-bit sample count overwritten (file bytes 18..25 hold sr/ch/bps/total)."""
    v = int.from_bytes(data[18:26], "big")
    v = (v & ~((1 << 36) - 1)) | total
    return data[:18] + v.to_bytes(8, "big") + data[26:]


def test_untrusted_streaminfo_sample_count(tmp_path):
    """the 36-bit sample count of STREAMINFO is untrusted.  A crafted count that the file's bytes cannot
    hold raises FlacError before anything is allocated (it used to size an np.empty of up to 2^36 * channels * 4 bytes);
    a count of ZERO (legal: streamed encoders do not know the length) still decodes, and wav_length counts the frames."""
    rng = np.random.default_rng(5)
    pcm = (2000 * np.sin(np.arange(9000)[:, None] * 0.01) + rng.integers(-30, 30, (9000, 1))).astype(np.int32)
    good = flac.encode(pcm, 44100, 16)
    for native_flag in (None, False):
        with pytest.raises(flac.FlacError, match="cannot hold"):
            flac.decode(_patched_total(good, (1 << 36) - 1), use_native=native_flag)
    unknown = _patched_total(good, 0)
    sr, got, bps = flac.decode(unknown, verify=False)
    assert sr == 44100 and bps == 16 and np.array_equal(got, pcm)
    p = str(tmp_path / "stream.flac")
    open(p, "wb").write(unknown)
    assert flac.info(p)[3] == 0 and audio_io.wav_length(p) == 9000 == len(audio_io.load_wav(p))


def test_native_handle_is_published_once_to_racing_threads(native_codec):
    """restore_folder's decode workers are the first callers of flac.native(), all at once; every one of
    them must get the C library (the loader used to publish ``False`` first, and racing threads fell back to the
    Python codec)."""
    import threading
    flac._NATIVE = None
    seen, gate = [], threading.Barrier(8)

    def worker():
        gate.wait()
        seen.append(flac.native())

    ts = [threading.Thread(target=worker) for _ in range(8)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert len(seen) == 8 and all(h is not None and h is seen[0] for h in seen)
