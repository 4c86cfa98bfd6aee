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
