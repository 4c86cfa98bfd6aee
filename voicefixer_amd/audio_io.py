"""File I/O at the edge of the path (host side, not timed, not on the kernel path).

Mirrors voicefixer/tools/wav.py: ``save_wave`` (:9-37, int16 truncation) and the
``librosa.load(path, sr=44100)`` call of voicefixer/base.py:47-49.  librosa / soundfile are not
available offline, so WAV files are read with scipy/stdlib and resampled with a polyphase
filter (librosa would use soxr_hq: resampled inputs are NOT bit-identical to the reference's;
44.1 kHz inputs are).  FLAC needs soundfile and is rejected loudly.
"""
import numpy as np

SR = 44100


def to_int16(frames):
    """voicefixer/tools/wav.py:27-34: x * 2^15 if max <= 1, then a truncating astype(np.short)
    (no rounding, no dither; values outside int16 wrap exactly like numpy's cast)."""
    frames = np.array(frames, copy=True)
    if np.max(frames) <= 1 and frames.dtype in (np.float32, np.float16, np.float64):
        frames = frames * 2 ** 15
    with np.errstate(invalid="ignore"):
        return frames.astype(np.short)


def save_wave(frames, fname, sample_rate=SR):
    """(1, N) or (N,) float waveform -> PCM16 WAV (voicefixer/tools/wav.py:9-37)."""
    frames = np.asarray(frames)
    if frames.ndim == 1:
        frames = frames[..., None]
    elif frames.ndim == 2 and frames.shape[0] < frames.shape[1] and frames.shape[0] <= 2:
        frames = frames.T  # (channels, N) -> (N, channels), as the reference's (1, N) output
    pcm = to_int16(frames)
    if not str(fname).lower().endswith(".wav"):
        raise RuntimeError("only .wav output is supported offline (the reference writes via soundfile): %s" % fname)
    from scipy.io import wavfile
    wavfile.write(fname, sample_rate, pcm if pcm.shape[1] > 1 else pcm[:, 0])


def wav_length(path, sample_rate=SR):
    """Number of samples ``load_wav(path, sample_rate)`` will return, from the header alone (memory-mapped read)."""
    from scipy.io import wavfile
    sr, data = wavfile.read(path, mmap=True)
    n = int(data.shape[0])
    if sr == sample_rate:
        return n
    from math import gcd
    g = gcd(int(sr), int(sample_rate))
    up, down = sample_rate // g, sr // g
    return -(-n * up // down)  # resample_poly: ceil(n * up / down)


def load_wav(path, sample_rate=SR, mono=True):
    """Decode + (if needed) resample + downmix, float32 in [-1, 1] (librosa.load semantics)."""
    if not str(path).lower().endswith(".wav"):
        raise RuntimeError("only .wav input is supported offline (no libsndfile/FLAC decoder): %s" % path)
    from scipy.io import wavfile
    sr, data = wavfile.read(path)
    if data.dtype == np.int16:
        x = data.astype(np.float32) / 32768.0
    elif data.dtype == np.int32:
        x = data.astype(np.float32) / 2147483648.0
    elif data.dtype == np.uint8:
        x = (data.astype(np.float32) - 128.0) / 128.0
    else:
        x = data.astype(np.float32)
    if x.ndim == 2:
        x = x.mean(axis=1) if mono else x.T
    if sr != sample_rate:
        from math import gcd
        from scipy.signal import resample_poly
        g = gcd(int(sr), int(sample_rate))
        x = resample_poly(x, sample_rate // g, sr // g, axis=-1).astype(np.float32)
    return np.ascontiguousarray(x, dtype=np.float32)
