"""File I/O at the edge of the path (host side, not timed, not on the kernel path).

Mirrors voicefixer/tools/wav.py: ``save_wave`` (:9-37, int16 truncation) and the
``librosa.load(path, sr=44100)`` call of voicefixer/base.py:47-49.  librosa / soundfile are not
available offline, so WAV files are read with scipy/stdlib, FLAC files (the format of the reference's own
test fixtures, test/test.py:45-75) with the decoder / encoder of ``flac.py``, and other sample rates are resampled with a
linear-phase polyphase filter designed to the published soxr "HQ" recipe librosa.load uses by default (pass band to
0.913 of the lower Nyquist frequency, stop band from that Nyquist frequency on, 125 dB rejection: ``resample_hq``).  Same
grade, not the same coefficients: resampled inputs agree with the reference's to the filter ripple, not bit for bit;
44.1 kHz inputs are untouched.
"""
import struct

import numpy as np

SR = 44100
FORMATS = (".wav", ".flac")   # what ``save_wave`` can write (the reference asks soundfile.available_formats())


def to_int16(frames):
    """voicefixer/tools/wav.py:27-34: x * 2^15 if max <= 1, then a truncating astype(np.short)
    (no rounding, no dither; values outside int16 wrap exactly like numpy's cast)."""
    frames = np.array(frames, copy=True)
    if np.max(frames) <= 1 and frames.dtype in (np.float32, np.float16, np.float64):
        frames = frames * 2 ** 15
    with np.errstate(invalid="ignore"):
        return frames.astype(np.short)


def save_wave(frames, fname, sample_rate=SR):
    """(1, N) or (N,) float waveform -> PCM16 WAV or FLAC by extension (voicefixer/tools/wav.py:9-37 writes through
    soundfile, which picks the container from the extension the same way)."""
    frames = np.asarray(frames)
    if frames.ndim == 1:
        frames = frames[..., None]
    elif frames.ndim == 2 and frames.shape[0] < frames.shape[1] and frames.shape[0] <= 2:
        frames = frames.T  # (channels, N) -> (N, channels), as the reference's (1, N) output
    pcm = to_int16(frames)
    low = str(fname).lower()
    if low.endswith(".flac"):
        from . import flac
        flac.write(fname, pcm, sample_rate, 16)
        return
    if not low.endswith(".wav"):
        raise RuntimeError("output format not supported offline (WAV and FLAC are; the reference writes via "
                           "soundfile): %s" % fname)
    from scipy.io import wavfile
    wavfile.write(fname, sample_rate, pcm if pcm.shape[1] > 1 else pcm[:, 0])


def _riff_info(path):
    """(sample_rate, frames, frames the header promises) of a RIFF/WAVE file from its ``fmt `` and ``data`` chunk headers alone
    -- any bit depth (scipy's memory-mapped read refuses 24-bit PCM, a very common studio format)."""
    with open(path, "rb") as f:
        head = f.read(12)
        if len(head) < 12 or head[:4] not in (b"RIFF", b"RF64") or head[8:12] != b"WAVE":
            raise RuntimeError("not a RIFF/WAVE file: %s" % path)
        sr = block_align = None
        while True:
            ck = f.read(8)
            if len(ck) < 8:
                raise RuntimeError("WAV file without a data chunk: %s" % path)
            cid, size = ck[:4], struct.unpack("<I", ck[4:])[0]
            if cid == b"fmt ":
                fmt = f.read(size + (size & 1))
                _, _, sr, _, block_align, _ = struct.unpack("<HHIIHH", fmt[:16])
            elif cid == b"data":
                if not block_align:
                    raise RuntimeError("WAV data chunk before the fmt chunk: %s" % path)
                # the header PLANS (staging width of the folder job): never believe more than the file can hold -- a streamed /
                # RF64 file says 0xFFFFFFFF, a truncated or lying one promises bytes that are not there
                pos = f.tell()
                f.seek(0, 2)
                real = min(size, f.tell() - pos)
                return int(sr), int(real // block_align), int((real if size == 0xFFFFFFFF else size) // block_align)
            else:
                f.seek(size + (size & 1), 1)


def wav_length(path, sample_rate=SR, with_promise=False):
    """Number of samples ``load_wav(path, sample_rate)`` will return, from the file header (and the file size) alone;
    ``with_promise``: (that number, the number the header PROMISES) -- they differ for a truncated recording."""
    promised = None
    if str(path).lower().endswith(".flac"):
        from . import flac
        sr, _, _, n = flac.info(path)
        # STREAMINFO's total is a promise; a frame holds at most 65535 samples per channel in no fewer than ~8 bytes, which bounds
        # what a file of this size can decode to (the folder job sizes pinned staging from this number)
        import os
        if n == 0 or n > (os.path.getsize(path) // 8 + 1) * 65535:      # 0: legal for streamed encoders (unknown length)
            n = flac.read(path)[1].shape[0]     # count what is really there
    else:
        sr, n, promised = _riff_info(path)
    promised = n if promised is None else promised
    if sr != sample_rate:
        from math import gcd
        g = gcd(int(sr), int(sample_rate))
        up, down = sample_rate // g, sr // g
        n, promised = -(-n * up // down), -(-promised * up // down)  # resample_poly: ceil(n * up / down)
    return (n, promised) if with_promise else n


_HQ_FILTERS = {}


def resample_hq(x, sr_in, sr_out, use_native=None):
    """Band-limited rate conversion along the last axis, ceil(n * sr_out / sr_in) samples (librosa.load(sr=...) ->
    soxr_hq).  One Kaiser-windowed sinc at the common rate up * sr_in: pass band edge 0.913 and stop band edge 1.0 of
    the lower of the two Nyquist frequencies, 125 dB (beta = 0.1102 (A - 8.7), length from Kaiser's estimate), applied
    as a polyphase filter: one dot product per output sample in libvfx_audio.so (vfx_resample_poly_f32: float32, several
    times scipy's speed and without the interpreter lock, so the folder driver's workers scale), or scipy's upfirdn in
    float64 when that library is not built / ``use_native=False`` (the two agree to float32 rounding)."""
    from math import gcd
    g = gcd(int(sr_in), int(sr_out))
    up, down = int(sr_out) // g, int(sr_in) // g
    h = _HQ_FILTERS.get((up, down))
    if h is None:
        from scipy.signal import firwin
        m = max(up, down)                      # the lower Nyquist frequency is 1 / m of the common rate's
        att, f_pass, f_stop = 125.0, 0.913 / m, 1.0 / m
        taps = int(np.ceil((att - 7.95) / (2.285 * np.pi * (f_stop - f_pass)))) | 1    # odd: zero phase
        h = firwin(taps, 0.5 * (f_pass + f_stop), window=("kaiser", 0.1102 * (att - 8.7)))       # (unit DC gain; the gain `up` is applied below)
        h = (h, np.ascontiguousarray(h * up, dtype=np.float32))
        _HQ_FILTERS[(up, down)] = h
    from . import flac
    lib = flac.native() if use_native in (None, True) else None
    x = np.asarray(x)
    if lib is not None and x.shape[-1] > 0:
        import ctypes
        f32p = ctypes.POINTER(ctypes.c_float)
        xin = np.ascontiguousarray(x, dtype=np.float32).reshape(-1, x.shape[-1])
        n = xin.shape[1]
        ny = -(-n * up // down)
        out = np.empty((xin.shape[0], ny), dtype=np.float32)
        for r in range(xin.shape[0]):
            rc = lib.vfx_resample_poly_f32(xin[r].ctypes.data_as(f32p), n, h[1].ctypes.data_as(f32p), h[1].shape[0], up, down,
                                           out[r].ctypes.data_as(f32p), ny)
            if rc:
                raise RuntimeError("vfx_resample_poly_f32 failed (%d)" % rc)
        return out.reshape(x.shape[:-1] + (ny,))
    from scipy.signal import resample_poly
    return resample_poly(np.asarray(x, dtype=np.float64), up, down, axis=-1, window=h[0]).astype(np.float32)


def load_wav(path, sample_rate=SR, mono=True):
    """Decode + (if needed) resample + downmix, float32 in [-1, 1] (librosa.load semantics)."""
    low = str(path).lower()
    if low.endswith(".flac"):
        from . import flac
        sr, pcm, bps = flac.read(path)
        x = pcm.astype(np.float32) / float(1 << (bps - 1))     # soundfile's float conversion
        x = x[:, 0] if x.shape[1] == 1 else x
    elif low.endswith(".wav"):
        from scipy.io import wavfile
        sr, data = wavfile.read(path)
        if data.dtype == np.int16:
            x = data.astype(np.float32) / 32768.0
        elif data.dtype == np.int32:
            x = data.astype(np.float32) / 2147483648.0      # (scipy delivers 24-bit PCM left-justified in int32)
        elif data.dtype == np.uint8:
            x = (data.astype(np.float32) - 128.0) / 128.0
        else:
            x = data.astype(np.float32)
    else:
        raise RuntimeError("input format not supported offline (WAV and FLAC are): %s" % path)
    if x.ndim == 2:
        x = x.mean(axis=1) if mono else x.T
    if sr != sample_rate:
        x = resample_hq(x, sr, sample_rate)
    return np.ascontiguousarray(x, dtype=np.float32)
