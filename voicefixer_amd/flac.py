"""FLAC decode / encode without libsndfile (host side, file I/O at the edge of the path; not timed).

The reference reads and writes ``.flac`` through librosa / soundfile (voicefixer/base.py:47-49,
voicefixer/tools/wav.py:36-37; its only test does, test/test.py:45-75).  Neither library nor libFLAC exists in this
image, so the subset of the format those files use is implemented here from the format specification:

  decode  -- every subframe type (CONSTANT, VERBATIM, FIXED order 0-4, LPC order 1-32), both Rice coding methods with
             partitions and escape codes, wasted bits, the four channel assignments (independent, left/side, right/side,
             mid/side), 4..32 bits per sample, fixed and variable block sizes.  The frame-header CRC-8, the frame
             CRC-16 and the STREAMINFO MD5 of the decoded PCM are VERIFIED (a mismatch raises): every decode is
             self-checking, and the CRC code the encoder shares is pinned by the reference's own files.
  encode  -- a valid stream a stock decoder accepts: FIXED order-2 prediction, one Rice partition per subframe
             (parameter chosen per subframe; VERBATIM where Rice would be longer), independent channels, block size
             4096, MD5 and total-sample count in STREAMINFO, CRC-8 / CRC-16 on every frame.

Two implementations of the frame level, bit-exact against each other in both directions (tests/test_flac.py):
  * this module -- the specification in Python (Rice decoding is sequential by nature: a loop over precomputed per-bit
    tables, ~14x real time per thread, under the interpreter lock);
  * ``libvfx_audio.so`` (voicefixer_amd/csrc_host/vfx_flac.c, C ABI include/vfx_audio.h, built by
    ``__graft_entry__.build()``) -- the same decoder / encoder in C, several hundred times real time per thread and
    called through ctypes WITHOUT the interpreter lock, which is what lets restore_folder's worker pool keep up with the
    device.  ``decode`` / ``encode`` use it when it is there (``VFX_FLAC_NATIVE=0``: never); metadata blocks,
    STREAMINFO, the MD5 check and argument checking are shared Python code either way.
"""
import ctypes
import threading
import hashlib
import os
import struct

import numpy as np

from ._dev import dev_env


class FlacError(RuntimeError):
    pass


# ---------------------------------------------------------------------------------------------------- native codec
_NATIVE = None          # None = not looked for yet, False = unavailable / switched off, else the ctypes handle
_NATIVE_ERRORS = {2: "FLAC: lost frame sync at byte %d", 3: "FLAC: frame header CRC-8 mismatch at byte %d",
                  4: "FLAC: frame CRC-16 mismatch in the frame at byte %d", 5: "FLAC: reserved field in the frame at byte %d",
                  6: "FLAC: Rice code runs past the end of the stream (frame at byte %d)",
                  7: "FLAC: channel count changes mid-stream (frame at byte %d)",
                  8: "FLAC: more samples than STREAMINFO announces (frame at byte %d)", 9: "FLAC: out of memory (frame at byte %d)"}


def build_native(verbose=False):
    """Compile libvfx_audio.so in-tree (``make -C voicefixer_amd/csrc_host``: one C file, gcc); returns its path."""
    import subprocess
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run(["make", "-C", os.path.join(here, "csrc_host")], capture_output=True, text=True)
    if verbose or r.returncode != 0:
        print(r.stdout[-2000:])
        print(r.stderr[-2000:])
    if r.returncode != 0:
        raise FlacError("building libvfx_audio.so failed (gcc)")
    global _NATIVE
    _NATIVE = None
    return os.path.join(here, "libvfx_audio.so")


_NATIVE_LOCK = threading.Lock()


def native():
    """ctypes handle of libvfx_audio.so, or None (not built, or VFX_FLAC_NATIVE=0).  Thread-safe: restore_folder's
    decode workers may be the first callers, all at once -- the handle is built under a lock and published only when
    its prototypes are complete, so no thread can see a half-initialised state and fall back to the Python codec."""
    global _NATIVE
    if _NATIVE is None:
        with _NATIVE_LOCK:
            if _NATIVE is None:
                _NATIVE = _load_native() or False
    return _NATIVE or None


def _load_native():
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libvfx_audio.so")
    if dev_env("VFX_FLAC_NATIVE", "1") == "0" or not os.path.exists(path):
        return None
    h = ctypes.CDLL(path)
    u8p, i32p = ctypes.POINTER(ctypes.c_ubyte), ctypes.POINTER(ctypes.c_int)
    ull, ullp = ctypes.c_ulonglong, ctypes.POINTER(ctypes.c_ulonglong)
    h.vfx_audio_version.restype = ctypes.c_int
    h.vfx_flac_decode_frames.restype = ctypes.c_int
    h.vfx_flac_decode_frames.argtypes = [ctypes.c_char_p, ull, ull, ctypes.c_int, ctypes.c_int, i32p, ull, ullp,
                                         ctypes.c_int, ullp]
    h.vfx_flac_encode_frames.restype = ctypes.c_longlong
    h.vfx_flac_encode_frames.argtypes = [i32p, ull, ctypes.c_int, ctypes.c_int, ctypes.c_int, u8p, ull,
                                         ctypes.POINTER(ctypes.c_uint), ctypes.POINTER(ctypes.c_uint)]
    f32p = ctypes.POINTER(ctypes.c_float)
    h.vfx_resample_poly_f32.restype = ctypes.c_int      # (audio_io.resample_hq; same library)
    h.vfx_resample_poly_f32.argtypes = [f32p, ull, f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, f32p, ull]
    return h if h.vfx_audio_version() >= 100 else None


def _decode_frames_native(h, data, pos, nch, bps0, total, verify):
    out = np.empty((total, nch), dtype=np.int32)
    done, err_at = ctypes.c_ulonglong(0), ctypes.c_ulonglong(0)
    rc = h.vfx_flac_decode_frames(data, len(data), pos, nch, bps0, out.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), total,
                                  ctypes.byref(done), 1 if verify else 0, ctypes.byref(err_at))
    if rc:
        raise FlacError(_NATIVE_ERRORS.get(rc, "FLAC: decoder error %d (frame at byte %%d)" % rc) % err_at.value)
    return out[:done.value]


def _crc_table(poly, bits):
    top = 1 << (bits - 1)
    mask = (1 << bits) - 1
    tab = []
    for i in range(256):
        c = i << (bits - 8)
        for _ in range(8):
            c = ((c << 1) ^ poly) & mask if c & top else (c << 1) & mask
        tab.append(c)
    return tab


_CRC8 = _crc_table(0x07, 8)
_CRC16 = _crc_table(0x8005, 16)


def _crc8(data):
    c = 0
    for b in data:
        c = _CRC8[c ^ b]
    return c


def _crc16(data):
    c = 0
    for b in data:
        c = ((c << 8) & 0xFFFF) ^ _CRC16[(c >> 8) ^ b]
    return c


# ---------------------------------------------------------------------------------------------------------- decoder
class _Bits:
    """MSB-first bit reader over a bytes object; ``pos`` is a bit index."""

    def __init__(self, data):
        self.data = data
        self.pos = 0
        self._next_one = None
        self._win = None

    def read(self, n):
        if n == 0:
            return 0
        p = self.pos
        b0, b1 = p >> 3, (p + n + 7) >> 3
        v = int.from_bytes(self.data[b0:b1], "big")
        v >>= (b1 << 3) - (p + n)
        self.pos = p + n
        return v & ((1 << n) - 1)

    def read_signed(self, n):
        v = self.read(n)
        return v - (1 << n) if n and v >> (n - 1) else v

    def align(self):
        self.pos = (self.pos + 7) & ~7

    # tables for the Rice loop: next_one[p] = position of the first 1 bit at or after p; win[p] = the 32 bits from p
    def tables(self):
        if self._next_one is None:
            raw = np.frombuffer(self.data, np.uint8)
            bits = np.unpackbits(raw)
            n = bits.shape[0]
            idx = np.where(bits == 1, np.arange(n, dtype=np.int64), np.int64(n))
            self._next_one = np.minimum.accumulate(idx[::-1])[::-1]
            pad = np.concatenate([raw, np.zeros(8, np.uint8)]).astype(np.uint64)
            words = np.zeros(raw.shape[0] + 1, np.uint64)          # 64 bits starting at every byte
            for k in range(8):
                words |= pad[k:k + words.shape[0]] << np.uint64(56 - 8 * k)
            self._words = words
        return self._next_one, self._words


def _rice_partition(br, n, k, out, o0):
    """n Rice(k)-coded residuals starting at br.pos -> out[o0:o0+n]."""
    next_one, words = br.tables()
    total = len(br.data) << 3
    # vectorised in rounds is not possible (every code's start depends on the previous one's length): plain loop over
    # Python lists of the two tables restricted to this partition's reach
    p = br.pos
    no = next_one
    wd = words
    kmask = (1 << k) - 1
    sh0 = 64 - k
    for i in range(o0, o0 + n):
        e = int(no[p])
        if e >= total:
            raise FlacError("FLAC: Rice code runs past the end of the stream")
        q = e - p
        p = e + 1
        if k:
            r = (int(wd[p >> 3]) >> (sh0 - (p & 7))) & kmask
            p += k
            u = (q << k) | r
        else:
            u = q
        out[i] = (u >> 1) ^ -(u & 1)
    br.pos = p


def _residual(br, blocksize, order, out):
    method = br.read(2)
    if method > 1:
        raise FlacError("FLAC: reserved residual coding method")
    pbits = 4 if method == 0 else 5
    esc = (1 << pbits) - 1
    porder = br.read(4)
    nparts = 1 << porder
    o = order
    for part in range(nparts):
        n = (blocksize >> porder) - (order if part == 0 else 0)
        if n < 0:
            raise FlacError("FLAC: partition smaller than the predictor order")
        k = br.read(pbits)
        if k == esc:
            nb = br.read(5)
            for i in range(o, o + n):
                out[i] = br.read_signed(nb) if nb else 0
        else:
            _rice_partition(br, n, k, out, o)
        o += n


_FIXED = {0: (), 1: (1,), 2: (2, -1), 3: (3, -3, 1), 4: (4, -6, 4, -1)}


def _predict(out, order, coefs, shift, n):
    """out[:order] warm-up, out[order:] residuals -> samples in place (integer arithmetic of the format: the prediction
    is an arithmetic right shift of the integer dot product)."""
    if order == 0:
        return
    if shift == 0 and coefs == _FIXED.get(order):
        # fixed predictors: small constant integer taps, unrolled
        x = list(out[:n])
        if order == 1:
            for i in range(1, n):
                x[i] += x[i - 1]
        elif order == 2:
            for i in range(2, n):
                x[i] += 2 * x[i - 1] - x[i - 2]
        elif order == 3:
            for i in range(3, n):
                x[i] += 3 * x[i - 1] - 3 * x[i - 2] + x[i - 3]
        else:
            for i in range(4, n):
                x[i] += 4 * x[i - 1] - 6 * x[i - 2] + 4 * x[i - 3] - x[i - 4]
        out[:n] = x
        return
    x = list(out[:n])
    c = list(coefs)
    rng = range(order)
    for i in range(order, n):
        s = 0
        for j in rng:
            s += c[j] * x[i - 1 - j]
        x[i] += s >> shift
    out[:n] = x


def _subframe(br, blocksize, bps):
    if br.read(1):
        raise FlacError("FLAC: subframe padding bit set")
    typ = br.read(6)
    wasted = 0
    if br.read(1):
        wasted = 1
        while br.read(1) == 0:
            wasted += 1
        bps -= wasted
    out = [0] * blocksize
    if typ == 0:
        out = [br.read_signed(bps)] * blocksize
    elif typ == 1:
        for i in range(blocksize):
            out[i] = br.read_signed(bps)
    elif 8 <= typ <= 12:
        order = typ - 8
        for i in range(order):
            out[i] = br.read_signed(bps)
        _residual(br, blocksize, order, out)
        _predict(out, order, _FIXED[order], 0, blocksize)
    elif typ >= 32:
        order = typ - 31
        for i in range(order):
            out[i] = br.read_signed(bps)
        prec = br.read(4) + 1
        if prec == 16:
            raise FlacError("FLAC: reserved LPC precision")
        shift = br.read_signed(5)
        if shift < 0:
            raise FlacError("FLAC: negative LPC shift")
        coefs = tuple(br.read_signed(prec) for _ in range(order))
        _residual(br, blocksize, order, out)
        _predict(out, order, coefs, shift, blocksize)
    else:
        raise FlacError("FLAC: reserved subframe type %d" % typ)
    if wasted:
        out = [v << wasted for v in out]
    return out


_BLOCK = {1: 192, 2: 576, 3: 1152, 4: 2304, 5: 4608, 8: 256, 9: 512, 10: 1024, 11: 2048, 12: 4096, 13: 8192, 14: 16384,
          15: 32768}
_BPS = {1: 8, 2: 12, 4: 16, 5: 20, 6: 24, 7: 32}


def decode(data, verify=True, use_native=None):
    """bytes of a FLAC file -> (sample_rate, int32 array (n, channels), bits per sample).  ``use_native``: None = the C
    frame decoder when it is built, False = the Python one, True = the C one (None if unavailable -> Python)."""
    if data[:4] != b"fLaC":
        raise FlacError("not a FLAC stream (missing fLaC marker)")
    pos = 4
    info = None
    while True:
        if pos + 4 > len(data):
            raise FlacError("FLAC: the stream ends inside the metadata blocks")
        hdr = data[pos]
        length = int.from_bytes(data[pos + 1:pos + 4], "big")
        body = data[pos + 4:pos + 4 + length]
        pos += 4 + length
        if pos > len(data) or (hdr & 0x7F == 0 and length < 34):
            raise FlacError("FLAC: the stream ends inside the metadata blocks")
        if hdr & 0x7F == 0:
            v = int.from_bytes(body[10:18], "big")
            info = {"sr": v >> 44, "ch": ((v >> 41) & 7) + 1, "bps": ((v >> 36) & 31) + 1, "total": v & ((1 << 36) - 1),
                    "md5": body[18:34], "max_block": int.from_bytes(body[2:4], "big") or 65535}
        if hdr & 0x80:
            break
    if info is None:
        raise FlacError("FLAC: no STREAMINFO block")
    nch, bps0 = info["ch"], info["bps"]
    # STREAMINFO's 36-bit sample count is untrusted input: a frame is at least 9 bytes (header, one subframe byte,
    # CRC-16) and holds at most max_block samples, so the file cannot contain more than this -- a crafted count must
    # not size an allocation
    if info["total"] > ((len(data) - pos) // 9 + 1) * info["max_block"]:
        raise FlacError("FLAC: STREAMINFO announces %d samples, the %d bytes of frames cannot hold them" % (info["total"], len(data) - pos))
    h = native() if use_native is None else (native() if use_native else None)
    if h is not None and info["total"] > 0 and 4 <= bps0 <= 32:
        pcm = _decode_frames_native(h, bytes(data), pos, nch, bps0, info["total"], verify)
        return _finish_decode(pcm, info, bps0, verify)
    br = _Bits(data)
    br.pos = pos << 3
    chans = [[] for _ in range(nch)]
    end = len(data) << 3
    while br.pos + 16 <= end:
        start = br.pos >> 3
        if br.read(14) != 0x3FFE:
            raise FlacError("FLAC: lost frame sync at byte %d" % start)
        br.read(1)
        br.read(1)                      # blocking strategy (the frame / sample number is not needed to decode in order)
        bcode, scode = br.read(4), br.read(4)
        cassign, zcode = br.read(4), br.read(3)
        br.read(1)
        first = br.read(8)              # UTF-8-style coded number: skip its continuation bytes
        extra = 0
        while first & 0x80 and extra < 7:
            first = (first << 1) & 0xFF
            extra += 1
        for _ in range(max(extra - 1, 0)):
            br.read(8)
        if bcode == 6:
            blocksize = br.read(8) + 1
        elif bcode == 7:
            blocksize = br.read(16) + 1
        elif bcode in _BLOCK:
            blocksize = _BLOCK[bcode]
        else:
            raise FlacError("FLAC: reserved block size code")
        if scode == 12:
            br.read(8)
        elif scode in (13, 14):
            br.read(16)
        crc = br.read(8)
        if verify and _crc8(data[start:(br.pos >> 3) - 1]) != crc:
            raise FlacError("FLAC: frame header CRC-8 mismatch at byte %d" % start)
        bps = _BPS.get(zcode, bps0) if zcode else bps0
        if cassign < 8:
            if cassign + 1 != nch:
                raise FlacError("FLAC: channel count changes mid-stream")
            sub = [_subframe(br, blocksize, bps) for _ in range(nch)]
        elif cassign == 8:              # left, side
            left = _subframe(br, blocksize, bps)
            side = _subframe(br, blocksize, bps + 1)
            sub = [left, [l - s for l, s in zip(left, side)]]
        elif cassign == 9:              # side, right
            side = _subframe(br, blocksize, bps + 1)
            right = _subframe(br, blocksize, bps)
            sub = [[s + r for s, r in zip(side, right)], right]
        elif cassign == 10:             # mid, side
            mid = _subframe(br, blocksize, bps)
            side = _subframe(br, blocksize, bps + 1)
            left, right = [], []
            for m, s in zip(mid, side):
                m = (m << 1) | (s & 1)
                left.append((m + s) >> 1)
                right.append((m - s) >> 1)
            sub = [left, right]
        else:
            raise FlacError("FLAC: reserved channel assignment")
        br.align()
        end_of_frame = br.pos >> 3
        if verify and _crc16(data[start:end_of_frame]) != br.read(16):
            raise FlacError("FLAC: frame CRC-16 mismatch in the frame at byte %d" % start)
        br.pos = (end_of_frame + 2) << 3
        for c in range(nch):
            chans[c].extend(sub[c])
    pcm = np.array(chans, dtype=np.int64).T
    return _finish_decode(pcm, info, bps0, verify)


def _finish_decode(pcm, info, bps0, verify):
    if info["total"] and pcm.shape[0] != info["total"]:
        raise FlacError("FLAC: decoded %d samples, STREAMINFO says %d" % (pcm.shape[0], info["total"]))
    if verify and any(info["md5"]):
        if hashlib.md5(_pcm_bytes(pcm, bps0)).digest() != info["md5"]:
            raise FlacError("FLAC: MD5 of the decoded audio does not match STREAMINFO")
    return info["sr"], pcm.astype(np.int32), bps0


def _pcm_bytes(pcm, bps):
    """Interleaved little-endian signed samples, (bps + 7) // 8 bytes each: what the STREAMINFO MD5 is taken over."""
    nbytes = (bps + 7) // 8
    flat = np.ascontiguousarray(pcm, dtype=np.int64).reshape(-1)
    if nbytes == 2:
        return flat.astype("<i2").tobytes()
    if nbytes == 4:
        return flat.astype("<i4").tobytes()
    if nbytes == 1:
        return flat.astype("i1").tobytes()
    b = flat.astype("<i4").view(np.uint8).reshape(-1, 4)[:, :3]
    return np.ascontiguousarray(b).tobytes()


def read(path):
    """-> (sample_rate, int32 (n, channels), bits per sample)."""
    with open(path, "rb") as f:
        return decode(f.read())


def info(path):
    """(sample_rate, channels, bits per sample, total samples) from the STREAMINFO block alone."""
    with open(path, "rb") as f:
        head = f.read(42)
    if head[:4] != b"fLaC" or head[4] & 0x7F != 0:
        raise FlacError("not a FLAC stream")
    v = int.from_bytes(head[18:26], "big")
    return v >> 44, ((v >> 41) & 7) + 1, ((v >> 36) & 31) + 1, v & ((1 << 36) - 1)


# ---------------------------------------------------------------------------------------------------------- encoder
def _utf8_number(v):
    """The frame number in the format's extended UTF-8 coding (up to 36 bits)."""
    if v < 0x80:
        return bytes([v])
    for nbytes, lead in ((2, 0xC0), (3, 0xE0), (4, 0xF0), (5, 0xF8), (6, 0xFC), (7, 0xFE)):
        if v < 1 << (5 * nbytes + 1 if nbytes < 7 else 36):
            tail = [0x80 | ((v >> (6 * k)) & 0x3F) for k in range(nbytes - 2, -1, -1)]
            return bytes([lead | (v >> (6 * (nbytes - 1)))] + tail)
    raise FlacError("FLAC: frame number too large")


def _pack_subframe(x, bps):
    """One channel of one block -> (uint8 bit array of the subframe).  FIXED order 2 + Rice (one partition) or VERBATIM."""
    n = x.shape[0]
    if n > 2:
        res = x[2:] - 2 * x[1:-1] + x[:-2]
        u = np.where(res >= 0, 2 * res, -2 * res - 1).astype(np.int64)
        mean = float(u.mean()) if u.size else 0.0
        k = max(0, min(14, int(np.floor(np.log2(mean + 1.0)))))
        best = None
        for kk in {max(k - 1, 0), k, min(k + 1, 14)}:
            bits = int(((u >> kk) + 1 + kk).sum())
            if best is None or bits < best[0]:
                best = (bits, kk)
        rice_bits, k = best
        rice_total = 8 + 2 * bps + 2 + 4 + 4 + rice_bits
        if rice_total < 8 + n * bps and int((u >> k).max(initial=0)) < (1 << 20) and np.abs(res).max(initial=0) < (1 << 31):
            arr = np.zeros(rice_total, np.uint8)
            head = (0b0001010 << 1)                       # padding 0, type 001010 (FIXED order 2), no wasted bits
            _put(arr, 0, head, 8)
            _put(arr, 8, int(x[0]) & ((1 << bps) - 1), bps)
            _put(arr, 8 + bps, int(x[1]) & ((1 << bps) - 1), bps)
            o = 8 + 2 * bps
            _put(arr, o, 0, 2)                            # Rice coding method 0 (4-bit parameters)
            _put(arr, o + 2, 0, 4)                        # partition order 0
            _put(arr, o + 6, k, 4)
            o += 10
            q = u >> k
            lens = q + 1 + k
            starts = o + np.concatenate([[0], np.cumsum(lens)[:-1]])
            arr[starts + q] = 1                           # unary: q zeros, then the stop bit
            r = u & ((1 << k) - 1)
            for b in range(k):
                arr[starts + q + 1 + b] = (r >> (k - 1 - b)) & 1
            return arr
    arr = np.zeros(8 + n * bps, np.uint8)
    _put(arr, 0, 0b00000010, 8)                           # VERBATIM
    v = (x.astype(np.int64) & ((1 << bps) - 1))
    for b in range(bps):
        arr[8 + b + bps * np.arange(n)] = (v >> (bps - 1 - b)) & 1
    return arr


def _put(arr, off, value, nbits):
    for b in range(nbits):
        arr[off + b] = (value >> (nbits - 1 - b)) & 1


def encode(pcm, sample_rate, bps=16, blocksize=4096, use_native=None):
    """int array (n,) or (n, channels), values inside the signed ``bps``-bit range -> bytes of a FLAC file.
    ``use_native`` as in ``decode`` (both encoders write the same bytes)."""
    pcm = np.asarray(pcm)
    if pcm.ndim == 1:
        pcm = pcm[:, None]
    pcm = pcm.astype(np.int64)
    n, nch = pcm.shape
    if not 1 <= nch <= 8:
        raise FlacError("FLAC: 1..8 channels")
    if n and (pcm.max() >= 1 << (bps - 1) or pcm.min() < -(1 << (bps - 1))):
        raise FlacError("FLAC: sample outside the %d-bit range" % bps)
    zcode = {8: 1, 12: 2, 16: 4, 20: 5, 24: 6}.get(bps)
    if zcode is None:
        raise FlacError("FLAC: unsupported bits per sample %d" % bps)
    h = native() if use_native is None else (native() if use_native else None)
    if h is not None and n:
        p32 = np.ascontiguousarray(pcm, dtype=np.int32)
        cap = n * nch * 4 + 32 * (n // blocksize + 2) + 64
        buf = np.empty(cap, dtype=np.uint8)
        mn, mx = ctypes.c_uint(0), ctypes.c_uint(0)
        nb = h.vfx_flac_encode_frames(p32.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), n, nch, bps, blocksize,
                                      buf.ctypes.data_as(ctypes.POINTER(ctypes.c_ubyte)), cap, ctypes.byref(mn), ctypes.byref(mx))
        if nb < 0:
            raise FlacError("FLAC: encoder error %d" % -nb)
        return _stream(pcm, sample_rate, nch, bps, n, blocksize, mn.value, mx.value, buf[:nb].tobytes())
    frames = []
    min_f, max_f = 1 << 24, 0
    for fi, s0 in enumerate(range(0, n, blocksize)):
        blk = pcm[s0:s0 + blocksize]
        bs = blk.shape[0]
        hdr = bytearray()
        hdr += bytes([0xFF, 0xF8])                        # sync, reserved 0, fixed block size stream
        hdr.append((7 << 4) | 0)                          # block size: 16-bit value follows; sample rate: from STREAMINFO
        hdr.append(((nch - 1) << 4) | (zcode << 1))
        hdr += _utf8_number(fi)
        hdr += struct.pack(">H", bs - 1)
        hdr.append(_crc8(hdr))
        body = np.concatenate([_pack_subframe(blk[:, c], bps) for c in range(nch)])
        frame = bytes(hdr) + np.packbits(body).tobytes()  # (packbits pads the last byte with zeros = the alignment padding)
        frame += struct.pack(">H", _crc16(frame))
        frames.append(frame)
        min_f, max_f = min(min_f, len(frame)), max(max_f, len(frame))
    if not frames:
        min_f = max_f = 0
    return _stream(pcm, sample_rate, nch, bps, n, blocksize, min_f, max_f, b"".join(frames))


def _stream(pcm, sample_rate, nch, bps, n, blocksize, min_f, max_f, frames):
    """fLaC marker + STREAMINFO (the only metadata block) + the frames."""
    md5 = hashlib.md5(_pcm_bytes(pcm, bps)).digest()
    v = (sample_rate << 44) | ((nch - 1) << 41) | ((bps - 1) << 36) | n
    si = struct.pack(">HH", blocksize, blocksize) + min_f.to_bytes(3, "big") + max_f.to_bytes(3, "big") + \
        v.to_bytes(8, "big") + md5
    return b"fLaC" + bytes([0x80]) + len(si).to_bytes(3, "big") + si + frames


def write(path, pcm, sample_rate, bps=16):
    with open(path, "wb") as f:
        f.write(encode(pcm, sample_rate, bps))
