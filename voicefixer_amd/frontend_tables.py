"""Host-computed constant tables of the analysis front-end.

* periodic hann window and FFT twiddles (float64 -> float32);
* the 128-bin HTK mel filterbank of the reference, voicefixer/tools/mel_scale.py:147-238
  (``melscale_fbanks(1025, 0, 22050, 128, 44100, norm=None, 'htk')``), evaluated with the
  same float32 torch expressions in the same order so that its support set -- the mel bin
  indexing -- is bit-identical (sha256 5a05d24b... pinned in tests/golden/filterbank.npz),
  then stored banded: per mel bin the inclusive [lo, hi] FFT-bin range and its coefficients;
* the slaney-normalised HTK filterbank of ``Vocoder.oracle`` (librosa.filters.mel as called by
  voicefixer/vocoder/model/util.py:115-123), float64 -> float32, banded the same way.  Checked in
  tests/test_librosa_like.py against the reference-held ``Config.mel_weight_torch`` table and against the
  independent formulation in oracle/librosa_like.py.
"""
import math

import numpy as np
import torch

N_FFT = 2048
N_MELS = 128
SR = 44100


def mel_filterbank():
    all_freqs = torch.linspace(0, SR // 2, N_FFT // 2 + 1)
    m_min = 2595.0 * math.log10(1.0 + (0.0 / 700.0))
    m_max = 2595.0 * math.log10(1.0 + (float(SR // 2) / 700.0))
    m_pts = torch.linspace(m_min, m_max, N_MELS + 2)
    f_pts = 700.0 * (10.0 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    zero = torch.zeros(1)
    down_slopes = (-1.0 * slopes[:, :-2]) / f_diff[:-1]
    up_slopes = slopes[:, 2:] / f_diff[1:]
    return torch.max(zero, torch.min(down_slopes, up_slopes))


def slaney_mel_basis():
    """librosa.filters.mel(sr=44100, n_fft=2048, n_mels=128, fmin=0, fmax=22050, htk=True, norm="slaney")
    -> float32 (128, 1025); librosa 0.10 formulation (ramps / fdiff in float64, cast at the end)."""
    fftfreqs = np.linspace(0, SR / 2.0, N_FFT // 2 + 1)
    mmin, mmax = 0.0, 2595.0 * np.log10(1.0 + (SR // 2) / 700.0)
    mel_f = 700.0 * (10.0 ** (np.linspace(mmin, mmax, N_MELS + 2) / 2595.0) - 1.0)
    fdiff = np.diff(mel_f)
    ramps = np.subtract.outer(mel_f, fftfreqs)
    w = np.zeros((N_MELS, N_FFT // 2 + 1))
    for i in range(N_MELS):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        w[i] = np.maximum(0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2:N_MELS + 2] - mel_f[:N_MELS])
    return (w * enorm[:, None]).astype(np.float32)


def banded(fb):
    fb = (fb.numpy() if hasattr(fb, "numpy") else np.asarray(fb)).astype(np.float32)  # (1025, 128)
    lo = np.zeros(N_MELS, np.int32)
    hi = np.zeros(N_MELS, np.int32)
    off = np.zeros(N_MELS, np.int32)
    coef = []
    for m in range(N_MELS):
        nz = np.nonzero(fb[:, m])[0]
        lo[m], hi[m] = nz[0], nz[-1]
        off[m] = len(coef)
        coef.extend(fb[lo[m]:hi[m] + 1, m].tolist())
    return lo, hi, off, np.asarray(coef, np.float32)


def dense(lo, hi, off, coef):
    """Inverse of ``banded``: the (1025, 128) float32 matrix a banded table represents."""
    fb = np.zeros((N_FFT // 2 + 1, N_MELS), np.float32)
    for m in range(N_MELS):
        fb[lo[m]:hi[m] + 1, m] = coef[off[m]:off[m] + hi[m] - lo[m] + 1]
    return fb


def oracle_tables():
    """Banded slaney filterbank for vfx_frontend_init_oracle."""
    return banded(np.ascontiguousarray(slaney_mel_basis().T))


def tables():
    n = np.arange(N_FFT, dtype=np.float64)
    win = (0.5 - 0.5 * np.cos(2.0 * np.pi * n / N_FFT)).astype(np.float32)
    m = np.arange(N_FFT // 2, dtype=np.float64)
    ang = -2.0 * np.pi * m / N_FFT
    tw = np.stack([np.cos(ang), np.sin(ang)], axis=1).astype(np.float32).reshape(-1)
    lo, hi, off, coef = banded(mel_filterbank())
    return (np.ascontiguousarray(win), np.ascontiguousarray(tw), lo, hi, off, coef)
