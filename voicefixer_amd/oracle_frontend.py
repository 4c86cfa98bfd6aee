"""Host-side front-end of ``Vocoder.oracle`` (voicefixer/vocoder/base.py:58-77).

The reference computes the "ground-truth" mel on the CPU with librosa/numpy before handing it
to the generator: |librosa.stft(wav/max|wav|, n_fft=2048, hop=441)| -> librosa.filters.mel(
sr=44100, n_fft=2048, n_mels=128, fmin=0, fmax=22050, htk=True) [slaney-normalised] ->
normalize(amp_to_db(.) - 20) -> pre() (model/util.py:39-66,83-94,115-128).  librosa is not
installed offline, so its two functions are restated here from their published definitions
(librosa 0.10.1, the version pinned in the reference Dockerfile: centre padding mode
"constant"); this front-end is a file-I/O-edge helper, not part of the timed path, and its
parity against real librosa is UNPINNED (no librosa to compare with).
"""
import numpy as np
import torch

N_FFT, HOP, N_MELS, SR = 2048, 441, 128, 44100


def stft_mag(wav):
    """|librosa.stft(wav, n_fft=2048, hop_length=441, win_length=None, center=True,
    pad_mode='constant')| -> (1025, T)."""
    x = np.pad(np.asarray(wav, np.float32), (N_FFT // 2, N_FFT // 2), mode="constant")
    n = np.arange(N_FFT)
    win = (0.5 - 0.5 * np.cos(2.0 * np.pi * n / N_FFT)).astype(np.float32)  # scipy hann, sym=False
    T = 1 + (len(x) - N_FFT) // HOP
    idx = np.arange(N_FFT)[None, :] + HOP * np.arange(T)[:, None]
    frames = x[idx] * win[None, :]
    return np.abs(np.fft.rfft(frames, axis=1)).T.astype(np.float32)


def mel_basis():
    """librosa.filters.mel(sr=44100, n_fft=2048, n_mels=128, fmin=0, fmax=22050, htk=True,
    norm='slaney') -> (128, 1025)."""
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


def wav_to_cond(wav):
    """float waveform (N,) -> normalised, tail-padded condition tensor (1, 128, T + T%2 + 4)."""
    wav = np.asarray(wav, np.float32)
    wav = wav / np.max(np.abs(wav))
    mel = np.dot(mel_basis(), stft_mag(wav))                       # linear_to_mel
    min_level = np.exp(-100 / 20 * np.log(10))
    S = 20 * np.log10(np.maximum(min_level, np.abs(mel))) - 20      # amp_to_db(.) - 20
    c = np.clip((2 * 4.0) * ((S - (-115)) / 115) - 4.0, -4.0, 4.0)  # normalize
    cond = torch.FloatTensor(c.T.copy()).unsqueeze(0).transpose(1, 2)  # pre(): (1, 128, T)
    pad_tail = cond.size(-1) % 2 + 4
    return torch.cat([cond, torch.zeros([1, N_MELS, pad_tail]) + -4.0], dim=-1)
