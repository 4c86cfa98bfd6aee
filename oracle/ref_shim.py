"""TEST INFRASTRUCTURE ONLY -- never imported by the product path.

Imports the *reference's own* Python modules from /root/reference (read-only) so the
oracle in ``oracle/oracle.py`` can be validated against them and golden vectors can be
generated (``oracle/make_golden.py``).  This only works inside the build container;
/root/reference does not exist on the GPU box and nothing in ``tests -m gpu``,
``smoke()`` or ``bench.py`` may call into this file at run time.

What is shimmed:
  * ``librosa`` (+ ``.display``, ``.filters``), ``soundfile``, ``torchlibrosa`` are
    not installed: stub modules are registered in ``sys.modules``.
  * ``torchlibrosa.stft.STFT`` is re-stated from its published semantics
    (torchlibrosa 0.0.7..0.1.0, ``stft.py``): reflect-pad n_fft//2, two Conv1d with
    windowed DFT basis (periodic hann, ``fftbins=True``), output (B,1,T,n_fft//2+1).
    The reference calls it at voicefixer/tools/modules/fDomainHelper.py:23-31,78,82.
  * import-time checkpoint download (voicefixer/vocoder/__init__.py:17-23,
    voicefixer/restorer/__init__.py:27-33) is defused by pointing HOME at a scratch
    directory that already holds seeded random-weight checkpoints.
"""
import os
import sys
import types
import math

import numpy as np
import torch

REFERENCE_ROOT = "/root/reference"


def reference_available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "voicefixer"))


# --------------------------------------------------------------------------------------
# torchlibrosa.stft restatement (published algorithm; package is not vendored)
# --------------------------------------------------------------------------------------
class _STFT(torch.nn.Module):
    def __init__(self, n_fft=2048, hop_length=None, win_length=None, window="hann",
                 center=True, pad_mode="reflect", freeze_parameters=True):
        super().__init__()
        assert pad_mode in ("constant", "reflect")
        self.n_fft = n_fft
        self.hop_length = hop_length if hop_length is not None else n_fft // 4
        self.win_length = win_length if win_length is not None else n_fft
        self.center = center
        self.pad_mode = pad_mode
        assert window == "hann" and self.win_length == n_fft
        n = np.arange(n_fft)
        # librosa.filters.get_window('hann', n_fft, fftbins=True) == periodic hann
        win = 0.5 - 0.5 * np.cos(2.0 * np.pi * n / n_fft)
        out_channels = n_fft // 2 + 1
        # DFT matrix W[n, k] = exp(-2*pi*i*n*k/N)
        x, y = np.meshgrid(np.arange(n_fft), np.arange(n_fft))
        omega = np.exp(-2 * np.pi * 1j / n_fft)
        W = np.power(omega, x * y)
        self.conv_real = torch.nn.Conv1d(1, out_channels, n_fft, stride=self.hop_length,
                                         padding=0, dilation=1, groups=1, bias=False)
        self.conv_imag = torch.nn.Conv1d(1, out_channels, n_fft, stride=self.hop_length,
                                         padding=0, dilation=1, groups=1, bias=False)
        self.conv_real.weight.data = torch.Tensor(
            np.real(W[:, 0:out_channels] * win[:, None]).T)[:, None, :]
        self.conv_imag.weight.data = torch.Tensor(
            np.imag(W[:, 0:out_channels] * win[:, None]).T)[:, None, :]
        if freeze_parameters:
            for p in self.parameters():
                p.requires_grad = False

    def forward(self, input):
        x = input[:, None, :]
        if self.center:
            x = torch.nn.functional.pad(x, pad=(self.n_fft // 2, self.n_fft // 2),
                                        mode=self.pad_mode)
        real = self.conv_real(x)
        imag = self.conv_imag(x)
        real = real[:, None, :, :].transpose(2, 3)
        imag = imag[:, None, :, :].transpose(2, 3)
        return real, imag


class _ISTFT(torch.nn.Module):
    """Never executed on the inference path (fDomainHelper.py:33-41 only constructs it)."""

    def __init__(self, *a, **k):
        super().__init__()

    def forward(self, *a, **k):  # pragma: no cover
        raise NotImplementedError("ISTFT is not on the mode-0 path")


def _magphase(real, imag):  # pragma: no cover - unused on the path
    mag = (real ** 2 + imag ** 2) ** 0.5
    return mag, real / torch.clamp(mag, 1e-10, np.inf), imag / torch.clamp(mag, 1e-10, np.inf)


def install_stubs():
    """Register stub modules for the reference's missing third-party imports."""
    if "torchlibrosa" in sys.modules and getattr(sys.modules["torchlibrosa"], "_vfx_stub", False):
        return

    def mod(name):
        m = types.ModuleType(name)
        m._vfx_stub = True
        sys.modules[name] = m
        return m

    librosa = mod("librosa")
    display = mod("librosa.display")
    filters = mod("librosa.filters")
    librosa.display = display
    librosa.filters = filters

    def _unavailable(*a, **k):
        raise RuntimeError("librosa is stubbed: this call is outside the parity path")

    # The three librosa transforms on the path are bound to oracle/librosa_like (restated from librosa's published
    # algorithms, pinned against scipy.signal and reference-held constants by tests/test_librosa_like.py), with
    # librosa's own signatures and defaults.  With them the REFERENCE'S OWN numpy code runs unmodified in this
    # container: VoiceFixer.remove_higher_frequency (base.py:87-104, mode 1) and the front half of Vocoder.oracle
    # (vocoder/base.py:58-77 + model/util.py:39-66,83-94,115-128) -- oracle/make_golden.py turns their outputs into
    # fixtures.  Anything else librosa offers stays unavailable and raises.
    from oracle import librosa_like as ll

    def _stft(y, n_fft=2048, hop_length=None, win_length=None, window="hann", center=True, dtype=None,
              pad_mode="constant"):
        assert n_fft == ll.N_FFT and win_length in (None, n_fft) and window == "hann" and center
        assert pad_mode == "constant"   # librosa >= 0.10 default (the reference's Dockerfile pins 0.10.1)
        return ll.stft(y, hop_length if hop_length is not None else n_fft // 4)

    def _istft(stft_matrix, hop_length=None, win_length=None, n_fft=None, window="hann", center=True, dtype=None,
               length=None):
        assert stft_matrix.shape[0] == ll.N_FFT // 2 + 1 and window == "hann" and center and length is None
        return ll.istft(stft_matrix, hop_length if hop_length is not None else ll.N_FFT // 4)

    def _mel(sr, n_fft, n_mels=128, fmin=0.0, fmax=None, htk=False, norm="slaney", dtype=np.float32):
        assert (sr, n_fft, n_mels, fmin, htk, norm) == (ll.SR, ll.N_FFT, ll.N_MELS, 0, True, "slaney")
        assert fmax in (None, ll.SR // 2, ll.SR / 2.0)
        return ll.mel_basis()

    def _load(path, sr=22050, mono=True, offset=0.0, duration=None, **kw):
        """librosa.load for files ALREADY at ``sr`` (no resampler here): float32 in [-1, 1], (n,) or (channels, n)."""
        assert offset == 0.0 and duration is None
        low = str(path).lower()
        if low.endswith(".flac"):
            from voicefixer_amd import flac
            rate, pcm, bps = flac.read(path)
            x = pcm.astype(np.float32).T / float(1 << (bps - 1))
        else:
            from scipy.io import wavfile
            rate, data = wavfile.read(path)
            assert data.dtype == np.int16
            x = (data.astype(np.float32) / 32768.0).T
        assert rate == sr, "the shimmed librosa.load does not resample"
        if x.ndim == 2 and (mono or x.shape[0] == 1):
            x = x.mean(axis=0) if x.shape[0] > 1 else x[0]
        return np.ascontiguousarray(x), rate

    librosa.load = _load
    librosa.stft = _stft
    librosa.istft = _istft
    filters.mel = _mel
    display.specshow = _unavailable

    sf = mod("soundfile")
    sf.written = []                      # (fname, frames, samplerate) of every soundfile.write the reference issued

    def _sf_write(fname, frames, samplerate=44100, **kw):
        sf.written.append((fname, np.array(frames, copy=True), samplerate))

    sf.write = _sf_write
    sf.read = _unavailable
    sf.available_formats = lambda: {"WAV": "WAV (Microsoft)", "FLAC": "FLAC (Free Lossless Audio Codec)"}

    tl = mod("torchlibrosa")
    tls = mod("torchlibrosa.stft")
    tls.STFT = _STFT
    tls.ISTFT = _ISTFT
    tls.magphase = _magphase
    tl.stft = tls
    tl.__version__ = "0.0.7"

    for name in ("progressbar", "git"):
        if name not in sys.modules:
            mod(name)


def _ckpt_paths(home):
    voc = os.path.join(home, ".cache/voicefixer/synthesis_module/44100/model.ckpt-1490000_trimed.pt")
    ana = os.path.join(home, ".cache/voicefixer/analysis_module/checkpoints/vf.ckpt")
    return voc, ana


def prepare_home(home, vocoder_sd=None, restorer_sd=None, legacy_keys=False):
    """Write checkpoints (or placeholders) under a scratch HOME; returns the two paths."""
    voc, ana = _ckpt_paths(home)
    os.makedirs(os.path.dirname(voc), exist_ok=True)
    os.makedirs(os.path.dirname(ana), exist_ok=True)
    if vocoder_sd is None:
        if not os.path.exists(voc):
            open(voc, "wb").close()
    else:
        torch.save({"generator": vocoder_sd}, voc)
    if restorer_sd is None:
        if not os.path.exists(ana):
            open(ana, "wb").close()
    else:
        torch.save(restorer_sd, ana)
    return voc, ana


_IMPORTED = {}


def import_reference(home):
    """Import the reference package with HOME redirected to ``home``.

    Returns the ``voicefixer`` package module.  Must be called with checkpoints (or
    placeholders) already present in ``home`` -- see ``prepare_home``.
    """
    if not reference_available():
        raise RuntimeError("/root/reference is not present (GPU box?)")
    install_stubs()
    os.environ["HOME"] = home
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    sys.dont_write_bytecode = True
    import matplotlib
    matplotlib.use("Agg")
    if "voicefixer" in _IMPORTED:
        # Config.ckpt was frozen at first import; re-point it at the new home.
        from voicefixer.vocoder.config import Config
        Config.ckpt = _ckpt_paths(home)[0]
        return _IMPORTED["voicefixer"]
    import importlib
    # only import sub-modules lazily: voicefixer/__init__ pulls everything in
    pkg = importlib.import_module("voicefixer")
    _IMPORTED["voicefixer"] = pkg
    return pkg


def build_reference_models(home, vocoder_sd, restorer_sd):
    """Instantiate the reference's VoiceFixer wrapper (base.py:10-30) with given weights.

    ``restorer_sd`` uses the key space of restorer.model.VoiceFixer.state_dict() restricted
    to ``generator.*`` (denoiser + unet); the vocoder weights go through the vocoder ckpt.
    """
    prepare_home(home, vocoder_sd, restorer_sd)
    pkg = import_reference(home)
    from voicefixer.vocoder.config import Config
    Config.ckpt = _ckpt_paths(home)[0]
    # VoiceFixer.__init__ (base.py:15-18) resolves ~ at construction time.
    vf = pkg.VoiceFixer()
    vf.eval()
    return vf
