"""CPU ORACLE -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A functional (no nn.Module, no librosa/torchlibrosa) CPU restatement of the reference
VoiceFixer mode-0 inference path, written against plain state dicts in the reference's
key space.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this; the product path (``voicefixer_amd``) never does
and fails loudly when its HIP library is missing.

The reference is a pure-Python/PyTorch package (zero native sources), so
the restatement is Python on torch CPU tensors -- the same ATen arithmetic the reference
CPU path executes -- rather than C.  ``dtype=torch.float64`` runs the identical algorithm
in double precision for error budgeting.

Parity pinning status
---------------------
* Pinned against the reference's OWN modules executed in the build container
  (``oracle/ref_shim.py``, ``tests/test_oracle_vs_reference.py``) with seeded weights, and
  against golden vectors generated from them (``tests/golden/*.npz``,
  ``oracle/make_golden.py``).
* The reference's only golden *audio* (test/utterance/target/*.flac, test/test.py:27-35)
  needs the Zenodo checkpoints and a FLAC decoder, neither of which exists offline:
  **real-weight golden parity is unpinned**.  torchlibrosa's STFT (un-vendored dep) is
  restated from its published algorithm (``ref_shim._STFT``).

Every function cites the reference lines it follows (paths relative to /root/reference).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

N_FFT = 2048
HOP = 441
N_MELS = 128
SR = 44100
SEG_LENGTH = 44100 * 30  # voicefixer/base.py:117


# --------------------------------------------------------------------------------------
# A.1 / A.2  STFT magnitude and HTK mel filterbank
# --------------------------------------------------------------------------------------
def mel_filterbank():
    """voicefixer/tools/mel_scale.py:147-238 with n_freqs=1025, f_min=0, f_max=22050,
    n_mels=128, norm=None, mel_scale='htk' -- float32 torch ops in the same order so the
    support set is bit-identical."""
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
    return torch.max(zero, torch.min(down_slopes, up_slopes))  # (1025, 128)


def stft_mag(wav, dtype=torch.float32):
    """fDomainHelper.py:81-110 + torchlibrosa STFT (center, reflect, periodic hann):
    wav (B, N) -> sp (B, T, 1025), T = 1 + N//441, sp = sqrt(clamp(re^2+im^2, 1e-8)).

    FFT formulation; equals the reference's conv-DFT to 7e-7 relative."""
    x = wav.to(dtype)
    xp = F.pad(x[:, None, :], (N_FFT // 2, N_FFT // 2), mode="reflect")[:, 0]
    n = torch.arange(N_FFT, dtype=torch.float64)
    win = (0.5 - 0.5 * torch.cos(2.0 * math.pi * n / N_FFT)).to(dtype)
    frames = xp.unfold(-1, N_FFT, HOP)  # (B, T, 2048)
    X = torch.fft.rfft(frames * win, dim=-1)
    power = X.real ** 2 + X.imag ** 2
    return torch.clamp(power, min=1e-8) ** 0.5


def wav_to_mel(wav, dtype=torch.float32):
    """voicefixer/base.py:78-85 (_pre): (B, N) -> mel (B, 1, T, 128) linear magnitudes."""
    sp = stft_mag(wav, dtype)
    fb = mel_filterbank().to(dtype)
    return torch.matmul(sp, fb)[:, None]  # mel_scale.py:73


# --------------------------------------------------------------------------------------
# A.3  restorer (denoiser + ResUNet)
# --------------------------------------------------------------------------------------
def _bn_scalar(x, sd, p):
    """eval BatchNorm2d(1) (restorer/model.py:69-99 members 0,3,9,13 and BN_GRU.bn)."""
    w, b = sd[p + ".weight"], sd[p + ".bias"]
    m, v = sd[p + ".running_mean"], sd[p + ".running_var"]
    return (x - m) / torch.sqrt(v + 1e-5) * w + b


def _gru_dir(x, w_ih, w_hh, b_ih, b_hh, reverse):
    """One direction of torch.nn.GRU (gate order r,z,n), h0 = 0.  x: (B, T, I)."""
    B, T, _ = x.shape
    H = w_hh.shape[1]
    gi = x @ w_ih.t() + b_ih
    h = x.new_zeros(B, H)
    out = x.new_zeros(B, T, H)
    steps = range(T - 1, -1, -1) if reverse else range(T)
    for t in steps:
        gh = h @ w_hh.t() + b_hh
        i_r, i_z, i_n = gi[:, t].chunk(3, dim=1)
        h_r, h_z, h_n = gh.chunk(3, dim=1)
        r = torch.sigmoid(i_r + h_r)
        z = torch.sigmoid(i_z + h_z)
        n = torch.tanh(i_n + r * h_n)
        h = (1.0 - z) * n + z * h
        out[:, t] = h
    return out


def bn_gru(x, sd, p):
    """restorer/model.py:22-62 BN_GRU: scalar BN then 2-layer bidirectional GRU(512->256).
    x: (B, 1, T, 512) -> (B, 1, T, 512)."""
    x = _bn_scalar(x, sd, p + ".bn")[:, 0]
    for layer in (0, 1):
        outs = []
        for suf, rev in (("", False), ("_reverse", True)):
            outs.append(_gru_dir(
                x, sd["%s.gru.weight_ih_l%d%s" % (p, layer, suf)],
                sd["%s.gru.weight_hh_l%d%s" % (p, layer, suf)],
                sd["%s.gru.bias_ih_l%d%s" % (p, layer, suf)],
                sd["%s.gru.bias_hh_l%d%s" % (p, layer, suf)], rev))
        x = torch.cat(outs, dim=-1)
    return x[:, None]


def denoiser(mel, sd, p="denoiser"):
    """restorer/model.py:69-99 (Dropout is identity in eval): (B,1,T,128) -> mask."""
    x = _bn_scalar(mel, sd, p + ".0")
    x = F.relu(F.linear(x, sd[p + ".1.weight"], sd[p + ".1.bias"]))
    x = _bn_scalar(x, sd, p + ".3")
    x = F.relu(F.linear(x, sd[p + ".4.weight"], sd[p + ".4.bias"]))
    x = bn_gru(x, sd, p + ".7")
    x = bn_gru(x, sd, p + ".8")
    x = F.relu(_bn_scalar(x, sd, p + ".9"))
    x = F.linear(x, sd[p + ".11.weight"], sd[p + ".11.bias"])
    x = F.relu(_bn_scalar(x, sd, p + ".13"))
    x = F.linear(x, sd[p + ".15.weight"], sd[p + ".15.bias"])
    return torch.sigmoid(x)


def _bn2d(x, sd, p):
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"],
                        sd[p + ".weight"], sd[p + ".bias"], training=False, eps=1e-5)


def conv_block_res(x, sd, p):
    """restorer/modules.py:68-76."""
    origin = x
    x = F.conv2d(F.leaky_relu(_bn2d(x, sd, p + ".bn1"), 0.01), sd[p + ".conv1.weight"], padding=1)
    x = F.conv2d(F.leaky_relu(_bn2d(x, sd, p + ".bn2"), 0.01), sd[p + ".conv2.weight"], padding=1)
    if (p + ".shortcut.weight") in sd:
        return F.conv2d(origin, sd[p + ".shortcut.weight"], sd[p + ".shortcut.bias"]) + x
    return origin + x


def encoder_block(x, sd, p):
    """restorer/modules.py:98-104."""
    for k in (1, 2, 3, 4):
        x = conv_block_res(x, sd, "%s.conv_block%d" % (p, k))
    return F.avg_pool2d(x, kernel_size=(2, 2)), x


def decoder_block(x, skip, sd, p):
    """restorer/modules.py:149-157 (prune drops the last time row only, :141-147)."""
    x = F.conv_transpose2d(F.relu(_bn2d(x, sd, p + ".bn1")), sd[p + ".conv1.weight"], stride=2)
    x = x[:, :, 0:-1, :]
    x = torch.cat((x, skip), dim=1)
    for k in (2, 3, 4, 5):
        x = conv_block_res(x, sd, "%s.conv_block%d" % (p, k))
    return x


def unet(x, sd, p="unet"):
    """restorer/model_kqq_bn.py:130-181: (B,2,T,128) -> (B,1,T,128)."""
    T = x.shape[2]
    pad_len = int(np.ceil(T / 64)) * 64 - T
    x = F.pad(x, pad=(0, 0, 0, pad_len))
    x = x[..., 0: x.shape[-1] - 1]
    skips = []
    for b in range(1, 7):
        x, s = encoder_block(x, sd, "%s.encoder_block%d" % (p, b))
        skips.append(s)
    x = conv_block_res(x, sd, p + ".conv_block7")
    for b in range(1, 7):
        x = decoder_block(x, skips[6 - b], sd, "%s.decoder_block%d" % (p, b))
    x = conv_block_res(x, sd, p + ".after_conv_block1")
    x = F.conv2d(x, sd[p + ".after_conv2.weight"], sd[p + ".after_conv2.bias"])
    x = F.pad(x, pad=(0, 1))
    return x[:, :, 0:T, :]


def to_log(x):
    """tools/pytorch_util.py:18-22 (the assert is a host sync; mel >= 0 by construction)."""
    return torch.log10(torch.clip(x, min=1e-8))


def from_log(x):
    """tools/pytorch_util.py:25-27."""
    return 10 ** torch.clip(x, max=5)


def restorer_forward(mel, sd, return_all=False):
    """restorer/model.py:103-120 Generator.forward: mel (B,1,T,128) -> log10-mel (B,1,T,128).
    ``sp`` is unused by the reference."""
    mask = denoiser(mel, sd)
    clean = mask * mel
    x = to_log(clean)
    unet_in = torch.cat([to_log(mel), x], dim=1)
    unet_out = unet(unet_in, sd)
    out = unet_out + x
    if return_all:
        return {"mel": out, "mask": mask, "clean": clean, "unet_out": unet_out, "x": x}
    return out


# --------------------------------------------------------------------------------------
# A.4 / A.5  vocoder
# --------------------------------------------------------------------------------------
def mel_weight(dtype=torch.float32):
    """vocoder/config.py:294-316: a*exp(b*k), k = linspace(1,128,128) (float32)."""
    k = torch.linspace(1, N_MELS, steps=N_MELS)
    return (18.8927416350036 * torch.exp(0.0269863588184314 * k)).to(dtype)


def mel_to_cond(mel):
    """vocoder/base.py:51-54 + model/util.py:8-36,69-80:
    mel (B,1,T,128) linear -> normalised cond (B,128,T') with T' = T + T%2 + 4."""
    m = mel / mel_weight(mel.dtype)[None, None, None, :]
    min_level = torch.exp(torch.tensor(-100.0) / 20 * torch.log(torch.tensor(10.0))).to(mel.dtype)
    S = 20 * torch.log10(torch.maximum(min_level, torch.abs(m))) - 20.0
    c = torch.clip((2 * 4.0) * ((S - (-115)) / 115) - 4.0, -4.0, 4.0)
    c = c[:, 0].transpose(1, 2)
    pad_tail = c.shape[-1] % 2 + 4
    tail = torch.zeros(c.shape[0], N_MELS, pad_tail, dtype=c.dtype) + -4.0
    return torch.cat([c, tail], dim=-1)


def _wn(sd, p):
    """effective weight of a weight-normed conv: g*v/||v|| over all dims but 0."""
    g = sd[p + ".parametrizations.weight.original0"]
    v = sd[p + ".parametrizations.weight.original1"]
    norm = v.reshape(v.shape[0], -1).norm(dim=1).reshape([v.shape[0]] + [1] * (v.dim() - 1))
    return v * (g / norm)


def _canon(sd):
    if any(k.endswith(".weight_g") for k in sd):
        out = {}
        for k, v in sd.items():
            if k.endswith(".weight_g"):
                k = k[:-9] + ".parametrizations.weight.original0"
            elif k.endswith(".weight_v"):
                k = k[:-9] + ".parametrizations.weight.original1"
            out[k] = v
        return out
    return sd


def vocoder_generator(cond, sd, stages=None):
    """vocoder/model/generator.py:127-145 with the layout of :33-54,72-100 and
    model/modules.py:501-528 (UpsampleNet, dead skip_conv omitted: no_skip=True) and
    :592-609 (ResStack).  cond (B,128,T') -> wav (B,1,441*T').

    ``stages``: optional dict that receives intermediate activations by name."""
    sd = _canon(sd)
    x = cond
    for i in (0, 2, 4, 6, 8):
        x = F.elu(F.conv1d(x, _wn(sd, "condnet.%d" % i), sd["condnet.%d.bias" % i], padding=1))
    if stages is not None:
        stages["condnet"] = x
    x = F.pad(x, (3, 3), mode="reflect")
    x = F.leaky_relu(F.conv1d(x, _wn(sd, "generator.1"), sd["generator.1.bias"]), 0.2)
    if stages is not None:
        stages["pre"] = x
    for j, s in enumerate((7, 7, 3, 3)):
        up = "generator.%d" % (3 + 3 * j)
        rs = "generator.%d" % (4 + 3 * j)
        x = x + torch.sin(x)
        x = F.conv_transpose1d(x, _wn(sd, up + ".layer"), sd[up + ".layer.bias"], stride=s,
                               padding=s // 2 + s % 2, output_padding=s % 2)
        if stages is not None:
            stages["up%d" % (j + 1)] = x
        for i in range(8):
            d = 3 ** i
            y = F.conv1d(F.leaky_relu(x, 0.01), _wn(sd, "%s.layers.%d.1" % (rs, i)),
                         sd["%s.layers.%d.1.bias" % (rs, i)], dilation=d, padding=d)
            y = F.conv1d(F.leaky_relu(y, 0.01), _wn(sd, "%s.layers.%d.3" % (rs, i)),
                         sd["%s.layers.%d.3.bias" % (rs, i)], padding=1)
            x = x + y
        x = F.leaky_relu(x, 0.2)
        if stages is not None:
            stages["res%d" % (j + 1)] = x
    x = F.pad(x, (3, 3), mode="reflect")
    x = torch.tanh(F.conv1d(x, _wn(sd, "generator.16"), sd["generator.16.bias"]))
    return x


def vocoder_forward(mel, sd):
    """vocoder/base.py:42-56 Vocoder.forward: mel (B,1,T,128) linear -> (B,1,441*T')."""
    assert mel.shape[-1] == 128
    return vocoder_generator(mel_to_cond(mel), sd)


# --------------------------------------------------------------------------------------
# A.7  driver
# --------------------------------------------------------------------------------------
def trim_center(est, ref_len):
    """voicefixer/base.py:63-76 (_trim_center), est (..., L) -> (..., ref_len)."""
    L = est.shape[-1]
    diff = abs(L - ref_len)
    if L == ref_len:
        return est
    if L > ref_len:
        h = int(diff // 2)
        est = est[..., h:-h] if h > 0 else est[..., h:]
        return est[..., :ref_len]
    return est  # reference trims the *ref* in this branch; never taken


def restore_inmem(wav, voc_sd, res_sd, dtype=torch.float32, vocoder_func=None):
    """voicefixer/base.py:107-139 restore_inmem, mode 0: wav np/tensor (N,) -> (1, N)."""
    wav = torch.as_tensor(np.asarray(wav), dtype=dtype)
    cast = (lambda d: {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in d.items()})
    voc_sd, res_sd = cast(_canon(voc_sd)), cast(res_sd)
    res = []
    break_point = SEG_LENGTH
    while break_point < wav.shape[0] + SEG_LENGTH:
        segment = wav[break_point - SEG_LENGTH: break_point]
        mel = wav_to_mel(segment[None], dtype)
        denoised = from_log(restorer_forward(mel, res_sd))
        out = vocoder_forward(denoised, voc_sd) if vocoder_func is None else vocoder_func(denoised)
        peak = torch.max(torch.abs(out))
        if peak > 1.0:
            out = out / peak
        out = trim_center(out, segment.shape[0])
        res.append(out)
        break_point += SEG_LENGTH
    return torch.cat(res, -1).squeeze(0).numpy()


def remove_higher_frequency(wav, ratio=0.95, stft_fn=None, istft_fn=None):
    """voicefixer/base.py:87-104.  ``librosa.stft(wav)`` / ``librosa.istft(stft)`` (0.10.x defaults: n_fft 2048,
    hop 512, periodic hann, center=True, pad_mode="constant") come from oracle/librosa_like.py, the numpy
    restatement that tests/test_librosa_like.py pins against scipy.signal.stft/istft; ``stft_fn``/``istft_fn``
    let that test substitute the scipy transforms for the whole function.  Everything between the two transforms
    is the reference's own numpy code, line for line.  Returns (filtered wav of length 512*(N//512), cut-off bin)."""
    from . import librosa_like
    EPS = 1e-8
    wav = np.asarray(wav, np.float32)
    stft = (stft_fn or (lambda y: librosa_like.stft(y, 512)))(wav)  # (1025, T) complex64
    real, img = np.real(stft), np.imag(stft)
    mag = (real ** 2 + img ** 2) ** 0.5
    cos, sin = real / (mag + EPS), img / (mag + EPS)
    spec = np.abs(stft)
    feature = spec.copy()
    feature = np.log10(feature + EPS)
    feature[feature < 0] = 0
    energy_level = np.sum(feature, axis=1)
    threshold = np.sum(energy_level) * ratio
    curent_level, i = energy_level[0], 0
    # (the reference's loop condition `i < energy_level.shape[0]` would index energy_level[1025]; it cannot be
    # reached for ratio < 1 because the running level equals the total at i = 1024)
    while i < energy_level.shape[0] - 1 and curent_level < threshold:
        curent_level += energy_level[i + 1, ...]
        i += 1
    spec[i:, ...] = np.zeros_like(spec[i:, ...])
    stft2 = spec * cos + 1j * spec * sin
    y = (istft_fn or (lambda S: librosa_like.istft(S, 512)))(stft2)
    return np.asarray(y, np.float32), i


def to_int16(frames):
    """tools/wav.py:27-34 save_wave quantisation: *2^15 if max<=1, truncating astype."""
    frames = np.array(frames, copy=True)
    if np.max(frames) <= 1:
        frames = frames * 2 ** 15
    return frames.astype(np.short)
