"""Checkpoint handling for the MI355X VoiceFixer path.

Mirrors the weight handling of the reference:
  * ``Vocoder._load_pretrain`` voicefixer/vocoder/base.py:24-32 (``{"generator": sd}``),
    ``load_try`` voicefixer/vocoder/model/util.py:97-107
  * ``VoiceFixer.__init__`` voicefixer/base.py:23-30 (flat state dict, key filter,
    ``strict=False``)
  * weight-norm is *never removed* in the reference (generator.py:153-160 swallows the
    ValueError), so ``w = g * v / ||v||`` is recomputed every forward.  Here it is folded
    once at load (numerically identical up to 1 ulp).

This module also owns the *manifests* (key -> shape) of both state dicts so that seeded
synthetic checkpoints can be generated without the reference being importable (GPU box).
tests/test_manifest.py checks the manifests against the reference's own state_dict()
key/shape sets when /root/reference is present.

Everything here is host-side preparation (runs once at load, on CPU tensors).
"""
from collections import OrderedDict
import math

import torch

# --------------------------------------------------------------------------------------
# architecture constants (voicefixer/vocoder/config.py:15-20, restorer/model_kqq_bn.py)
# --------------------------------------------------------------------------------------
N_MELS = 128
COND_CHANNELS = 512
VOC_CHANNELS = 1024
UPSAMPLE_SCALES = (7, 7, 3, 3)
RESSTACK_DEPTH = 8
UNET_ENC = ((2, 32), (32, 64), (64, 128), (128, 256), (256, 384), (384, 384))
UNET_DEC = ((384, 384), (384, 384), (384, 256), (256, 128), (128, 64), (64, 32))


def _wn_keys(prefix, legacy):
    if legacy:
        return prefix + ".weight_g", prefix + ".weight_v"
    return (prefix + ".parametrizations.weight.original0",
            prefix + ".parametrizations.weight.original1")


def vocoder_manifest(legacy=False):
    """Ordered {key: shape} of vocoder.model.generator.Generator(128).state_dict().

    ``legacy=True`` yields the pre-parametrization key style (``weight_g``/``weight_v``)
    that the published Zenodo checkpoint uses.
    """
    m = OrderedDict()

    def wn_conv(prefix, w_shape, bias_n):
        g, v = _wn_keys(prefix, legacy)
        m[prefix + ".bias"] = (bias_n,)
        m[g] = (w_shape[0], 1, 1)
        m[v] = tuple(w_shape)

    cin = N_MELS
    for i in (0, 2, 4, 6, 8):
        wn_conv("condnet.%d" % i, (COND_CHANNELS, cin, 3), COND_CHANNELS)
        cin = COND_CHANNELS
    wn_conv("generator.1", (VOC_CHANNELS, COND_CHANNELS, 7), VOC_CHANNELS)
    c = VOC_CHANNELS
    for j, s in enumerate(UPSAMPLE_SCALES):
        up = "generator.%d" % (3 + 3 * j)
        rs = "generator.%d" % (4 + 3 * j)
        m[up + ".skip_conv.weight"] = (c // 2, c, 1)
        m[up + ".skip_conv.bias"] = (c // 2,)
        wn_conv(up + ".layer", (c, c // 2, 2 * s), c // 2)  # ConvTranspose1d: (Cin, Cout, k)
        c //= 2
        for i in range(RESSTACK_DEPTH):
            wn_conv("%s.layers.%d.1" % (rs, i), (c, c, 3), c)
            wn_conv("%s.layers.%d.3" % (rs, i), (c, c, 3), c)
    wn_conv("generator.16", (1, c, 7), 1)
    return m


def _bn(m, prefix, n):
    m[prefix + ".weight"] = (n,)
    m[prefix + ".bias"] = (n,)
    m[prefix + ".running_mean"] = (n,)
    m[prefix + ".running_var"] = (n,)
    m[prefix + ".num_batches_tracked"] = ()


def _conv_block(m, prefix, cin, cout):
    m[prefix + ".conv1.weight"] = (cout, cin, 3, 3)
    _bn(m, prefix + ".bn1", cin)
    m[prefix + ".conv2.weight"] = (cout, cout, 3, 3)
    _bn(m, prefix + ".bn2", cout)
    if cin != cout:
        m[prefix + ".shortcut.weight"] = (cout, cin, 1, 1)
        m[prefix + ".shortcut.bias"] = (cout,)


def restorer_manifest():
    """Ordered {key: shape} of restorer.model.Generator(128,1025,2).state_dict()
    (denoiser + unet); in vf.ckpt these keys carry the prefix ``generator.``."""
    m = OrderedDict()
    H = 2 * N_MELS
    _bn(m, "denoiser.0", 1)
    m["denoiser.1.weight"] = (2 * N_MELS, N_MELS)
    m["denoiser.1.bias"] = (2 * N_MELS,)
    _bn(m, "denoiser.3", 1)
    m["denoiser.4.weight"] = (4 * N_MELS, 2 * N_MELS)
    m["denoiser.4.bias"] = (4 * N_MELS,)
    for idx in (7, 8):
        _bn(m, "denoiser.%d.bn" % idx, 1)
        for layer in (0, 1):
            for suf in ("", "_reverse"):
                m["denoiser.%d.gru.weight_ih_l%d%s" % (idx, layer, suf)] = (3 * H, 4 * N_MELS)
                m["denoiser.%d.gru.weight_hh_l%d%s" % (idx, layer, suf)] = (3 * H, H)
                m["denoiser.%d.gru.bias_ih_l%d%s" % (idx, layer, suf)] = (3 * H,)
                m["denoiser.%d.gru.bias_hh_l%d%s" % (idx, layer, suf)] = (3 * H,)
    _bn(m, "denoiser.9", 1)
    m["denoiser.11.weight"] = (4 * N_MELS, 4 * N_MELS)
    m["denoiser.11.bias"] = (4 * N_MELS,)
    _bn(m, "denoiser.13", 1)
    m["denoiser.15.weight"] = (N_MELS, 4 * N_MELS)
    m["denoiser.15.bias"] = (N_MELS,)
    for b, (cin, cout) in enumerate(UNET_ENC, start=1):
        p = "unet.encoder_block%d" % b
        _conv_block(m, p + ".conv_block1", cin, cout)
        for k in (2, 3, 4):
            _conv_block(m, p + ".conv_block%d" % k, cout, cout)
    _conv_block(m, "unet.conv_block7", 384, 384)
    for b, (cin, cout) in enumerate(UNET_DEC, start=1):
        p = "unet.decoder_block%d" % b
        m[p + ".conv1.weight"] = (cin, cout, 3, 3)  # ConvTranspose2d: (Cin, Cout, kh, kw)
        _bn(m, p + ".bn1", cin)
        _conv_block(m, p + ".conv_block2", 2 * cout, cout)
        for k in (3, 4, 5):
            _conv_block(m, p + ".conv_block%d" % k, cout, cout)
    _conv_block(m, "unet.after_conv_block1", 32, 32)
    m["unet.after_conv2.weight"] = (1, 32, 1, 1)
    m["unet.after_conv2.bias"] = (1,)
    return m


# --------------------------------------------------------------------------------------
# seeded synthetic checkpoints (no network => no Zenodo weights)
# --------------------------------------------------------------------------------------
def _fill(shape, gen, std):
    return torch.randn(shape, generator=gen, dtype=torch.float32) * std


def seeded_vocoder_state(seed=1234, legacy=False, gain_sigma=0.0):
    """Random-weight vocoder state dict with the exact key/shape set of the reference.

    Scales are chosen so activations stay O(1) through the 4 residual stacks and the
    final tanh is not saturated (keeps the parity comparison meaningful).
    ``gain_sigma`` > 0: HEAVY-TAILED weight-norm gains -- every output row's ``g`` is multiplied by
    exp(sigma z - sigma^2), z ~ N(0, 1): log-normal, mean square 1 (the layer's output energy is unchanged on average, so
    the path stays in range), median exp(-sigma^2), rows up to ~exp(3 sigma - sigma^2) -- what trained weight-normed
    checkpoints look like rather than the flat gains of the default (parity tests of round 3)."""
    gen = torch.Generator().manual_seed(seed)
    sd = OrderedDict()
    man = vocoder_manifest(legacy)
    g_suffix = ".weight_g" if legacy else ".parametrizations.weight.original0"
    v_suffix = ".weight_v" if legacy else ".parametrizations.weight.original1"
    for k, shape in man.items():
        if k.endswith(g_suffix):
            sd[k] = None  # filled once v is known
        elif k.endswith(v_suffix):
            base = k[: -len(v_suffix)]
            is_up = base.endswith(".layer")
            if is_up:
                # ConvTranspose1d (Cin, Cout, 2s): each output sees 2 taps x Cin
                fan_in = shape[0] * 2
                gain = 0.8
            else:
                fan_in = shape[1] * shape[2]
                gain = 1.0
                if ".layers." in base and base.endswith(".3"):
                    gain = 0.35  # second conv of a residual branch: keep the sum bounded
                elif ".layers." in base:
                    gain = 1.4
                elif base == "generator.16":
                    gain = 0.15
            v = _fill(shape, gen, gain / math.sqrt(fan_in))
            sd[k] = v
            norm = v.reshape(shape[0], -1).norm(dim=1).reshape(shape[0], 1, 1)
            jitter = 0.9 + 0.2 * torch.rand((shape[0], 1, 1), generator=gen)
            if gain_sigma > 0:
                jitter = jitter * torch.exp(gain_sigma * torch.randn((shape[0], 1, 1), generator=gen) - gain_sigma ** 2)
            sd[base + g_suffix] = norm * jitter
        elif k.endswith(".skip_conv.weight"):
            sd[k] = _fill(shape, gen, 1.0 / math.sqrt(shape[1]))
        else:  # biases
            sd[k] = _fill(shape, gen, 0.05)
    return sd


def seeded_restorer_state(seed=4321, gain_sigma=0.0):
    """Random-weight denoiser+UNet state dict (keys of restorer.model.Generator);
    BN running stats are randomised so eval-BN is non-trivial.
    ``gain_sigma`` > 0: the BatchNorm scales (the per-channel gains of this model) become log-normal with mean square 1,
    exp(sigma z - sigma^2), as in ``seeded_vocoder_state``."""
    gen = torch.Generator().manual_seed(seed)
    sd = OrderedDict()
    man = restorer_manifest()
    bn_prefixes = {k[: -len(".running_mean")] for k in man if k.endswith(".running_mean")}
    for k, shape in man.items():
        parent, leaf = k.rsplit(".", 1)
        if parent in bn_prefixes:
            if leaf == "num_batches_tracked":
                sd[k] = torch.tensor(1000, dtype=torch.int64)
            elif leaf == "running_mean":
                sd[k] = _fill(shape, gen, 0.1)
            elif leaf == "running_var":
                sd[k] = 0.75 + 0.5 * torch.rand(shape, generator=gen)
            elif leaf == "weight":
                sd[k] = 0.8 + 0.4 * torch.rand(shape, generator=gen)
                if gain_sigma > 0 and len(shape) == 1 and shape[0] > 1:   # (not the scalar BatchNorm2d(1) of the denoiser)
                    sd[k] = sd[k] * torch.exp(gain_sigma * torch.randn(shape, generator=gen) - gain_sigma ** 2)
            else:
                sd[k] = _fill(shape, gen, 0.1)
        elif ".gru." in k:
            sd[k] = (torch.rand(shape, generator=gen) * 2 - 1) / math.sqrt(2 * N_MELS)
        elif k.startswith("denoiser") and leaf == "weight":
            sd[k] = _fill(shape, gen, 1.0 / math.sqrt(shape[1]))
        elif leaf == "bias":
            sd[k] = _fill(shape, gen, 0.05)
        elif k.endswith("after_conv2.weight"):
            sd[k] = _fill(shape, gen, 0.3 / math.sqrt(shape[1]))
        elif ".decoder_block" in k and k.endswith(".conv1.weight") and ".conv_block" not in k:
            # ConvTranspose2d 3x3 s2: every output sees <= 4 taps x Cin
            sd[k] = _fill(shape, gen, 1.0 / math.sqrt(shape[0] * 2.25))
        elif leaf == "weight" and len(shape) == 4:
            fan_in = shape[1] * shape[2] * shape[3]
            gain = 0.2 if k.endswith("conv2.weight") else 1.0
            sd[k] = _fill(shape, gen, gain / math.sqrt(fan_in))
        else:
            raise AssertionError("unhandled manifest key " + k)
    return sd


def check_state(sd, manifest, what):
    """Raise if ``sd`` lacks a manifest key or has a wrong shape."""
    for k, shape in manifest.items():
        if k not in sd:
            raise KeyError("%s checkpoint is missing key %s" % (what, k))
        if tuple(sd[k].shape) != tuple(shape):
            raise ValueError("%s key %s has shape %s, expected %s"
                             % (what, k, tuple(sd[k].shape), tuple(shape)))


def normalise_vocoder_keys(sd):
    """Accept both key styles (legacy weight_g/_v and parametrized original0/1)."""
    out = OrderedDict()
    for k, v in sd.items():
        if k.endswith(".weight_g"):
            k = k[:-len(".weight_g")] + ".parametrizations.weight.original0"
        elif k.endswith(".weight_v"):
            k = k[:-len(".weight_v")] + ".parametrizations.weight.original1"
        out[k] = v
    return out


def fold_weight_norm(g, v):
    """w = g * v / ||v||, norm over every dim but 0 (torch weight_norm dim=0;
    for ConvTranspose1d dim 0 is C_in -- SURVEY.md A.6)."""
    norm = v.reshape(v.shape[0], -1).norm(dim=1).reshape([v.shape[0]] + [1] * (v.dim() - 1))
    return v * (g / norm)


def bn_affine(sd, prefix, eps=1e-5):
    """eval-mode BatchNorm as y = x*scale + shift."""
    scale = sd[prefix + ".weight"] / torch.sqrt(sd[prefix + ".running_var"] + eps)
    shift = sd[prefix + ".bias"] - sd[prefix + ".running_mean"] * scale
    return scale.float().contiguous(), shift.float().contiguous()
