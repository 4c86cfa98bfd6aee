"""Launch plans of the two models of the path, expressed as sequences of libvfx_hip calls.

``VocoderEngine``  = vocoder/model/generator.py:127-145 (+ modules.py:501-528, 592-609)
``RestorerEngine`` = restorer/model.py:103-120 (denoiser :69-99, BN_GRU :22-62) +
                     restorer/model_kqq_bn.py:130-181 (ResUNet) + restorer/modules.py
``Pipeline``       = voicefixer/base.py:78-85,123-135 for a batch of equal-length segments.

Host work done ONCE at construction (CPU torch, "plumbing"): weight-norm folding, eval
BatchNorm folding (into the neighbouring conv/linear weights where algebraically exact,
otherwise kept as a fused pre-activation), packing to the [slab][CinPad][Cout] layout, H2D.
At run time every FLOP is executed by a libvfx_hip kernel; torch only owns the buffers.

Data layouts in HBM (all float32):
  * 1-D activations: (B, C, Lp) channel-major, Lp = L rounded up to 4;
  * UNet maps: (B, C, H*P) pitch maps, P = 128 >> level (W = P-1 valid columns);
  * mel / logmel / denoised: (B, T, 128) frame-major (the reference's (B,1,T,128));
  * GRU x-projections: (B, T, 1536) frame-major.
"""
import math

import torch

from . import ops, packing, weights
from ._lib import (PRE_NONE, PRE_LRELU, PRE_AFFINE_LRELU, POST_NONE, POST_LRELU, POST_ELU, POST_TANH,
                   POST_SIGMOID, POST_LRELU_SNAKE, PAD_ZERO, PAD_REFLECT, VfxError)


class GruHandoffMissed(VfxError):
    """The two-CU GRU's bounded spin expired (a partner workgroup was not resident in time)."""


def _up4(n):
    return (n + 3) // 4 * 4


# Guard bands (readable slack around every row of a conv INPUT, see include/vfx_hip.h): with a
# guard >= largest tap offset + one tile (256) + 8 every tile of a launch runs the branch-free
# interior kernel; only reflect-padded and channel-tail launches still need the general one.
G_TILE = 256 + 8
G_DIL = 3 ** 7 + G_TILE  # largest ResStack dilation


def _rows(B, Cn, L, guard, dev, rows=None):
    """Guarded (B, C, L) activation buffer; ``rows`` (device int32 (B,)) tags it with the valid length of every batch
    item (ragged batches): a kernel that READS the buffer pads at each row's own end and skips the tiles past it."""
    v = ops.guarded(B, Cn, L, guard, dev)
    if rows is not None:
        ops.with_rows(v, rows)
    return v


class RaggedRows:
    """Per-utterance lengths of a ragged batch at every stage of the path, as device int32 vectors (one H2D copy).
    Utterance b has n_b samples -> T_b = 1 + n_b // 441 frames; the ResUNet sees Tp_b = T_b rounded up to 64 rows
    (restorer/model_kqq_bn.py:150-153 pads to a multiple of 2^6), level k of it (Tp_b >> k) rows of pitch 128 >> k;
    the vocoder sees Tc_b = T_b + T_b % 2 + 4 frames (vocoder/base.py:40-47) and 441 * Tc_b samples."""

    VOC_MULTS = (1, 7, 49, 147, 441)

    def __init__(self, lengths, device):
        n = torch.as_tensor(list(lengths), dtype=torch.int64)
        T = 1 + n // 441
        Tp = (T + 63) // 64 * 64
        Tc = T + T % 2 + 4
        table = [n, T] + [(Tp >> k) << (7 - k) for k in range(7)] + [Tc * m for m in self.VOC_MULTS]
        dev = torch.stack(table).to(torch.int32).to(device)
        self.n, self.T = dev[0], dev[1]
        self.unet = [dev[2 + k] for k in range(7)]          # valid elements of a level-k pitch map, per row
        self.voc = {int(m): dev[9 + i] for i, m in enumerate(self.VOC_MULTS)}   # upsampling factor -> valid length
        self.T_max, self.n_max = int(T.max()), int(n.max())
        self.B = len(n)


def _dev(t, device):
    return t.contiguous().float().to(device)


def _wpair(w_packed, device):
    """(w_packed, w_direct) on the device: the [slab][CinPad][Cout] layout of conv_taps_kernel and the
    [slab][CinPad/8][Cout][8] A-operand layout of convw_kernel / vfx_resblock_f32 (packing.pack_direct)."""
    return _dev(w_packed, device), _dev(packing.pack_direct(w_packed), device)


from ._dev import dev_env as _dev_env
# development switches (honoured under VFX_DEV=1 only, _dev.py): VFX_FUSE=0 runs every ResStack layer as two launches, VFX_CONVW=0 (read by the library)
# keeps every launch on the first-generation kernel
_FUSE = _dev_env("VFX_FUSE", "1") != "0"
# The C = 64 layers with dilation >= 81 run as TWO Winograd F(4,3) launches (convwg4_kernel), in place, instead of the fused layer whose dilated
# half is a direct sum there (a block of 4 d positions does not fit its tile): 18 GB instead of 10 GB through HBM per layer, but half the
# products in the dilated half -- step 219.2 -> 217.6 ms alternating on one box (VFX_UNFUSE_WIDE=0 restores the fused form).
_UNFUSE_WIDE = _dev_env("VFX_UNFUSE_WIDE", "1") != "0"
FUSE_MAX_C = 128  # ResStack stages with at most this many channels CAN run one fused launch per layer
# ResStack stages with at least this many channels run their two k = 3 convolutions per layer as two Winograd F(4,3)
# launches (convwg4_kernel: half the fp32 MFMAs of the direct sum; the dilation-1 one moves its quads as 16-byte
# vectors).  Measured per step at batch 32, one box, round 3: C = 128 on F(4,3) for both convolutions 235.3 ms, with the
# round-2 F(2,3) kernel for the first 238.5 ms; C = 64 as two F(4,3) launches 235.2 ms against 235.3 ms fused (its
# second launch reads the intermediate AND the residual from HBM) -- so C = 64 keeps the fused layer, whose dilation-1
# half then moved to F(4,3) on the LDS tile as well (229.0 -> 223.5 ms; F(2,3) is its fallback for unaligned rows).
# VFX_WINO_MIN_C=64 / 0: development switch.
WINO_MIN_C = int(_dev_env("VFX_WINO_MIN_C", "128"))
WINO2D = _dev_env("VFX_WINO2D", "1") != "0"       # the 3x3 convolutions of the ResUNet as Winograd F(4,3) (development switch)


# Run-time arithmetic switch (voicefixer_amd/selfcheck.py): False = no launch is offered its Winograd-transformed weights, so
# every k = 3 / 3x3 convolution runs as the direct sum (the fused C = 64 layer: both halves direct).  The weights stay packed.
_ARITH = {"winograd": True}


def set_winograd(on):
    """True (default): k = 3 / 3x3 convolutions as Winograd F(4,3) where the kernels take them; False: direct sums everywhere.
    Same operands, same fp32 accumulation -- only the order of the sums (and the products formed) differs."""
    _ARITH["winograd"] = bool(on)


def _wg(w):
    return w if _ARITH["winograd"] else None


class VocoderEngine:
    """TFGAN-style 44.1 kHz generator: cond (B,128,T') -> wav (B,1,441*T')."""

    def __init__(self, state, device="cuda", math="f32"):
        sd = weights.normalise_vocoder_keys(state)
        weights.check_state(sd, weights.vocoder_manifest(), "vocoder")
        self.device = device
        self.set_math(math)

        def wn(prefix):
            return weights.fold_weight_norm(sd[prefix + ".parametrizations.weight.original0"].float(),
                                            sd[prefix + ".parametrizations.weight.original1"].float())

        self.condnet = []
        for i in (0, 2, 4, 6, 8):
            p = "condnet.%d" % i
            wc = packing.pack_conv1d(wn(p))
            wcg = (_dev(packing.pack_wino4(wc), device)
                   if WINO_MIN_C > 0 and wc.shape[1] % 32 == 0 and wc.shape[2] % 64 == 0 else None)
            self.condnet.append(_wpair(wc, device) + (_dev(sd[p + ".bias"], device), wcg))
        self.pre = (_dev(packing.pack_conv1d(wn("generator.1")), device), _dev(sd["generator.1.bias"], device))
        self.stages = []
        for j, s in enumerate(weights.UPSAMPLE_SCALES):
            up = "generator.%d.layer" % (3 + 3 * j)
            rs = "generator.%d" % (4 + 3 * j)
            upp = packing.pack_convtr1d(wn(up))
            # (w_packed, w_direct, bias, Winograd F(3,2) planes per output phase for convtw_kernel)
            upw = _wpair(upp, device) + (_dev(sd[up + ".bias"], device), _dev(packing.pack_wino32_tr(upp, s), device))
            layers = []
            cst = weights.VOC_CHANNELS >> (j + 1)
            wino = WINO_MIN_C > 0 and cst >= WINO_MIN_C and cst % 64 == 0
            for i in range(weights.RESSTACK_DEPTH):
                a = "%s.layers.%d.1" % (rs, i)
                b = "%s.layers.%d.3" % (rs, i)
                wa, wb = packing.pack_conv1d(wn(a)), packing.pack_conv1d(wn(b))
                # (w1, w1d, b1, w2, w2d, b2, w2 as F(2,3) for the fused layer's LDS half, w1 / w2 as F(4,3) for two launches --
                # the fused C = 64 layer takes w2 as F(4,3) too: its LDS half runs on it, F(2,3) is the fallback)
                layers.append(_wpair(wa, device) + (_dev(sd[a + ".bias"], device),) +
                              _wpair(wb, device) + (_dev(sd[b + ".bias"], device),) +
                              (_dev(packing.pack_wino(wb), device) if cst <= FUSE_MAX_C and not wino else None,) +
                              ((_dev(packing.pack_wino4(wa), device), _dev(packing.pack_wino4(wb), device)) if wino
                               else ((_dev(packing.pack_wino4(wa), device), _dev(packing.pack_wino4(wb), device)) if cst == 64
                                     else (None, None))))
            self.stages.append((s, upw, layers))
        self.post = (_dev(packing.pack_cout1(wn("generator.16")), device), _dev(sd["generator.16.bias"], device))
        self.act_elu = ops.Act(post=POST_ELU)
        self.act_pre = ops.Act(post=POST_LRELU_SNAKE, post_slope=0.2)
        self.act_c1 = ops.Act(pre=PRE_LRELU, pre_slope=0.01, post=POST_LRELU, post_slope=0.01)
        self.act_none = ops.Act()
        self.act_last_snake = ops.Act(post=POST_LRELU_SNAKE, post_slope=0.2)
        self.act_last = ops.Act(post=POST_LRELU, post_slope=0.2)

    def set_math(self, math):
        """"f32" (default: exact fp32 MFMA) or "bf16x3" (opt-in VFX_MATH_BF16X3: split-bf16 products with fp32
        accumulation; bf16 weight planes are packed lazily, once per layer)."""
        if math not in ("f32", "bf16x3"):
            raise ValueError("math must be 'f32' or 'bf16x3'")
        self.math = math
        if not hasattr(self, "_w3"):
            self._w3 = {}

    def _x3(self, w):
        if self.math != "bf16x3":
            return None
        key = w.data_ptr()
        if key not in self._w3:
            p = packing.pack_x3(w.cpu())
            self._w3[key] = None if p is None else p.to(w.device)
        return self._w3[key]

    def forward_cond(self, cond, Tc, stages=None, ragged=None):
        """cond: device (B,128,>=Tc) channel-major.  Returns (wav buffer (B,1,Lp), L = 441*Tc).
        ``ragged`` (RaggedRows): Tc is the largest row, row b has ragged.voc[1][b] valid frames."""
        B = cond.shape[0]
        dev = cond.device
        rows = (lambda mult: ragged.voc[mult]) if ragged is not None else (lambda mult: None)
        a = _rows(B, weights.COND_CHANNELS, Tc, G_TILE, dev, rows(1))
        b = _rows(B, weights.COND_CHANNELS, Tc, G_TILE, dev, rows(1))
        x = cond
        for i, (w, wd, bias, wg) in enumerate(self.condnet):
            y = a if i % 2 == 0 else b
            ops.conv1d(x, w, bias, y, Tc, 3, 1, PAD_ZERO, self.act_elu, w3=self._x3(w), wd=wd,
                       wg4=_wg(wg) if self.math == "f32" else None)
            x = y
        if stages is not None:
            stages["condnet"] = x[:, :, :Tc]
        # pre: ReflectionPad1d(3) + Conv1d k7 + LeakyReLU(0.2); the next UpsampleNet's x+sin(x) is fused here
        h = _rows(B, weights.VOC_CHANNELS, Tc, G_TILE, dev, rows(1))
        ops.conv1d(x, self.pre[0], self.pre[1], h, Tc, 7, 1, PAD_REFLECT, self.act_pre)
        L = Tc
        c = weights.VOC_CHANNELS
        nst = len(self.stages)
        mult = 1
        for j, (s, upw, layers) in enumerate(self.stages):
            Lo = L * s
            c //= 2
            mult *= s
            # (the fused kernel addresses one batch item with 32-bit byte offsets: rows of more than ~3 minutes at the last
            # stage fall back to the two-launch form, whose first-generation kernel has no such limit)
            wino = layers[0][7] is not None and c >= WINO_MIN_C and self.math == "f32" and _ARITH["winograd"]
            fused = (_FUSE and self.math == "f32" and c <= FUSE_MAX_C and not wino and
                     c * (_up4(Lo) + 2 * (G_DIL + 4)) * 4 < 2 ** 31 - 2 ** 21)
            xs = _rows(B, c, Lo, G_DIL, dev, rows(mult))
            ys = _rows(B, c, Lo, G_DIL if fused else G_TILE, dev, rows(mult))
            ops.convtr1d(h, upw[0], upw[2], xs, L, s, self.act_none, w3=self._x3(upw[0]), wd=upw[1],
                         wg4=_wg(upw[3]) if self.math == "f32" else None)
            if stages is not None:
                stages["up%d" % (j + 1)] = xs[:, :, :Lo].clone()
            # The fused C = 64 stage runs its WIDELY dilated layers (d > 27: a block of 4 d positions does not fit the fused tile) as two
            # Winograd F(4,3) launches each, in place on xs, from the first such layer on.  Decided once per stage: the index must be
            # even (the fused ping-pong has the data back in xs there, and everything after it runs in place), and the launch must be
            # large enough for convwg4_kernel to take it (it declines fewer than 512 tiles: B * Lo / 256) -- a short single utterance
            # keeps the fused layer instead of falling to two non-Winograd launches.
            first_unfused = None
            if (fused and _UNFUSE_WIDE and c == 64 and _ARITH["winograd"] and all(l[7] is not None and l[8] is not None for l in layers)
                    and B * Lo >= 1 << 18):
                wide = [i for i in range(len(layers) - 1) if 3 ** i > 27 and i % 2 == 0]
                first_unfused = wide[0] if wide else None
                assert first_unfused is None or first_unfused % 2 == 0
            for i, (w1, w1d, b1, w2, w2d, b2, w2g, w1g4, w2g4) in enumerate(layers):
                last = i == len(layers) - 1
                if fused and (first_unfused is None or i < first_unfused):
                    # one launch per layer, intermediate tile in LDS; input and output ping-pong between xs and ys
                    # (a tile reads its neighbours' input columns, so the update cannot be in place)
                    post, pslope = POST_NONE, 0.0
                    if last:
                        post, pslope = (POST_LRELU if j == nst - 1 else POST_LRELU_SNAKE), 0.2
                    src, dst = (xs, ys) if i % 2 == 0 else (ys, xs)
                    ops.resblock(src, dst, w1d, b1, w2d, b2, Lo, 3 ** i, 0.01, post, pslope, w2g=_wg(w2g), w2g4=_wg(w2g4),
                                 w1g4=_wg(w1g4))
                    continue
                if not wino and not fused:
                    w1g4 = w2g4 = None
                ops.conv1d(xs, w1, b1, ys, Lo, 3, 3 ** i, PAD_ZERO, self.act_c1, w3=self._x3(w1), wd=w1d, wg4=_wg(w1g4))
                act = self.act_none if not last else (self.act_last if j == nst - 1 else self.act_last_snake)
                ops.conv1d(ys, w2, b2, xs, Lo, 3, 1, PAD_ZERO, act, res=xs, w3=self._x3(w2), wd=w2d, wg4=_wg(w2g4))  # residual updated in place
            assert len(layers) % 2 == 0  # the fused ping-pong ends in xs
            h = xs
            L = Lo
            del ys
        wav = torch.empty((B, 1, _up4(L)), device=dev)
        ops.conv1d_cout1(h, self.post[0], self.post[1], wav, L, 7, PAD_REFLECT, POST_TANH)
        return wav, L

    def forward(self, mel, T, ragged=None, stages=None):
        """mel: device (B,T,128) linear, non-normalised (Vocoder.forward semantics); ``ragged``: row b holds
        ragged.T[b] <= T frames.  ``stages`` (optional dict) receives "cond", "condnet", "up1" .. "up4"."""
        B = mel.shape[0]
        Tc = T + T % 2 + 4
        if ragged is None:
            cond = _rows(B, weights.N_MELS, Tc, G_TILE, mel.device)
            ops.mel_to_cond(mel, cond, T)
            if stages is not None:
                stages["cond"] = cond[:, :, :Tc].clone()
            return self.forward_cond(cond, Tc, stages=stages)
        cond = _rows(B, weights.N_MELS, Tc, G_TILE, mel.device, ragged.voc[1])
        ops.mel_to_cond(mel, cond, T, ragged.T)
        return self.forward_cond(cond, Tc, ragged=ragged)


class _ConvBlock:
    """ConvBlockRes (restorer/modules.py:7-76) with bn2 folded into conv1."""

    def __init__(self, sd, p, device, pad_cin=None):
        s1, sh1 = weights.bn_affine(sd, p + ".bn1")
        s2, sh2 = weights.bn_affine(sd, p + ".bn2")
        w1 = sd[p + ".conv1.weight"].float() * s2.reshape(-1, 1, 1, 1)
        self.cin = w1.shape[1]
        if pad_cin is not None and pad_cin > self.cin:
            # zero filler input channels (packed weights are zero-padded to 8 anyway): scale 1, shift 0
            extra = pad_cin - self.cin
            s1 = torch.cat([s1, torch.ones(extra)])
            sh1 = torch.cat([sh1, torch.zeros(extra)])
            self.cin = pad_cin
        self.cout = w1.shape[0]
        wp1, wp2 = packing.pack_conv2d(w1), packing.pack_conv2d(sd[p + ".conv2.weight"])
        self.w1, self.w1d = _wpair(wp1, device)
        self.b1 = _dev(sh2, device)
        self.w2, self.w2d = _wpair(wp2, device)
        # Winograd F(4,3) along the map rows (convwg4s_kernel: 64-channel blocks with Cin % 32 == 0, or Cout = 32 -- UNet
        # level 0 -- with Cin % 16 == 0); what it declines (the 2 -> 32 entry convolution, the deep levels' small
        # launches) runs on the direct kernels
        wino4 = WINO2D and self.cout % 32 == 0
        cin_step4 = 32 if self.cout % 64 == 0 else 16
        self.w1g4 = _dev(packing.pack_wino4_2d(wp1), device) if wino4 and self.cin % cin_step4 == 0 else None
        self.w2g4 = _dev(packing.pack_wino4_2d(wp2), device) if wino4 and self.cout % cin_step4 == 0 else None
        self.act1 = ops.Act(pre=PRE_AFFINE_LRELU, pre_slope=0.01, scale=_dev(s1, device), shift=_dev(sh1, device),
                            post=POST_LRELU, post_slope=0.01)
        self.shortcut = None
        if (p + ".shortcut.weight") in sd:
            self.shortcut = (_dev(packing.pack_conv2d(sd[p + ".shortcut.weight"]), device),
                             _dev(sd[p + ".shortcut.bias"], device))
        self.x3 = (None, None, None)

    def set_math(self, math):
        """bf16 (hi, lo) weight planes for VFX_MATH_BF16X3 (packed once, on first use)."""
        if math != "bf16x3":
            self.x3 = (None, None, None)
            return
        if not hasattr(self, "_x3_planes"):
            def pk(w):
                q = packing.pack_x3(w.cpu()) if w.shape[1] % 32 == 0 else None
                return None if q is None else q.to(w.device)
            self._x3_planes = (pk(self.w1), pk(self.w2), pk(self.shortcut[0]) if self.shortcut is not None else None)
        self.x3 = self._x3_planes

    def run(self, x, y1, out, H, lp):
        """x (B,Cin,HP) -> out (B,Cout,HP); y1 scratch (B,Cout,HP).  ``out`` may alias ``x`` when there
        is no shortcut (the residual is read and written at the same position by the same thread)."""
        if self.shortcut is not None:
            ops.conv2d(x, self.shortcut[0], self.shortcut[1], out, H, lp, 1, None, cin=self.cin, w3=self.x3[2])
            res = out
        else:
            res = x
        ops.conv2d(x, self.w1, self.b1, y1, H, lp, 3, self.act1, cin=self.cin, w3=self.x3[0], wd=self.w1d, wg4=_wg(self.w1g4))
        ops.conv2d(y1, self.w2, None, out, H, lp, 3, None, res=res, w3=self.x3[1], wd=self.w2d, wg4=_wg(self.w2g4))


class RestorerEngine:
    """mel (B,T,128) -> (logmel, denoised mel), both (B,T,128)."""

    def __init__(self, state, device="cuda"):
        sd = {k: v for k, v in state.items()}
        weights.check_state(sd, weights.restorer_manifest(), "restorer")
        self.device = device
        f = lambda k: sd[k].float()

        def bn_scalar(p):
            a = f(p + ".weight") / torch.sqrt(f(p + ".running_var") + 1e-5)
            return a.item(), (f(p + ".bias") - f(p + ".running_mean") * a).item()

        def lin_in_bn(wk, bk, a, b):
            """Linear applied to (a*x + b): W' = a W, bias' = bias + b * W.sum(1)."""
            W, bias = f(wk), f(bk)
            return W * a, bias + b * W.sum(dim=1)

        a0, b0 = bn_scalar("denoiser.0")
        W, bias = lin_in_bn("denoiser.1.weight", "denoiser.1.bias", a0, b0)
        self.l1 = (_dev(packing.pack_linear(W), device), _dev(bias, device))
        a1, b1 = bn_scalar("denoiser.3")
        W, bias = lin_in_bn("denoiser.4.weight", "denoiser.4.bias", a1, b1)
        self.l2 = (_dev(packing.pack_linear(W), device), _dev(bias, device))
        self.grus = []
        for idx in (7, 8):
            a, b = bn_scalar("denoiser.%d.bn" % idx)
            layers = []
            for layer in (0, 1):
                Ws, bs, whh, bhh = [], [], [], []
                for suf in ("", "_reverse"):
                    p = "denoiser.%d.gru." % idx
                    Wih, bih = f(p + "weight_ih_l%d%s" % (layer, suf)), f(p + "bias_ih_l%d%s" % (layer, suf))
                    if layer == 0:
                        bih = bih + b * Wih.sum(dim=1)
                        Wih = Wih * a
                    Ws.append(Wih)
                    bs.append(bih)
                    whh.append(f(p + "weight_hh_l%d%s" % (layer, suf)))
                    bhh.append(f(p + "bias_hh_l%d%s" % (layer, suf)))
                layers.append((_dev(packing.pack_linear(torch.cat(Ws, 0)), device), _dev(torch.cat(bs, 0), device),
                               _dev(packing.pack_gru_whh(whh[0], whh[1], *ops.gru_layout()), device),
                               _dev(torch.stack(bhh), device),
                               _dev(torch.stack([whh[0].t().contiguous(), whh[1].t().contiguous()]), device)))
            self.grus.append(layers)
        a4, b4 = bn_scalar("denoiser.9")
        a5, b5 = bn_scalar("denoiser.13")
        W, bias = f("denoiser.11.weight"), f("denoiser.11.bias")
        self.l3 = (_dev(packing.pack_linear(W * a5), device), _dev(bias * a5 + b5, device))
        self.act_l3 = ops.Act(pre=PRE_AFFINE_LRELU, pre_slope=0.0, scale=torch.full((512,), a4, device=device),
                              shift=torch.full((512,), b4, device=device), post=POST_LRELU, post_slope=0.0)
        self.l4 = (_dev(packing.pack_linear(f("denoiser.15.weight")), device), _dev(f("denoiser.15.bias"), device))
        self.act_relu = ops.Act(post=POST_LRELU, post_slope=0.0)
        self.act_sigmoid = ops.Act(post=POST_SIGMOID)
        self.gru_err = None   # device flag: set by vfx_gru_bidir2_f32 if a partner workgroup timed out
        self._gru_keep = []
        # utterances per two-CU GRU launch (4 workgroups of 512 threads each, one per CU): a launch must be able
        # to become resident next to the launches of the caller's other streams (Pipeline.set_streams)
        self.gru_group = ops.GRU2_MAX_B
        self.gru_single = False     # True: recurrences run on vfx_gru_bidir_f32 (one workgroup per sequence, no hand-off)
        self._force_gru_miss = 0    # test hook: raise the hand-off flag after the next n two-CU launches

        self.enc = []
        for b in range(1, 7):
            self.enc.append([_ConvBlock(sd, "unet.encoder_block%d.conv_block%d" % (b, k), device,
                                        pad_cin=8 if (b == 1 and k == 1) else None)
                             for k in (1, 2, 3, 4)])
        self.center = _ConvBlock(sd, "unet.conv_block7", device)
        self.dec = []
        for b in range(1, 7):
            p = "unet.decoder_block%d" % b
            s, sh = weights.bn_affine(sd, p + ".bn1")
            act = ops.Act(pre=PRE_AFFINE_LRELU, pre_slope=0.0, scale=_dev(s, device), shift=_dev(sh, device))
            self.dec.append((_dev(packing.pack_convtr2d(sd[p + ".conv1.weight"]), device), act,
                             [_ConvBlock(sd, "%s.conv_block%d" % (p, k), device) for k in (2, 3, 4, 5)]))
        self.after = _ConvBlock(sd, "unet.after_conv_block1", device)
        self.after2 = (_dev(packing.pack_cout1(sd["unet.after_conv2.weight"]), device),
                       _dev(sd["unet.after_conv2.bias"], device))

    # -- denoiser -------------------------------------------------------------------
    def denoiser(self, mel, T, t_rows=None):
        """mel (B,T,128) -> mask channel-major (B,128,Tp4).  ``t_rows`` (device int32 (B,)): frames of every row of a
        ragged batch -- only the recurrence sees them (the reverse direction starts at each row's own last frame), the
        linear layers are frame-wise."""
        B, dev = mel.shape[0], mel.device
        Tp4 = _up4(T)
        x0 = _rows(B, 128, T, G_TILE, dev)
        ops.tm_to_cm(mel, x0, T, 128)
        x1 = _rows(B, 256, T, G_TILE, dev)
        ops.conv1d(x0, self.l1[0], self.l1[1], x1, T, 1, act=self.act_relu)
        x = _rows(B, 512, T, G_TILE, dev)
        ops.conv1d(x1, self.l2[0], self.l2[1], x, T, 1, act=self.act_relu)
        gi = torch.empty((B, T, 1536), device=dev)
        if self.gru_err is None:
            self.gru_err = torch.zeros(1, dtype=torch.int32, device=dev)
        keep = []
        for layers in self.grus:
            for (wih, bih, whh_packed, bhh, whh_t) in layers:
                ops.conv1d(x, wih, bih, gi.transpose(1, 2), T, 1)
                y = _rows(B, 512, T, G_TILE, dev)
                # two CUs per sequence (W_hh resident in registers) while all 4*B workgroups fit on the
                # chip; larger batches are walked in resident-sized groups
                for b0 in range(0, B, self.gru_group):
                    b1 = min(B, b0 + self.gru_group)
                    yv = y[b0:b1]
                    yv._vfx_guard = getattr(y, "_vfx_guard", 0)
                    if t_rows is not None:
                        ops.with_rows(yv, t_rows[b0:b1])
                    if self.gru_single:
                        ops.gru_bidir(gi[b0:b1], whh_packed, bhh, yv, T)      # one workgroup per sequence: no hand-off to miss
                    else:
                        keep.append(ops.gru_bidir2(gi[b0:b1], whh_t, bhh, yv, T, self.gru_err))
                        if self._force_gru_miss:                              # test hook (see Pipeline.run_checked)
                            self._force_gru_miss -= 1
                            self.gru_err.fill_(1)
                x = y
        self._gru_keep = keep  # mailboxes stay referenced until the next forward
        x3 = _rows(B, 512, T, G_TILE, dev)
        ops.conv1d(x, self.l3[0], self.l3[1], x3, T, 1, act=self.act_l3)
        mask = torch.empty((B, 128, Tp4), device=dev)
        ops.conv1d(x3, self.l4[0], self.l4[1], mask, T, 1, act=self.act_sigmoid)
        if t_rows is not None:
            ops.with_rows(mask, t_rows)
        return mask

    def set_math(self, math):
        """"f32" or "bf16x3" for the UNet's 3x3 / 1x1 convolutions (the denoiser and the transposed convolutions
        stay fp32: 1 % of the FLOPs)."""
        if math not in ("f32", "bf16x3"):
            raise ValueError("math must be 'f32' or 'bf16x3'")
        for grp in list(self.enc) + [[self.center]] + [d[2] for d in self.dec] + [[self.after]]:
            for blk in grp:
                blk.set_math(math)

    # -- ResUNet ---------------------------------------------------------------------
    def unet(self, u, Tp, ragged=None):
        """u (B,2,Tp*128) pitch map -> (B,1,Tp*128).  ``ragged``: row b is a map of ragged.unet[0][b] / 128 <= Tp rows;
        every 3x3 convolution zero-pads below the row's own last map row, exactly as it does for the utterance alone."""
        B, dev = u.shape[0], u.device
        lv = (lambda lp: ragged.unet[7 - lp]) if ragged is not None else (lambda lp: None)
        x = u
        cats = []
        H, lp = Tp, 7
        for blocks in self.enc:
            cout = blocks[0].cout
            HP = H << lp
            G = (1 << lp) + 1 + G_TILE
            cat = _rows(B, 2 * cout, HP, G, dev, lv(lp))
            skip = cat[:, cout:]
            a = _rows(B, cout, HP, G, dev, lv(lp))
            y1 = _rows(B, cout, HP, G, dev, lv(lp))
            blocks[0].run(x, y1, a, H, lp)
            blocks[1].run(a, y1, a, H, lp)
            blocks[2].run(a, y1, a, H, lp)
            blocks[3].run(a, y1, skip, H, lp)
            cats.append((cat, H, lp))
            pooled = _rows(B, cout, (H // 2) << (lp - 1), (1 << (lp - 1)) + 1 + G_TILE, dev, lv(lp - 1))
            ops.avgpool2x2(skip, pooled, H, lp)
            x = pooled
            H //= 2
            lp -= 1
        y1 = _rows(B, x.shape[1], H << lp, (1 << lp) + 1 + G_TILE, dev, lv(lp))
        self.center.run(x, y1, x, H, lp)
        for (wt, act, blocks) in self.dec:
            cat, Hs, lps = cats.pop()
            cout = blocks[0].cout
            assert Hs == 2 * H and lps == lp + 1
            ops.convtr2d_3x3s2(x, wt, cat[:, :cout], H, lp, act)
            H, lp = Hs, lps
            HP = H << lp
            G = (1 << lp) + 1 + G_TILE
            a = _rows(B, cout, HP, G, dev, lv(lp))
            y1 = _rows(B, cout, HP, G, dev, lv(lp))
            blocks[0].run(cat, y1, a, H, lp)
            for blk in blocks[1:]:
                blk.run(a, y1, a, H, lp)
            x = a
        self.after.run(x, y1, x, H, lp)
        out = torch.empty((B, 1, H << lp), device=dev)
        ops.conv1d_cout1(x, self.after2[0], self.after2[1], out, H << lp, 1, PAD_ZERO, POST_NONE, lp)
        return out

    def forward(self, mel, T, debug=None, ragged=None):
        """``ragged`` (RaggedRows): row b of mel holds ragged.T[b] <= T frames; rows of logmel / denoised are valid up
        to their own frame count."""
        B, dev = mel.shape[0], mel.device
        Tp = (T + 63) // 64 * 64
        mask = self.denoiser(mel, T, None if ragged is None else ragged.T)
        u = _rows(B, 8, Tp * 128, 128 + 1 + G_TILE, dev,  # 2 real + 6 zero channels: no channel tail
                  None if ragged is None else ragged.unet[0])
        ops.unet_input(mel, mask, u, T, Tp)
        uo = self.unet(u, Tp, ragged)
        logmel = torch.empty((B, T, 128), device=dev)
        den = torch.empty((B, T, 128), device=dev)
        ops.unet_output(uo, u, mel, mask, logmel, den, T, Tp)
        if debug is not None:
            debug["mask"] = mask[:, :, :T]
            debug["unet_out"] = uo[:, 0, :Tp * 128].reshape(B, Tp, 128)[:, :T]
        return logmel, den


class Pipeline:
    """wav batch (B,N) on the device -> restored (B,N): voicefixer/base.py:123-135 for
    B equal-length segments at once (per-utterance peak rule)."""

    def __init__(self, vocoder_state, restorer_state, device="cuda", math="f32"):
        if not torch.cuda.is_available():
            raise VfxError("no HIP device visible: the MI355X path has no CPU fallback")
        self.device = device
        self.vocoder = VocoderEngine(vocoder_state, device, math)
        self.restorer = RestorerEngine(restorer_state, device)
        self.restorer.set_math(math)
        self.math = math

    def set_math(self, math):
        """"f32" (default) or "bf16x3" (opt-in split-bf16 MFMA products, fp32 accumulation)."""
        self.vocoder.set_math(math)
        self.restorer.set_math(math)
        self.math = math

    def set_streams(self, streams):
        """Tell the engine how many HIP streams issue ``restore`` concurrently: the two-CU GRU keeps one workgroup
        per CU resident, so its launches are sized to 256 CUs / (4 workgroups per utterance x streams)."""
        self._n_streams = max(1, int(streams))
        self.restorer.gru_group = max(1, min(ops.GRU2_MAX_B, 256 // (4 * self._n_streams)))

    def check(self):
        """Read the device-side error flags (ONE 4-byte D2H copy; call it where the result crosses to the host,
        i.e. where the API synchronises anyway).  The two-CU GRU raises its flag when a partner workgroup did not
        answer within the bounded spin (vfx_gru.hip); the frames after that point were never written, so the
        waveform must not be returned: ``GruHandoffMissed`` -- ``run_checked`` turns it into a re-run."""
        flag = self.restorer.gru_err
        if flag is not None and int(flag.item()) != 0:
            flag.zero_()
            raise GruHandoffMissed("vfx_gru_bidir2_f32: a partner workgroup missed the bounded hand-off spin "
                                   "(GRU output incomplete); the result of this call was discarded")

    def run_checked(self, fn):
        """``fn()`` (one API call's worth of launches, returning HOST data) followed by ``check()``.  If the two-CU GRU
        reported a missed hand-off, the call is not lost: it is issued again with the recurrences on
        ``vfx_gru_bidir_f32`` -- one workgroup per sequence, no inter-workgroup traffic, nothing to miss (3.6 instead of
        2.2 us per step) -- and the two-CU kernel is back for the next call."""
        out = fn()
        try:
            self.check()
            return out
        except GruHandoffMissed:
            self.restorer.gru_single = True
            try:
                out = fn()
                self.check()
                self.gru_retries = getattr(self, "gru_retries", 0) + 1
                return out
            finally:
                self.restorer.gru_single = False

    def stage_report(self, wav, N):
        """The path's intermediates for one batch (selfcheck / parity tests): dict of device tensors -- "mel" (B,T,128),
        "mask" (B,T,128), "unet_out" (B,T,128), "logmel", "denoised", "cond" (B,128,T'), "condnet" (B,512,T'),
        "up1" .. "up4" (the four ConvTranspose1d outputs), "wav" (B,N): what ``restore`` returns."""
        mel, T = self.wav_to_mel(wav, N)
        dbg, st = {}, {}
        logmel, den = self.restorer.forward(mel, T, debug=dbg)
        y, Ly = self.vocoder.forward(den, T, stages=st)
        out = torch.empty((wav.shape[0], min(N, Ly)), device=wav.device)
        ws = torch.empty((wav.shape[0],), dtype=torch.int32, device=wav.device)
        ops.post(y[:, 0], Ly, out, out.shape[1], ws)
        rep = {"mel": mel, "mask": dbg["mask"].transpose(1, 2).contiguous(), "unet_out": dbg["unet_out"].contiguous(),
               "logmel": logmel, "denoised": den, "wav": out}
        rep.update({k: v.clone() for k, v in st.items()})
        return rep

    def wav_to_mel(self, wav, N):
        B = wav.shape[0]
        T = 1 + N // 441
        mel = torch.empty((B, T, 128), device=wav.device)
        ops.stft_mel(wav, mel, N)
        return mel, T

    # ---- HIP-graph replay for repeated (batch, length) shapes --------------------------------------------------
    def enable_graphs(self, max_shapes=4, max_batch=4):
        """Opt in: ``restore`` of a (B <= max_batch, N) shape seen before replays ONE captured HIP graph of its ~300
        kernel launches instead of issuing them one by one from Python.  The graph owns its
        intermediate buffers (torch's graph-private pool: ~0.4 GB per utterance of 10 s), so at most ``max_shapes``
        shapes are kept (least recently used first out).  Results are bit-identical to the eager path.  Meant for
        latency-sensitive small batches (single utterances, equal-length streaming chunks); at batch 32 the launches
        are <1 % of the step and the eager path is used."""
        self._graphs = {}
        self._graph_cap = (int(max_shapes), int(max_batch))

    def disable_graphs(self):
        self._drop_graphs(None)
        self._graphs = None

    def _drop_graphs(self, key):
        """Destroy one captured graph (``key``) or all of them (None).  Destroying a graph releases its private memory pool; the
        device is drained first so that nothing queued can still be reading from it (graph destruction is rare: disable, eviction
        of the least recently used shape)."""
        graphs = getattr(self, "_graphs", None)
        if not graphs:
            return
        torch.cuda.synchronize(self.device)
        if key is None:
            graphs.clear()
        else:
            graphs.pop(key, None)
        torch.cuda.synchronize(self.device)

    def _capture(self, B, N):
        dev = self.device
        static_in = torch.zeros((B, N), device=dev)
        s = torch.cuda.Stream(device=dev)
        s.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(s):
            # two eager passes on the capture stream: every first-use allocation of the library (tap tables, split-K
            # workspace of THIS stream, front-end tables) happens here, none during capture
            self._restore_eager(static_in, N)
            self._restore_eager(static_in, N)
        s.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            static_out = self._restore_eager(static_in, N)
        torch.cuda.current_stream(dev).wait_stream(s)
        return [g, static_in, static_out, None]   # [3]: event recorded after the last user's clone of static_out

    def restore(self, wav, N, vocoder_func=None):
        """wav: device float32 (B, >=N).  Returns device (B, N)."""
        graphs = getattr(self, "_graphs", None)
        # A graph entry owns ONE static input / output pair per (B, N) shape.  While several streams issue ``restore``
        # concurrently (restore_batch / restore_folder: set_streams(> 1)) two of them could hold the same entry at once
        # -- stream 1's copy into static_in is unordered against stream 0's replay -- so the replay path is bypassed
        # there; sequential users on DIFFERENT streams are ordered through the entry's event.
        # While ``gru_single`` is set (run_checked's re-run after a missed two-CU hand-off) the replay path is bypassed
        # too: a captured graph has the two-CU kernel baked in (replaying it would not be the miss-proof re-run), and a
        # shape first seen during a re-run would be captured with the slower one-workgroup kernel for good.
        if graphs is not None and vocoder_func is None and ops.PROFILE is None and wav.shape[0] <= self._graph_cap[1] \
                and N >= 1025 and getattr(self, "_n_streams", 1) == 1 and not self.restorer.gru_single:
            key = (wav.shape[0], N)
            ent = graphs.pop(key, None)
            if ent is None:
                while len(graphs) >= self._graph_cap[0]:
                    self._drop_graphs(next(iter(graphs)))
                ent = self._capture(*key)
            graphs[key] = ent  # most recently used last
            g, static_in, static_out, last_use = ent
            cur = torch.cuda.current_stream(self.device)
            if last_use is not None:
                cur.wait_event(last_use)   # the previous user (possibly on another stream) has finished with the pair
            static_in.copy_(wav[:, :N])
            g.replay()
            out = static_out.clone()
            ev = torch.cuda.Event()
            ev.record(cur)
            ent[3] = ev
            return out
        return self._restore_eager(wav, N, vocoder_func)

    def restore_rows(self, wav, lengths, force_ragged=False):
        """A RAGGED batch: utterances of different sample counts in one launch sequence.  wav device float32
        (B, >= max(lengths)), lengths a list of ints (each >= 1025).  Every kernel whose result depends on where a
        sequence ends takes the per-row lengths (RaggedRows): the STFT reflects at each row's end, the GRU's reverse
        direction starts at each row's last frame, every convolution zero- or reflect-pads at the row's own end (and
        skips the tiles past it), the peak rule and the centre trim work on the row's own samples -- so row b equals
        what ``restore`` returns for that utterance alone (same arithmetic; tile shapes may differ with the batch
        size, which moves fp32 sums by ~1e-7).  Returns device (B, max(lengths)); row b is valid up to lengths[b]
        (zero beyond).  Equal lengths take the plain batched path unless ``force_ragged`` (tests)."""
        if min(lengths) < 1025:
            raise VfxError("segment of %d samples is too short for the reflect-padded STFT (needs > 1024)" % min(lengths))
        B = wav.shape[0]
        if len(lengths) != B:
            raise VfxError("restore_rows: %d lengths for %d rows" % (len(lengths), B))
        if min(lengths) == max(lengths) and not force_ragged:
            return self._restore_eager(wav, lengths[0])
        rg = RaggedRows(lengths, wav.device)
        T = rg.T_max
        mel = torch.empty((B, T, 128), device=wav.device)
        ops.stft_mel_rows(wav, mel, rg.n, T)
        _, den = self.restorer.forward(mel, T, ragged=rg)
        y, Ly = self.vocoder.forward(den, T, ragged=rg)
        out = torch.zeros((B, rg.n_max), device=wav.device)
        ws = torch.empty((B,), dtype=torch.int32, device=wav.device)
        ops.post_rows(y[:, 0], Ly, out, rg.n, rg.n_max, ws, ly_rows=rg.voc[441])
        return out

    def _restore_eager(self, wav, N, vocoder_func=None):
        if N < 1025:
            raise VfxError("segment of %d samples is too short for the reflect-padded STFT (needs > 1024); "
                           "the reference raises inside torch reflect-pad here" % N)
        B = wav.shape[0]
        mel, T = self.wav_to_mel(wav, N)
        _, den = self.restorer.forward(mel, T)
        if vocoder_func is None:
            y, Ly = self.vocoder.forward(den, T)
            y = y[:, 0]
        else:
            y = vocoder_func(den[:, None])  # plugin hook: (B,1,T,128) -> (B,1,samples)
            y = y.to(wav.device).float().contiguous()[:, 0]
            Ly = y.shape[-1]
        # _trim_center (base.py:63-76): an estimate LONGER than the segment is centre-cropped to N samples; a
        # SHORTER one (only a foreign your_vocoder_func can produce it) is returned as it is, Ly samples
        n_out = min(N, Ly)
        out = torch.empty((B, n_out), device=wav.device)
        ws = torch.empty((B,), dtype=torch.int32, device=wav.device)
        ops.post(y, Ly, out, n_out, ws)
        return out
