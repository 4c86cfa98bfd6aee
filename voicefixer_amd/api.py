"""Drop-in Python surface of the reference for the restore / vocoder path.

    from voicefixer_amd import VoiceFixer, Vocoder          # == from voicefixer import ...

``VoiceFixer`` mirrors voicefixer/base.py:10-146 and ``Vocoder`` mirrors
voicefixer/vocoder/base.py:10-77: same constructor behaviour (checkpoint locations and the
"Error 0"/"Error 1" RuntimeErrors), same method names, argument meaning and return types, the
``your_vocoder_func`` plugin hook (base.py:126-129) and the 30 s hard-cut segmentation
(base.py:117-137).  Everything between the waveform going in and the waveform coming out runs
in libvfx_hip on the MI355X; there is NO CPU implementation behind this API:

  * ``cuda=True``  -> tensors returned on the HIP device where the reference would return CUDA tensors;
  * ``cuda=False`` -> same kernels, results copied back to host tensors (the reference would
    compute on the CPU; numerically equivalent within the parity tolerance).  Without a visible
    device either setting raises -- nothing silently falls back.
  * ``mode=0`` and ``mode=1`` (``remove_higher_frequency`` pre-filter, base.py:87-104, run on the
    device by ``vfx_hf_cut_f32``).  ``mode=2`` (train-mode BatchNorm/Dropout, nondeterministic and
    exempt from the reference's own check, test/test.py:58) raises NotImplementedError.
"""
import os

import numpy as np
import torch
import torch.nn as nn

from . import audio_io, engine, weights
from ._lib import VfxError

SEG_LENGTH = 44100 * 30  # voicefixer/base.py:117

ANALYSIS_CKPT = ".cache/voicefixer/analysis_module/checkpoints/vf.ckpt"
VOCODER_CKPT = ".cache/voicefixer/synthesis_module/44100/model.ckpt-1490000_trimed.pt"


def _device():
    if not torch.cuda.is_available():
        raise RuntimeError("Error: no HIP device found; voicefixer_amd has no CPU fallback "
                           "(the reference raises 'You set cuda=True but no cuda device found.' here)")
    return torch.device("cuda", torch.cuda.current_device())


def _load_vocoder_state(path):
    ckpt = torch.load(path, map_location="cpu")
    return ckpt["generator"]  # voicefixer/vocoder/base.py:26-27


def _load_restorer_state(path):
    """vf.ckpt is a flat state dict of restorer.model.VoiceFixer; the engine needs the
    ``generator.*`` (denoiser + unet) entries (voicefixer/base.py:23-29 key filter)."""
    sd = torch.load(path, map_location="cpu")
    if "state_dict" in sd and not any(k.startswith("generator.") for k in sd):
        sd = sd["state_dict"]
    out = {k[len("generator."):]: v for k, v in sd.items() if k.startswith("generator.")}
    voc = {k[len("vocoder.model."):]: v for k, v in sd.items() if k.startswith("vocoder.model.")}
    return out, (voc or None)  # vf.ckpt may overwrite the vocoder weights


def plan_batches(sorted_lengths, batch_size, ragged_ratio=0.5, ragged=True):
    """Cut a list of ASCENDING sample counts into batches: ("ragged", [positions]) for runs of utterances of
    1025..SEG_LENGTH samples whose shortest member has >= ragged_ratio of the frames (1 + n // 441) of the longest
    (Pipeline.restore_rows), ("samples", [positions]) for runs of exactly equal length otherwise (files of several
    segments, plugin vocoders, too-short files -- the last raise in the pipeline as the reference does)."""
    plan = []
    i, n_items = 0, len(sorted_lengths)
    while i < n_items:
        n0 = sorted_lengths[i]
        j = i + 1
        if ragged and 1025 <= n0 <= SEG_LENGTH:
            t0 = 1 + n0 // 441
            while (j < n_items and j - i < batch_size and sorted_lengths[j] <= SEG_LENGTH and
                   t0 >= ragged_ratio * (1 + sorted_lengths[j] // 441)):
                j += 1
            plan.append(("ragged", list(range(i, j))))
        else:
            while j < n_items and j - i < batch_size and sorted_lengths[j] == n0:
                j += 1
            plan.append(("samples", list(range(i, j))))
        i = j
    return plan


def plan_stream_chunks(n, chunk, overlap, min_tail=1024):
    """Chunk starts / lengths of the overlap-add streaming mode: chunks of ``chunk`` samples every
    ``chunk - overlap`` samples; the last one is shorter.  A tail that would be too short to restore
    (<= overlap + ``min_tail`` samples: one cross-fade plus the reflect pad of the STFT; mode 1 passes 1535 because its
    pre-filter first shortens a chunk to 512 * (len // 512) samples) is merged into its predecessor."""
    if chunk <= overlap + 1024 or overlap < 0:
        raise ValueError("chunk must exceed overlap + 1024 samples")
    hop = chunk - overlap
    plan = []
    start = 0
    while True:
        length = min(chunk, n - start)
        plan.append([start, length])
        if start + length >= n:
            break
        start += hop
    if len(plan) > 1 and plan[-1][1] <= overlap + min_tail:
        last = plan.pop()
        plan[-1][1] = last[0] + last[1] - plan[-1][0]
    return [tuple(c) for c in plan]


class Vocoder(nn.Module):
    """44.1 kHz TFGAN-style universal vocoder (voicefixer/vocoder/base.py)."""

    def __init__(self, sample_rate=44100, _state=None):
        super().__init__()
        if sample_rate != 44100:
            raise RuntimeError("Error: Vocoder currently only support 44100 samplerate.")  # config.py:28-31
        self.rate = sample_rate
        if _state is None:
            path = os.path.join(os.path.expanduser("~"), VOCODER_CKPT)
            if not os.path.exists(path):
                raise RuntimeError(
                    "Error 1: The checkpoint for synthesis module / vocoder (model.ckpt-1490000_trimed) is not "
                    "found in ~/.cache/voicefixer/synthesis_module/44100. There is no network in this build; "
                    "place the Zenodo file there (https://zenodo.org/record/5600188).")
            _state = _load_vocoder_state(path)
        self._state = _state
        self._engine = None

    @classmethod
    def from_state(cls, state):
        """Build from an in-memory generator state dict (either weight-norm key style)."""
        return cls(44100, _state=state)

    def _get_engine(self):
        if self._engine is None:
            self._engine = engine.VocoderEngine(self._state, _device(), getattr(self, "math", "f32"))
        return self._engine

    def set_math(self, math):
        """"f32" (default, exact) or "bf16x3" (opt-in split-bf16 products, see VoiceFixer.set_math)."""
        if math not in ("f32", "bf16x3"):
            raise ValueError("math must be 'f32' or 'bf16x3'")
        self.math = math
        if self._engine is not None:
            self._engine.set_math(math)

    def forward(self, mel, cuda=False):
        """mel: [B, 1, T, 128] linear, non-normalised -> [B, 1, 441*(T + T%2 + 4)]."""
        assert mel.size()[-1] == 128
        eng = self._get_engine()
        dev = eng.device
        m = mel.detach().to(dev, torch.float32)[:, 0].contiguous()
        T = m.shape[1]
        wav, L = eng.forward(m, T)
        out = wav[:, :, :L]
        return out if cuda else out.cpu()

    __call__ = forward  # usable directly as ``your_vocoder_func`` (with the default cuda=False)

    def oracle(self, fpath, out_path, cuda=False):
        """wav file -> ground-truth mel (librosa-style STFT + slaney HTK mel) -> vocoder -> wav file
        (voicefixer/vocoder/base.py:58-77); only the file decode/encode runs on the host."""
        from . import ops
        wav = audio_io.load_wav(fpath, self.rate, mono=False)
        if wav.ndim == 2:
            wav = np.ascontiguousarray(wav[0])  # read_wave(fpath)[..., 0]: the FIRST channel, not a down-mix (vocoder/base.py:61)
        eng = self._get_engine()
        w = torch.from_numpy(wav)[None].to(eng.device)
        N = w.shape[1]
        mel, T = ops.oracle_mel(w, N)            # wav/max|wav| -> |STFT| -> slaney mel, on the device
        Tc = T + T % 2 + 4
        cond = ops.guarded(1, 128, Tc, engine.G_TILE, eng.device)
        ops.mel_to_cond_plain(mel, cond, T)      # amp_to_db - 20, normalize, pre()
        wav_re, L = eng.forward_cond(cond, Tc)
        audio_io.save_wave((wav_re[:, 0, :L] * 2 ** 15).cpu().numpy(), out_path, sample_rate=self.rate)


class _RestorerHandle(nn.Module):
    """What reference callers reach through ``VoiceFixer._model`` (voicefixer/base.py:13,109; test/streamlit.py:40-42 does
    ``list(vf._model.parameters())[0].is_cuda`` and ``vf._model = vf._model.to(device)``): an ``nn.Module`` with a
    ``vocoder`` attribute whose call is the restorer's forward ``(sp, mel_noisy) -> {"mel": log-mel, ...}``
    (restorer/model.py:102-120, 395-405).  The weights live in the engine's packed HIP buffers, so ``.to()`` only moves
    the one placeholder parameter that makes ``parameters()`` non-empty; compute always runs on the MI355X."""

    def __init__(self, owner):
        super().__init__()
        object.__setattr__(self, "_owner", owner)   # (not a sub-module: no cycle in nn.Module's registry)
        self.placeholder = nn.Parameter(torch.zeros(1), requires_grad=False)
        super().train(False)                        # base.py:30: the reference puts the restorer in eval mode at load time

    @property
    def vocoder(self):
        return self._owner._vocoder

    def train(self, mode=True):
        """nn.Module.train() recurses into children, so ``vf.train()`` / ``vf.eval()`` on the owning VoiceFixer (or any
        wrapper that toggles modes) must not raise here: the flag is stored, and asking for a forward pass in train mode
        (the reference's mode 2, base.py:115) is what fails."""
        return super().train(mode)

    @torch.no_grad()
    def forward(self, sp, mel_orig):
        """mel_orig: [B, 1, T, 128] linear mel -> {"mel": log10 of the restored mel [B, 1, T, 128], "clean", "noisy"};
        ``sp`` is ignored exactly as in the reference (restorer/model.py:102: the argument is unused)."""
        if self.training:
            raise NotImplementedError("mode 2 (BatchNorm / Dropout in train mode, base.py:115) is nondeterministic and "
                                      "not built; call .eval() first")
        assert mel_orig.size()[-1] == 128
        pipe = self._owner._get_pipe()
        m = mel_orig.detach().to(pipe.device, torch.float32)[:, 0].contiguous()
        dbg = {}
        logmel, _ = pipe.restorer.forward(m, m.shape[1], debug=dbg)
        pipe.check()
        out = {"mel": logmel[:, None], "noisy": mel_orig,
               "clean": (dbg["mask"].transpose(1, 2) * m)[:, None]}
        out["unet_out"] = out["lstm_out"] = dbg["unet_out"][:, None]
        dev = self.placeholder.device
        return {k: (v if v.device == dev else v.to(dev)) for k, v in out.items()}


class BatchSourceError(RuntimeError):
    """The iterator that FEEDS ``_restore_batches_isolated`` raised (staging allocation, decode planning): not a fault of a
    device batch, so no row-by-row re-issue can recover it -- the batches already issued are finished first, then this
    is raised; ``restore_folder`` records every file it never got to as failed."""


class VoiceFixer(nn.Module):
    """General speech restoration, inference path (voicefixer/base.py)."""

    def __init__(self, _states=None):
        super().__init__()
        if _states is None:
            path = os.path.join(os.path.expanduser("~"), ANALYSIS_CKPT)
            if not os.path.exists(path):
                raise RuntimeError(
                    "Error 0: The checkpoint for analysis module (vf.ckpt) is not found in "
                    "~/.cache/voicefixer/analysis_module/checkpoints. There is no network in this build; "
                    "place the Zenodo file there (https://zenodo.org/record/5600188/files/vf.ckpt).")
            restorer_state, voc_override = _load_restorer_state(path)
            vocoder = Vocoder(44100)
            if voc_override:
                merged = dict(weights.normalise_vocoder_keys(vocoder._state))
                merged.update(weights.normalise_vocoder_keys(voc_override))
                vocoder = Vocoder.from_state(merged)
        else:
            vocoder_state, restorer_state = _states
            vocoder = Vocoder.from_state(vocoder_state)
        self._vocoder = vocoder
        self._restorer_state = restorer_state
        self._pipe = None
        self._model = _RestorerHandle(self)   # the reference's attribute name (base.py:13); see _RestorerHandle
        self.math = "f32"       # "bf16x3": opt-in fast contraction arithmetic (set_math)
        self.segment_batch = 8  # 30 s segments of one long input restored per launch (~1.3 GB of HBM each)

    @classmethod
    def from_state(cls, vocoder_state, restorer_state):
        """Build from in-memory state dicts (restorer keys without the ``generator.`` prefix)."""
        return cls(_states=(vocoder_state, restorer_state))

    def _get_pipe(self):
        if self._pipe is None:
            dev = _device()
            self._pipe = engine.Pipeline(self._vocoder._state, self._restorer_state, dev, self.math)
            self._vocoder._engine = self._pipe.vocoder
        return self._pipe

    def set_math(self, math):
        """Contraction arithmetic of the convolution family (extension; the reference is fp32 only).
        "f32" (default): exact fp32 products on the fp32 MFMA.  "bf16x3": every fp32 operand is split into
        two bf16 terms and x*w is evaluated as xh*wh + xh*wl + xl*wh on the bf16 MFMA with fp32 accumulation
        (per-product relative error <= 2^-16; end-to-end waveform difference ~2e-6 RMS, the size of an fp32
        summation-order change, against the 1e-3 parity bound)."""
        if math not in ("f32", "bf16x3"):
            raise ValueError("math must be 'f32' or 'bf16x3'")
        self.math = math
        if self._pipe is not None:
            self._pipe.set_math(math)

    def enable_graphs(self, max_shapes=4, max_batch=4):
        """Extension: replay a captured HIP graph for repeated (batch <= max_batch, length) shapes (single utterances of
        one length, the equal-length chunks of ``restore_stream``); bit-identical results, see engine.Pipeline."""
        self._get_pipe().enable_graphs(max_shapes, max_batch)

    def _load_wav(self, path, sample_rate, threshold=0.95):
        return audio_io.load_wav(path, sample_rate)

    @staticmethod
    def _check_mode(mode):
        if mode in (0, 1):
            return
        if mode == 2:
            raise NotImplementedError(
                "mode=2 (train-mode BatchNorm + Dropout) is nondeterministic in the reference and outside "
                "the MI355X path; modes 0 and 1 are implemented")
        raise ValueError("mode must be 0, 1 or 2")

    @torch.no_grad()
    def restore_inmem(self, wav_10k, cuda=False, mode=0, your_vocoder_func=None):
        """wav_10k: float32 numpy (N,) at 44.1 kHz -> float32 numpy (1, N).
        30 s hard-cut segments, no overlap, concatenated (voicefixer/base.py:117-138)."""
        self._check_mode(mode)
        pipe = self._get_pipe()
        wav = np.asarray(wav_10k, dtype=np.float32)
        n = wav.shape[0]
        # Segment boundaries exactly as the reference's while-loop (base.py:117-120,137): full 30 s
        # segments, then a shorter tail.  Segments are independent in mode 0 (no carried state), so all
        # full segments of a long file go through the path as ONE batch (the reference runs them one
        # by one); the hard cuts and their positions are unchanged.
        bounds = []
        break_point = SEG_LENGTH
        while break_point < n + SEG_LENGTH:
            lo = break_point - SEG_LENGTH
            bounds.append((lo, min(break_point, n)))
            break_point += SEG_LENGTH
        full = [b for b in bounds if b[1] - b[0] == SEG_LENGTH]
        tail = [b for b in bounds if b[1] - b[0] != SEG_LENGTH]

        def run():
            res = []
            for i in range(0, len(full), self.segment_batch):
                grp = full[i:i + self.segment_batch]
                seg = torch.from_numpy(np.stack([wav[a:b] for a, b in grp])).to(pipe.device)
                out = self._restore_segments(pipe, seg, SEG_LENGTH, mode, your_vocoder_func)
                res.extend(out[k:k + 1] for k in range(len(grp)))
            for a, b in tail:
                seg = torch.from_numpy(np.ascontiguousarray(wav[a:b]))[None].to(pipe.device)
                res.append(self._restore_segments(pipe, seg, b - a, mode, your_vocoder_func))
            return torch.cat(res, -1).cpu().numpy()  # (synchronises)

        return pipe.run_checked(run)   # (device error flags are read here; a missed GRU hand-off re-runs the call)

    @staticmethod
    def _restore_segments(pipe, seg, n, mode, your_vocoder_func):
        """One batch of equal-length segments through the path; mode 1 first shortens every segment to
        512*(n//512) samples by the device-side high-frequency cut (base.py:121-122)."""
        if mode == 1:
            from . import ops
            seg, _ = ops.hf_cut(seg, n, 0.95)
            n = seg.shape[1]
        return pipe.restore(seg, n, your_vocoder_func)

    def _streams(self, streams):
        """The SAME stream objects on every call: torch's caching allocator keeps one block pool per stream, so fresh
        streams per call (torch hands them out round-robin from 32) would strand a batch's worth of HBM in a new pool
        each time until the allocator has to flush everything (measured: a 5x slower call after ~6 calls)."""
        pipe = self._get_pipe()
        if not hasattr(self, "_stream_pool"):
            self._stream_pool = []
        while len(self._stream_pool) < max(1, int(streams)):
            self._stream_pool.append(torch.cuda.Stream(device=pipe.device))
        return self._stream_pool[:max(1, int(streams))]

    def _issue_batch(self, pipe, stream, item, mode, your_vocoder_func):
        """Queue ONE batch on ``stream``: H2D of its pinned staging tensor, the launch sequence, D2H of the result into
        a pinned tensor, an event.  Nothing here waits for the device."""
        from . import ops
        tag, kind, host, lens = item
        lens = list(lens)
        if kind not in ("ragged", "samples") or len(lens) != host.shape[0] or max(lens) > host.shape[1]:
            raise ValueError("restore_batches: item must be (tag, 'ragged' | 'samples', host (B, >= max(lens)), lens (B))")
        if kind == "ragged" and your_vocoder_func is not None:
            raise ValueError("restore_batches: a plugin vocoder takes 'samples' batches (equal lengths); plan_batches(ragged=False) cuts them")
        if kind == "samples" and min(lens) != max(lens):
            raise ValueError("restore_batches: a 'samples' batch holds rows of ONE length")
        with torch.cuda.stream(stream):
            seg = host.to(pipe.device, non_blocking=True)
            if kind == "ragged":
                if mode == 1:
                    if min(lens) < 1536:
                        raise VfxError("mode 1 shortens a file to 512 * (n // 512) samples: %d samples leave too few for the "
                                       "reflect-padded STFT (needs > 1024 after the cut)" % min(lens))
                    new_lens = [512 * (n // 512) for n in lens]
                    # (rows are written in place into ONE buffer of exactly the width the longest cut row needs; what lies behind a
                    # row's own end is never read -- every kernel takes the per-row lengths -- so it is not cleared)
                    cut = torch.empty((len(lens), max(new_lens)), dtype=torch.float32, device=seg.device)
                    done = {}
                    for r in range(len(lens)):           # the cut-off is a per-file quantity (base.py:87-104);
                        if r in done:                    # rows of EQUAL length share one launch (vfx_hf_cut_f32 takes B rows)
                            continue
                        same = [q for q in range(r, len(lens)) if lens[q] == lens[r]]
                        if same == list(range(r, r + len(same))):
                            y, _ = ops.hf_cut(seg[r:r + len(same), :lens[r]], lens[r], 0.95)
                            cut[r:r + len(same), :y.shape[1]] = y
                        else:
                            for q in same:
                                y, _ = ops.hf_cut(seg[q:q + 1, :lens[q]], lens[q], 0.95)
                                cut[q, :y.shape[1]] = y[0]
                        for q in same:
                            done[q] = True
                    lens = new_lens
                    seg = cut
                full = pipe.restore_rows(seg, lens)
                lens_out = lens
            else:
                n = lens[0]
                parts = [self._restore_segments(pipe, seg[:, s0:s0 + SEG_LENGTH], min(SEG_LENGTH, n - s0), mode,
                                                your_vocoder_func) for s0 in range(0, n, SEG_LENGTH)]
                full = parts[0] if len(parts) == 1 else torch.cat(parts, -1)
                lens_out = [full.shape[-1]] * len(lens)
            out_host = torch.empty(tuple(full.shape), dtype=torch.float32, pin_memory=True)
            out_host.copy_(full, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(stream)
        return [item, out_host, lens_out, ev]

    @torch.no_grad()
    def restore_batches(self, batches, your_vocoder_func=None, streams=2, mode=0):
        """The device stage of folder inference as a GENERATOR: ``batches`` yields ``(tag, kind, host, lens)`` --
        ``host`` a pinned float32 (B, >= max(lens)) staging tensor whose row r holds ``lens[r]`` samples, ``kind``
        "ragged" (one launch sequence with per-row lengths, Pipeline.restore_rows) or "samples" (equal lengths: files
        of several 30 s segments, plugin vocoders) as ``plan_batches`` cuts them -- and the generator yields
        ``(tag, out_host, lens_out)`` in the same order, ``out_host`` a pinned (B, >= max(lens_out)) tensor.
        Batches go round-robin to ``streams`` HIP streams (the low-occupancy phases of one batch -- GRU recurrence, deep
        UNet levels -- overlap the convolutions of the next) and the host runs one batch per stream AHEAD of the device:
        H2D, the ~600 launches and the D2H of a batch are queued while earlier batches compute, so the device never waits
        for the host and the caller (restore_folder: decode / encode workers) works on other batches meanwhile.
        The two-CU GRU's error flag is read when a batch's result crosses to the host; a missed hand-off drains the
        batches in flight and re-issues them on the one-workgroup GRU kernel (Pipeline.run_checked's rule)."""
        from collections import deque
        from .engine import GruHandoffMissed
        self._check_mode(mode)
        pipe = self._get_pipe()
        pool = self._streams(streams)
        main = torch.cuda.current_stream(pipe.device)
        for st in pool:
            st.wait_stream(main)
        inflight = deque()
        pipe.set_streams(len(pool))
        ok = False
        try:
            try:
                pipe.check()                  # a flag that is already set belongs to an earlier, unchecked launch (a direct
            except GruHandoffMissed:          # pipe.restore user): check() has cleared it, and none of THIS call's work failed
                pass
            nb = 0

            def finish_oldest():
                rec = inflight[0]
                rec[3].synchronize()
                try:
                    pipe.check()
                except GruHandoffMissed:
                    # which of the batches in flight raised it cannot be told: drain them all, re-issue every one of
                    # them with the recurrences on vfx_gru_bidir_f32 (nothing to miss), one after the other
                    torch.cuda.synchronize(pipe.device)
                    pipe.restorer.gru_err.zero_()
                    pipe.restorer.gru_single = True
                    try:
                        for q in range(len(inflight)):
                            inflight[q] = self._issue_batch(pipe, pool[0], inflight[q][0], mode, your_vocoder_func)
                        torch.cuda.synchronize(pipe.device)
                        pipe.check()
                    finally:
                        pipe.restorer.gru_single = False
                    pipe.gru_retries = getattr(pipe, "gru_retries", 0) + 1
                    rec = inflight[0]
                inflight.popleft()
                return rec[0][0], rec[1], rec[2]

            for item in batches:
                inflight.append(self._issue_batch(pipe, pool[nb % len(pool)], item, mode, your_vocoder_func))
                nb += 1
                while len(inflight) > len(pool) + 1:
                    yield finish_oldest()
            while inflight:
                yield finish_oldest()
            ok = True
        finally:
            # in every case: drain the side streams and give the GRU its single-stream launch size back; when a batch
            # raised (a too-short file, an out-of-memory, a plugin vocoder error) or the caller abandoned the
            # generator, whatever the queued launches leave in the device-side error flag belongs to THIS call -- drop
            # it here so that a later, unrelated call does not inherit it
            torch.cuda.synchronize(pipe.device)
            pipe.set_streams(1)
            if not ok and pipe.restorer.gru_err is not None:
                pipe.restorer.gru_err.zero_()

    @torch.no_grad()
    def restore_batch(self, wavs, your_vocoder_func=None, batch_size=32, streams=2, ragged_ratio=0.5, mode=0):
        """Batched folder inference (not in the reference, which loops files at B=1,
        voicefixer/__main__.py:187-212): list of float32 numpy (N_i,) -> list of (1, N_i)  (mode 1: (1, 512*(N_i//512))
        per 30 s segment, as ``restore_inmem`` returns it).
        Utterances of up to 30 s go through RAGGED batches: the length-sorted list is cut into runs of up to
        ``batch_size`` files whose shortest member has at least ``ragged_ratio`` of the frames of the longest, and a run
        is ONE launch sequence in which every kernel takes the per-row lengths (Pipeline.restore_rows) -- each row is
        what restoring that utterance alone returns, the tiles past a row's end are skipped, only the buffers are
        sized for the longest row.  Longer files (several 30 s segments) and plugin vocoders are bucketed by exact
        length, one batched launch sequence per segment index.  The batches run through ``restore_batches`` (round-robin
        on ``streams`` HIP streams, pinned staging in both directions, the host one batch per stream ahead).
        ``mode=1``: every file (every 30 s segment of it) first goes through the device-side high-frequency cut
        (base.py:121-122, ``vfx_hf_cut_f32``) exactly as ``restore_inmem(mode=1)`` does it."""
        self._check_mode(mode)
        order = sorted(range(len(wavs)), key=lambda i: len(wavs[i]))
        outs = [None] * len(wavs)
        plan = plan_batches([len(wavs[k]) for k in order], batch_size, ragged_ratio, ragged=your_vocoder_func is None)

        def staged():
            for kind, grp in plan:
                idx = [order[g] for g in grp]
                lens = [len(wavs[k]) for k in idx]
                # rows are padded in a PINNED staging buffer (torch caches pinned blocks) and uploaded without
                # blocking the host, so that the next batch is staged while this one's copy and kernels run
                host = torch.empty((len(idx), max(lens)), dtype=torch.float32, pin_memory=True)
                hv = host.numpy()
                for r, k in enumerate(idx):
                    hv[r, :lens[r]] = wavs[k]
                    hv[r, lens[r]:] = 0.0
                yield idx, kind, host, lens

        for idx, out_host, lens_out in self.restore_batches(staged(), your_vocoder_func, streams, mode):
            ov = out_host.numpy()
            for r, k in enumerate(idx):
                outs[k] = ov[r:r + 1, :lens_out[r]].copy()   # (the pinned block goes back to torch's host cache)
        return outs

    @torch.no_grad()
    def restore_stream(self, wav, chunk_seconds=30.0, overlap_seconds=1.0, batch_size=8, mode=0,
                       your_vocoder_func=None, on_chunk=None):
        """Long-form restoration with overlap-add (BASELINE config 5; NOT in the reference, whose 30 s segments
        are hard-cut -- ``restore_inmem`` keeps that behaviour): chunks of ``chunk_seconds`` every
        ``chunk_seconds - overlap_seconds``, each restored independently (equal-length chunks are batched),
        consecutive chunks cross-faded linearly over the overlap.  ``on_chunk(start, samples)`` is called with every
        finished stretch of output in order (bounded latency: the first call comes after the first batch).
        ``mode=1``: the high-frequency cut (base.py:121-122) runs per chunk; it returns 512 * (len // 512) samples
        aligned at the chunk's start, so the chunk length is rounded down to a multiple of 512 (every full chunk keeps
        its length) and only the last chunk loses its sub-512 tail -- the output is that much shorter, as the
        reference's mode-1 output is.  Returns float32 numpy (1, N')."""
        self._check_mode(mode)
        pipe = self._get_pipe()
        wav = np.asarray(wav, dtype=np.float32)
        n = wav.shape[0]
        chunk, ov = int(round(chunk_seconds * 44100)), int(round(overlap_seconds * 44100))
        if mode == 1:
            chunk -= chunk % 512
        plan = plan_stream_chunks(n, chunk, ov, 1535 if mode == 1 else 1024)
        out = np.zeros((1, n), np.float32)
        fade_in = (np.arange(ov, dtype=np.float32) / max(ov, 1))[None]
        done = 0  # output is final below this sample
        n_out = n
        i = 0
        while i < len(plan):
            length = plan[i][1]
            grp = [c for c in plan[i:i + batch_size] if c[1] == length]
            seg = torch.from_numpy(np.stack([wav[a:a + length] for a, _ in grp])).to(pipe.device)
            res = pipe.run_checked(lambda: self._restore_segments(pipe, seg, length, mode, your_vocoder_func).cpu().numpy())
            got = res.shape[1]          # == length in mode 0; 512 * (length // 512) in mode 1
            for (a, _), y in zip(grp, res):
                y = y[None]
                if a > 0:  # cross-fade with what the previous chunk left in the overlap
                    out[:, a:a + ov] = out[:, a:a + ov] * (1.0 - fade_in) + y[:, :ov] * fade_in
                    out[:, a + ov:a + got] = y[:, ov:]
                else:
                    out[:, :got] = y
                last = a + length >= n
                if last:
                    n_out = a + got
                final = a + got - ov if not last else n_out
                if on_chunk is not None and final > done:
                    on_chunk(done, out[:, done:final].copy())
                done = max(done, final)
            i += len(grp)
        return out[:, :n_out]

    MIN_SAMPLES = {0: 1025, 1: 1536}   # shortest restorable file: the reflect-padded STFT needs > 1024 samples (mode 1: after the cut to 512 * (n // 512))

    def _restore_batches_isolated(self, items, failed, your_vocoder_func, streams, mode):
        """``restore_batches`` with per-row fault isolation (the folder job's device stage): when a batch raises -- a
        length a kernel refuses, an allocation that does not fit, a plugin vocoder error -- the batches that were in
        flight are re-issued ROW BY ROW, every row that still fails is recorded as ``(tag, reason)`` in ``failed`` and
        the stream of batches continues; the job loses the failing file, nothing else (the reference's serial loop,
        voicefixer/__main__.py:187-212, keeps every file it finished before a bad one)."""
        from collections import deque
        src = iter(items)
        pending = deque()
        src_exc = []           # what the batch SOURCE raised (a generator that has raised is finished: nothing more will come)

        def feed():
            while not src_exc:
                try:
                    it = next(src)
                except StopIteration:
                    return
                except Exception as e:    # noqa: BLE001 -- not a device fault: the batches already issued finish (or are re-issued
                    src_exc.append(e)     # row by row), then the error goes to the caller, who knows which files it never saw
                    return
                pending.append(it)
                yield it

        while True:
            try:
                for tag, out_host, lens_out in self.restore_batches(feed(), your_vocoder_func, streams, mode):
                    pending.popleft()
                    yield tag, out_host, lens_out
                if src_exc:
                    raise BatchSourceError("the batch source failed after %s: %s" % (type(src_exc[0]).__name__, src_exc[0])) from src_exc[0]
                return
            except BatchSourceError:
                raise
            except (KeyboardInterrupt, GeneratorExit):
                raise
            except Exception as exc:    # noqa: BLE001 -- whatever the batch raised costs the rows that raise it again, alone
                bad = list(pending)
                pending.clear()
                if not bad:
                    raise
                first = "%s: %s" % (type(exc).__name__, exc)
                for tag, kind, host, lens in bad:
                    for r in range(len(tag)):
                        one = (tag[r:r + 1], kind, host[r:r + 1, :max(int(lens[r]), 1)], [lens[r]])
                        try:
                            for t1, o1, l1 in self.restore_batches(iter([one]), your_vocoder_func, streams, mode):
                                yield t1, o1, l1
                        except (KeyboardInterrupt, GeneratorExit):
                            raise
                        except Exception as e1:    # noqa: BLE001
                            failed.append((tag[r], "%s: %s" % (type(e1).__name__, e1) if str(e1) else first))

    def restore_folder(self, infolder, outfolder, mode=0, batch_size=32, io_threads=None, your_vocoder_func=None,
                       name_suffix="", extensions=(".wav",), rank=None, world=None, streams=2, ahead=3, stats=None,
                       skip_existing=False):
        """Folder inference (the reference's CLI loop, voicefixer/__main__.py:176-212: every ``*.wav`` of
        ``infolder`` -> same file name in ``outfolder``), batched, pipelined and -- with ``world`` > 1 -- sharded over
        one process per GPU (BASELINE configs[2] and [3]).

        Every rank lists the folder and reads the lengths from the file HEADERS (cheap, no decoding), so all ranks hold
        the same work list without exchanging anything; ``dist.deal_files`` deals the files longest-first to the least
        loaded rank (equal sample totals to within one file, whatever the length distribution -- NOT contiguous blocks of
        the sorted list, which would give one rank all the long files); a rank cuts ITS files, sorted by length, into
        ragged batches (``plan_batches``) and streams them: a thread pool decodes / resamples / down-mixes the files of
        the next ``ahead`` batches straight into pinned staging rows, ``restore_batches`` keeps one batch per HIP stream
        running and one more queued, and the pool encodes each finished batch to PCM16 from the pinned result --
        decode || restore || encode with the device never waiting.  No collective in the data path; every output file is
        written by exactly one rank.  ``rank`` / ``world`` default to the initialised ``torch.distributed`` group (or 0 / 1).

        A bad file costs THAT file (the reference's loop keeps every file it finished before a bad one; a batched, sharded
        job must not do worse): an unreadable header, a file too short to restore (< 1025 samples; mode 1: < 1536), a decode
        error in a worker, a row the device stage refuses (its batch is re-issued row by row) -- each is skipped, recorded
        as ``(file name, reason)`` in ``stats["failed"]`` and the job goes on; the header length only PLANS (staging
        width, dealing): a truncated file is restored at the length the decoder really returned.  Outputs are written
        to a temporary name and renamed, so a file in ``outfolder`` is always complete; ``skip_existing`` leaves files
        whose output already exists alone (resume after an interrupted job; listed in ``stats["skipped"]``).

        ``mode`` 0 or 1 (restore_batch); ``name_suffix`` goes between base name and extension (the CLI's ``-mode<k>``
        naming for ``--mode all``).  ``extensions``: which files of the folder are taken -- the reference's loop takes
        ``.wav`` only (the default, and what the CLI passes); ``(".wav", ".flac")`` adds FLAC inputs, written back as
        FLAC (the workers decode / resample / encode in libvfx_audio.so, several thousand x real time).
        ``io_threads``: decode / encode workers of this rank (default: the host's cores / (2 * world), 2..8).
        ``stats`` (optional dict) receives this rank's counters: files, audio seconds, wall seconds, summed worker
        seconds of decode and encode, seconds the device stage waited for decoded input, ``failed``, ``skipped``.
        Returns the list of file names THIS rank wrote."""
        import threading
        import time
        from collections import deque
        from concurrent.futures import ThreadPoolExecutor
        from . import dist as vdist, flac
        self._check_mode(mode)
        rank, world = vdist.rank_world(rank, world)
        if io_threads is None:
            io_threads = vdist.default_io_threads(world)
        io_threads = max(1, int(io_threads))
        min_len = self.MIN_SAMPLES[mode]
        files = sorted(f for f in os.listdir(infolder) if os.path.splitext(f)[-1] in tuple(extensions))
        os.makedirs(outfolder, exist_ok=True)
        paths = [os.path.join(infolder, f) for f in files]
        names = [("%s%s%s" % (os.path.splitext(f)[0], name_suffix, os.path.splitext(f)[1])) for f in files]
        if any(p.lower().endswith(".flac") for p in paths):
            flac.native()      # load the C codec once, before the workers race for it
        t_start = time.perf_counter()
        lock = threading.Lock()
        cnt = {"decode_s": 0.0, "encode_s": 0.0, "stall_s": 0.0}
        failed = []            # (index, reason): files this rank gave up on
        real_len = {}          # index -> samples the decoder returned
        truncated = []         # (index, header length, decoded length): restored at the decoded length

        def scan(i):
            """Planning length of file i from its header; None + reason when the header is unreadable.  A header that
            promises less than a restorable file (0 in a streamed / interrupted recording) is not believed: the file
            is decoded once to see what is really there."""
            try:
                n, promised = audio_io.wav_length(paths[i], 44100, with_promise=True)
                if promised != n:         # (a header that promises more than the file holds: planned at what is there)
                    with lock:
                        truncated.append((i, promised, n))
                if n < min_len:
                    n = len(audio_io.load_wav(paths[i], 44100))
                return n, None
            except Exception as e:    # noqa: BLE001 -- any unreadable file is this file's problem only
                return None, "%s: %s" % (type(e).__name__, e)

        def decode_into(i, row, n):
            """Worker: file i -> staging row (width n = the header's promise).  Returns the number of samples really
            there (<= n: a longer decode is cut at the staging width), or raises -- the caller drops the row."""
            t0 = time.perf_counter()
            x = audio_io.load_wav(paths[i], 44100)
            m = min(len(x), n)
            row[:m] = x[:m]
            row[m:] = 0.0
            with lock:
                cnt["decode_s"] += time.perf_counter() - t0
                if len(x) != n and not any(t[0] == i for t in truncated):
                    truncated.append((i, n, len(x)))
            return m

        def encode_from(row, i):
            t0 = time.perf_counter()
            final = os.path.join(outfolder, names[i])
            part = os.path.join(outfolder, ".part-%d-%s" % (os.getpid(), names[i]))   # (same extension: save_wave picks the container from it)
            try:
                audio_io.save_wave(row, part, 44100)
                os.replace(part, final)
            except BaseException:
                if os.path.exists(part):
                    os.remove(part)
                raise
            with lock:
                cnt["encode_s"] += time.perf_counter() - t0

        written, skipped, done = [], [], []
        with ThreadPoolExecutor(max_workers=io_threads) as pool:
            scanned = list(pool.map(scan, range(len(files))))
            # every rank must deal from the SAME list: inside an initialised process group of this world size the scans are
            # all-gathered and a file any rank could not read is dropped by all (dist.agree_on_scan); without a group (explicit
            # rank / world) the ranks rely on seeing the same headers.  Each dropped file is REPORTED by one rank
            if world > 1 and vdist.dist.is_available() and vdist.dist.is_initialized() and vdist.dist.get_world_size() == world:
                scanned = vdist.agree_on_scan(scanned)
            usable = []
            for i, (n, why) in enumerate(scanned):
                if why is None and n < min_len:
                    why = "too short to restore: %d samples at 44.1 kHz (mode %d needs >= %d)" % (n, mode, min_len)
                if why is not None:
                    if i % world == rank:
                        failed.append((i, why))
                else:
                    usable.append(i)
            lengths = {i: scanned[i][0] for i in usable}
            owner = vdist.deal_files([lengths[i] for i in usable], world)
            mine = sorted((i for i, o in zip(usable, owner) if o == rank), key=lambda i: (lengths[i], i))
            if skip_existing:
                # applied AFTER the deal and to this rank's own files only: the deal depends on nothing but the input headers,
                # so ranks that look at the output folder at different moments still agree on who owns what
                skipped = [names[i] for i in mine if os.path.exists(os.path.join(outfolder, names[i]))]
                mine = [i for i in mine if not os.path.exists(os.path.join(outfolder, names[i]))]
            plan = plan_batches([lengths[i] for i in mine], batch_size, ragged=your_vocoder_func is None)

            def submit_decode(b):
                kind, grp = plan[b]
                idx = [mine[g] for g in grp]
                lens = [lengths[i] for i in idx]
                try:
                    host = torch.empty((len(idx), max(lens)), dtype=torch.float32, pin_memory=self._pin_memory())
                except Exception as e:    # noqa: BLE001 -- a staging block that cannot be had (host memory, pinning) costs THIS batch
                    for i in idx:
                        failed.append((i, "staging for a batch of %d x %d samples: %s: %s" % (len(idx), max(lens), type(e).__name__, e)))
                    return idx, kind, None, lens, []
                hv = host.numpy()
                return idx, kind, host, lens, [pool.submit(decode_into, i, hv[r], lens[r]) for r, i in enumerate(idx)]

            def decoded():
                queue = [submit_decode(b) for b in range(min(ahead, len(plan)))]
                for b in range(len(plan)):
                    idx, kind, host, lens, futs = queue.pop(0)
                    if b + ahead < len(plan):
                        queue.append(submit_decode(b + ahead))
                    if host is None:          # (its staging block could not be allocated: the files are already in `failed`)
                        continue
                    t0 = time.perf_counter()
                    keep, real = [], []
                    for r, f in enumerate(futs):
                        try:
                            m = f.result()
                            if m < min_len:
                                raise RuntimeError("too short to restore: the header promised %d samples, the decoder "
                                                   "returned %d (mode %d needs >= %d)" % (lens[r], m, mode, min_len))
                            keep.append(r)
                            real.append(m)
                            real_len[idx[r]] = m
                        except Exception as e:    # noqa: BLE001 -- a decode error costs this file only
                            failed.append((idx[r], "%s: %s" % (type(e).__name__, e)))
                    cnt["stall_s"] += time.perf_counter() - t0
                    if not keep:
                        continue
                    if len(keep) < len(idx):          # (rare) drop the failed rows: the batch shrinks, the others go on
                        host = host[keep].contiguous()
                        if self._pin_memory():
                            host = host.pin_memory()
                        idx = [idx[r] for r in keep]
                    if kind == "samples" and min(real) != max(real):
                        # equal-length bucket (several 30 s segments, plugin vocoder) with a truncated member: one batch per row
                        for r in range(len(idx)):
                            yield idx[r:r + 1], kind, host[r:r + 1, :real[r]], real[r:r + 1]
                        continue
                    yield idx, kind, host, real

            writes = deque()           # per batch: the futures of its rows (their views keep the batch's pinned result alive)
            dev_failed = []

            def drain(limit):
                while len(writes) > limit:
                    for i, w in writes.popleft():
                        try:
                            w.result()
                            written.append(names[i])
                            done.append(i)
                        except Exception as e:    # noqa: BLE001 -- a full disk / unwritable name costs this file
                            failed.append((i, "%s: %s" % (type(e).__name__, e)))

            source_error = None
            try:
                for idx, out_host, lens_out in self._restore_batches_isolated(decoded(), dev_failed, your_vocoder_func, streams, mode):
                    ov = out_host.numpy()
                    writes.append([(i, pool.submit(encode_from, ov[r:r + 1, :lens_out[r]], i)) for r, i in enumerate(idx)])
                    drain(ahead + 2)       # bounded backlog: pinned results do not pile up behind a slow disk
            except BatchSourceError as e:  # the source died: every batch it had handed over has been finished and is written below
                source_error = str(e)
            drain(0)
            failed.extend(dev_failed)
            # the books must balance: every file dealt to this rank is written, failed or skipped -- a file nobody accounted
            # for (the batch source died before it got there) is reported as failed, never dropped in silence
            seen = set(done) | {i for i, _ in failed}
            for i in mine:
                if i not in seen:
                    failed.append((i, "not processed: %s" % (source_error or "the device stage ended early")))
        if stats is not None:
            stats.update(rank=rank, world=world, files=len(written), folder_files=len(files), batches=len(plan),
                         audio_s=sum(real_len[i] for i in done) / 44100.0, wall_s=time.perf_counter() - t_start,
                         decode_worker_s=cnt["decode_s"], encode_worker_s=cnt["encode_s"],
                         device_waited_for_decode_s=cnt["stall_s"], io_threads=io_threads,
                         failed=sorted((files[i], why) for i, why in failed), skipped=sorted(skipped),
                         truncated=sorted((files[i], n, m) for i, n, m in truncated if i in real_len or i in mine))
        return sorted(written)

    @staticmethod
    def _pin_memory():
        return torch.cuda.is_available()

    def restore(self, input, output, cuda=False, mode=0, your_vocoder_func=None):
        wav_10k = self._load_wav(input, sample_rate=44100)
        out_np_wav = self.restore_inmem(wav_10k, cuda=cuda, mode=mode, your_vocoder_func=your_vocoder_func)
        audio_io.save_wave(out_np_wav, fname=output, sample_rate=44100)
