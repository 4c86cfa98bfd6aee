#!/usr/bin/env python
"""bench.py -- throughput of the MI355X VoiceFixer restore path (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch 32] [--seconds 10]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path (STFT->mel -> denoiser+ResUNet -> vocoder -> peak/trim)
over one batch of ``--batch`` synthetic ``--seconds``-second 44.1 kHz utterances already resident
in HBM (BASELINE configs[2]: batched folder restore, batch 32 x 10 s, mode 0).  With N > 1 each
rank owns its own batch (utterances shard embarrassingly; no data-path collective): weak scaling,
``value`` = all ranks' audio seconds / max-over-ranks wall time.

The JSON line also carries
  roofline      -- the dominant kernel (one conv_taps_kernel<BM,BL,..,KC> instance): algorithmic
                   FLOPs per launch / average launch duration from HIP events on the launch
                   stream, against the 157.3 TFLOP/s fp32-MFMA peak;
  cpu_baseline  -- the CPU oracle (oracle/oracle.py, a port of the reference path onto the same
                   torch-CPU operators) timed on this box's host cores on ONE utterance (rank 0,
                   N = 1 only).  Test infrastructure used as a reported baseline, never shipped.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md
BF16_MFMA_PEAK_TFLOPS = 2500.0  # dense (no 2:1 sparsity), same guide
SR = 44100


def synth_batch(batch, n, seed, device):
    """Speech-like synthetic input: low-passed noise + 3 harmonic sines, peak < 0.9."""
    g = torch.Generator().manual_seed(seed)
    noise = torch.randn((batch, n + 64), generator=g)
    kern = torch.hann_window(33, periodic=False)
    kern = (kern / kern.sum())[None, None]
    low = torch.nn.functional.conv1d(noise[:, None], kern, padding=16)[:, 0, :n]
    t = torch.arange(n, dtype=torch.float64) / SR
    f0 = 110.0 + 20.0 * torch.arange(batch, dtype=torch.float64)[:, None]
    tone = sum(a * torch.sin(2 * torch.pi * (k * f0) * t[None]) for k, a in ((1, 0.15), (2, 0.08), (3, 0.04)))
    wav = (0.3 * low + tone.float()).float()
    wav = wav / wav.abs().amax(dim=1, keepdim=True) * 0.8
    return wav.to(device).contiguous()


def path_macs(n):
    """Algorithmic MACs of one utterance of n samples."""
    T = 1 + n // 441
    Tp = (T + 63) // 64 * 64
    Tc = T + T % 2 + 4
    return 488784832 * Tc + 92894304 * Tp + 5210112 * T


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--seconds", type=float, default=10.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--streams", type=int, default=1,
                    help="issue consecutive steps (batches) round-robin on this many HIP streams")
    ap.add_argument("--no-events", action="store_true",
                    help="development: time the steps without the per-launch HIP events (no roofline object)")
    ap.add_argument("--no-bf16x3", action="store_true",
                    help="skip the auxiliary timing of the opt-in bf16x3 arithmetic (reported beside the fp32 headline)")
    ap.add_argument("--math", choices=["f32", "bf16x3"], default="f32",
                    help="contraction arithmetic: exact fp32 MFMA (default, the headline) or the opt-in split-bf16 "
                         "products with fp32 accumulation (DESIGN.md 3.4)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (the product path has no CPU fallback)")
    torch.cuda.set_device(local_rank % torch.cuda.device_count())
    dev = torch.device("cuda", torch.cuda.current_device())
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=dev)  # nccl == RCCL on ROCm

    from voicefixer_amd import engine, ops, weights

    n = int(round(args.seconds * SR))
    pipe = engine.Pipeline(weights.seeded_vocoder_state(1234), weights.seeded_restorer_state(4321), dev, args.math)
    wav = synth_batch(args.batch, n, 1000 + rank, dev)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    streams = [torch.cuda.Stream(device=dev) for _ in range(max(1, args.streams))]

    def run_steps(k):
        last = None
        for i in range(k):
            if len(streams) == 1:
                last = pipe.restore(wav, n)
            else:  # consecutive batches on alternating streams: one batch's GRU overlaps the other's convolutions
                with torch.cuda.stream(streams[i % len(streams)]):
                    last = pipe.restore(wav, n)
        return last

    out = run_steps(args.warmup)
    barrier()
    if not args.no_events:
        # pre-created timing events (creating one costs ~10 us of host time: visible in a launch-bound batch-1 run)
        ops.EVENT_POOL = [torch.cuda.Event(enable_timing=True) for _ in range(800 * max(args.steps, 1))]
        for e in ops.EVENT_POOL[:8]:
            e.record()  # first use of an event allocates its backing object
        torch.cuda.synchronize()
    ops.PROFILE = None if args.no_events else []
    t0 = time.perf_counter()
    out = run_steps(args.steps)
    barrier()
    dt = time.perf_counter() - t0
    prof, ops.PROFILE = ops.PROFILE, None
    ops.EVENT_POOL = None
    if prof is None:
        if rank == 0:
            print(json.dumps({"ms_per_step": round(dt / args.steps * 1e3, 3), "events": False}), flush=True)
        return
    assert torch.isfinite(out).all()

    if dist is not None:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    # ---- roofline bookkeeping for the dominant conv_taps_kernel instance (this rank) ----
    by_tile = {}
    stft_bytes, stft_secs, stft_n = 0, 0.0, 0
    for tile, macs, e0, e1 in prof:
        if tile == -1:  # the STFT->mel front-end: `macs` carries its algorithmic bytes
            stft_bytes += macs
            stft_secs += e0.elapsed_time(e1) * 1e-3
            stft_n += 1
            continue
        d = by_tile.setdefault(tile, [0, 0, 0.0])
        d[0] += 1
        d[1] += macs
        d[2] += e0.elapsed_time(e1) * 1e-3
    conv_time = sum(d[2] for d in by_tile.values())
    conv_macs = sum(d[1] for d in by_tile.values())
    dom = max(by_tile.items(), key=lambda kv: kv[1][2])
    tile, (launches, macs, secs) = dom
    achieved = 2.0 * macs / secs / 1e12
    # HBM bytes per launch of that kernel: PMC counters cannot be read live; they come from the committed
    # rocprofv3 --pmc passes over this same command (profiles/r01_pmc_hbm_traffic_bench_b32.json)
    traffic = None
    tfile = os.path.join(ROOT, "profiles", "r01_pmc_hbm_traffic_bench_b32.json")
    if os.path.exists(tfile) and args.batch == 32 and abs(args.seconds - 10.0) < 1e-9:
        # the instance family <BM, BL, WGM, WGL, KC, interior, *>: launch-weighted mean over its staging-slot variants
        want = "conv_taps_kernel<%d, %d," % (tile // 100000, tile // 100 % 1000)
        num = den = 0
        for kname, rec in json.load(open(tfile))["kernels"].items():
            if want in kname and ", %d, true" % (tile % 100) in kname:
                num += rec["hbm_bytes_per_launch"] * rec["launches"]
                den += rec["launches"]
        traffic = int(num / den) if den else None
    x3_dom = args.math == "bf16x3" and tile % 100 == 16
    # bf16x3 instance: three bf16 MFMA products per algorithmic product -> peak = dense bf16 peak / 3
    peak = BF16_MFMA_PEAK_TFLOPS / 3.0 if x3_dom else FP32_MFMA_PEAK_TFLOPS
    if x3_dom:
        traffic = None
    roofline = {
        "bound": "mfma", "achieved": round(achieved, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
        "frac": round(achieved / peak, 4), "traffic": traffic,
        "kernel": ("conv_x3_kernel<BM=%d,BL=%d,KC=%d,*>" if x3_dom else "conv_taps_kernel<BM=%d,BL=%d,KC=%d,FAST,*>")
                  % (tile // 100000, tile // 100 % 1000, tile % 100),
        "launches_per_step": launches // args.steps,
        "avg_launch_ms": round(secs / launches * 1e3, 4),
        "algorithmic_gflop_per_launch": round(2.0 * macs / launches / 1e9, 3),
        "all_conv_kernels": {"achieved": round(2.0 * conv_macs / conv_time / 1e12, 2),
                             "time_share_of_step": round(conv_time / dt, 4) if world == 1 else None},
    }

    if stft_n:
        # reported for completeness: the front-end moves the algorithmic minimum of bytes
        # but is bound by its in-LDS FFT, not by HBM
        roofline["stft_mel_kernel"] = {"bound": "hbm", "achieved": round(stft_bytes / stft_secs / 1e9, 1),
                                       "peak": 8000.0, "unit": "GB/s",
                                       "frac": round(stft_bytes / stft_secs / 8.0e12, 4),
                                       "avg_launch_ms": round(stft_secs / stft_n * 1e3, 4)}
    audio_seconds = world * args.batch * args.seconds * args.steps
    value = audio_seconds / dt
    line = {
        "metric": "seconds-of-44.1kHz-audio restored per wall-second",
        "value": round(value, 2), "unit": "x real-time", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": args.math, "data": "synthetic",
        "config": {"workload": "batched folder restore (BASELINE configs[2]): one batch of %d x %.0f s 44.1 kHz "
                               "utterances per step, VoiceFixer.restore mode 0, seeded random weights"
                               % (args.batch, args.seconds),
                   "batch_per_gpu": args.batch, "utterance_seconds": args.seconds, "frames": 1 + n // 441,
                   "parallelism": "utterance sharding x%d (no data-path collective)" % world},
        "path_tflops": round(2.0 * path_macs(n) * args.batch * world * args.steps / dt / 1e12, 2),
        "roofline": roofline,
    }

    if world == 1 and args.math == "f32" and not args.no_bf16x3:
        # auxiliary, NOT the headline: the same workload with the opt-in split-bf16 contraction (three bf16 MFMA
        # products per fp32 product, fp32 accumulation), and its waveform distance from the fp32 run
        pipe.set_math("bf16x3")
        run_steps(2)  # (packs the bf16 weight planes, creates the tables / workspaces of the new launch shapes)
        barrier()
        t1 = time.perf_counter()
        out3 = run_steps(args.steps)
        barrier()
        dt3 = time.perf_counter() - t1
        pipe.set_math("f32")
        line["bf16x3_optin"] = {
            "value": round(args.batch * args.seconds * args.steps / dt3, 2), "unit": "x real-time",
            "ms_per_step": round(dt3 / args.steps * 1e3, 3),
            "rms_vs_f32": float(torch.sqrt(torch.mean((out3 - out) ** 2))),
            "note": "opt-in VoiceFixer.set_math('bf16x3'); parity bound 1e-3 RMS"}

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import oracle  # checker/baseline only
        vsd, rsd = weights.seeded_vocoder_state(1234), weights.seeded_restorer_state(4321)
        threads = torch.get_num_threads()
        w1 = wav[0].cpu().numpy()
        with torch.no_grad():
            oracle.restore_inmem(w1[: 2 * SR], vsd, rsd)  # thread-pool / allocator warm-up (2 s)
            c0 = time.perf_counter()
            ref = oracle.restore_inmem(w1, vsd, rsd)
            cdt = time.perf_counter() - c0
        err = float(torch.sqrt(torch.mean((out[0].cpu() - torch.from_numpy(ref[0])) ** 2)))
        line["cpu_baseline"] = {"value": round(args.seconds / cdt, 3), "unit": "x real-time", "cores": threads,
                                "kind": "port",
                                "sample": "1 utterance of %.0f s (utterance 0 of the batch), B=1 sequential like "
                                          "voicefixer/__main__.py:187-212; %.1f s of CPU time" % (args.seconds, cdt),
                                "rms_vs_gpu": err}
    if rank == 0:
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
