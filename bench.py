#!/usr/bin/env python
"""bench.py -- throughput of the MI355X VoiceFixer restore path (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch 32] [--seconds 10]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Launched PLAIN with ``--gpus N`` (N > 1, no WORLD_SIZE in the environment) the script starts the ranks itself: it
re-executes under ``torch.distributed.run --nnodes=1 --nproc-per-node min(N, visible devices)`` on 127.0.0.1 (one
process per GPU, RCCL), says so loudly on stderr when fewer than N devices are visible, and the JSON line carries
``n_gpus`` = ranks that really ran, ``requested_gpus``, ``visible_devices``, ``rccl`` and the device of every rank.

One "step" = one pass of the hot path (STFT->mel -> denoiser+ResUNet -> vocoder -> peak/trim)
over one batch of ``--batch`` synthetic ``--seconds``-second 44.1 kHz utterances already resident
in HBM (BASELINE configs[2]: batched folder restore, batch 32 x 10 s, mode 0); consecutive steps
take DIFFERENT batches (a ring of three resident batches).  With N > 1 each rank owns its own
batches (utterances shard embarrassingly; no data-path collective): weak scaling,
``value`` = all ranks' audio seconds / max-over-ranks wall time.

``host_to_host`` (same JSON line) is the metric as SURVEY.md 8(d) defines it: pinned host waveforms in,
pinned host waveforms out, H2D and D2H on copy streams, the compute alternating over the same HIP streams as the timed
region, everything inside the timed region.  The host runs up to THREE steps ahead of the oldest unfinished D2H (three slots of
pinned input / output and device input, what ``VoiceFixer.restore_batches`` keeps in flight: one batch per stream running and one
queued) and never synchronises on the issue path, so two batches overlap on the device as they do in the resident loop.  The bench
contract defines ``value`` with the inputs already resident in HBM ("the PCIe-inclusive rate ... is never value"); the host-to-host
figure therefore sits beside it as ``value_host_to_host``.

``--scatter`` (BASELINE configs[3], launched under torch.distributed.run): rank 0 owns 256 x N utterances,
``dist.restore_sharded`` scatters them over RCCL (backend "nccl"), every rank restores its 256 in batches of
``--batch``, rank 0 gathers; the JSON line then reports the whole job (scatter + compute + gather).

The JSON line also carries
  roofline      -- the dominant kernel FAMILY (what one regex over rocprofv3's kernel names selects; at batch 32 the
                   Winograd F(4,3) ResStack convolutions convwg4[p]_kernel<4,1,...>): executed FLOPs per launch / average
                   launch duration from HIP events on the launch stream (single-stream leg: un-overlapped durations),
                   against the 157.3 TFLOP/s fp32-MFMA peak at 2.4 GHz; ``traffic`` from the committed PMC passes;
  clocks        -- shader clock and socket power sampled (sysfs hwmon, else rocm-smi) DURING the timed region and during the
                   single-stream leg: the dominant kernels run at the 1400 W power cap and the clock the firmware grants under it
                   differs per chip (2.07-2.26 GHz seen), so a round-over-round delta below ~6 % is only readable next to these;
  cpu_baseline  -- the CPU oracle (oracle/oracle.py, a port of the reference path onto the same
                   torch-CPU operators) timed on this box's host cores on ONE utterance (rank 0,
                   N = 1 only); ``reference_over_port`` = the measured time ratio of the reference's own modules to the port
                   (tools/cpu_reference_vs_port.py, build container, where /root/reference exists).  Test infrastructure
                   used as a reported baseline, never shipped.
With N > 1 ranks the line also carries ``one_rank_leg`` (rank 0 alone, the other ranks idle at a barrier, same K steps) and
``scaling_efficiency_self_measured`` = value / (N x that leg's value): a convenience for reading a log, the driver computes its own.
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
TRAFFIC_PROFILE = "r06_pmc_hbm_traffic_bench_b32.json"  # tools/profile_round.sh -> tools/pmc_summary.py
REFERENCE_VS_PORT = "r06_cpu_reference_vs_port.json"     # tools/cpu_reference_vs_port.py (build container)


class ClockSampler:
    """Shader clock (MHz) and socket power (W) of one device, sampled by a host thread while a timed region runs.  sysfs hwmon
    (freq1_input = sclk in Hz, power1_input / power1_average in microwatts) when the amdgpu driver exposes it, else
    ``rocm-smi --showpower --showclocks`` (slower: one sample per ~0.5 s).  Reporting only -- nothing is throttled or set."""

    def __init__(self, dev_index=0, period=0.05):
        import glob
        import threading
        self.period, self.samples, self.source = period, [], None
        self._stop, self._thread = threading.Event(), None
        self._freq = self._power = None
        cards = sorted(glob.glob("/sys/class/drm/card[0-9]*/device/hwmon/hwmon*"))
        cards = [c for c in cards if os.path.exists(os.path.join(c, "freq1_input"))]
        # the HIP device's sysfs card by PCI address: a box may show more cards in sysfs than the process may use (a one-GPU lease
        # of an eight-GPU node), and HIP index 0 is not card0 then
        want = None
        try:
            pr = torch.cuda.get_device_properties(dev_index)
            want = "%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
        except (AttributeError, RuntimeError, AssertionError):
            pass
        match = [c for c in cards if want and os.path.basename(os.path.realpath(os.path.join(c, "..", ".."))) == want]
        if match or (cards and len(cards) == 1):
            h = (match or cards)[0]
            self._freq = os.path.join(h, "freq1_input")
            for nm in ("power1_input", "power1_average"):
                if os.path.exists(os.path.join(h, nm)):
                    try:
                        int(open(os.path.join(h, nm)).read())
                        self._power = os.path.join(h, nm)
                        break
                    except (OSError, ValueError):
                        pass
            self.source = "sysfs %s (freq1_input, %s)" % (h, os.path.basename(self._power) if self._power else "no power file")
        self._smi = None
        if self._freq is None or self._power is None:
            import shutil
            self._smi = shutil.which("rocm-smi") or ("/opt/rocm/bin/rocm-smi" if os.path.exists("/opt/rocm/bin/rocm-smi") else None)
            if self._smi:
                self.source = (self.source + " + " if self.source else "") + "rocm-smi --showpower --showclocks -d %d" % dev_index
                self.period = max(period, 0.4)
        self.dev_index = dev_index

    def _read(self):
        sclk = watts = None
        try:
            if self._freq:
                sclk = int(open(self._freq).read()) / 1e6
            if self._power:
                watts = int(open(self._power).read()) / 1e6
        except (OSError, ValueError):
            pass
        if (sclk is None or watts is None) and self._smi:
            import re
            import subprocess
            try:
                out = subprocess.run([self._smi, "--showpower", "--showclocks", "-d", str(self.dev_index)],
                                     capture_output=True, text=True, timeout=5).stdout
                m = re.search(r"sclk clock level:?\s*\d*:?\s*\(?(\d+)Mhz", out)
                if sclk is None and m:
                    sclk = float(m.group(1))
                m = re.search(r"Power \(W\):\s*([0-9.]+)", out)
                if watts is None and m:
                    watts = float(m.group(1))
            except (OSError, subprocess.SubprocessError):
                pass
        return sclk, watts

    def __enter__(self):
        import threading
        if self.source is None:
            return self
        self.samples = []

        def loop():
            while not self._stop.is_set():
                self.samples.append((time.perf_counter(),) + self._read())
                self._stop.wait(self.period)

        self._stop.clear()
        self._thread = threading.Thread(target=loop, daemon=True)
        self._thread.start()
        return self

    def __exit__(self, *exc):
        if self._thread is not None:
            self._stop.set()
            self._thread.join(10)
            self._thread = None
        return False

    def summary(self, t0=None, t1=None):
        """{"sclk_mhz": {mean, min, max, n}, "socket_power_w": {...}, "source"} over the samples taken inside [t0, t1]."""
        if self.source is None:
            return {"source": None, "note": "neither sysfs hwmon nor rocm-smi available"}
        rows = [r for r in self.samples if (t0 is None or r[0] >= t0) and (t1 is None or r[0] <= t1)]

        def stat(vals):
            vals = [v for v in vals if v is not None]
            if not vals:
                return None
            return {"mean": round(sum(vals) / len(vals), 1), "min": round(min(vals), 1), "max": round(max(vals), 1), "n": len(vals)}

        return {"sclk_mhz": stat([r[1] for r in rows]), "socket_power_w": stat([r[2] for r in rows]), "source": self.source}


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


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.startswith("model name"):
                    return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(args, w1, gpu_out):
    """The CPU oracle (a port of the reference path onto the same torch-CPU operators; the reference itself cannot
    travel to the GPU box) on ONE utterance of the batch, B = 1 like voicefixer/__main__.py:187-212.
    ATen's default thread count on a 2 x 64-core host oversubscribes this small-operator path, so the thread count is
    TUNED first (a sweep on a 2 s clip of the same utterance) and the reported ``value`` is the median of
    ``--cpu-reps`` repetitions of the full utterance at the best count; the all-threads figure (what
    ``torch.set_num_threads(os.cpu_count())`` style defaults give) is reported beside it."""
    from oracle import oracle  # checker / reported baseline only -- never on the product path
    from voicefixer_amd import weights
    vsd, rsd = weights.seeded_vocoder_state(1234), weights.seeded_restorer_state(4321)
    default_threads = torch.get_num_threads()
    clip = w1[: 2 * SR]

    def timed(x, threads):
        torch.set_num_threads(threads)
        c0 = time.perf_counter()
        y = oracle.restore_inmem(x, vsd, rsd)
        return time.perf_counter() - c0, y

    with torch.no_grad():
        timed(clip, default_threads)  # thread-pool / allocator warm-up
        sweep = {}
        for th in sorted({t for t in (4, 8, 16, 32, 64, default_threads) if t <= max(default_threads, 8)}):
            timed(clip, th)
            sweep[th] = round(timed(clip, th)[0], 3)
        best = min(sweep, key=sweep.get)
        times = []
        for _ in range(max(1, args.cpu_reps)):
            dt, ref = timed(w1, best)
            times.append(dt)
        all_threads_s = timed(w1, default_threads)[0] if best != default_threads else None
        torch.set_num_threads(default_threads)
    med = sorted(times)[len(times) // 2]
    err = float(torch.sqrt(torch.mean((gpu_out - torch.from_numpy(ref[0])) ** 2)))
    out = {"value": round(args.seconds / med, 3), "unit": "x real-time", "cores": best, "kind": "port",
           "cpu_model": cpu_model(), "os_cpu_count": os.cpu_count(), "torch": torch.__version__,
           "repetitions_s": [round(t, 2) for t in times],
           "thread_sweep_2s_clip_s": {str(k): v for k, v in sweep.items()},
           "sample": "1 utterance of %.0f s (one utterance of the last timed batch), B=1 sequential like "
                     "voicefixer/__main__.py:187-212, at the best thread count of a sweep over a 2 s clip; median of "
                     "%d repetitions; %.1f s of CPU time for them" % (args.seconds, len(times), sum(times)),
           "rms_vs_gpu": err}
    # kind = "port": /root/reference cannot travel to the GPU box.  The measured equivalence behind the label: the reference's OWN
    # modules against this port on the same cores (tools/cpu_reference_vs_port.py, taken in the build container this round)
    rp = os.path.join(ROOT, "profiles", REFERENCE_VS_PORT)
    if os.path.exists(rp):
        try:
            doc = json.load(open(rp))
            out["reference_over_port"] = {
                "time_ratio": round(1.0 / doc["port_over_reference_time"], 4),
                "reference_x_real_time_estimate": round(out["value"] * doc["port_over_reference_time"], 3),
                "measured_on": "%s, %d threads (build container, profiles/%s)" % (doc["cpu_model"], doc["threads"], REFERENCE_VS_PORT),
                "rms_port_vs_reference": doc["rms_port_vs_reference"],
                "note": "reference time / port time for one 10 s utterance on the same cores; > 1 = the port is the faster, i.e. the more "
                        "demanding, baseline (it skips the dead UpsampleNet.skip_conv and uses torch.stft)"}
        except (OSError, ValueError, KeyError, ZeroDivisionError):
            pass
    if all_threads_s is not None:
        out["all_threads"] = {"value": round(args.seconds / all_threads_s, 3), "cores": default_threads,
                              "seconds": round(all_threads_s, 2)}
    return out


def host_to_host_leg(vf, args, n, dev, n_streams, stream_pool=None):
    """SURVEY.md 8(d) / BASELINE.md 4.6: host-resident float32 waveforms -> host-resident float32 waveforms, through the PRODUCT's
    own device stage: ``VoiceFixer.restore_batches`` (api.py) fed with pinned (B, n) batches, a different batch every step.  Per
    batch it queues H2D, the ~600 launches and the D2H into a pinned result ON THE BATCH'S COMPUTE STREAM (no copy streams: with
    five streams on the runtime's four hardware queues a copy stream's event wait shares a queue with -- and stalls -- a compute
    stream, which is how round 5's hand-rolled leg lost the two-stream gain), batches alternate over ``n_streams`` HIP streams
    and the host runs ``n_streams + 1`` batches ahead.  Everything inside the timed region; file decode / encode excluded (the
    disk-to-disk figure is ``--synth-folder``)."""
    B, steps = args.batch, max(args.steps, 2)
    NS = n_streams + 2
    if stream_pool:        # the HIP streams of the timed region (a process that keeps creating streams ends up with several of them on
        vf._stream_pool = list(stream_pool)     # one of the runtime's four hardware queues, where they serialise)
    host_in = [synth_batch(B, n, 5000 + 31 * k, "cpu").pin_memory() for k in range(NS)]
    lens = [n] * B

    def feed(k):
        for i in range(k):
            yield i, "ragged", host_in[i % NS], lens

    def run(k):
        seen, last = 0, None
        for tag, out_host, lens_out in vf.restore_batches(feed(k), streams=n_streams):
            assert tag == seen and out_host.shape == (B, n)
            seen, last = seen + 1, out_host
        assert seen == k
        return last

    run(NS)
    t0 = time.perf_counter()
    last = run(steps)
    dt = time.perf_counter() - t0
    assert torch.isfinite(last).all() and float(last.abs().max()) > 1e-3
    return {"value": round(B * args.seconds * steps / dt, 2), "unit": "x real-time",
            "ms_per_step": round(dt / steps * 1e3, 3), "steps": steps, "compute_streams": n_streams, "batches_in_flight": n_streams + 1,
            "pcie_bytes_per_step": 2 * B * n * 4,
            "note": "pinned host waveform -> pinned host waveform through VoiceFixer.restore_batches (the folder job's device stage: H2D, "
                    "launches and D2H of a batch on its own compute stream, %d streams, the host %d batches ahead), a different batch per "
                    "step, the generator's start-up and drain inside the timed region (SURVEY.md 8(d)); file decode/encode excluded"
                    % (n_streams, n_streams + 1)}


def scatter_job(args, pipe, n, rank, world, dev, dist):
    """BASELINE configs[3]: rank 0 owns ``utterances_per_gpu * world`` utterances (device-resident on rank 0),
    scatter over RCCL -> every rank restores its block in batches of ``--batch`` -> gather on rank 0."""
    from voicefixer_amd import dist as vdist
    per = args.utterances_per_gpu
    n_utt = per * world
    # rank 0 builds the job in chunks of 32 (the synthetic generator is CPU torch)
    src = None
    if rank == 0:
        src = torch.cat([synth_batch(min(32, n_utt - i), n, 2000 + i, dev) for i in range(0, n_utt, 32)], 0)
    timing = {}
    # warm-up: one batch through the path on every rank (tables, workspaces, allocator)
    pipe.restore(synth_batch(min(args.batch, per), n, 77 + rank, dev), n)
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = vdist.restore_sharded(lambda w: pipe.restore(w, n), src, n, dev, batch_size=args.batch, timing=timing)
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    pipe.check()
    t = torch.tensor([dt, timing.get("scatter_s", 0.0), timing.get("compute_s", 0.0), timing.get("gather_s", 0.0)],
                     device=dev, dtype=torch.float64)
    allt = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(allt, t)
    if rank == 0:
        assert out.shape == (n_utt, n) and torch.isfinite(out).all()
        dmax = max(float(x[0]) for x in allt)
        line = {
            "metric": "seconds-of-44.1kHz-audio restored per wall-second",
            "value": round(n_utt * args.seconds / dmax, 2), "unit": "x real-time", "n_gpus": world,
            "steps": (per + args.batch - 1) // args.batch, "warmup": 1,
            "ms_per_step": round(dmax * 1e3 / ((per + args.batch - 1) // args.batch), 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.math, "data": "synthetic",
            "config": {"workload": "batched folder restore sharded via RCCL scatter/gather (BASELINE configs[3]): rank 0 "
                                   "owns %d x %.0f s utterances, %d per GPU in batches of %d, mode 0, seeded random weights"
                                   % (n_utt, args.seconds, per, args.batch),
                       "batch_per_gpu": args.batch, "utterances": n_utt, "utterance_seconds": args.seconds,
                       "parallelism": "utterance sharding x%d, RCCL point-to-point scatter/gather only" % world},
            "requested_gpus": int(os.environ.get("VFX_BENCH_REQUESTED_GPUS", args.gpus)),
            "visible_devices": torch.cuda.device_count(),
            "rccl": {"backend": dist.get_backend(), "world_size": dist.get_world_size()},
            "per_rank": [{"rank": r, "total_s": round(float(x[0]), 4), "scatter_s": round(float(x[1]), 4),
                          "compute_s": round(float(x[2]), 4), "gather_s": round(float(x[3]), 4)}
                         for r, x in enumerate(allt)],
            "path_tflops": round(2.0 * path_macs(n) * n_utt / dmax / 1e12, 2),
        }
        _json_line_last(line)
    dist.destroy_process_group()


def folder_job(args, rank, world, dev, dist, dry=False):
    """BASELINE configs[2] / [3] as the PRODUCT runs them, disk to disk: a folder of ``--synth-folder`` x world synthetic
    utterances (PCM16 WAV on tmpfs; or ``--folder DIR``) goes through ``VoiceFixer.restore_folder`` -- every rank lists
    the folder, takes the files dist.deal_files deals it, decodes / restores / encodes them (ragged batches of
    ``--batch``) and writes its outputs; no data-path collective, one all-gather of per-rank counters.  Timed from the
    folder on disk to the last output file closed, barrier + max over ranks; beside it, in the same process, the
    HBM-resident rate of the headline bench (``hbm_resident``) so that the two can be divided.
    ``dry`` (``--dry-run``, tests/test_bench_launcher.py): the same job on CPU with the DEVICE STAGE replaced by the identity and gloo
    instead of RCCL -- folder preparation by rank 0, the barriers, the deal, decode / encode workers, the all-gather of the counters and
    the JSON line are the real code."""
    import shutil
    import numpy as np
    from voicefixer_amd import weights, audio_io, dist as vdist
    from voicefixer_amd.api import VoiceFixer
    n = int(round(args.seconds * SR))
    tag = os.environ.get("MASTER_PORT", str(os.getpid()))
    base = os.path.join("/dev/shm" if os.path.isdir("/dev/shm") else "/tmp", "vfx_bench_folder_%s" % tag)
    ind, outd, warm_in, warm_out = (os.path.join(base, d) for d in ("in", "out", "warm_in", "warm_out"))
    synthetic = not args.folder
    if rank == 0:
        shutil.rmtree(base, ignore_errors=True)
        os.makedirs(warm_in)
        t0 = time.perf_counter()
        if synthetic:
            os.makedirs(ind)
            n_files = args.synth_folder * world
            for i0 in range(0, n_files, 32):
                w = synth_batch(min(32, n_files - i0), n, 3000 + i0, "cpu").numpy()
                for r in range(w.shape[0]):
                    audio_io.save_wave(w[r:r + 1], os.path.join(ind, "utt%05d.wav" % (i0 + r)))
        w = synth_batch(min(args.batch, 32), n, 99, "cpu").numpy()      # warm-up folder: one batch per rank
        for r in range(w.shape[0] * world):
            audio_io.save_wave(w[r % w.shape[0]:r % w.shape[0] + 1], os.path.join(warm_in, "w%04d.wav" % r))
        print("bench.py: folder prepared in %.1f s under %s" % (time.perf_counter() - t0, base), file=sys.stderr, flush=True)
    if args.folder:
        ind = args.folder

    def barrier():
        if not dry:
            torch.cuda.synchronize()
        if dist is not None and dist.is_initialized():
            dist.barrier()
            if not dry:
                torch.cuda.synchronize()

    if dry:
        class _StubDevice(VoiceFixer):        # test hook: no checkpoints, no device -- the device stage hands the rows back
            def __init__(self):
                pass

            def restore_batches(self, batches, your_vocoder_func=None, streams=2, mode=0):
                for tag, kind, host, lens in batches:
                    yield tag, host.clone(), list(lens)

        vf, pipe = _StubDevice(), None
    else:
        vf = VoiceFixer.from_state(weights.seeded_vocoder_state(1234), weights.seeded_restorer_state(4321))
        pipe = vf._get_pipe()
    def stamp(what, t):
        print("bench.py: [%s] rank %d: %s %.1f s" % (time.strftime("%H:%M:%S"), rank, what, time.perf_counter() - t), file=sys.stderr, flush=True)

    t_ = time.perf_counter()
    barrier()      # (rank 0 has written the folders)
    stamp("pipeline built, folders ready after", t_)
    ext = (".wav", ".flac") if args.folder else (".wav",)
    t_ = time.perf_counter()
    vf.restore_folder(warm_in, warm_out, batch_size=args.batch, io_threads=args.io_threads or None, rank=rank, world=world, extensions=ext)
    barrier()
    stamp("warm-up folder", t_)
    st = {}
    t0 = time.perf_counter()
    vf.restore_folder(ind, outd, batch_size=args.batch, io_threads=args.io_threads or None, rank=rank, world=world, stats=st,
                      streams=args.folder_streams, extensions=ext)
    barrier()
    dt = time.perf_counter() - t0
    # the HBM-resident rate of the same build in the same process (bench.py's headline loop: consecutive batches alternating over
    # the same number of streams as the folder job's device stage, no per-launch events)
    hbm_ms = float("nan")
    if not dry:
        x = synth_batch(args.batch, n, 1000 + rank, dev)
        pool = vf._streams(args.folder_streams)
        pipe.set_streams(len(pool))
        for st_ in pool:
            st_.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(st_):
                pipe.restore(x, n)
        torch.cuda.synchronize()
        nrep = max(2 * len(pool), args.steps)
        t1 = time.perf_counter()
        for i in range(nrep):
            with torch.cuda.stream(pool[i % len(pool)]):
                pipe.restore(x, n)
        torch.cuda.synchronize()
        hbm_ms = (time.perf_counter() - t1) / nrep * 1e3
        pipe.set_streams(1)
        pipe.check()
    keys = ["files", "audio_s", "wall_s", "decode_worker_s", "encode_worker_s", "device_waited_for_decode_s", "batches"]
    allr = vdist.gather_counters([st[k] for k in keys] + [dt, 0.0 if dry else hbm_ms, float(dev.index) if dev is not None else -1.0],
                                 dev if (dist is not None and not dry and dist.is_initialized() and dist.get_backend() == "nccl") else None)
    try:
        my_cores = sorted(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        my_cores = []
    slices = vdist.gather_objects({"cores": _ranges(my_cores), "n_cores": len(my_cores), "n_physical": len(vdist.physical_cores(my_cores)),
                                   "pinned": vdist._PINNED is not None, "io_threads": st.get("io_threads"),
                                   "torch_threads": torch.get_num_threads()})
    if rank == 0:
        outs = sorted(os.listdir(outd))
        assert len(outs) == int(sum(x[0] for x in allr)) == st["folder_files"], (len(outs), st["folder_files"])
        y = audio_io.load_wav(os.path.join(outd, outs[-1]))
        assert np.isfinite(y).all() and np.abs(y).max() > 1e-3
        dmax = max(x[7] for x in allr)
        audio = sum(x[1] for x in allr)
        hbm_value = None if dry else world * args.batch * args.seconds / (max(x[8] for x in allr) * 1e-3)
        line = {
            "metric": "seconds-of-44.1kHz-audio restored per wall-second",
            # n_gpus counts DISTINCT devices (an --oversubscribe rehearsal puts several ranks on one); `ranks` = processes that ran
            "value": round(audio / dmax, 2), "unit": "x real-time", "n_gpus": world if dry else len({int(x[9]) for x in allr}), "ranks": world,
            "steps": int(max(x[6] for x in allr)), "warmup": 1, "ms_per_step": round(dmax * 1e3 / max(1, max(x[6] for x in allr)), 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.math, "data": "synthetic" if synthetic else "folder",
            "value_definition": "DISK TO DISK: from the folder of PCM16 files on tmpfs to the last restored file written "
                                "(decode, H2D, restore, D2H, int16 encode all inside the timed region); hbm_resident = the "
                                "headline bench loop (inputs resident in HBM) in the same process, for the ratio",
            "config": {"workload": "batched folder restore, disk to disk (BASELINE configs[%d]): %d files of %.0f s, %d per GPU, "
                                   "ragged batches of %d, VoiceFixer.restore_folder mode 0, seeded random weights"
                                   % (3 if world > 1 else 2, int(sum(x[0] for x in allr)), args.seconds, int(allr[0][0]), args.batch),
                       "batch_per_gpu": args.batch, "utterance_seconds": args.seconds, "io_threads": st.get("io_threads", args.io_threads),
                       "streams": args.folder_streams, "host_cores": os.cpu_count(), "folder": base if synthetic else args.folder,
                       "parallelism": "files dealt to %d rank(s) by dist.deal_files (no data-path collective; one all-gather of counters)" % world,
                       **({"oversubscribed": "%d ranks share %d device(s): a rehearsal of the launch, the deal and the per-rank I/O, NOT a scaling "
                                             "measurement" % (world, len({int(x[9]) for x in allr}))}
                          if (not dry and len({int(x[9]) for x in allr}) < world) else {})},
            "hbm_resident": None if dry else {"value": round(hbm_value, 2), "ms_per_step": round(max(x[8] for x in allr), 3)},
            "disk_to_disk_over_hbm_resident": None if dry else round(audio / dmax / hbm_value, 4),
            "decode_worker_s": round(sum(x[3] for x in allr), 3), "encode_worker_s": round(sum(x[4] for x in allr), 3),
            "decode_x_realtime_per_thread": round(audio / max(sum(x[3] for x in allr), 1e-9), 1),
            "encode_x_realtime_per_thread": round(audio / max(sum(x[4] for x in allr), 1e-9), 1),
            "requested_gpus": int(os.environ.get("VFX_BENCH_REQUESTED_GPUS", args.gpus)), "visible_devices": torch.cuda.device_count(),
            "rccl": {"backend": dist.get_backend() if (dist is not None and dist.is_initialized()) else None, "world_size": world},
            **({"dry_run": True} if dry else {}),
            "per_rank": [{"rank": r, "device": "dry" if dry else "cuda:%d" % int(x[9]), "files": int(x[0]), "audio_s": round(x[1], 1), "batches": int(x[6]),
                          "wall_s": round(x[7], 4), "folder_s": round(x[2], 4), "decode_worker_s": round(x[3], 3),
                          "encode_worker_s": round(x[4], 3), "device_waited_for_decode_s": round(x[5], 4),
                          "cpu": slices[r] if r < len(slices) else None} for r, x in enumerate(allr)],
            "lib_build_id": None if dry else _lib_build_id(),
        }
        _json_line_last(line)
        shutil.rmtree(base, ignore_errors=True)
    if dist is not None and dist.is_initialized():
        dist.destroy_process_group()


def _ranges(ids):
    """[0, 1, 2, 3, 128, 129] -> "0-3,128-129" (a rank's logical CPUs: its physical cores and their SMT siblings)."""
    out, i = [], 0
    while i < len(ids):
        j = i
        while j + 1 < len(ids) and ids[j + 1] == ids[j] + 1:
            j += 1
        out.append(str(ids[i]) if i == j else "%d-%d" % (ids[i], ids[j]))
        i = j + 1
    return ",".join(out) or None


def _lib_build_id():
    from voicefixer_amd import _lib
    return _lib.lib().vfx_build_id().decode()


def _json_line_last(line):
    """Print the ONE JSON line as the last thing this job writes to stdout: librccl prints a version banner through C
    stdio (block-buffered on a pipe, so it would surface at exit, AFTER the line) -- flush that first, print, then send
    whatever native code still writes to fd 1 to stderr."""
    import ctypes
    try:
        ctypes.CDLL(None).fflush(None)
    except OSError:
        pass
    print(json.dumps(line), flush=True)
    try:
        ctypes.CDLL(None).fflush(None)
        os.dup2(2, 1)
    except OSError:
        pass


def self_launch(args, argv):
    """``python bench.py --gpus N`` started WITHOUT a launcher (no WORLD_SIZE): become the launcher.  One process per
    GPU under torch.distributed.run on 127.0.0.1, N clamped to the visible device count (loudly); the folder job this
    scales is voicefixer/__main__.py:187-212 (the launcher runs unchanged on 1..8 visible devices)."""
    visible = args.gpus if args.dry_run else torch.cuda.device_count()
    nproc = max(1, min(args.gpus, visible))
    if args.oversubscribe and not args.dry_run and visible >= 1:
        # rehearsal of the N-rank job on FEWER devices (a one-GPU box): all N launcher ranks run, rank r on device r % visible, the
        # counters travel over gloo (RCCL refuses two ranks on one device).  What it shows: the launch, the deal, per-rank I/O pools
        # and CPU slices, every file written once -- not a scaling number (the ranks share the device).
        nproc = args.gpus
        os.environ["VFX_BENCH_OVERSUBSCRIBE"] = "1"
        print("bench.py: --oversubscribe: %d ranks on %d visible device(s), gloo for the counters" % (nproc, visible), file=sys.stderr, flush=True)
    elif nproc < args.gpus:
        print("bench.py: WARNING: --gpus %d requested but only %d HIP device(s) visible -> running %d rank(s); "
              "n_gpus in the JSON line is the number of ranks that ran" % (args.gpus, visible, nproc),
              file=sys.stderr, flush=True)
    if nproc <= 1:
        return False  # fall through to the single-process path (reports n_gpus = 1, requested_gpus = N)
    from voicefixer_amd import dist as vdist
    vdist.exec_ranks(nproc, [os.path.abspath(__file__)] + argv, {"VFX_BENCH_REQUESTED_GPUS": str(args.gpus)})


def dry_run(args, rank, world):
    """CPU rehearsal of the N-rank launch (tests/test_bench_launcher.py): gloo instead of RCCL, no device work --
    what is exercised is the self-launch, the rendezvous, the barrier / max-over-ranks reduction and the JSON line."""
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group(backend="gloo")
    if args.synth_folder or args.folder:      # the folder job with a stub device stage (see folder_job)
        return folder_job(args, rank, world, None, dist, dry=True)
    t0 = time.perf_counter()
    time.sleep(0.01 * (rank + 1))
    dt = torch.tensor([time.perf_counter() - t0, float(rank)], dtype=torch.float64)
    allr = [torch.zeros_like(dt) for _ in range(world)]
    if world > 1:
        dist.barrier()
        dist.all_gather(allr, dt)
    else:
        allr = [dt]
    if rank == 0:
        dmax = max(float(x[0]) for x in allr)
        print(json.dumps({
            "metric": "seconds-of-44.1kHz-audio restored per wall-second", "dry_run": True, "value": None,
            "n_gpus": world, "requested_gpus": int(os.environ.get("VFX_BENCH_REQUESTED_GPUS", args.gpus)),
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dmax * 1e3, 3),
            "rccl": {"backend": dist.get_backend() if world > 1 else None, "world_size": world},
            "per_rank": [{"rank": int(x[1]), "device": "dry"} for x in allr]}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--seconds", type=float, default=10.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--streams", type=int, default=2,
                    help="issue consecutive steps (one batch each) round-robin on this many HIP streams: 2 = the product "
                         "configuration (VoiceFixer.restore_batch / restore_folder: one batch's low-occupancy phases -- GRU "
                         "recurrence, deep UNet levels -- overlap the next batch's convolutions); the single-stream figure "
                         "and the per-kernel roofline (clean, un-overlapped launch durations) come from a second leg")
    ap.add_argument("--no-events", action="store_true",
                    help="development: time the steps without the per-launch HIP events (no roofline object)")
    ap.add_argument("--no-bf16x3", action="store_true",
                    help="skip the auxiliary timing of the opt-in bf16x3 arithmetic (reported beside the fp32 headline)")
    ap.add_argument("--no-host-leg", action="store_true", help="skip the host-to-host (PCIe inclusive) timing")
    ap.add_argument("--graph", action="store_true",
                    help="development (use with --no-events): replay a captured HIP graph per step (Pipeline.enable_graphs)")
    ap.add_argument("--scatter", action="store_true",
                    help="BASELINE configs[3]: rank 0 owns --utterances-per-gpu x N utterances, scatter/gather over RCCL")
    ap.add_argument("--utterances-per-gpu", type=int, default=256)
    ap.add_argument("--cpu-reps", type=int, default=2, help="timed repetitions of the CPU baseline (median reported)")
    ap.add_argument("--math", choices=["f32", "bf16x3"], default="f32",
                    help="contraction arithmetic: exact fp32 MFMA (default, the headline) or the opt-in split-bf16 "
                         "products with fp32 accumulation (DESIGN.md 3.4)")
    ap.add_argument("--synth-folder", type=int, default=0, metavar="FILES_PER_GPU",
                    help="BASELINE configs[2]/[3] disk to disk: FILES_PER_GPU x N synthetic --seconds utterances as PCM16 WAV on "
                         "tmpfs through VoiceFixer.restore_folder (ranks share the folder, dist.deal_files deals the files)")
    ap.add_argument("--folder", type=str, default="", help="the same job on an existing folder of .wav / .flac files")
    ap.add_argument("--io-threads", type=int, default=0,
                    help="decode / encode workers per rank of the folder job (0 = dist.default_io_threads: host cores / (2 * ranks), 2..8)")
    ap.add_argument("--folder-streams", type=int, default=2, help="HIP streams of the folder job's device stage")
    ap.add_argument("--oversubscribe", action="store_true",
                    help="with --gpus N on a box with fewer devices: run all N ranks anyway (rank r on device r %% visible, gloo for the "
                         "counters) -- a rehearsal of the N-rank folder job, not a scaling measurement")
    ap.add_argument("--dry-run", action="store_true",
                    help="test hook: rehearse the N-rank launch on CPU (gloo, no device work)")
    ap.add_argument("--dump-stacks-after", type=float, default=0.0,
                    help="development: faulthandler dumps every thread's stack to stderr after this many seconds (where is a slow run?)")
    args = ap.parse_args()
    if args.dump_stacks_after > 0:
        import faulthandler
        faulthandler.dump_traceback_later(args.dump_stacks_after, repeat=True, file=sys.stderr)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args, sys.argv[1:])   # does not return when it re-executes under torch.distributed.run
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if rank != 0:
        os.dup2(2, 1)   # only rank 0 owns stdout (the JSON line); whatever native libraries print elsewhere goes to stderr
    requested = int(os.environ.get("VFX_BENCH_REQUESTED_GPUS", args.gpus))
    if args.dry_run:
        if world > 1 and (args.synth_folder or args.folder):
            from voicefixer_amd import dist as vdist_
            vdist_.pin_rank_cpus(int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("LOCAL_WORLD_SIZE", world)))
        return dry_run(args, rank, world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (the product path has no CPU fallback)")
    torch.cuda.set_device(local_rank % torch.cuda.device_count())
    dev = torch.device("cuda", torch.cuda.current_device())
    dist = None
    if world > 1:
        # every rank of a node gets its own slice of physical cores, BEFORE the process group exists (its threads inherit the mask):
        # the folder job runs decode / encode workers next to the interpreter, and every job packs ~100 M weights on the host at
        # start-up -- eight ranks with a host's worth of ATen threads each would fight over the cores (`python -m voicefixer_amd
        # --gpus N` does the same, __main__.py)
        from voicefixer_amd import dist as vdist_
        vdist_.pin_rank_cpus(int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("LOCAL_WORLD_SIZE", world)))
    if world > 1 or args.scatter or ((args.synth_folder or args.folder) and "WORLD_SIZE" in os.environ):
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if os.environ.get("VFX_BENCH_OVERSUBSCRIBE") == "1" and world > torch.cuda.device_count():
            dist.init_process_group(backend="gloo")   # (--oversubscribe: several ranks share a device, which RCCL refuses)
        else:
            dist.init_process_group(backend="nccl", device_id=dev)  # nccl == RCCL on ROCm

    if args.synth_folder or args.folder:
        return folder_job(args, rank, world, dev, dist)
    from voicefixer_amd import engine, ops, weights, _lib
    build_id = _lib.lib().vfx_build_id().decode()

    n = int(round(args.seconds * SR))
    # the product's facade owns the pipeline (VoiceFixer.restore_batches is the host-to-host leg's device stage); the resident legs
    # drive its engine directly
    from voicefixer_amd.api import VoiceFixer
    vf = VoiceFixer.from_state(weights.seeded_vocoder_state(1234), weights.seeded_restorer_state(4321))
    vf.math = args.math
    pipe = vf._get_pipe()
    assert isinstance(pipe, engine.Pipeline) and pipe.device == dev, (pipe.device, dev)
    if args.scatter:
        return scatter_job(args, pipe, n, rank, world, dev, dist)
    if args.graph:
        pipe.enable_graphs(max_shapes=1, max_batch=args.batch)
    NRING = 3  # resident input batches; step i restores ring[i % NRING]
    ring = [synth_batch(args.batch, n, 1000 + 97 * k + rank, dev) for k in range(NRING)]
    wav = ring[0]

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    streams = [torch.cuda.Stream(device=dev) for _ in range(max(1, args.streams))]
    if args.graph:
        streams = streams[:1]     # (graph replay is a single-stream feature: Pipeline.restore bypasses it otherwise)

    def run_steps(k, pool=None):
        pool = streams if pool is None else pool
        pipe.set_streams(len(pool))   # (sizes the two-CU GRU launches so that every stream's launch can become resident)
        last = None
        for i in range(k):
            if len(pool) == 1 and pool[0] is None:
                last = pipe.restore(ring[i % NRING], n)
            else:  # consecutive batches on alternating streams: one batch's GRU overlaps the other's convolutions
                with torch.cuda.stream(pool[i % len(pool)]):
                    last = pipe.restore(ring[i % NRING], n)
        if len(pool) > 1:             # the caller's stream joins every side stream (what restore_batches' drain does)
            for st in pool:
                torch.cuda.current_stream(dev).wait_stream(st)
        return last

    SINGLE = [None]                   # the single-stream leg: torch's current stream
    if len(streams) == 1:
        streams = SINGLE
    out = run_steps(args.warmup)
    barrier()
    # ---- (N > 1 ranks) one-rank leg: rank 0 runs the K steps ALONE while the other ranks wait at the barrier -- the N = 1 rate of
    # this very job on this very node, so that the line can say what N ranks made of it (no events, same streams)
    one_rank = None
    if dist is not None and world > 1:
        if rank == 0:
            t1 = time.perf_counter()
            run_steps(args.steps)
            torch.cuda.synchronize()
            d1 = time.perf_counter() - t1
            one_rank = {"value": round(args.batch * args.seconds * args.steps / d1, 2), "unit": "x real-time",
                        "ms_per_step": round(d1 / args.steps * 1e3, 3), "steps": args.steps,
                        "note": "rank 0 alone on its device, every other rank idle at a barrier; same build, same streams"}
        barrier()
    sampler = ClockSampler(dev.index)
    if not args.no_events:
        # pre-created timing events (creating one costs ~10 us of host time: visible in a launch-bound batch-1 run)
        ops.EVENT_POOL = [torch.cuda.Event(enable_timing=True) for _ in range(800 * max(args.steps, 1) * (2 if len(streams) > 1 else 1))]
        for e in ops.EVENT_POOL[:8]:
            e.record()  # first use of an event allocates its backing object
        torch.cuda.synchronize()
    ops.PROFILE = None if args.no_events else []
    with sampler:
        t0 = time.perf_counter()
        out = run_steps(args.steps)
        barrier()
        dt = time.perf_counter() - t0
    clocks = {"timed_region": sampler.summary(t0 + 0.25 * dt, t0 + dt)}   # (the first quarter: the clock is still ramping)
    prof, ops.PROFILE = ops.PROFILE, None
    if prof is None:
        ops.EVENT_POOL = None
        if rank == 0:
            print(json.dumps({"ms_per_step": round(dt / args.steps * 1e3, 3), "events": False, "streams": len(streams)}), flush=True)
        return
    assert torch.isfinite(out).all()
    pipe.check()  # device-side error flags (two-CU GRU hand-off)
    # ---- second leg (only when the timed region ran on several streams): the same K steps on ONE stream.  Its wall time is
    # `single_stream`; its per-launch HIP events are what the roofline is computed from -- in the multi-stream timed region the
    # kernels of two batches share the chip, so an event bracket there also contains the other stream's work (reported beside
    # it as roofline.timed_region, not hidden)
    prof_timed, dt_single = None, None
    if len(streams) > 1:
        run_steps(1, SINGLE)
        barrier()
        prof_timed, ops.PROFILE = prof, []
        with sampler:
            t1 = time.perf_counter()
            run_steps(args.steps, SINGLE)
            barrier()
            dt_single = time.perf_counter() - t1
        clocks["single_stream_leg"] = sampler.summary(t1 + 0.25 * dt_single, t1 + dt_single)
        prof, ops.PROFILE = ops.PROFILE, None
        pipe.check()
        pipe.set_streams(len(streams))
    ops.EVENT_POOL = None
    last_idx = (args.steps - 1) % NRING  # ``out`` is the restoration of ring[last_idx]

    per_rank = [{"rank": 0, "device": "cuda:%d" % dev.index, "wall_s": round(dt, 4)}]
    if dist is not None:
        xdev = dev if dist.get_backend() == "nccl" else None      # (the --oversubscribe rehearsal runs its collectives on gloo: host tensors)
        mine = torch.tensor([dt, float(dev.index)], device=xdev, dtype=torch.float64)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = [{"rank": r, "device": "cuda:%d" % int(x[1]), "wall_s": round(float(x[0]), 4)}
                    for r, x in enumerate(allr)]
        dt = max(float(x[0]) for x in allr)   # max over ranks

    # ---- roofline bookkeeping for the dominant MFMA kernel family (this rank) ----
    # vfx_last_conv_tile() = BM*100000 + BL*100 + code; code 51/52/54: convw_kernel (1-D, chunk depth 8/16/32),
    # 59: convw_kernel 3x3 on pitch maps, 61/62/64: convw_kernel fused ResStack layer, 96: resblk4_kernel (fused layer, both halves F(4,3)), 16: conv_x3_kernel,
    # 83: convtw_kernel (ConvTranspose1d, Winograd F(3,2)),
    # 80 / 81 / 82: convwg4_kernel / convwg4p_kernel / convwg4x_kernel (Winograd F(4,3), 1-D; one tile per workgroup / persistent / persistent with the re-blocked wave tile), 88: convwg4s_kernel (Winograd F(4,3), 3x3 on pitch maps), 71/72/74 | 91/92/94: fused layer with a Winograd F(2,3) | F(4,3) second half,
    # anything else: conv_taps_kernel with KC = code.  A family = what one regex over rocprofv3's kernel names selects,
    # so that profiles/*kernel_stats*.csv can be averaged over exactly the same launches.
    import re

    def family(tile):
        bm, bl, code = tile // 100000, tile // 100 % 1000, tile % 100
        if code in (51, 52, 54):
            return ("w1d", bm, bl), "convw_kernel<%d,%d,*,*,NT=2|3,*,0>" % (bm, bl), \
                   r"convw_kernel<%d, %d, \d+, \d+, [23], \d+, 0>" % (bm, bl)
        if code == 59:
            return ("w2d", bm, bl), "convw_kernel<%d,%d,*,*,NT=9,*,0>" % (bm, bl), \
                   r"convw_kernel<%d, %d, \d+, \d+, 9, \d+, 0>" % (bm, bl)
        if code in (61, 62, 64):
            return ("wfused", bm, bl), "convw_kernel<%d,%d,*,*,3,*,1> (fused ResStack layer)" % (bm, bl), \
                   r"convw_kernel<%d, %d, \d+, \d+, 3, \d+, 1>" % (bm, bl)
        if code in (71, 72, 74):
            return ("wfusedw", bm, bl), "convw_kernel<%d,%d,*,*,3,*,2> (fused ResStack layer, Winograd second half)" % (bm, bl), \
                   r"convw_kernel<%d, %d, \d+, \d+, 3, \d+, 2>" % (bm, bl)
        if code in (91, 92, 94):
            return ("wfusedw4", bm, bl), "convw_kernel<%d,%d,*,*,3,*,3> (fused ResStack layer, Winograd F(4,3) second half)" % (bm, bl), \
                   r"convw_kernel<%d, %d, \d+, \d+, 3, \d+, 3>" % (bm, bl)
        if code == 96:
            return ("wfusedw44", bm, bl), "resblk4_kernel (fused ResStack layer, both halves Winograd F(4,3))", r"resblk4_kernel"
        if code == 88:
            wgm = bm // 32
            return ("wino4", bm, bl, 3, "s"), "convwg4s_kernel<%d,%d,*> (3x3 as Winograd F(4,3) along the map rows, kernel columns share one staged tile)" % (
                wgm, 4 // wgm), r"convwg4s_kernel<%d, %d, \d+[,>]" % (wgm, 4 // wgm)
        if code == 82 or (code in (80, 81) and bm == 128):
            # one family: the Winograd F(4,3) convolutions of the C >= 128 ResStacks and the condnet -- convwg4x_kernel (round 6: 128 channels x 64 quads
            # per workgroup, 32 x 64 x 6 per wave, deferred epilogue) for the two launches of a ResStack layer, convwg4[p]_kernel<4,1> (128 x 32) for the rest
            return ("wino4", 128, 128), ("convwg4x_kernel + convwg4[p]_kernel<4,1,*> (Winograd F(4,3) along the dilated axis, 128 output channels per workgroup; "
                                         "x = 32 x 64 x 6 wave tile with the deferred epilogue, p = persistent workgroups)"), \
                   r"(convwg4x_kernel<|convwg4p?_kernel<4, 1, (true|false)[,>])"
        if code in (80, 81):   # 81: convwg4p_kernel, the persistent form of the same tile (one family: the same arithmetic on the same tile)
            wgm = bm // 32
            return ("wino4", bm, bl), "convwg4[p]_kernel<%d,%d,*> (Winograd F(4,3), %d ch x %d output quads; p = persistent workgroups, pipeline across tiles)" % (
                wgm, 4 // wgm, bm, bl // 4), r"convwg4p?_kernel<%d, %d, (true|false)[,>]" % (wgm, 4 // wgm)
        if code == 83:
            wgm = bm // 32
            return ("tw32", bm, bl), "convtw_kernel<%d,%d> (ConvTranspose1d as Winograd F(3,2) along the input axis, one output phase per workgroup)" % (
                wgm, 4 // wgm), r"convtw_kernel<%d, %d>" % (wgm, 4 // wgm)
        if code == 16:
            return ("x3", bm, bl), "conv_x3_kernel<%d,%d,*>" % (bm, bl), r"conv_x3_kernel<%d, %d," % (bm, bl)
        return ("taps", bm, bl, code), "conv_taps_kernel<%d,%d,*,*,KC=%d,*>" % (bm, bl, code), \
               r"conv_taps_kernel<%d, %d, \d+, \d+, %d," % (bm, bl, code)

    # Arithmetic a family EXECUTES per multiply-accumulate of the direct convolution: the Winograd F(4,3) kernels form
    # 6 products per four outputs where the direct sum has 12.  `achieved` / `frac` below count executed MFMA
    # work (what the matrix pipe can be compared with); the direct-convolution equivalent is reported next to it.
    def exec_factor(key):
        if key[0] == "wino4":
            return 0.5         # six products per four outputs; the direct sum has twelve
        if key[0] == "wfusedw":
            return 5.0 / 6.0   # the dilated half direct (3 products per output), the dilation-1 half Winograd F(2,3) (2)
        if key[0] == "wfusedw4":
            return 0.75        # the dilated half direct (3 products per output), the dilation-1 half Winograd F(4,3) (1.5)
        if key[0] == "wfusedw44":
            return 0.5         # both halves Winograd F(4,3)
        if key[0] == "tw32":
            return 2.0 / 3.0   # four products per three outputs of a phase; the direct sum has six
        return 1.0

    def by_family(events):
        fams, sb, ss, sn = {}, 0, 0.0, 0
        for tile, macs, e0, e1 in events:
            if tile == -1:  # the STFT->mel front-end: `macs` carries its algorithmic bytes
                sb += macs
                ss += e0.elapsed_time(e1) * 1e-3
                sn += 1
                continue
            key, name, rx = family(tile)
            d = fams.setdefault(key, [0, 0, 0.0, name, rx, exec_factor(key)])
            d[0] += 1
            d[1] += macs
            d[2] += e0.elapsed_time(e1) * 1e-3
        return fams, sb, ss, sn

    by_fam, stft_bytes, stft_secs, stft_n = by_family(prof)
    dt_roof = dt if dt_single is None else dt_single      # the wall time of the leg the events belong to
    conv_time = sum(d[2] for d in by_fam.values())
    conv_macs = sum(d[1] * d[5] for d in by_fam.values())
    conv_macs_direct = sum(d[1] for d in by_fam.values())
    launches, macs, secs, kname, krx, xf = max(by_fam.values(), key=lambda d: d[2])
    achieved = 2.0 * macs * xf / secs / 1e12
    # HBM bytes per launch of that family: PMC counters cannot be read live; they come from the committed rocprofv3
    # --pmc passes over this same command (tools/profile_round.sh -> profiles/<TRAFFIC_PROFILE>)
    # -- and only when that summary was taken with THIS library: the summary carries vfx_build_id() of the build it
    # profiled; a kernel change without a re-profile reports traffic = null and says why
    traffic, traffic_note = None, None
    tfile = os.path.join(ROOT, "profiles", TRAFFIC_PROFILE)
    if os.path.exists(tfile) and args.batch == 32 and abs(args.seconds - 10.0) < 1e-9:
        doc = json.load(open(tfile))
        if doc.get("lib_build_id") != build_id:
            traffic_note = ("profiles/%s was taken with library build %s, the loaded library is %s: HBM traffic not "
                            "reported (re-run tools/profile_round.sh)" % (TRAFFIC_PROFILE, doc.get("lib_build_id"), build_id))
            print("bench.py: WARNING: " + traffic_note, file=sys.stderr, flush=True)
        else:
            num = den = 0
            for kn, rec in doc["kernels"].items():
                if re.search(krx, kn):
                    num += rec["hbm_bytes_per_launch"] * rec["launches"]
                    den += rec["launches"]
            traffic = int(num / den) if den else None
    x3_dom = kname.startswith("conv_x3")
    # bf16x3 instance: three bf16 MFMA products per algorithmic product -> peak = dense bf16 peak / 3
    peak = BF16_MFMA_PEAK_TFLOPS / 3.0 if x3_dom else FP32_MFMA_PEAK_TFLOPS
    if x3_dom:
        traffic = None
    roofline = {
        "bound": "mfma", "achieved": round(achieved, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
        "frac": round(achieved / peak, 4), "traffic": traffic,
        **({"traffic_note": traffic_note} if traffic_note else {}),
        "kernel": kname, "kernel_name_regex": krx,
        "launches_per_step": launches // args.steps,
        "avg_launch_ms": round(secs / launches * 1e3, 4),
        "algorithmic_gflop_per_launch": round(2.0 * macs * xf / launches / 1e9, 3),
        "all_conv_kernels": {"achieved": round(2.0 * conv_macs / conv_time / 1e12, 2),
                             "direct_equivalent": round(2.0 * conv_macs_direct / conv_time / 1e12, 2),
                             "time_share_of_step": round(conv_time / dt_roof, 4) if world == 1 else None},
        "families": {d[3]: {"launches_per_step": d[0] // args.steps, "ms_per_step": round(d[2] / args.steps * 1e3, 2),
                            "tflops": round(2.0 * d[1] * d[5] / d[2] / 1e12, 1),
                            **({"direct_equivalent_tflops": round(2.0 * d[1] / d[2] / 1e12, 1)} if d[5] != 1.0 else {})}
                     for d in sorted(by_fam.values(), key=lambda d: -d[2])[:8]},
    }
    if prof_timed is not None:
        fam2 = by_family(prof_timed)[0]
        dom = next((d for d in fam2.values() if d[4] == krx), None)
        roofline["measured_on"] = ("the single-stream leg (K steps on one stream right after the timed region): un-overlapped launch "
                                   "durations.  roofline.timed_region = the same family's event brackets inside the %d-stream timed "
                                   "region, where a bracket also contains whatever the other stream ran meanwhile" % len(streams))
        if dom is not None:
            roofline["timed_region"] = {"streams": len(streams), "avg_launch_ms": round(dom[2] / dom[0] * 1e3, 4),
                                        "achieved": round(2.0 * dom[1] * dom[5] / dom[2] / 1e12, 2),
                                        "frac": round(2.0 * dom[1] * dom[5] / dom[2] / 1e12 / peak, 4)}
    if xf != 1.0:
        roofline["algorithm"] = ("Winograd F(4,3) along the dilated axis: 6 fp32 MFMA products per 4 outputs instead of 12; "
                                 "achieved / frac / algorithmic_gflop_per_launch count the EXECUTED products"
                                 if xf == 0.5 else "ConvTranspose1d as Winograd F(3,2): four products per three outputs instead of six" if abs(xf - 2.0 / 3.0) < 1e-9 else "fused ResStack layer: dilated half direct, dilation-1 half Winograd on the LDS tile "
                                 "(F(4,3): 3 of 4 products, F(2,3): 5 of 6); achieved / frac count the EXECUTED products")
        roofline["direct_conv_gflop_per_launch"] = round(2.0 * macs / launches / 1e9, 3)
        roofline["direct_equivalent_tflops"] = round(2.0 * macs / secs / 1e12, 2)
        # SURVEY.md 8(d) defines `achieved` on the direct convolution's FLOPs: that fraction too (it exceeds what the
        # EXECUTED products allow by 1 / (executed share) because Winograd forms fewer products, not because work is skipped)
        roofline["frac_direct_equivalent"] = round(2.0 * macs / secs / 1e12 / peak, 4)
    else:
        roofline["frac_direct_equivalent"] = roofline["frac"]

    if stft_n:
        # The front-end moves the algorithmic minimum of HBM bytes (4 N + 512 T per utterance) but that is not what bounds
        # it: a frame's 2048-point radix-2 Stockham FFT lives in LDS (11 passes, every pass reads and writes 2048 complex
        # values and reads 1024 twiddles), so the roof it is priced against is the LDS data path -- 128 B / clk / CU
        # (ds_read_b64 / ds_write_b64, MI355X_MICROARCH.md "LDS") x 256 CUs x 2.4 GHz = 78.6 TB/s.  The HBM figure
        # stays beside it for the record.
        T_ = 1 + n // 441
        lds_bytes = stft_n * args.batch * T_ * (11 * (2048 * 16 + 1024 * 8) + 2048 * 8 + 1025 * 4 + 2018 * 8)
        lds_peak = 128 * 256 * 2.4e9
        roofline["stft_mel_kernel"] = {"bound": "lds", "achieved": round(lds_bytes / stft_secs / 1e12, 2),
                                       "peak": round(lds_peak / 1e12, 1), "unit": "TB/s (LDS)",
                                       "frac": round(lds_bytes / stft_secs / lds_peak, 4),
                                       "hbm_gbps": round(stft_bytes / stft_secs / 1e9, 1),
                                       "hbm_frac_of_8TBps": round(stft_bytes / stft_secs / 8.0e12, 4),
                                       "avg_launch_ms": round(stft_secs / stft_n * 1e3, 4)}
    audio_seconds = world * args.batch * args.seconds * args.steps
    value = audio_seconds / dt
    line = {
        "metric": "seconds-of-44.1kHz-audio restored per wall-second",
        # n_gpus counts DISTINCT devices (an --oversubscribe rehearsal puts several ranks on one); `ranks` = processes that ran
        "value": round(value, 2), "unit": "x real-time", "n_gpus": len({pr["device"] for pr in per_rank}), "ranks": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": args.math, "data": "synthetic",
        "value_definition": ("whole-job audio seconds per wall second with the inputs resident in HBM when the timed region "
                             "starts and the outputs left there (the bench contract); value_host_to_host = the same from "
                             "pinned host waveforms to pinned host waveforms, H2D / D2H inside the timed region (SURVEY.md 8(d))"),
        "config": {"workload": ("batched folder restore (BASELINE configs[2]): one batch of %d x %.0f s 44.1 kHz "
                                "utterances per step, VoiceFixer.restore mode 0, seeded random weights"
                                % (args.batch, args.seconds)) if args.batch > 1 else
                               ("VoiceFixer.restore mode 0, single %.0f s 44.1 kHz mono utterance per step (BASELINE configs[1]%s), "
                                "seeded random weights" % (args.seconds, "" if abs(args.seconds - 10.0) < 1e-9 else " at another length")),
                   "batch_per_gpu": args.batch, "utterance_seconds": args.seconds, "frames": 1 + n // 441,
                   "arithmetic": ("fp32 operands, fp32 MFMA accumulation everywhere; k=3 / 3x3 convolutions evaluated as Winograd "
                                  "F(4,3) (the C=64 stage: one fused launch per layer with both halves F(4,3) for dilations <= 27, two F(4,3) launches for the wider ones): half "
                                  "of the direct sum's products, rounding ~3x the direct sum's (DESIGN.md 3.0b; engine.set_winograd(False) / --selfcheck run the direct sums)") if args.math == "f32"
                                 else "opt-in split-bf16 products (three bf16 MFMAs per fp32 product), fp32 accumulation",
                   "parallelism": "utterance sharding x%d (no data-path collective)" % world},
        "streams": len(streams),
        "path_tflops": round(2.0 * path_macs(n) * args.batch * world * args.steps / dt / 1e12, 2),
        "requested_gpus": requested, "visible_devices": torch.cuda.device_count(),
        "rccl": {"backend": dist.get_backend() if dist is not None else None, "world_size": world},
        "per_rank": per_rank,
        "lib_build_id": build_id,
        "roofline": roofline,
        "clocks": dict(clocks, note="shader clock / socket power of rank 0's device while the K timed steps ran (first quarter of the "
                                    "region dropped: ramp).  The convolution families run at the package power cap, so achieved = "
                                    "busy share x granted clock and boxes differ by what clock the cap buys them (DESIGN.md 3.0): "
                                    "compare rounds at equal sclk, or normalise by it"),
    }
    if one_rank is not None:
        line["one_rank_leg"] = one_rank
        line["scaling_efficiency_self_measured"] = round(value / (world * one_rank["value"]), 4)
    if dt_single is not None:
        dts = dt_single
        if dist is not None:
            t = torch.tensor([dt_single], device=dev if dist.get_backend() == "nccl" else None, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dts = float(t.item())
        line["single_stream"] = {"value": round(audio_seconds / dts, 2), "unit": "x real-time",
                                 "ms_per_step": round(dts / args.steps * 1e3, 3), "steps": args.steps,
                                 "two_streams_over_single": round(dts / dt, 4),
                                 "note": "the same K steps issued on one stream (what `value` was before round 5); the roofline's "
                                         "launch durations are this leg's"}
    if requested != world:
        line["note_gpus"] = ("--gpus %d was requested, %d rank(s) ran (visible HIP devices: %d)"
                             % (requested, world, torch.cuda.device_count()))

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

    if world == 1 and not args.no_host_leg:
        real = [st for st in streams if st is not None]
        line["host_to_host"] = host_to_host_leg(vf, args, n, dev, len(real) or 1, real)
        # SURVEY.md 8(d)'s host-waveform -> host-waveform figure, at the top level next to `value` (which the bench
        # contract defines with inputs resident in HBM)
        line["value_host_to_host"] = line["host_to_host"]["value"]

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(args, ring[last_idx][0].cpu().numpy(), out[0].cpu())
    if rank == 0:
        _json_line_last(line)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
