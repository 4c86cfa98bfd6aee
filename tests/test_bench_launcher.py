"""bench.py launched PLAIN with --gpus N must start N ranks by itself (SURVEY.md 8(e) caveat: "the launcher must run
unchanged on 1...8 visible devices and print the device count").  Rehearsed on CPU: --dry-run swaps RCCL for gloo and
skips the device work; the self-launch, rendezvous on 127.0.0.1, barrier and JSON line are the real code."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*extra, env=None):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run", "--steps", "2", "--warmup", "1",
                        *extra], capture_output=True, text=True, timeout=300, env=e)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout       # rank 0 prints ONE JSON line
    assert r.stdout.strip().splitlines()[-1] == lines[0], r.stdout   # ... and nothing follows it on stdout
    return json.loads(lines[0]), r.stderr


def test_plain_launch_with_gpus_2_runs_two_ranks():
    line, _ = _run("--gpus", "2")
    assert line["n_gpus"] == 2 and line["requested_gpus"] == 2
    assert line["rccl"]["world_size"] == 2 and line["rccl"]["backend"] == "gloo"
    assert sorted(r["rank"] for r in line["per_rank"]) == [0, 1]


def test_plain_launch_single():
    line, _ = _run("--gpus", "1")
    assert line["n_gpus"] == 1 and line["per_rank"] == [{"rank": 0, "device": "dry"}]


def test_under_an_external_launcher_the_script_does_not_relaunch():
    # the driver's form: WORLD_SIZE already set by torch.distributed.run -> bench.py must use it, not nest a launcher
    line, _ = _run("--gpus", "8", env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert line["n_gpus"] == 1 and line["requested_gpus"] == 8


def test_gpus_flag_is_read():
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "args.gpus" in src and "torch.distributed.run" in src


def test_folder_job_on_two_ranks_with_a_stub_device_stage():
    """``bench.py --gpus 2 --synth-folder F`` (BASELINE configs[3] disk to disk) rehearsed on CPU: the self-launch, rank 0 writing the
    synthetic folder, the barriers, dist.deal_files, the decode / encode workers of restore_folder, the all-gather of the per-rank
    counters and the JSON line are the real code; only the device stage is a stub (identity) and RCCL is gloo."""
    line, _ = _run("--gpus", "2", "--synth-folder", "5", "--seconds", "0.25", "--batch", "4", "--io-threads", "2")
    assert line["dry_run"] is True and line["n_gpus"] == 2 and line["rccl"] == {"backend": "gloo", "world_size": 2}
    files = [r["files"] for r in line["per_rank"]]
    assert sum(files) == 10 and files == [5, 5] and [r["rank"] for r in line["per_rank"]] == [0, 1]
    assert abs(sum(r["audio_s"] for r in line["per_rank"]) - 10 * 0.25) < 0.11 and line["value"] > 0
    assert all(r["decode_worker_s"] > 0 and r["encode_worker_s"] > 0 and r["batches"] == 2 for r in line["per_rank"])
    assert "configs[3]" in line["config"]["workload"] and line["hbm_resident"] is None and line["ranks"] == 2
    # the ranks of the folder job are pinned to disjoint core slices before their process group exists, and
    # their I/O pools / ATen pools are sized to the slice
    cpus = [r["cpu"] for r in line["per_rank"]]
    if len(os.sched_getaffinity(0)) >= 4:
        assert all(c["pinned"] for c in cpus) and cpus[0]["cores"] != cpus[1]["cores"]
        assert all(c["torch_threads"] == min(c["n_physical"], 16) for c in cpus)


def test_clock_sampler_and_core_ranges_do_not_need_a_gpu():
    """bench.ClockSampler (sclk / socket power beside every bench line) must never break a run: without a HIP device it finds no hwmon of
    its own (or only rocm-smi, which reports nothing) and says so; its summary keeps the samples inside the window it is asked for."""
    import time
    sys.path.insert(0, ROOT)
    import bench
    smp = bench.ClockSampler(0, period=0.01)
    with smp:
        time.sleep(0.05)
    doc = smp.summary()
    assert "source" in doc and (doc["source"] is None or doc.get("sclk_mhz") is None or doc["sclk_mhz"]["n"] >= 1)
    smp.source, smp.samples = "fake", [(1.0, 2100.0, 1300.0), (2.0, 2200.0, None), (3.0, None, 1400.0), (9.0, 999.0, 999.0)]
    doc = smp.summary(0.5, 3.5)
    assert doc["sclk_mhz"] == {"mean": 2150.0, "min": 2100.0, "max": 2200.0, "n": 2}
    assert doc["socket_power_w"] == {"mean": 1350.0, "min": 1300.0, "max": 1400.0, "n": 2}
    assert bench._ranges([0, 1, 2, 3, 128, 129, 200]) == "0-3,128-129,200" and bench._ranges([]) is None
