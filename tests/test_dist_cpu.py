"""world_size-2 gloo tests of the N>1 path (sharding + scatter/gather plumbing) on CPU.
The GPU compute is replaced by a deterministic stand-in: what is under test is that every
utterance is processed exactly once, by the right rank, and comes back in order."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from voicefixer_amd import dist as vdist


def test_shard_range_covers_everything():
    for n in (0, 1, 5, 8, 2048, 2049):
        for world in (1, 2, 3, 8):
            blocks = [vdist.shard_range(n, r, world) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            for (a, b), (c, d) in zip(blocks, blocks[1:]):
                assert b == c
            sizes = [b - a for a, b in blocks]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_utt, n, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(0)
        wavs = torch.randn((n_utt, n), generator=g) if rank == 0 else None
        seen = []

        def fake_restore(batch):  # stand-in for Pipeline.restore: tags the rank, keeps the data
            seen.append(batch.shape[0])
            return batch * 2.0 + (rank + 1)

        out = vdist.restore_sharded(fake_restore, wavs, n, torch.device("cpu"), batch_size=2)
        lo, hi = vdist.shard_range(n_utt, rank, world)
        assert sum(seen) == hi - lo and all(s <= 2 for s in seen)
        if rank == 0:
            expect = wavs * 2.0
            for r in range(world):
                a, b = vdist.shard_range(n_utt, r, world)
                expect[a:b] += r + 1
            q.put(bool(torch.equal(out, expect)))
        else:
            assert out is None
        # timing reduction used by bench.py: max over ranks
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        assert t.item() == world
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_utt", [5, 1])
def test_scatter_restore_gather_world2(n_utt):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_utt, 64, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=10) is True


# ---- folder inference sharded over ranks (VoiceFixer.restore_folder(rank, world), python -m voicefixer_amd --gpus N) ----

def test_deal_files_balances_a_ragged_folder():
    """dist.deal_files: longest first to the least loaded rank.  On a ragged 5..30 s folder every rank's audio seconds
    are within 5 % of the mean (in fact within one file), an equal-length folder is dealt round-robin, every file has
    exactly one owner and the answer depends on nothing but the lengths (every rank computes it alone)."""
    import numpy as np
    rng = np.random.default_rng(3)
    for n_files, world in ((256, 2), (2048, 8), (37, 4), (3, 8)):
        lens = [int(x) for x in rng.integers(5 * 44100, 30 * 44100, n_files)]
        owner = vdist.deal_files(lens, world)
        assert owner == vdist.deal_files(list(lens), world) and len(owner) == n_files and set(owner) <= set(range(world))
        tot = [sum(l for l, o in zip(lens, owner) if o == r) for r in range(world)]
        if n_files >= 4 * world:
            assert max(tot) - min(tot) <= max(lens)
            assert max(abs(t - sum(tot) / world) for t in tot) < 0.05 * sum(tot) / world
    owner = vdist.deal_files([441000] * 2048, 8)
    assert [owner.count(r) for r in range(8)] == [256] * 8 and owner[:16] == list(range(8)) * 2
    # contiguous blocks of the length-sorted list (shard_range) would NOT balance: that is why the folder path does not use them
    lens = sorted(int(x) for x in rng.integers(5 * 44100, 30 * 44100, 256))
    lo, hi = vdist.shard_range(256, 0, 2)
    assert sum(lens[lo:hi]) < 0.75 * sum(lens[hi:])


def _make_ragged_folder(ind, n_files, seed=0):
    import numpy as np
    from scipy.io import wavfile
    rng = np.random.default_rng(seed)
    os.makedirs(ind, exist_ok=True)
    lens = {}
    for k in range(n_files):
        n = int(rng.integers(5 * 441, 30 * 441))        # the 5..30 s distribution at 1/100 scale
        name = "utt%03d.wav" % k
        x = (2000 * np.sin(np.arange(n) * (0.01 + 0.001 * k)) + rng.integers(-20, 20, n)).astype(np.int16)
        wavfile.write(os.path.join(ind, name), 44100, x)
        lens[name] = n
    return lens


class _StubDevice:
    """Mixed into VoiceFixer: the device stage replaced by x -> -x (no checkpoints, no device; the host side -- dealing,
    planning, decode / encode workers, counters -- is what runs)."""

    def __init__(self):
        pass

    def restore_batches(self, batches, your_vocoder_func=None, streams=2, mode=0):
        for tag, kind, host, lens in batches:
            yield tag, -host, list(lens)


def _folder_worker(rank, world, port, ind, outd, q, use_cli):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world))
    from voicefixer_amd import api

    class Stub(_StubDevice, api.VoiceFixer):
        pass

    if use_cli:      # the console script as torch.distributed.run would start it on every rank
        from voicefixer_amd import __main__ as cli
        api.VoiceFixer = Stub
        rc = cli.main(["-ifdr", ind, "-ofdr", outd, "--gpus", str(world), "--dist-backend", "gloo", "--batch-size", "4"])
        q.put((rank, rc, None))
        return
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        st = {}
        names = Stub().restore_folder(ind, outd, batch_size=4, io_threads=2, stats=st)   # rank / world from the group
        per_rank = vdist.gather_counters([st["files"], st["audio_s"]])
        q.put((rank, names, per_rank))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("use_cli", [False, True])
def test_folder_sharded_over_two_ranks_every_file_written_once(tmp_path, use_cli):
    """World-2 gloo run of the sharded folder job with the stub device stage: every file of a ragged folder is written
    exactly once (the two ranks' name lists are disjoint and cover the folder), each output is ITS input through the
    stub (so no row got mixed up in staging / batching), and the ranks' audio seconds agree within 5 %."""
    import numpy as np
    from voicefixer_amd import audio_io
    ind, outd = str(tmp_path / "in"), str(tmp_path / "out")
    lens = _make_ragged_folder(ind, 41)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_folder_worker, args=(r, 2, port, ind, outd, q, use_cli)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert sorted(os.listdir(outd)) == sorted(lens)
    for name, n in lens.items():
        y, x = audio_io.load_wav(os.path.join(outd, name)), audio_io.load_wav(os.path.join(ind, name))
        assert y.shape == (n,) and np.abs(y + x).max() <= 1.0 / 32768 + 1e-7, name
    if use_cli:
        assert sorted(g[1] for g in got) == [0, 0]
        return
    a, b = (set(g[1]) for g in sorted(got))
    assert not (a & b) and a | b == set(lens) and abs(len(a) - len(b)) <= 3
    per_rank = got[0][2]
    assert per_rank == got[1][2] and sum(int(x[0]) for x in per_rank) == 41
    secs = [x[1] for x in per_rank]
    assert abs(secs[0] - secs[1]) < 0.05 * (sum(secs) / 2)
    assert abs(sum(secs) - sum(lens.values()) / 44100.0) < 1e-6


def test_restore_folder_explicit_rank_and_world_without_a_process_group(tmp_path):
    """rank= / world= given explicitly (no torch.distributed group): two calls in one process write disjoint halves."""
    from voicefixer_amd import api

    class Stub(_StubDevice, api.VoiceFixer):
        pass

    ind, outd = str(tmp_path / "in"), str(tmp_path / "out")
    lens = _make_ragged_folder(ind, 9, seed=1)
    a = Stub().restore_folder(ind, outd, batch_size=2, io_threads=2, rank=0, world=2)
    assert set(os.listdir(outd)) == set(a) and 0 < len(a) < 9
    b = Stub().restore_folder(ind, outd, batch_size=2, io_threads=2, rank=1, world=2)
    assert not (set(a) & set(b)) and set(a) | set(b) == set(lens)
    with pytest.raises(ValueError):
        Stub().restore_folder(ind, outd, rank=2, world=2)


# ---- per-file fault isolation and resume of the (sharded) folder job ----

def _add_bad_files(ind):
    """Four inputs a folder job must survive: a RIFF header cut off before its data chunk, a 500-sample file, a file whose
    header parses but whose decoder raises (a format tag scipy refuses), and a PCM16 file whose data chunk holds fewer
    samples than its header promises (an interrupted recording: restored at the length that is really there)."""
    import struct
    import numpy as np
    from scipy.io import wavfile
    with open(os.path.join(ind, "bad_header.wav"), "wb") as f:
        f.write(b"RIFF\x24\x00\x00\x00WAVEfmt ")
    wavfile.write(os.path.join(ind, "bad_short.wav"), 44100, (1000 * np.sin(np.arange(500) * 0.1)).astype(np.int16))
    n = 3000
    fmt = struct.pack("<HHIIHH", 0x0002, 1, 44100, 44100 * 2, 2, 16)      # ADPCM tag: wav_length reads it, scipy raises
    with open(os.path.join(ind, "bad_decode.wav"), "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", 36 + 2 * n) + b"WAVE" + b"fmt " + struct.pack("<I", 16) + fmt +
                b"data" + struct.pack("<I", 2 * n) + bytes(2 * n))
    x = (2000 * np.sin(np.arange(4000) * 0.02)).astype(np.int16)
    wavfile.write(os.path.join(ind, "truncated.wav"), 44100, x)
    raw = open(os.path.join(ind, "truncated.wav"), "rb").read()
    with open(os.path.join(ind, "truncated.wav"), "wb") as f:
        f.write(raw[:44 + 2 * 2900])            # header still says 4000 samples, 2900 are there
    return {"bad_header.wav", "bad_short.wav", "bad_decode.wav"}, {"truncated.wav": 2900}


def _faulty_worker(rank, world, port, ind, outd, q, use_cli):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world))
    import warnings
    warnings.simplefilter("ignore")          # (scipy warns about the truncated data chunk)
    from voicefixer_amd import api

    class Stub(_StubDevice, api.VoiceFixer):
        pass

    if use_cli:
        import contextlib
        import io
        from voicefixer_amd import __main__ as cli
        api.VoiceFixer = Stub
        err = io.StringIO()
        with contextlib.redirect_stderr(err):
            rc = cli.main(["-ifdr", ind, "-ofdr", outd, "--gpus", str(world), "--dist-backend", "gloo", "--batch-size", "4"])
        q.put((rank, rc, err.getvalue()))
        return
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        st = {}
        names = Stub().restore_folder(ind, outd, batch_size=4, io_threads=2, stats=st)
        every = vdist.gather_objects((st["failed"], st["truncated"]))       # the collective after the job: nobody hangs
        vdist.gather_counters([st["files"], st["audio_s"]])
        q.put((rank, names, every))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("use_cli", [False, True])
def test_folder_job_survives_bad_files_on_two_ranks(tmp_path, use_cli):
    """One truncated header, one 500-sample file, one file that fails in the decoder, one file shorter than its header says:
    every OTHER output is written exactly once, the short-data file is restored at its real length, the three bad ones are
    listed with a reason (each by exactly one rank), both ranks reach the collectives, the CLI exits with status 2."""
    import numpy as np
    from voicefixer_amd import audio_io
    ind, outd = str(tmp_path / "in"), str(tmp_path / "out")
    lens = _make_ragged_folder(ind, 19, seed=5)
    bad, trunc = _add_bad_files(ind)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_faulty_worker, args=(r, 2, port, ind, outd, q, use_cli)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    good = dict(lens, **trunc)
    assert sorted(os.listdir(outd)) == sorted(good)             # (no .part-* leftovers either)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for name, n in good.items():
            y, x = audio_io.load_wav(os.path.join(outd, name)), audio_io.load_wav(os.path.join(ind, name))
            assert y.shape == (n,) and x.shape == (n,) and np.abs(y + x).max() <= 1.0 / 32768 + 1e-7, name
    if use_cli:
        assert [g[1] for g in got] == [2, 2]
        for name in bad:
            assert "FAILED %s" % name in got[0][2] and "FAILED %s" % name not in got[1][2]    # rank 0 speaks for the job, once per file
        assert "3 file(s) failed" in got[0][2]
        return
    a, b = set(got[0][1]), set(got[1][1])
    assert not (a & b) and a | b == set(good)
    assert got[0][2] == got[1][2]
    failed = [f for fs, _ in got[0][2] for f in fs]
    assert sorted(n for n, _ in failed) == sorted(bad)
    why = dict(failed)
    assert "data chunk" in why["bad_header.wav"] and "too short" in why["bad_short.wav"] and why["bad_decode.wav"]
    assert [t for _, ts in got[0][2] for t in ts] == [("truncated.wav", 4000, 2900)]


def test_a_failing_batch_is_reissued_row_by_row(tmp_path):
    """The device stage raises for every batch that contains file `utt003` and for that file alone: the rows that shared
    its batch (and the batches in flight behind it) are still written, utt003 is listed with the device's reason."""
    from voicefixer_amd import api

    class Refuses(api.VoiceFixer):
        calls = []

        def __init__(self):
            pass

        def restore_batches(self, batches, your_vocoder_func=None, streams=2, mode=0):
            held = []
            for tag, kind, host, lens in batches:           # one batch 'in flight' behind the one that is finished,
                held.append((tag, host, lens))              # as the real generator keeps them
                if len(held) > 1:
                    t, h, l = held.pop(0)
                    Refuses.calls.append(list(t))
                    if 3 in t:
                        raise RuntimeError("vfx_conv1d_f32 failed with code 1")
                    yield t, -h, list(l)
            for t, h, l in held:
                Refuses.calls.append(list(t))
                if 3 in t:
                    raise RuntimeError("vfx_conv1d_f32 failed with code 1")
                yield t, -h, list(l)

    ind, outd = str(tmp_path / "in"), str(tmp_path / "out")
    lens = _make_ragged_folder(ind, 13, seed=2)
    st = {}
    names = Refuses().restore_folder(ind, outd, batch_size=4, io_threads=2, stats=st)
    assert sorted(names) == sorted(n for n in lens if n != "utt003.wav") == sorted(os.listdir(outd))
    assert st["failed"] == [("utt003.wav", "RuntimeError: vfx_conv1d_f32 failed with code 1")]
    assert [3] in Refuses.calls and st["files"] == 12


def test_skip_existing_resumes_a_folder_job(tmp_path):
    from voicefixer_amd import api

    class Stub(_StubDevice, api.VoiceFixer):
        pass

    ind, outd = str(tmp_path / "in"), str(tmp_path / "out")
    lens = _make_ragged_folder(ind, 9, seed=4)
    assert len(Stub().restore_folder(ind, outd, batch_size=4, io_threads=2)) == 9
    for gone in ("utt002.wav", "utt007.wav"):
        os.remove(os.path.join(outd, gone))
    before = {n: os.stat(os.path.join(outd, n)).st_mtime_ns for n in os.listdir(outd)}
    st = {}
    again = Stub().restore_folder(ind, outd, batch_size=4, io_threads=2, stats=st, skip_existing=True)
    assert again == ["utt002.wav", "utt007.wav"] and len(st["skipped"]) == 7 and st["failed"] == []
    assert all(os.stat(os.path.join(outd, n)).st_mtime_ns == t for n, t in before.items())
    assert sorted(os.listdir(outd)) == sorted(lens)
    # two ranks, sequentially, second look at a folder the first already wrote into: the deal does not depend on the outputs
    for gone in ("utt001.wav", "utt002.wav", "utt003.wav", "utt004.wav"):
        os.remove(os.path.join(outd, gone))
    a = Stub().restore_folder(ind, outd, batch_size=4, io_threads=2, rank=0, world=2, skip_existing=True)
    b = Stub().restore_folder(ind, outd, batch_size=4, io_threads=2, rank=1, world=2, skip_existing=True)
    assert sorted(a + b) == ["utt001.wav", "utt002.wav", "utt003.wav", "utt004.wav"]


def test_io_thread_default_and_cpu_slices():
    assert vdist.default_io_threads(1, cores=128) == 8 and vdist.default_io_threads(8, cores=128) == 8
    assert vdist.default_io_threads(8, cores=64) == 4 and vdist.default_io_threads(8, cores=8) == 2
    assert vdist.pin_rank_cpus(0, 1) is None
    # a PINNED rank takes half of its own slice (round 5 divided the slice by the world size again: 32 // 16 = 2 workers)
    try:
        vdist._PINNED = list(range(32))
        assert vdist.default_io_threads(8) == 8
        vdist._PINNED = list(range(6))
        assert vdist.default_io_threads(8) == 3
    finally:
        vdist._PINNED = None


def _smt(c):      # a host whose logical CPUs n/2 .. n-1 are the SMT siblings of 0 .. n/2-1 (Linux's numbering on the GPU boxes)
    n = len(os.sched_getaffinity(0)) // 2 * 2
    return "%d,%d" % (c % (n // 2), c % (n // 2) + n // 2) if c < n else str(c)


def _pin_worker(q, smt):
    import torch as t
    mine = vdist.pin_rank_cpus(1, 2, _smt if smt else None)
    q.put((mine, sorted(os.sched_getaffinity(0)), t.get_num_threads(), vdist.default_io_threads(2)))


@pytest.mark.parametrize("smt", [False, True])
def test_pin_rank_cpus_binds_the_slice_and_sizes_the_thread_pools(smt):
    """Round 6: the slice is cut in PHYSICAL cores.  With the SMT numbering of the GPU boxes (siblings of 0..N-1 are N..2N-1) a
    contiguous cut of logical ids gave rank 1 the hardware threads of rank 0's cores."""
    cores = sorted(os.sched_getaffinity(0))
    if len(cores) < 8:
        pytest.skip("needs eight cores")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_pin_worker, args=(q, smt))
    p.start()
    mine, aff, nthreads, io = q.get(timeout=120)
    p.join(60)
    assert mine == aff
    if smt:
        n = len(cores) // 2 * 2
        phys = [(cores[i], cores[i + n // 2]) for i in range(n // 2)]       # what _smt describes (cores = 0..n-1 here)
        lo, hi = vdist.shard_range(len(phys), 1, 2)
        assert mine == sorted(c for pr in phys[lo:hi] for c in pr)
        assert nthreads == hi - lo and io == max(2, (hi - lo) // 2)
        # ... and rank 0's slice shares no physical core with it
        lo0, hi0 = vdist.shard_range(len(phys), 0, 2)
        assert not {c for pr in phys[lo0:hi0] for c in pr} & set(mine)
    else:
        lo, hi = vdist.shard_range(len(cores), 1, 2)
        assert mine == cores[lo:hi] and nthreads == min(len(mine), vdist.MAX_RANK_TORCH_THREADS) and io == max(2, min(8, len(mine) // 2))
    assert vdist.physical_cores([0, 1, 2, 3, 4, 5, 6, 7], lambda c: "%d,%d" % (c % 4, c % 4 + 4)) == [(0, 4), (1, 5), (2, 6), (3, 7)]
    assert vdist.physical_cores([1, 5, 6], lambda c: "%d,%d" % (c % 4, c % 4 + 4)) == [(1, 5), (6,)]


def _scan_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        scanned = [(1000, None), (2000, None), (None, "RuntimeError: not a RIFF/WAVE file: c.wav"), (4000, None)]
        if rank == 1:
            scanned[1] = (None, "OSError: [Errno 5] Input/output error")     # a transient read error on ONE rank
            scanned[3] = (3900, None)                                        # (and a header read while the file was growing)
        q.put((rank, vdist.agree_on_scan(scanned)))
    finally:
        dist.destroy_process_group()


def test_ranks_agree_on_the_scan_before_they_deal():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_scan_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert got[0] == got[1] == [(1000, None), (None, "rank 1: OSError: [Errno 5] Input/output error"),
                                (None, "RuntimeError: not a RIFF/WAVE file: c.wav"), (3900, None)]
    assert vdist.agree_on_scan([(5, None)]) == [(5, None)]       # no process group: unchanged


def test_a_failing_batch_source_is_reported_not_dropped(tmp_path):
    """the iterator that FEEDS the device stage raises at batch 3 of 6 -- the three batches already handed
    over are yielded, then BatchSourceError is raised (round 5 returned quietly: yielded [0, 1, 2], failed [], exit 0)."""
    from voicefixer_amd import api

    class Stub(_StubDevice, api.VoiceFixer):
        pass

    def source():
        for b in range(6):
            if b == 3:
                raise MemoryError("pinned staging for batch 3")
            yield [b], "ragged", torch.full((1, 8), float(b)), [8]

    failed, seen = [], []
    with pytest.raises(api.BatchSourceError, match="MemoryError: pinned staging for batch 3"):
        for tag, out, lens in Stub()._restore_batches_isolated(source(), failed, None, 2, 0):
            seen.append(tag[0])
    assert seen == [0, 1, 2] and failed == []


def test_folder_job_accounts_for_every_file_when_staging_fails(tmp_path, monkeypatch):
    """A staging block that cannot be allocated costs the files of THAT batch (listed in stats["failed"]); if the source of
    batches dies altogether, every file it never reached is listed too -- written + failed + skipped always cover the deal."""
    from voicefixer_amd import api

    class Stub(_StubDevice, api.VoiceFixer):
        n_alloc = 0

        @staticmethod
        def _pin_memory():
            Stub.n_alloc += 1
            if Stub.n_alloc == 2:
                raise MemoryError("cannot pin 2 GB")
            return False

    ind, outd = str(tmp_path / "in"), str(tmp_path / "out")
    lens = _make_ragged_folder(ind, 12, seed=5)
    st = {}
    names = Stub().restore_folder(ind, outd, batch_size=4, io_threads=2, stats=st)
    lost = [f for f, why in st["failed"]]
    assert len(names) == 8 and len(lost) == 4 and sorted(names + lost) == sorted(lens)
    assert all("staging for a batch of 4 x" in why and "cannot pin 2 GB" in why for _, why in st["failed"])

    class Dies(_StubDevice, api.VoiceFixer):
        def _restore_batches_isolated(self, items, failed, your_vocoder_func, streams, mode):
            it = iter(items)
            first = next(it)
            yield first[0], -first[2], list(first[3])
            raise api.BatchSourceError("the batch source failed after OSError: worker pool gone")

    st = {}
    names = Dies().restore_folder(ind, str(tmp_path / "out2"), batch_size=4, io_threads=2, stats=st)
    assert len(names) == 4 and len(st["failed"]) == 8 and sorted(names + [f for f, _ in st["failed"]]) == sorted(lens)
    assert all(why.startswith("not processed: the batch source failed after OSError") for _, why in st["failed"])


def test_a_lying_wav_header_cannot_size_the_staging(tmp_path):
    """The RIFF data chunk promises 400 MB, the file holds 2000 samples: the planning length is what the file can hold."""
    import numpy as np
    import struct
    from scipy.io import wavfile
    from voicefixer_amd import audio_io
    p = str(tmp_path / "liar.wav")
    wavfile.write(p, 44100, np.zeros(2000, np.int16))
    raw = bytearray(open(p, "rb").read())
    k = raw.index(b"data")
    raw[k + 4:k + 8] = struct.pack("<I", 400_000_000)
    open(p, "wb").write(bytes(raw))
    assert audio_io.wav_length(p) == 2000
