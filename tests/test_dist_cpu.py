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
