"""Multi-GPU folder inference: one process per GPU (``torch.distributed``; backend "nccl" is
RCCL over xGMI on ROCm, "gloo" for the CPU tests).

The path shards embarrassingly: utterances are independent, so each rank
restores a contiguous block of the (length-sorted) work list with NO collective in the data
path.  RCCL is used only for the trivial movement of the batch when the caller holds all
inputs on rank 0: one scatter of inputs and one gather of outputs per job.
"""
import numpy as np
import torch
import torch.distributed as dist


def shard_range(n_items, rank, world):
    """Contiguous block of rank ``rank``: sizes differ by at most one, blocks cover [0, n)."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def rank_world(rank=None, world=None):
    """(rank, world) of a sharded job: the arguments if given, else the initialised ``torch.distributed`` group, else (0, 1)."""
    if world is None:
        if dist.is_available() and dist.is_initialized():
            return dist.get_rank(), dist.get_world_size()
        return 0, 1
    rank = 0 if rank is None else int(rank)
    if not 0 <= rank < int(world):
        raise ValueError("rank %d outside world of %d" % (rank, world))
    return rank, int(world)


def deal_files(costs, world):
    """Deal items with the given costs (samples of every file of a folder) to ``world`` ranks: longest first, each to
    the rank with the smallest total so far (ties: fewer items, then lower rank).  Returns owner[i].  Every rank
    computes the same answer from the same header lengths -- no exchange; totals differ by at most one item's cost
    (LPT rule), so a ragged 5..30 s folder loads all ranks alike, and an equal-length folder is dealt round-robin.
    (``shard_range`` -- contiguous blocks -- stays what the in-memory scatter uses, where all rows are equally long.)"""
    import heapq
    owner = [0] * len(costs)
    heap = [(0, 0, r) for r in range(int(world))]
    for i in sorted(range(len(costs)), key=lambda i: (-costs[i], i)):
        total, count, r = heapq.heappop(heap)
        owner[i] = r
        heapq.heappush(heap, (total + max(int(costs[i]), 0), count + 1, r))
    return owner


def gather_counters(values, device=None, group=None):
    """The ONE collective of a sharded folder job: every rank contributes a flat list of floats (its counters), every
    rank gets the list of all ranks' lists.  ``device``: where the exchange tensor lives (the rank's GPU for RCCL)."""
    t = torch.tensor([float(v) for v in values], dtype=torch.float64, device=device)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return [t.tolist()]
    allt = [torch.zeros_like(t) for _ in range(dist.get_world_size(group))]
    dist.all_gather(allt, t, group=group)
    return [x.tolist() for x in allt]


def gather_objects(obj, group=None):
    """All ranks' picklable ``obj`` as a list, on every rank (the folder job's failure / skip lists: names and reasons
    are not numbers).  World size 1 / no group: ``[obj]``."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return [obj]
    out = [None] * dist.get_world_size(group)
    dist.all_gather_object(out, obj, group=group)
    return out


_PINNED = None      # the core slice pin_rank_cpus gave this process (None: not pinned)


def default_io_threads(world=1, cores=None):
    """Decode / encode workers per rank of a folder job, between 2 and 8: half of the cores this RANK has.  After
    ``pin_rank_cpus`` that is half of the rank's own slice (the slice is already cores / ranks-on-this-node: dividing it by the
    world size again, as round 5 did, left every multi-rank job at the minimum of 2); unpinned, the host's cores /
    (2 * ranks on this node) -- LOCAL_WORLD_SIZE when a launcher set it, else ``world``.  Eight ranks x eight workers would
    fight over a 64-core host while one rank's workers need about 0.3 worker-seconds per 2560 s of audio
    (profiles/r04_folder_256x10s.json).  ``cores``: the host's core count, for tests (treated as unpinned)."""
    import os
    if cores is None and _PINNED is not None:
        return max(2, min(8, len(physical_cores(_PINNED)) // 2))
    if cores is None:
        try:
            cores = len(os.sched_getaffinity(0))
        except (AttributeError, OSError):
            cores = os.cpu_count() or 8
    try:
        local_world = int(os.environ.get("LOCAL_WORLD_SIZE", "") or world)
    except ValueError:
        local_world = world
    return max(2, min(8, cores // (2 * max(1, int(local_world)))))


def physical_cores(cpus, read_siblings=None):
    """Group the logical CPUs ``cpus`` by physical core (sysfs ``topology/thread_siblings_list``): a sorted list of tuples, one
    per core, each holding that core's hardware threads that are in ``cpus``.  Without sysfs every CPU is its own core.
    ``read_siblings(cpu) -> "0,128"`` replaces the sysfs read (tests)."""
    seen, cores = set(), []
    allowed = set(cpus)
    for c in sorted(cpus):
        if c in seen:
            continue
        sib = [c]
        try:
            if read_siblings is not None:
                txt = read_siblings(c)
            else:
                with open("/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list" % c) as f:
                    txt = f.read().strip()
            sib = []
            for part in txt.split(","):
                a, _, b = part.partition("-")
                sib.extend(range(int(a), int(b or a) + 1))
            sib = sorted(x for x in sib if x in allowed) or [c]
        except (OSError, ValueError):
            sib = [c]
        seen.update(sib)
        cores.append(tuple(sib))
    return sorted(cores)


MAX_RANK_TORCH_THREADS = 16    # ATen's intra-op pool of a rank: this path's host-side torch work (weight packing at load, batch staging)
                               # is many small operators -- 16 threads is the fastest count on the GPU boxes' 2 x 64-core host
                               # (bench.py cpu_baseline.thread_sweep: 16 beats 32, 64 and 128), and a rank's slice is smaller anyway


def pin_rank_cpus(local_rank, local_world, read_siblings=None):
    """Give rank ``local_rank`` of ``local_world`` on this node its own slice of the cores this process may run on, so the ranks'
    interpreter threads and I/O workers do not migrate across each other's caches.  The slice is cut in PHYSICAL cores (both
    hardware threads of a core go to the same rank): Linux numbers the SMT siblings of cores 0..N-1 as N..2N-1, so a contiguous
    cut of the logical ids -- what round 5 did -- hands rank 1 of 2 exactly the siblings of rank 0's cores, and two OpenMP pools
    spinning on the same physical cores made the weight packing at load take minutes instead of seconds (measured on the
    2 x 64-core GPU box: `bench.py --gpus 2 --oversubscribe` 7 min -> see profiles/README.md, round 6).
    Call it BEFORE ``init_process_group`` and before the first torch operator: ``sched_setaffinity(0, ...)`` binds the calling
    thread, and only threads created AFTERWARDS (gloo / RCCL progress threads, ATen's intra-op pool, the I/O workers) inherit the
    mask; ``torch.set_num_threads`` is set to the slice's physical cores, at most MAX_RANK_TORCH_THREADS.
    Returns the slice (sorted list of logical core ids), or None when there is nothing to split (one rank, fewer physical cores
    than 2 per rank, no affinity API)."""
    import os
    global _PINNED
    try:
        cpus = sorted(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        return None
    local_world = int(local_world)
    cores = physical_cores(cpus, read_siblings)
    if local_world <= 1 or len(cores) < 2 * local_world:
        return None
    lo, hi = shard_range(len(cores), int(local_rank), local_world)
    mine = sorted(c for core in cores[lo:hi] for c in core)
    try:
        os.sched_setaffinity(0, mine)
    except OSError:
        return None
    _PINNED = mine
    try:
        torch.set_num_threads(max(1, min(hi - lo, MAX_RANK_TORCH_THREADS)))
    except RuntimeError:
        pass
    return mine


def agree_on_scan(scanned, group=None):
    """Every rank scanned the folder's headers on its own; a transient read error on ONE rank (a network file system) would give
    the ranks different work lists and different deals -- files restored twice or by nobody.  All-gather the per-rank
    ``[(length | None, reason | None), ...]`` lists and keep, for every file, rank 0's answer unless some rank could not read
    it (then the file is unusable for everybody, with that rank's reason): all ranks deal from the same list.  World size 1 /
    no process group: the list comes back unchanged."""
    every = gather_objects(list(scanned), group)
    if len(every) == 1:
        return list(scanned)
    if any(len(e) != len(every[0]) for e in every):
        raise RuntimeError("the ranks list different folders (%s files)" % ", ".join(str(len(e)) for e in every))
    out = []
    for i in range(len(every[0])):
        bad = next(((r, e[i][1]) for r, e in enumerate(every) if e[i][1] is not None), None)
        if bad is not None:
            out.append((None, bad[1] if bad[0] == 0 else "rank %d: %s" % bad))
        else:
            out.append((min(e[i][0] for e in every), None))
    return out


def free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def exec_ranks(nproc, target, env_extra=None):
    """Become the launcher of an N-rank single-node job: replace this process by ``python -m torch.distributed.run
    --nnodes=1 --nproc-per-node nproc --master-addr 127.0.0.1 --master-port <free> <target...>`` (``target``: a script
    path or ``["-m", module]`` followed by its arguments).  One process per GPU; HSA_ENABLE_IPC_MODE_LEGACY=0 selects the
    dmabuf IPC this host driver needs for RCCL across processes.  Does not return."""
    import os
    import sys
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // (2 * nproc))))
    env.update(env_extra or {})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port())] + list(target)
    sys.stdout.flush()
    sys.stderr.flush()
    os.execve(sys.executable, cmd, env)


def scatter_utterances(wavs_on_root, n_samples, device, group=None, root=0):
    """Rank ``root`` holds a float32 tensor (n_utt, n_samples); every rank receives its block
    (shard_range) as a device tensor.  Point-to-point sends: xGMI is a full mesh, a ring buys
    nothing for a one-shot fan-out."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    meta = torch.zeros(1, dtype=torch.int64, device=device)
    if rank == root:
        meta[0] = wavs_on_root.shape[0]
    dist.broadcast(meta, src=root, group=group)
    n_utt = int(meta.item())
    lo, hi = shard_range(n_utt, rank, world)
    mine = torch.empty((hi - lo, n_samples), dtype=torch.float32, device=device)
    if rank == root:
        reqs = []
        for r in range(world):
            a, b = shard_range(n_utt, r, world)
            if r == root:
                mine.copy_(wavs_on_root[a:b])
            elif b > a:
                reqs.append(dist.isend(wavs_on_root[a:b].to(device).contiguous(), dst=r, group=group))
        for q in reqs:
            q.wait()
    elif hi > lo:
        dist.recv(mine, src=root, group=group)
    return mine, (lo, hi), n_utt


def gather_utterances(mine, n_utt, n_samples, device, group=None, root=0):
    """Inverse of scatter_utterances: rank ``root`` returns (n_utt, n_samples), others None."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    if rank == root:
        out = torch.empty((n_utt, n_samples), dtype=torch.float32, device=device)
        for r in range(world):
            a, b = shard_range(n_utt, r, world)
            if r == root:
                out[a:b].copy_(mine)
            elif b > a:
                buf = torch.empty((b - a, n_samples), dtype=torch.float32, device=device)
                dist.recv(buf, src=r, group=group)
                out[a:b].copy_(buf)
        return out
    if mine.shape[0] > 0:
        dist.send(mine.contiguous(), dst=root, group=group)
    return None


def restore_sharded(restore_fn, wavs_on_root, n_samples, device, batch_size=32, group=None, root=0, timing=None):
    """scatter -> each rank runs ``restore_fn(batch (b, n_samples)) -> (b, n_samples)`` over its block
    in batches of ``batch_size`` -> gather on ``root``.  ``restore_fn`` is Pipeline.restore on GPUs.
    ``timing`` (optional dict) receives this rank's ``scatter_s`` / ``compute_s`` / ``gather_s`` wall times (each
    phase ends with a device synchronise when the tensors live on a GPU)."""
    import time

    def sync():
        if timing is not None and torch.device(device).type == "cuda":
            torch.cuda.synchronize(device)

    t0 = time.perf_counter()
    mine, _, n_utt = scatter_utterances(wavs_on_root, n_samples, device, group, root)
    sync()
    t1 = time.perf_counter()
    outs = [restore_fn(mine[i:i + batch_size]) for i in range(0, mine.shape[0], batch_size)]
    local = torch.cat(outs, 0) if outs else mine
    sync()
    t2 = time.perf_counter()
    out = gather_utterances(local, n_utt, n_samples, device, group, root)
    sync()
    if timing is not None:
        timing.update(scatter_s=t1 - t0, compute_s=t2 - t1, gather_s=time.perf_counter() - t2)
    return out
