"""Sharding of a batch of independent JPEG images over the GPUs of one node.

Images share nothing on the data path: every rank decodes its own shard with its own
decoder, stream and HBM slabs, and there is no collective on the data path.  torch.distributed is
used only to agree on the wall-clock of a step (barrier + MAX of the elapsed time) and to sum the
work done (SUM of the pixel counts); both helpers work on any backend (gloo on CPU, nccl = RCCL on GPU).
"""


def partition(costs, world_size):
    """Greedy longest-processing-time split of image indices by cost (scan bytes or pixels).
    Returns a list of `world_size` index lists; deterministic for equal inputs on every rank."""
    order = sorted(range(len(costs)), key=lambda i: (-costs[i], i))
    shards = [[] for _ in range(world_size)]
    load = [0] * world_size
    for i in order:
        r = min(range(world_size), key=lambda k: (load[k], k))
        shards[r].append(i)
        load[r] += costs[i]
    for s in shards:
        s.sort()
    return shards


def init_group(backend, local_rank=None, timeout_s=120.0):
    """Joins the process group the launcher's environment describes (RANK / WORLD_SIZE / MASTER_*) with an EXPLICIT
    timeout, and returns the backend that is really in use.  "nccl" (= RCCL on ROCm) is initialised eagerly on the rank's
    own device, so that a rank that cannot reach its peers fails here, within the timeout, and not inside the first
    barrier of a timed region; if RCCL cannot be brought up at all the group falls back to gloo — the data path has no
    collective, only the clock agreement needs a group — and the caller's JSON line says which one it was."""
    import datetime
    import sys
    import torch
    import torch.distributed as dist
    to = datetime.timedelta(seconds=float(timeout_s))
    if backend == "nccl":
        try:
            dist.init_process_group(backend="nccl", timeout=to, device_id=torch.device("cuda", int(local_rank or 0)))
            return "nccl"
        except Exception as e:
            print("shard: RCCL init failed (%s); using gloo for the barrier / clock agreement" % e, file=sys.stderr)
            if dist.is_initialized():
                dist.destroy_process_group()
            backend = "gloo"
    dist.init_process_group(backend=backend, timeout=to)
    return backend


def ranks_seen(device=None):
    """How many ranks actually take part: an all-reduce of ones.  1 without a process group."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return 1
    t = torch.ones(1, dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return int(t.item())


def fail_fast(what, fn, *args, **kw):
    """Runs a collective; a failure (a peer that is gone, a timeout) ends THIS process with a non-zero code at once —
    no clean-up that could block on the broken group — so that the launcher sees the job fail within seconds of the
    group's timeout instead of hanging in teardown."""
    import os
    import sys
    try:
        return fn(*args, **kw)
    except BaseException as e:  # noqa: BLE001 - anything a backend throws, SystemExit of a watchdog included
        print("shard: %s failed: %s: %s" % (what, type(e).__name__, e), file=sys.stderr)
        sys.stderr.flush()
        sys.stdout.flush()
        os._exit(3)


def agree(elapsed_s, units, device=None, alone_too=False):
    """(max over ranks of elapsed_s, sum over ranks of units).  No-op without a process group, and for a group of one rank
    unless `alone_too` (a rehearsal of the collectives on a one-GPU box)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and not alone_too):
        return float(elapsed_s), float(units)
    t = torch.tensor([float(elapsed_s)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    u = torch.tensor([float(units)], dtype=torch.float64, device=device)
    dist.all_reduce(u, op=dist.ReduceOp.SUM)
    return float(t.item()), float(u.item())


def gather_per_rank(rank, device_index, elapsed_s, steps, device=None, alone_too=False):
    """Every rank's own step time, for rank 0's JSON line: [{"rank", "device", "ms_per_step"}, ...] in rank order.  MAX-of-time
    (agree) hides which rank was the slow one; a straggler shows here.  One all-gather of three numbers, outside the timed region."""
    import torch
    import torch.distributed as dist
    mine = [float(rank), float(device_index), float(elapsed_s) / max(1, int(steps)) * 1e3]
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and not alone_too):
        rows = [mine]
    else:
        t = torch.tensor(mine, dtype=torch.float64, device=device)
        out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
        dist.all_gather(out, t)
        rows = [[float(x) for x in o.tolist()] for o in out]
    return [{"rank": int(r[0]), "device": int(r[1]), "ms_per_step": round(r[2], 4)} for r in rows]
