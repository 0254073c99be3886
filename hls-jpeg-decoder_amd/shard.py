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


def agree(elapsed_s, units, device=None):
    """(max over ranks of elapsed_s, sum over ranks of units).  No-op without a process group."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(elapsed_s), float(units)
    t = torch.tensor([float(elapsed_s)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    u = torch.tensor([float(units)], dtype=torch.float64, device=device)
    dist.all_reduce(u, op=dist.ReduceOp.SUM)
    return float(t.item()), float(u.item())
