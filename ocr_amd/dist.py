"""Image-wise sharding over the GPUs of a node: one process per GPU, no collective in the data
path.  Pages are independent (detection, NMS and recognition never look at another page), so a
rank processes its own pages and the per-page results (a few KB: boxes + strings) are gathered on
the host with `all_gather_object`; the only other collectives are a barrier and a MAX all-reduce
used for timing.  Works with the `nccl` (= RCCL) backend on GPUs and with `gloo` on CPU.
"""
import os

import torch


def init_from_env(backend=None, device=None):
    """Initialises torch.distributed from RANK / WORLD_SIZE / MASTER_* when launched by torchrun.
    Returns (rank, world, dist-or-None)."""
    if 'RANK' not in os.environ or 'WORLD_SIZE' not in os.environ:
        return 0, 1, None
    import torch.distributed as dist
    if not dist.is_initialized():
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        kwargs = {}
        if backend == 'nccl' and device is not None:
            kwargs['device_id'] = device
        dist.init_process_group(backend, **kwargs)
    return dist.get_rank(), dist.get_world_size(), dist


def shard_indices(n_items, rank, world):
    """Round-robin assignment of items to ranks (item i -> rank i % world)."""
    return list(range(rank, n_items, world))


def gather_in_order(local_results, n_items, dist):
    """local_results: the results of shard_indices(n_items, rank, world) in that order.
    Returns the full list in item order on every rank (host gather of small objects)."""
    if dist is None:
        return list(local_results)
    world = dist.get_world_size()
    parts = [None] * world
    dist.all_gather_object(parts, list(local_results))
    out = [None] * n_items
    for r, part in enumerate(parts):
        for k, i in enumerate(range(r, n_items, world)):
            out[i] = part[k]
    return out


def max_over_ranks(value, dist, device=None):
    """MAX all-reduce of a host float (the elapsed time of the slowest rank)."""
    if dist is None:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device if device is not None else 'cpu')
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
