"""Image-wise sharding over the GPUs of a node: one process per GPU, no collective in the data
path.  Pages are independent (detection, NMS and recognition never look at another page), so a
rank processes the page batches it takes and the per-page results (a few KB: boxes + strings) are
gathered on the host; the only other collectives are a barrier and a MAX all-reduce used for
timing.  Works with the `nccl` (= RCCL) backend on GPUs and with `gloo` on CPU.

Assignment is dynamic: post-NMS word counts (and with them the recogniser's work) vary from page to
page, so ranks take batches from one shared queue (`PageQueue`, an atomic counter in a
torch.distributed TCPStore on rank 0) instead of owning a fixed slice.  The reference has no
counterpart: it runs one image at a time in one process (utils/threading/stream.py:227-272).
"""
import os
import socket

import torch


class Group:
    """What `init_from_env` returns: rank, world size, the torch.distributed module (None when single process),
    a gloo group for host-side object gathers and the shared key-value store of the job."""

    def __init__(self, rank=0, world=1, dist=None, host_group=None, store=None):
        self.rank, self.world, self.dist, self.host_group, self.store = rank, world, dist, host_group, store

    def __iter__(self):                      # (rank, world, dist) = init_from_env(...)
        return iter((self.rank, self.world, self.dist))

    def __eq__(self, other):
        return tuple(self) == tuple(other)


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def init_from_env(backend=None, device=None):
    """Initialises torch.distributed from RANK / WORLD_SIZE / MASTER_* when launched by torchrun.
    Returns a `Group` (unpacks as (rank, world, dist-or-None))."""
    if 'RANK' not in os.environ or 'WORLD_SIZE' not in os.environ:
        return Group()
    import torch.distributed as dist
    if not dist.is_initialized():
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        kwargs = {}
        if backend == 'nccl' and device is not None:
            kwargs['device_id'] = device
        dist.init_process_group(backend, **kwargs)
    rank, world = dist.get_rank(), dist.get_world_size()
    # host-side collectives (python objects) never go through the GPUs
    host_group = dist.new_group(backend='gloo') if dist.get_backend() != 'gloo' else dist.group.WORLD
    # a store of the job's own for the page queue: rank 0 serves it on a free port it tells the others
    port = [_free_port() if rank == 0 else None]
    dist.broadcast_object_list(port, src=0, group=host_group)
    store = dist.TCPStore(os.environ.get('MASTER_ADDR', '127.0.0.1'), int(port[0]), world, is_master=(rank == 0),
                          wait_for_workers=True)
    return Group(rank, world, dist, host_group, store)


class PageQueue:
    """A shared queue of `n_items` work items (page batches) numbered 0..n_items-1: `take()` hands every item to
    exactly one rank, in order, whichever rank asks first; None when the queue is empty.  One atomic add on the
    job's store per item (~0.1 ms on loopback against ~70 ms of GPU work per batch of pages)."""

    def __init__(self, n_items, group=None, key='pages'):
        self.n_items = int(n_items)
        self.store = group.store if group is not None else None
        self.key = key
        self._local = 0
        self.taken = []

    def take(self):
        if self.store is None:
            i = self._local
            self._local += 1
        else:
            i = int(self.store.add(self.key, 1)) - 1
        if i >= self.n_items:
            return None
        self.taken.append(i)
        return i

    def __iter__(self):
        while True:
            i = self.take()
            if i is None:
                return
            yield i


def shard_indices(n_items, rank, world):
    """Static round-robin assignment of items to ranks (item i -> rank i % world)."""
    return list(range(rank, n_items, world))


def gather_results(local_pairs, n_items, group, dst=0):
    """local_pairs: [(item index, result), ...] of this rank, any order.  Host gather of the small per-item results
    to rank `dst`, which returns the full list in item order (every item exactly once, or it raises); the other
    ranks return None."""
    if group is None or group.dist is None:
        parts = [list(local_pairs)]
    else:
        dist = group.dist
        parts = [None] * group.world if group.rank == dst else None
        dist.gather_object(list(local_pairs), parts, dst=dst, group=group.host_group)
        if group.rank != dst:
            return None
    out = [None] * n_items
    seen = [False] * n_items
    for part in parts:
        for i, res in part:
            if seen[i]:
                raise RuntimeError('gather_results: item %d was processed twice' % i)
            seen[i] = True
            out[i] = res
    if not all(seen):
        raise RuntimeError('gather_results: items %s were not processed' % [i for i, s in enumerate(seen) if not s][:8])
    return out


def gather_in_order(local_results, n_items, dist):
    """Static sharding: local_results are the results of shard_indices(n_items, rank, world) in that order.
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
