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


_stores = []


def _serve_store(addr, world, tries=16):
    """Rank 0: opens the job's key-value store on a port the system picks and returns (store, port).  The port is the
    one the server really listens on (binding port 0 inside the server: nothing can take it between a probe and the
    listen); a torch without that feature gets a probed port and a few retries when somebody else was faster."""
    import torch.distributed as dist
    try:
        store = dist.TCPStore(addr, 0, world, is_master=True, wait_for_workers=False)
        if int(store.port) > 0:
            return store, int(store.port)
    except (RuntimeError, ValueError, TypeError):
        pass
    last = None
    for _ in range(tries):
        with socket.socket() as s:
            s.bind((addr if addr not in ('localhost',) else '127.0.0.1', 0))
            port = s.getsockname()[1]
        try:
            return dist.TCPStore(addr, port, world, is_master=True, wait_for_workers=False), port
        except RuntimeError as e:                 # address in use: somebody took the port after the probe
            last = e
    raise RuntimeError('could not open a TCPStore on %s: %s' % (addr, last))


class _stdout_to_stderr:
    """The gloo transport announces its connections on the C-level stdout; a job whose stdout is a result line keeps
    it clean by pointing file descriptor 1 at stderr while the process groups are being set up."""

    def __enter__(self):
        import sys
        sys.stdout.flush()
        self._saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        os.dup2(self._saved, 1)
        os.close(self._saved)


def init_from_env(backend=None, device=None):
    """Initialises torch.distributed from RANK / WORLD_SIZE / MASTER_* when launched by torchrun.
    Returns a `Group` (unpacks as (rank, world, dist-or-None))."""
    if 'RANK' not in os.environ or 'WORLD_SIZE' not in os.environ:
        return Group()
    with _stdout_to_stderr():
        return _init_from_env(backend, device)


def _init_from_env(backend, device):
    import torch.distributed as dist
    if not dist.is_initialized():
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        if backend == 'nccl':
            _init_rccl_or_gloo(dist, device)
        else:
            dist.init_process_group(backend)
    rank, world = dist.get_rank(), dist.get_world_size()
    # host-side collectives (python objects) never go through the GPUs
    host_group = dist.new_group(backend='gloo') if dist.get_backend() != 'gloo' else dist.group.WORLD
    # a store of the job's own for the page queues: rank 0 opens the server first and tells the others the port it
    # listens on
    addr = os.environ.get('MASTER_ADDR', '127.0.0.1')
    store, port = _serve_store(addr, world) if rank == 0 else (None, None)
    box = [port]
    dist.broadcast_object_list(box, src=0, group=host_group)
    if rank != 0:
        store = dist.TCPStore(addr, int(box[0]), world, is_master=False)
    _stores.append(store)                     # the server must outlive a caller that drops the Group
    dist.barrier(group=host_group)            # every client is connected before anybody goes on
    return Group(rank, world, dist, host_group, store)


def _init_rccl_or_gloo(dist, device):
    """RCCL for the job's two bracket collectives (a barrier, a MAX of one double).  The data path has no collective, so
    a node whose RCCL cannot come up (two ranks given the same GPU, a fabric that is down) still runs: every rank then
    sees the same failure while the communicator is being built, and the job's collectives go over gloo on the host
    instead.  (A second rendezvous needs a second port: MASTER_PORT + 1.)"""
    import datetime
    import sys
    kwargs = {'device_id': device} if device is not None else {}
    try:
        dist.init_process_group('nccl', timeout=datetime.timedelta(seconds=300), **kwargs)
        t = torch.zeros(1, device=device if device is not None else 'cuda')
        dist.all_reduce(t)                        # builds the communicator if the call above did not
        torch.cuda.synchronize()
        return
    except Exception as e:                        # noqa: BLE001 - whatever the backend raises, the job can go on without it
        sys.stderr.write('ocr_amd.dist: RCCL did not come up (%s: %s); the bracket collectives use gloo\n'
                         % (type(e).__name__, str(e).splitlines()[0][:200] if str(e) else ''))
    try:
        if dist.is_initialized():
            dist.destroy_process_group()
    except Exception:                             # noqa: BLE001
        pass
    os.environ['MASTER_PORT'] = str(int(os.environ.get('MASTER_PORT', '29500')) + 1)
    dist.init_process_group('gloo')


class QueueAborted(RuntimeError):
    """A work item failed on as many ranks as the job has: the queue gives up on every rank."""


class PageQueue:
    """A shared queue of `n_items` work items (page batches) numbered 0..n_items-1: `take()` hands every item to
    exactly one rank, in order, whichever rank asks first; None when every item is done.  One atomic add on the
    job's store per item (~0.1 ms on loopback against ~150 ms of GPU work per batch of pages).

    A rank that cannot finish an item (its GPU call raised) calls `give_back(i)`: the number goes to a list of
    returned items in the store and another rank takes it; `done(i)` reports a finished item.  A rank whose first-come
    items are used up keeps polling the returned list until all `n_items` are reported done, so a returned item
    always finds a taker.  `for i in queue:` reports `done` for an item when the loop body comes back for the next
    one without having given it back.  An item that fails on `max_attempts` ranks (default: every rank once, at
    least 2) aborts the queue on all ranks (`QueueAborted`).

    Every queue of a job has keys of its own in the store (a per-group sequence number, the same on all ranks as
    long as they create their queues in the same order), so a second queue starts at zero."""

    POLL_S = 0.005

    def __init__(self, n_items, group=None, key=None, max_attempts=None):
        self.n_items = int(n_items)
        self.store = group.store if group is not None else None
        self.rank = group.rank if group is not None else 0
        world = group.world if group is not None else 1
        if key is None:
            seq = getattr(group, '_queue_seq', 0) if group is not None else 0
            if group is not None:
                group._queue_seq = seq + 1
            key = 'pages%d' % seq
        self.key = key
        self.max_attempts = int(max_attempts) if max_attempts else max(2, world)
        self._local = 0
        self._local_returned = []
        self._local_done = 0
        self._slot = None          # index into the returned list this rank has reserved
        self._mine = set()         # items this rank gave back: it leaves them to the others
        self._attempts = {}
        self.taken = []
        self.given_back = []

    # -- store helpers (single process: plain attributes)
    def _add(self, name, n):
        return int(self.store.add('%s/%s' % (self.key, name), n))

    def take(self, block=True):
        """The next item, or None.  block=True: None means every item of the queue is done (the call waits, polling the
        list of returned items, while other ranks still work).  block=False: None also when nothing is to be had right
        now - for a caller that must first finish what it has in flight (a software pipeline) before it may wait."""
        if self.store is None:
            if self._local < self.n_items:
                i = self._local
                self._local += 1
            elif self._local_returned:
                i = self._local_returned.pop(0)
            else:
                return None
            self.taken.append(i)
            return i
        i = self._add('next', 1) - 1
        if i < self.n_items:
            self.taken.append(i)
            return i
        import time
        while True:
            if self._add('abort', 0):
                raise QueueAborted('PageQueue %s: an item failed on %d ranks' % (self.key, self.max_attempts))
            if self._slot is None and self._add('rtake', 0) < self._add('nret', 0):
                self._slot = self._add('rtake', 1) - 1
            if self._slot is not None and self._slot < self._add('nret', 0):
                i = int(self.store.get('%s/ret/%d' % (self.key, self._slot)))
                self._slot = None
                if i in self._mine:                    # my own failure: somebody else's turn (not counted as an attempt)
                    self._put_back(i)
                    time.sleep(4 * self.POLL_S)
                    continue
                self.taken.append(i)
                return i
            if self._add('done', 0) >= self.n_items or not block:
                return None
            time.sleep(self.POLL_S)

    def _put_back(self, i):
        slot = self._add('nret', 1) - 1
        self.store.set('%s/ret/%d' % (self.key, slot), str(int(i)))

    def give_back(self, i):
        """This rank could not finish item i: another rank takes it."""
        self.given_back.append(i)
        self._mine.add(i)
        if self.store is None:
            self._attempts[i] = self._attempts.get(i, 0) + 1
            if self._attempts[i] >= self.max_attempts:
                raise QueueAborted('PageQueue %s: item %d failed %d times' % (self.key, i, self._attempts[i]))
            self._local_returned.append(i)
            return
        if self._add('att/%d' % i, 1) >= self.max_attempts:
            self._add('abort', 1)
            raise QueueAborted('PageQueue %s: item %d failed on %d ranks' % (self.key, i, self.max_attempts))
        self._put_back(i)

    def done(self, i):
        if self.store is None:
            self._local_done += 1
        else:
            self._add('done', 1)

    def __iter__(self):
        while True:
            i = self.take()
            if i is None:
                return
            before = len(self.given_back)
            yield i
            if len(self.given_back) == before:
                self.done(i)


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
    on_host = device is None or dist.get_backend() == 'gloo'
    t = torch.tensor([float(value)], dtype=torch.float64, device='cpu' if on_host else device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
