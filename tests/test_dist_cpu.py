"""The multi-process path (image sharding, host gather, timing reduction) with 2 gloo ranks on CPU."""
import os
import socket
import sys

import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_items, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    from ocr_amd import dist as D
    r, w, dist = D.init_from_env(backend='gloo')
    assert (r, w) == (rank, world)
    mine = D.shard_indices(n_items, r, w)
    # a "page result" per item: what the pipeline returns per page (boxes + strings), here synthetic
    local = [{'page': i, 'texts': ['w%d_%d' % (i, k) for k in range(i % 3)], 'rank': r} for i in mine]
    dist.barrier()
    full = D.gather_in_order(local, n_items, dist)
    slowest = D.max_over_ranks(1.0 + rank, dist)
    dist.barrier()
    q.put((rank, mine, [f['page'] for f in full], [f['rank'] for f in full], slowest))
    dist.destroy_process_group()


def test_two_rank_sharding_and_gather():
    world, n_items = 2, 7
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_items, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    got.sort()
    assert got[0][1] == [0, 2, 4, 6] and got[1][1] == [1, 3, 5]           # disjoint, complete
    for rank, mine, pages, ranks, slowest in got:
        assert pages == list(range(n_items))                               # gathered in page order on every rank
        assert ranks == [i % world for i in range(n_items)]
        assert slowest == 2.0                                              # MAX over ranks


def _queue_worker(rank, world, port, n_items, q):
    """Dynamic assignment with uneven costs: rank 1 needs 8x as long per page batch as rank 0."""
    import time
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    from ocr_amd import dist as D
    grp = D.init_from_env(backend='gloo')
    queue = D.PageQueue(n_items, grp)
    grp.dist.barrier()
    local = []
    for i in queue:
        time.sleep(0.16 if rank == 1 else 0.02)
        local.append((i, {'page': i, 'texts': ['w%d' % i] * (i % 4), 'rank': rank}))
    full = D.gather_results(local, n_items, grp, dst=0)
    grp.dist.barrier()
    q.put((rank, queue.taken, None if full is None else [(f['page'], f['rank']) for f in full]))
    grp.dist.destroy_process_group()


def test_two_rank_dynamic_queue_balances_uneven_costs():
    world, n_items = 2, 27
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_queue_worker, args=(r, world, port, n_items, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, taken0, full), (_, taken1, none) = got
    assert none is None                                                    # only the destination rank holds the result
    assert sorted(taken0 + taken1) == list(range(n_items))                 # every batch exactly once
    assert len(taken0) >= 2 * len(taken1) >= 2                             # the fast rank took most of the work
    assert [p for p, _ in full] == list(range(n_items))                    # gathered in page order
    assert [r for _, r in full] == [0 if i in taken0 else 1 for i in range(n_items)]


def _retry_worker(rank, world, port, n_items, q):
    """Rank 1 fails item 3 (its GPU call 'raises'); rank 0 must end up with it.  Then a second queue on the same group
    starts at zero, and an item nobody can process aborts the queue on every rank."""
    import time
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    from ocr_amd import dist as D
    grp = D.init_from_env(backend='gloo')
    queue = D.PageQueue(n_items, grp)
    grp.dist.barrier()
    local, failed = [], []
    while True:
        i = queue.take()
        if i is None:
            break
        time.sleep(0.01)
        # item 3 fails on whichever rank gets it first, and only there
        if i == 3 and not grp.store.add('test/failed3', 0):
            grp.store.add('test/failed3', 1)
            failed.append(i)
            queue.give_back(i)
            continue
        local.append((i, {'page': i, 'rank': rank}))
        queue.done(i)
    full = D.gather_results(local, n_items, grp, dst=0)
    grp.dist.barrier()
    # a second queue of the same job: its own keys, starts at zero
    second = D.PageQueue(5, grp)
    assert second.key != queue.key
    got2 = list(second)
    full2 = D.gather_results([(i, i * i) for i in got2], 5, grp, dst=0)
    grp.dist.barrier()
    # an item that fails everywhere: QueueAborted on every rank instead of a hang
    third = D.PageQueue(2, grp)
    aborted = False
    try:
        for i in third:
            if i == 1:
                third.give_back(i)
    except D.QueueAborted:
        aborted = True
    grp.dist.barrier()
    q.put((rank, failed, queue.taken, None if full is None else [(f['page'], f['rank']) for f in full], full2, aborted))
    grp.dist.destroy_process_group()


def test_a_failed_batch_goes_back_to_the_queue_and_another_rank_takes_it():
    world, n_items = 2, 9
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_retry_worker, args=(r, world, port, n_items, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, failed0, taken0, full, full2, ab0), (_, failed1, taken1, none, none2, ab1) = got
    assert none is None and none2 is None
    assert sorted(failed0 + failed1) == [3]                                # exactly one rank failed it
    loser = 0 if failed0 else 1
    assert [p for p, _ in full] == list(range(n_items))                    # nothing lost, nothing twice
    assert dict(full)[3] == 1 - loser                                      # the OTHER rank processed the failed batch
    assert (taken0 if loser == 0 else taken1).count(3) == 1
    assert full2 == [0, 1, 4, 9, 16]
    assert ab0 and ab1


def _pipelined_worker(rank, world, port, n_items, q):
    """A consumer like the benchmark's: an item is finished two takes after it was taken (a software pipeline), so it
    must never WAIT for the queue while items are in flight: non-blocking takes while it has work, one blocking take
    when it has drained."""
    import time
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    from ocr_amd import dist as D
    grp = D.init_from_env(backend='gloo')
    queue = D.PageQueue(n_items, grp)
    grp.dist.barrier()
    local, pending = [], []
    while True:
        inflight = []
        while True:
            i = pending.pop() if pending else queue.take(block=False)
            if i is None:
                break
            inflight.append(i)
            time.sleep(0.01 * (1 + rank))
            if len(inflight) > 2:                     # the pipeline completes the oldest item
                j = inflight.pop(0)
                local.append((j, rank))
                queue.done(j)
        for j in inflight:                            # end of the feed: the pipeline drains
            local.append((j, rank))
            queue.done(j)
        i = queue.take(block=True)
        if i is None:
            break
        pending.append(i)
    full = D.gather_results(local, n_items, grp, dst=0)
    grp.dist.barrier()
    q.put((rank, len(local), None if full is None else len(full)))
    grp.dist.destroy_process_group()


def test_pipelined_consumers_finish():
    world, n_items = 2, 15
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_pipelined_worker, args=(r, world, port, n_items, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got[0][1] + got[1][1] == n_items and got[0][2] == n_items and got[1][2] is None


def test_gather_results_rejects_lost_and_duplicate_items():
    import pytest
    from ocr_amd import dist as D
    assert D.gather_results([(1, 'b'), (0, 'a')], 2, None) == ['a', 'b']
    with pytest.raises(RuntimeError):
        D.gather_results([(0, 'a')], 2, None)
    with pytest.raises(RuntimeError):
        D.gather_results([(0, 'a'), (0, 'b')], 2, None)
    q = D.PageQueue(3)
    assert list(q) == [0, 1, 2] and q.take() is None
    # single process: a returned item comes back once, a second failure of the same item gives up
    q = D.PageQueue(2)
    assert q.take() == 0
    q.give_back(0)
    assert q.take() == 1 and q.take() == 0
    with pytest.raises(D.QueueAborted):
        q.give_back(0)


def test_single_process_is_a_no_op():
    from ocr_amd import dist as D
    for k in ('RANK', 'WORLD_SIZE'):
        os.environ.pop(k, None)
    assert D.init_from_env() == (0, 1, None)
    assert D.shard_indices(5, 0, 1) == [0, 1, 2, 3, 4]
    assert D.gather_in_order(['a', 'b'], 2, None) == ['a', 'b']
    assert D.max_over_ranks(3.5, None) == 3.5
