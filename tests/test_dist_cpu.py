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


def test_single_process_is_a_no_op():
    from ocr_amd import dist as D
    for k in ('RANK', 'WORLD_SIZE'):
        os.environ.pop(k, None)
    assert D.init_from_env() == (0, 1, None)
    assert D.shard_indices(5, 0, 1) == [0, 1, 2, 3, 4]
    assert D.gather_in_order(['a', 'b'], 2, None) == ['a', 'b']
    assert D.max_over_ranks(3.5, None) == 3.5
