"""The N>1 path: videos / replica streams are independent (no data-path collective), only the timing barrier and the
max-over-ranks reduction of bench.py use torch.distributed.  Covered here with world_size-2 gloo processes on CPU."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import bench
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        videos = [f'v{i}' for i in range(7)]
        lengths = [50, 10, 40, 30, 20, 60, 5]
        mine = bench.shard_videos(videos, lengths, rank, world)
        elapsed = 1.0 + rank                                    # rank 1 is the slow one
        t = bench.max_over_ranks(elapsed, torch.device('cpu'))
        frames = bench.sum_over_ranks(100, torch.device('cpu'))
        q.put((rank, mine, t, frames))
    finally:
        dist.destroy_process_group()


def test_sharding_and_reduction_world2():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, v0, t0, f0), (r1, v1, t1, f1) = res
    assert set(v0) | set(v1) == {f'v{i}' for i in range(7)} and not (set(v0) & set(v1))
    assert v0[0] == 'v5' and v1[0] == 'v0'                      # longest-first, dealt round-robin
    assert t0 == t1 == 2.0 and f0 == f1 == 200


def test_shard_single_process():
    sys.path.insert(0, ROOT)
    import bench
    assert bench.shard_videos(['a', 'b', 'c'], [1, 3, 2], 0, 1) == ['b', 'c', 'a']
    assert bench.max_over_ranks(1.5, torch.device('cpu')) == 1.5
