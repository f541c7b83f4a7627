"""CPU, world_size 2 (gloo): the sharding + all-gather path used for N > 1 GPUs."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import dimx  # noqa: F401
    from dimx import dist as dd
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    r, w, _ = dd.init_from_env("gloo")
    assert (r, w) == (rank, world)
    B = 7                                    # odd on purpose: shards of 4 and 3 rows
    full = torch.arange(B * 5, dtype=torch.int32).view(B, 5)
    lo, hi = dd.shard_bounds(B, r, w)
    got = dd.all_gather_rows(full[lo:hi].clone())
    ok1 = torch.equal(got, full)
    even = torch.arange(8 * 3, dtype=torch.float32).view(8, 3)
    lo, hi = dd.shard_bounds(8, r, w)
    ok2 = torch.equal(dd.all_gather_rows(even[lo:hi].clone()), even)
    mx = dd.max_over_ranks(float(rank + 1))
    dd.barrier()
    q.put((rank, ok1, ok2, mx))
    dist.destroy_process_group()


def test_shard_and_allgather_world2():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    for rank, ok1, ok2, mx in res:
        assert ok1 and ok2 and mx == 2.0


def test_shard_bounds_partition():
    sys.path.insert(0, ROOT)
    import dimx  # noqa: F401
    from dimx import dist as dd
    for n in (1, 7, 256, 2048):
        for w in (1, 2, 3, 8):
            b = [dd.shard_bounds(n, r, w) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n and all(b[i][1] == b[i + 1][0] for i in range(w - 1))
