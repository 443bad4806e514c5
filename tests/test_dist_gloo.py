"""The N>1 path on CPU: world_size-2 (and 3) gloo process groups exercise the frame sharding and the
scatter/gather that are the only collectives of the multi-GPU path (ffmpeg_amd/dist.py)."""
import os
import socket

import numpy as np
import pytest

from ffmpeg_amd import dist as D


def test_shard_ranges_cover_exactly_once():
    for n in (0, 1, 2, 7, 8, 255, 256, 512):
        for world in (1, 2, 3, 4, 8):
            seen = []
            for r in range(world):
                lo, hi = D.shard_range(n, r, world)
                assert 0 <= lo <= hi <= n
                seen += list(range(lo, hi))
            assert seen == list(range(n))
            assert sum(D.shard_sizes(n, world)) == n
            assert max(D.shard_sizes(n, world)) <= -(-n // world) if n else True


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_frames, q):
    import torch
    import torch.distributed as dist
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    D.init_process_group("gloo")
    try:
        shape = (6, 10)
        full = torch.arange(n_frames * 60, dtype=torch.int32).reshape((n_frames,) + shape) if rank == 0 else \
            torch.empty((0,) + shape, dtype=torch.int32)
        shard = D.scatter_batch(full, n_frames)
        lo, hi = D.shard_range(n_frames, rank, world)
        assert shard.shape[0] == hi - lo
        if hi > lo:
            assert int(shard[0, 0, 0]) == lo * 60                 # the right frames arrived, in order
        # stand-in for the per-rank kernel launch: a per-frame function of the frame alone
        res = (shard.to(torch.int64) * 3 + 1).sum(dim=(1, 2), keepdim=False).reshape(-1, 1)
        out = D.gather_batch(res, n_frames)
        # the max-over-ranks timing reduction bench.py uses
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        assert float(t) == world
        if rank == 0:
            want = (full.to(torch.int64) * 3 + 1).sum(dim=(1, 2)).reshape(-1, 1)
            assert torch.equal(out, want)
            q.put("ok")
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_frames", [(2, 7), (2, 1), (3, 8), (2, 256)])
def test_scatter_process_gather_gloo(world, n_frames):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_frames, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) == "ok"
