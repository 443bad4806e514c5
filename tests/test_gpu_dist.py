"""N ranks == 1 rank, byte for byte, on the REAL kernels (SURVEY.md §4 last row): two processes share cuda:0 (the GPU box has
one device; the process group is gloo, so the collectives carry host tensors — on a multi-GPU node the same code runs one rank per
GPU over RCCL, see bench.py), rank 0 holds the batch: scatter -> libffhip batch entry points on the shard -> gather, compared with
the single-process result over the whole batch.  Scaler frames (the exact-2x kernel, odd shard sizes), IDCT block lists, and the
full search over a sequence with its 1-frame halo."""
import os
import socket

import numpy as np
import pytest

from ffmpeg_amd import dist as D

pytestmark = pytest.mark.gpu

SW, SH = 192, 108
PW, PH = 128, 64


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _convert(torch, S, y, uv):
    n = y.shape[0]
    ctx = S.SwsContext(SW, SH, 23, 2 * SW, 2 * SH, 23, S.SWS_BICUBIC)
    dst = [torch.zeros((n, 2 * SH, 2 * SW), dtype=torch.uint8, device="cuda:0"),
           torch.zeros((n, SH, 2 * SW), dtype=torch.uint8, device="cuda:0")]
    if n:
        ctx.scale_batch([y.cuda(), uv.cuda()], dst)
        torch.cuda.synchronize()
    ctx.close()
    return dst[0].cpu(), dst[1].cpu()


def _idct(torch, h264, planes, coefs):
    n = planes.shape[0]
    d = planes.cuda().reshape(n * PH, PW).contiguous()
    by, bx = np.meshgrid(np.arange(n * PH // 8), np.arange(PW // 8), indexing="ij")
    offs = torch.from_numpy((by * 8 * PW + bx * 8).astype(np.int32).ravel()).cuda()
    c = coefs.cuda().reshape(-1, 64).contiguous()
    if n:
        h264.idct_add_batch(h264.IDCT8, d, PW, offs, c)
        torch.cuda.synchronize()
    return d.reshape(n, PH, PW).cpu()


def _esa(torch, me, frames):
    k = frames.shape[0]
    nmb = (PW // 16) * (PH // 16)
    npair = max(k - 1, 0)
    mv = torch.zeros((npair, nmb * 2), dtype=torch.int16, device="cuda:0")
    cost = torch.zeros((npair, nmb), dtype=torch.int32, device="cuda:0")
    if npair:
        f = frames.cuda()
        me.esa_batch(f[1:].contiguous(), f[:-1].contiguous(), PW, PH, PW, PW * PH, npair, 16, 7, me.SAD, mv, cost)
        torch.cuda.synchronize()
    return mv.cpu(), cost.cpu()


def _worker(rank, world, port, n_frames, q):
    import torch
    import torch.distributed as dist
    from ffmpeg_amd import swscale as S, h264, me
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    D.init_process_group("gloo")
    try:
        root = rank == 0
        g = torch.Generator().manual_seed(11)
        y = torch.randint(0, 256, (n_frames, SH, SW), dtype=torch.uint8, generator=g)
        uv = torch.randint(0, 256, (n_frames, SH // 2, SW), dtype=torch.uint8, generator=g)
        planes = torch.randint(0, 256, (n_frames, PH, PW), dtype=torch.uint8, generator=g)
        coefs = torch.randint(-300, 300, (n_frames, (PH // 8) * (PW // 8), 64), dtype=torch.int16, generator=g)
        seq = torch.randint(0, 256, (n_frames, PH, PW), dtype=torch.uint8, generator=g)
        for f in range(1, n_frames):
            seq[f] = torch.roll(seq[f - 1], (f % 5 - 2, 2 - f % 4), (0, 1))

        def T(a):
            return a if root else torch.empty((0,) + tuple(a.shape[1:]), dtype=a.dtype)

        oy, ouv = _convert(torch, S, D.scatter_batch(T(y), n_frames), D.scatter_batch(T(uv), n_frames))
        gy, guv = D.gather_batch(oy, n_frames), D.gather_batch(ouv, n_frames)
        gp = D.gather_batch(_idct(torch, h264, D.scatter_batch(T(planes), n_frames), D.scatter_batch(T(coefs), n_frames)), n_frames)
        mv, cost = _esa(torch, me, D.scatter_frames_for_pairs(T(seq), n_frames))
        gmv, gco = D.gather_batch(mv, n_frames - 1), D.gather_batch(cost, n_frames - 1)
        if root:
            wy, wuv = _convert(torch, S, y, uv)
            assert torch.equal(gy, wy) and torch.equal(guv, wuv)
            assert torch.equal(gp, _idct(torch, h264, planes, coefs))
            wmv, wco = _esa(torch, me, seq)
            assert torch.equal(gmv, wmv) and torch.equal(gco, wco)
            q.put("ok")
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_frames", [(2, 7), (3, 8)])
def test_n_ranks_equal_one_rank_on_the_kernels(world, n_frames):
    import torch
    import torch.multiprocessing as mp
    assert torch.cuda.is_available()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_frames, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    assert q.get(timeout=5) == "ok"
