"""The N>1 path on CPU: world_size-2 (and 3) gloo process groups run the multi-GPU data path end to end — rank 0 holds the
batch, scatter -> per-rank processing -> gather (ffmpeg_amd/dist.py: the only collectives of the path) — with the ORACLE standing
in for the per-rank kernels (same per-item functions, so sharding bugs show as byte differences), and compare with the
unsharded result: scaler frames, IDCT block lists (whole planes per rank), full-search frame pairs with their 1-frame halo."""
import ctypes as C
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


def test_frame_pair_shards_carry_one_halo_frame():
    for n in (0, 1, 2, 3, 9, 512):
        for world in (1, 2, 3, 8):
            pairs = []
            for r in range(world):
                plo, phi, flo, fhi = D.shard_frame_pairs(n, r, world)
                pairs += list(range(plo, phi))
                if phi > plo:
                    assert (flo, fhi) == (plo, phi + 1) and fhi <= n      # frames p and p + 1 of every pair p are held
                else:
                    assert fhi == flo
            assert pairs == list(range(max(n - 1, 0)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


SW, SH = 32, 16            # scaler frames: nv12 32x16 -> 64x32
PW, PH = 64, 32            # planes of the IDCT / motion-search cases


def _sws_tables():
    import ffi
    from ffmpeg_amd import swscale as S
    ht = S.HostTables(SW, SH, ffi.PIX["nv12"], 2 * SW, 2 * SH, ffi.PIX["nv12"], ffi.SWS_BICUBIC)
    return ffi.make_otables(SW, SH, ffi.PIX["nv12"], 2 * SW, 2 * SH, ffi.PIX["nv12"], ffi.SWS_BICUBIC, ht.banks(), ht.coeffs())


def _scale_frames(t, y, uv):
    """oracle scaler over [n, rows, cols] numpy planes"""
    import ffi
    n = y.shape[0]
    oy, ouv = np.zeros((n, 2 * SH, 2 * SW), np.uint8), np.zeros((n, SH, 2 * SW), np.uint8)
    for f in range(n):
        sp, ss = ffi.planes([np.ascontiguousarray(y[f]), np.ascontiguousarray(uv[f])])
        dy, duv = np.zeros((2 * SH, 2 * SW), np.uint8), np.zeros((SH, 2 * SW), np.uint8)
        dp, ds = ffi.planes([dy, duv])
        assert ffi.oracle().ffo_sws_scale_frame(C.byref(t), sp, ss, dp, ds) == 2 * SH
        oy[f], ouv[f] = dy, duv
    return oy, ouv


def _idct_planes(planes, coefs):
    """oracle idct8_add of every 8x8 block of [n, PH, PW] planes; coefs [n, blocks, 64] are consumed"""
    import ffi
    out = planes.copy()
    for f in range(out.shape[0]):
        i = 0
        for by in range(PH // 8):
            for bx in range(PW // 8):
                blk = np.ascontiguousarray(coefs[f, i])
                ffi.oracle().ffo_h264_idct8_add(ffi.ptr(out[f, by * 8:, bx * 8:]), ffi.ptr(blk, ffi.i16p), PW)
                i += 1
    return out


def _esa_pairs(frames):
    """oracle full search of frame p + 1 in frame p for the pairs inside `frames` ([k, PH, PW]) -> mv [k-1, nmb, 2], cost"""
    import ffi
    k = frames.shape[0]
    nmb = (PW // 16) * (PH // 16)
    mv, cost = np.zeros((max(k - 1, 0), nmb, 2), np.int16), np.zeros((max(k - 1, 0), nmb), np.uint32)
    for p in range(k - 1):
        cur, ref = np.ascontiguousarray(frames[p + 1]), np.ascontiguousarray(frames[p])
        ffi.oracle().ffo_me_esa_frame(ffi.ptr(cur), ffi.ptr(ref), PW, PW, PH, 16, 3, 0, ffi.ptr(mv[p], ffi.i16p),
                                      cost[p].ctypes.data_as(C.POINTER(C.c_uint32)))
    return mv, cost


def _worker(rank, world, port, n_frames, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import torch
    import torch.distributed as dist
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    D.init_process_group("gloo")
    try:
        rng = np.random.default_rng(5)                                   # every rank could build the batch; only rank 0 uses it
        y = rng.integers(0, 256, (n_frames, SH, SW), dtype=np.uint8)
        uv = rng.integers(0, 256, (n_frames, SH // 2, SW), dtype=np.uint8)
        planes = rng.integers(0, 256, (n_frames, PH, PW), dtype=np.uint8)
        coefs = rng.integers(-300, 300, (n_frames, (PH // 8) * (PW // 8), 64)).astype(np.int16)
        seq = rng.integers(0, 256, (n_frames, PH, PW), dtype=np.uint8)
        for f in range(1, n_frames):
            seq[f] = np.roll(seq[f - 1], (f % 3 - 1, 1 - f % 2), (0, 1))
        root = rank == 0

        def T(a, item_shape, dt):
            return torch.from_numpy(a) if root else torch.empty((0,) + item_shape, dtype=dt)

        lo, hi = D.shard_range(n_frames, rank, world)
        t = _sws_tables()
        # 1. scaler frames
        sy = D.scatter_batch(T(y, (SH, SW), torch.uint8), n_frames)
        suv = D.scatter_batch(T(uv, (SH // 2, SW), torch.uint8), n_frames)
        assert sy.shape[0] == hi - lo
        oy, ouv = _scale_frames(t, sy.numpy(), suv.numpy())
        gy = D.gather_batch(torch.from_numpy(oy), n_frames)
        guv = D.gather_batch(torch.from_numpy(ouv), n_frames)
        # 2. IDCT block lists, whole planes per rank
        sp_ = D.scatter_batch(T(planes, (PH, PW), torch.uint8), n_frames)
        sc_ = D.scatter_batch(T(coefs, coefs.shape[1:], torch.int16), n_frames)
        gp = D.gather_batch(torch.from_numpy(_idct_planes(sp_.numpy(), sc_.numpy())), n_frames)
        # 3. full search over the sequence: pairs shard, frames travel with one halo frame
        plo, phi, flo, fhi = D.shard_frame_pairs(n_frames, rank, world)
        fr = D.scatter_frames_for_pairs(T(seq, (PH, PW), torch.uint8), n_frames)
        assert fr.shape[0] == fhi - flo
        mv, cost = _esa_pairs(fr.numpy())
        assert mv.shape[0] == phi - plo
        gmv = D.gather_batch(torch.from_numpy(mv), max(n_frames - 1, 0))
        gco = D.gather_batch(torch.from_numpy(cost.view(np.int32)), max(n_frames - 1, 0))
        # the max-over-ranks timing reduction bench.py uses
        tt = torch.tensor([float(rank + 1)], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        assert float(tt) == world
        if root:
            wy, wuv = _scale_frames(t, y, uv)
            assert np.array_equal(gy.numpy(), wy) and np.array_equal(guv.numpy(), wuv)
            assert np.array_equal(gp.numpy(), _idct_planes(planes, coefs))
            wmv, wco = _esa_pairs(seq)
            assert np.array_equal(gmv.numpy(), wmv) and np.array_equal(gco.numpy().view(np.uint32), wco)
            q.put("ok")
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_frames", [(2, 7), (2, 1), (3, 8), (2, 2)])
def test_scatter_process_gather_gloo(world, n_frames):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_frames, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert q.get(timeout=5) == "ok"
