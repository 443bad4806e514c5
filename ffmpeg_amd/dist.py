"""Frame-batch sharding over the GPUs of one node: one process per GPU, torch.distributed ("nccl" == RCCL
over xGMI on ROCm; "gloo" for the CPU test-suite).

The hot path has no cross-frame dependency: frames / macroblock lists / transforms are cut
into contiguous per-rank ranges and every rank runs the single-GPU batch entry points on its own range.
There is no reduction anywhere, so no all-reduce: the only collectives are the optional scatter of a batch
that originates on rank 0 and the gather of the results (for the motion search: 8 bytes per macroblock).
"""
import os


def shard_range(n_items, rank, world):
    """Contiguous block of ceil(n/world) items per rank (the last ranks may get fewer or none)."""
    per = -(-n_items // world) if world > 0 else n_items
    lo = min(rank * per, n_items)
    return lo, min(lo + per, n_items)


def shard_sizes(n_items, world):
    return [shard_range(n_items, r, world)[1] - shard_range(n_items, r, world)[0] for r in range(world)]


def shard_frame_pairs(n_frames, rank, world):
    """Motion search over a sequence: pair p searches frame p+1 in frame p, p in [0, n_frames-1).
    Pairs shard contiguously like everything else; a rank's pairs [plo, phi) read frames [plo, phi] — its own range plus ONE
    halo frame (the first reference frame of the next rank is this rank's last current frame).
    Returns (plo, phi, flo, fhi): the pair range and the half-open frame range to hold (empty when the rank has no pair)."""
    plo, phi = shard_range(max(n_frames - 1, 0), rank, world)
    if phi <= plo:
        return plo, phi, plo, plo
    return plo, phi, plo, phi + 1


def init_process_group(backend=None, device=None):
    """Reads RANK / WORLD_SIZE / MASTER_* from the environment (torchrun); 127.0.0.1 rendezvous by default."""
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1:
        return rank, world
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if not dist.is_initialized():
        kw = {"device_id": device} if (backend == "nccl" and device is not None) else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world


def scatter_ranges(full, ranges, src=0):
    """Rank `src` holds `full` ([n, ...]); rank r returns full[ranges[r][0]:ranges[r][1]] (ranges may overlap: halos).
    Implemented with point-to-point sends (xGMI is point-to-point: root egress is the bound, 7 links x ~153 GB/s)."""
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(), dist.get_world_size()
    lo, hi = ranges[rank]
    if rank == src:
        reqs = []
        for r in range(world):
            rlo, rhi = ranges[r]
            if r != src and rhi > rlo:
                reqs.append(dist.isend(full[rlo:rhi].contiguous(), dst=r))
        mine = full[lo:hi].clone()
        for q in reqs:
            q.wait()
        return mine
    meta = full  # non-root passes a template tensor of the per-item shape/dtype/device
    out = torch.empty((hi - lo,) + tuple(meta.shape[1:]), dtype=meta.dtype, device=meta.device)
    if hi > lo:
        dist.recv(out, src=src)
    return out


def scatter_batch(full, n_items, src=0):
    """Rank `src` holds `full` ([n_items, ...]); every rank returns its own [hi-lo, ...] shard (shard_range)."""
    import torch.distributed as dist
    world = dist.get_world_size()
    return scatter_ranges(full, [shard_range(n_items, r, world) for r in range(world)], src)


def scatter_frames_for_pairs(full, n_frames, src=0):
    """Frames of a sequence for the motion search: every rank gets the frames of its pairs, halo frame included."""
    import torch.distributed as dist
    world = dist.get_world_size()
    return scatter_ranges(full, [shard_frame_pairs(n_frames, r, world)[2:] for r in range(world)], src)


def gather_batch(shard, n_items, dst=0):
    """Inverse of scatter_batch: rank `dst` returns the [n_items, ...] concatenation in frame order, others None."""
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(), dist.get_world_size()
    if rank != dst:
        if shard.shape[0] > 0:
            dist.send(shard.contiguous(), dst=dst)
        return None
    out = torch.empty((n_items,) + tuple(shard.shape[1:]), dtype=shard.dtype, device=shard.device)
    for r in range(world):
        rlo, rhi = shard_range(n_items, r, world)
        if rhi <= rlo:
            continue
        if r == dst:
            out[rlo:rhi] = shard
        else:
            dist.recv(out[rlo:rhi], src=r)
    return out
