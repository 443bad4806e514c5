#!/usr/bin/env python3
"""bench.py — BASELINE.json's metric on its configuration, on N MI355X GPUs of one node.

  step      one pass of the hot path over one batch: 256 nv12 1920x1080 frames -> nv12 3840x2160,
            SWS_BICUBIC (BASELINE.json configs[1]), inputs and outputs resident in HBM.
  metric    Mpixels/s of OUTPUT pixels, whole job over all ranks (weak scaling: 256 frames per GPU).
  roofline  the dominant kernel k_sws_up2 (one launch per step): algorithmic bytes per launch
            (15,552,000 B/frame x frames) / its average duration measured with HIP
            events on the launch stream, against the 8 TB/s HBM3E peak; `traffic` = 2*FETCH_SIZE +
            WRITE_SIZE of the same kernel from two rocprofv3 PMC passes spawned by this run (or the
            committed profile, labelled); `achievable_GB/s` = the guide's measured 6.29 TB/s (or a
            better streaming probe of this box), the yardstick beside the spec peak.
  idct      BASELINE's second metric (IDCT Gblocks/s at 1/2/4/8 GPUs): h264 idct8_add over 32 4K luma
            planes per rank, summed over ranks, in the same line.
  cpu_baseline  the reference's own C path (oracle/_ref, kind "reference") or the oracle port, timed
            on the host cores over bounded samples of the same workloads (rank 0, N=1 only): swscale on
            1 thread / slice-threaded / frame-parallel on all cores, idct8_add on 1 thread / all cores.

N>1 is launched by torchrun (one process per GPU, RCCL): frames shard across ranks with no data-path
collective; the only collectives are the barriers and the MAX-over-ranks of the timed interval.  With
N>1 the line also carries `strong`: a rank-0 batch scattered over RCCL, converted and gathered back
(ffmpeg_amd/dist.py), each phase timed on its own — the path BASELINE's north_star describes.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

SRC_W, SRC_H, DST_W, DST_H = 1920, 1080, 3840, 2160
NV12 = 23
BYTES_PER_FRAME = SRC_W * SRC_H * 3 // 2 + DST_W * DST_H * 3 // 2   # 15,552,000
HBM_PEAK_GBS = 8000.0                                             # MI355X_MICROARCH.md: 8 TB/s spec
HBM_GUIDE_ACHIEVABLE_GBS = 6290.0                                 # same guide: 6.29 TB/s measured (float4 copy, 79 %)


def usable_cores():
    """what this process may actually run on at once: min(scheduler affinity, the container's CPU quota) — `cores` of every CPU leg is
    capped by it (a 256-thread leg under a 16-CPU cgroup quota is a 16-core measurement)"""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        if os.path.exists("/sys/fs/cgroup/cpu.max"):
            q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
            if q != "max":
                n = min(n, max(1, int(int(q) / int(per) + .5)))
        elif os.path.exists("/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, int(q / per + .5)))
    except (OSError, ValueError):
        pass
    return n


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown CPU"


def cpu_baseline(budget_s=5.0):
    """Bounded samples of the same work on the host cores (checker infrastructure, timed only)."""
    import ffi
    rng = np.random.default_rng(2)
    src = ffi.alloc_frame(NV12, SRC_W, SRC_H, rng)
    dst = ffi.alloc_frame(NV12, DST_W, DST_H)
    sp, ss = ffi.planes(src)
    dp, ds = ffi.planes(dst)
    cores = os.cpu_count() or 1
    px = DST_W * DST_H
    what = "nv12 %dx%d->%dx%d bicubic" % (SRC_W, SRC_H, DST_W, DST_H)
    if not ffi.have_ref():
        from ffmpeg_amd import swscale as S
        ht = S.HostTables(SRC_W, SRC_H, NV12, DST_W, DST_H, NV12, 4)
        t = ffi.make_otables(SRC_W, SRC_H, NV12, DST_W, DST_H, NV12, 4, ht.banks(), ht.coeffs())
        n, t0 = 0, time.perf_counter()
        while True:
            ffi.oracle().ffo_sws_scale_frame(C.byref(t), sp, ss, dp, ds)
            n += 1
            dt = time.perf_counter() - t0
            if (dt > budget_s and n >= 2) or n >= 200:
                break
        return {"value": round(n * px / dt / 1e6, 2), "unit": "Mpixels/s", "cores": 1, "kind": "port",
                "sample": "%d frames %s in %.1f s, 1 thread of %d host cores, pure C (oracle port)" % (n, what, dt, cores)}
    R = ffi.ref()

    def run_sws(threads, budget):
        ctx = R.ffref_sws_create(SRC_W, SRC_H, NV12, DST_W, DST_H, NV12, 4, threads)
        R.ffref_sws_scale(ctx, sp, ss, 0, SRC_H, dp, ds)            # warm-up
        n, t0 = 0, time.perf_counter()
        while True:
            R.ffref_sws_scale(ctx, sp, ss, 0, SRC_H, dp, ds)
            n += 1
            dt = time.perf_counter() - t0
            if (dt > budget and n >= 3) or n >= 2000:
                break
        R.ffref_sws_free(ctx)
        return {"value": round(n * px / dt / 1e6, 2), "unit": "Mpixels/s", "cores": threads,
                "sample": "%d frames %s in %.1f s, %d thread(s)" % (n, what, dt, threads)}

    usable = usable_cores()

    def run_sws_frame(threads, budget):
        """sws_scale_frame() on refcounted frames: the entry through which a threaded context reaches ff_sws_slice_worker on every
        slice thread (libswscale/swscale.c:1405-1420, 1645-1679); sws_scale() on such a context runs slice_ctx[0] alone (:1626-1643)"""
        R.ffref_frame_alloc.restype = C.c_void_p
        R.ffref_frame_alloc.argtypes = [C.c_int] * 3
        R.ffref_frame_plane.restype = C.c_void_p
        R.ffref_frame_plane.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int)]
        R.ffref_sws_scale_frame.argtypes = [C.c_void_p] * 3
        R.ffref_frame_free.argtypes = [C.c_void_p]
        fs, fd = R.ffref_frame_alloc(SRC_W, SRC_H, NV12), R.ffref_frame_alloc(DST_W, DST_H, NV12)
        for i, rows in ((0, SRC_H), (1, SRC_H // 2)):
            ls = C.c_int()
            pl = np.ctypeslib.as_array(C.cast(R.ffref_frame_plane(fs, i, C.byref(ls)), C.POINTER(C.c_uint8)), shape=(rows, ls.value))
            pl[:] = rng.integers(0, 256, pl.shape, dtype=np.uint8)
        ctx = R.ffref_sws_create(SRC_W, SRC_H, NV12, DST_W, DST_H, NV12, 4, threads)
        R.ffref_sws_scale_frame(ctx, fd, fs)                         # warm-up
        n, t0 = 0, time.perf_counter()
        while True:
            R.ffref_sws_scale_frame(ctx, fd, fs)
            n += 1
            dt = time.perf_counter() - t0
            if (dt > budget and n >= 3) or n >= 2000:
                break
        R.ffref_sws_free(ctx)
        R.ffref_frame_free(fs)
        R.ffref_frame_free(fd)
        return {"value": round(n * px / dt / 1e6, 2), "unit": "Mpixels/s", "cores": min(threads, usable), "threads": threads,
                "sample": "%d frames %s in %.1f s, sws_scale_frame on a context with %d slice thread(s)" % (n, what, dt, threads)}

    legs = {"sws_1_thread": run_sws(1, 2.0)}
    if hasattr(R, "ffref_sws_scale_frame"):
        legs["sws_slice_threads"] = run_sws_frame(max(2, min(usable, 64)), 3.0)
    # frame-parallel: one single-threaded context and one frame per thread (a batch of independent frames on all cores)
    if hasattr(R, "ffref_sws_scale_frames_mt"):
        nt = min(cores, 256)
        ctxs = (C.c_void_p * nt)(*[R.ffref_sws_create(SRC_W, SRC_H, NV12, DST_W, DST_H, NV12, 4, 1) for _ in range(nt)])
        dsts = [ffi.alloc_frame(NV12, DST_W, DST_H) for _ in range(nt)]
        dps = [ffi.planes(d)[0] for d in dsts]
        sarr = (C.c_void_p * nt)(*[C.cast(sp, C.c_void_p).value] * nt)
        darr = (C.c_void_p * nt)(*[C.cast(d, C.c_void_p).value for d in dps])
        reps = 2
        R.ffref_sws_scale_frames_mt(ctxs, sarr, ss, darr, ds, SRC_H, nt, 1)
        t0 = time.perf_counter()
        R.ffref_sws_scale_frames_mt(ctxs, sarr, ss, darr, ds, SRC_H, nt, reps)
        dt = time.perf_counter() - t0
        for c in ctxs:
            R.ffref_sws_free(c)
        legs["sws_frame_parallel"] = {"value": round(nt * reps * px / dt / 1e6, 2), "unit": "Mpixels/s", "cores": min(nt, usable), "threads": nt,
                                      "sample": "%d frames %s in %.1f s, %d single-threaded contexts side by side" % (nt * reps, what, dt, nt)}
        del dsts, dps
    # h264 idct8_add over 4K luma planes (129,600 blocks each): 1 thread, then a static split over all cores.  Persistent threads,
    # one untimed warm-up pass, then whole passes for >= 1 s on a clock inside the C runner (oracle/refbuild/ffref_shim.c)
    if hasattr(R, "ffref_h264_idct_batch_timed"):
        for key, th, planes in (("idct8_1_thread", 1, 2), ("idct8_all_cores", min(cores, 1024), 32)):
            n = planes * 129600
            pic = rng.integers(0, 256, (planes * 2160, 3840), dtype=np.uint8)
            by, bx = np.meshgrid(np.arange(planes * 270), np.arange(480), indexing="ij")
            off = (by * 8 * 3840 + bx * 8).astype(np.int32).ravel()
            blk = rng.integers(-512, 512, (n, 64), dtype=np.int16)
            secs, passes = C.c_double(0), C.c_int(0)
            got = R.ffref_h264_idct_batch_timed(1, ffi.ptr(pic), 3840, ffi.ptr(off, ffi.i32p), ffi.ptr(blk, ffi.i16p), n, th, 1.5,
                                                C.byref(secs), C.byref(passes))
            if got > 0 and secs.value > 0:
                legs[key] = {"value": round(n * passes.value / secs.value / 1e9, 5), "unit": "Gblocks/s", "cores": min(got, usable), "threads": got,
                             "sample": "%d passes over %d 8x8 blocks (ff_h264_idct8_add_8_c, %d 4K luma planes) in %.2f s after a warm-up "
                                       "pass, %d persistent thread(s) on thread-local (NUMA-local) copies of their share" % (passes.value, n, planes, secs.value, got)}
            del pic, off, blk
    # BASELINE's other configurations through the reference's own function pointers (oracle/refbuild/ffref_shim.c ffref_bench_leg: persistent
    # threads on thread-local data, one warm-up chunk, free-running chunks for `secs`): 1 thread and one thread per usable core
    if hasattr(R, "ffref_bench_leg"):
        R.ffref_bench_leg.restype = C.c_int
        R.ffref_bench_leg.argtypes = [C.c_int, C.c_int, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        allc = max(1, min(usable, 128))
        for leg, key, unit, div in ((0, "rgb24_4k", "Mpixels/s", 1e6), (1, "h264_qpel16_mixed", "Mpixels/s", 1e6),
                                    (2, "h264_v_loop_filter_luma", "Medges/s", 1e6), (3, "h264_h_loop_filter_luma", "Medges/s", 1e6),
                                    (4, "mdct1024_fwd", "Mtransforms/s", 1e6), (5, "mdct1024_inv", "Mtransforms/s", 1e6),
                                    (6, "me_esa_sad_r7", "MB-searches/s", 1.0), (7, "me_esa_satd_r7", "MB-searches/s", 1.0)):
            for th, suffix in ((1, "1_thread"), (allc, "all_cores")):
                units, secs = C.c_double(0), C.c_double(0)
                got = R.ffref_bench_leg(leg, th, 0.6, C.byref(units), C.byref(secs))
                if got > 0 and secs.value > 0 and units.value > 0:
                    legs["%s_%s" % (key, suffix)] = {"value": float("%.5g" % (units.value / secs.value / div)), "unit": unit, "cores": min(got, usable),
                                                     "threads": got, "sample": "%.4g units in %.2f s" % (units.value, secs.value)}
    best = max((legs[k] for k in legs if k.startswith("sws_")), key=lambda l: l["value"])
    out = {"value": best["value"], "unit": "Mpixels/s", "cores": best["cores"], "kind": "reference",
           "sample": best["sample"] + "; %d usable of %d host cores (%s), pure C (no SIMD asm: nasm absent)" % (usable, cores, cpu_model()),
           "host_cores": cores, "usable_cores": usable, "cpu_model": cpu_model()}
    # flat scalars (the driver's record keeps scalar members only): every leg's rate under <leg>_<unit>
    short = {"Mpixels/s": "Mpix", "Gblocks/s": "Gblocks", "Medges/s": "Medges", "Mtransforms/s": "Mtransforms", "MB-searches/s": "MBsearches"}
    out["legs"] = legs
    for k, l in legs.items():
        out["%s_%s" % (k, short.get(l["unit"], "rate"))] = l["value"]
    # what the box lets this process use: a container's CPU quota / cpuset bounds every sustained all-cores leg (a 9 ms burst is not throttled,
    # 1.5 s are: round 2's one-shot idct leg read 4x the sustained rate on such a box)
    try:
        out["sched_affinity_cores"] = len(os.sched_getaffinity(0))
        for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
            if os.path.exists(f):
                out["cgroup_cpu_limit"] = open(f).read().strip()
                break
    except OSError:
        pass
    return out


def measure_traffic(kname, frames):
    """HBM bytes per launch of the headline kernel, measured NOW: two child runs of this script (--pmc-child: the timed loop's
    launches only) under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` / `... WRITE_SIZE` — separate passes, kernel trace only, as
    MI355X_MICROARCH.md's HBM section prescribes — and 2 x FETCH_SIZE + WRITE_SIZE (KB units; the x2 is the guide's gfx950
    correction for wide coalesced reads).  Returns (bytes, source) or (None, None) when rocprofv3 is not on PATH or a pass fails."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3")
    if not exe:
        return None, None
    env = dict(os.environ, TMPDIR="/tmp")
    vals = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="ffhip_pmc_", dir="/tmp")
        try:
            cmd = [exe, "--kernel-trace", "--pmc", ctr, "--output-format", "csv", "-d", d, "-o", "case", "--", sys.executable,
                   os.path.join(ROOT, "bench.py"), "--pmc-child", "--frames", str(frames)]
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=240)
            got = []
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if kname.split("<")[0] in row["Kernel_Name"] and row["Counter_Name"] == ctr:
                        got.append(float(row["Counter_Value"]))
            if r.returncode != 0 or not got:
                return None, None
            vals[ctr] = sum(got) / len(got)
        except (OSError, subprocess.SubprocessError, KeyError, ValueError):
            return None, None
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return round((2 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024), \
        "measured in this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE, then WRITE_SIZE (2 passes), 2*FETCH + WRITE"


def frame_batches(torch, dev, _lib, src_shapes, dst_shapes, torch_alloc=False):
    """uint8 tensors of the given shapes: source batches in one range of ffhip_frames_alloc(), target batches in another (each plane batch on
    a 2 MiB boundary), or plain torch tensors"""
    if torch_alloc:
        return ([torch.empty(s_, dtype=torch.uint8, device=dev) for s_ in src_shapes], [torch.empty(s_, dtype=torch.uint8, device=dev) for s_ in dst_shapes])
    out = []
    for shapes in (src_shapes, dst_shapes):
        sizes = [int(np.prod(s_)) for s_ in shapes]
        offs, at = [], 0
        for z in sizes:
            offs.append(at)
            at += (z + (2 << 20) - 1) // (2 << 20) * (2 << 20)
        mem = _lib.FrameMemory(at)
        out.append([mem.tensor(s_, o) for s_, o in zip(shapes, offs)])
    return out[0], out[1]


def free_batches(torch, *batches):
    """gives the frame memory behind tensors of frame_batches() back now (torch tensors: to torch's cache)"""
    torch.cuda.synchronize()
    for ts in batches:
        for t in ts:
            m = getattr(t, "_ffhip_frames", None)
            if m is not None:
                m.close()


def place_best(torch, make, measure, free, tries, good_ms):
    """Where a buffer lies in HBM decides between two speeds for the same launch on this part (nv12 -> 4K: 0.58 and 0.63 - 0.67 of the peak; which
    physical pages an allocation gets, not its virtual address: profiles/r06_alloc_vmm_sweep_*.txt, r06_arena_offset_sweep.txt).  A resident
    converter places its pools once, by trial — so does the bench: up to `tries` placements, each measured with a few launches, the fastest
    kept (all are held until the last trial, so every trial gets other pages), stopping at the first that reaches good_ms.
    Returns (placement, [ms of every trial])."""
    cands, seen = [], []
    for _ in range(max(1, tries)):
        cand = make()
        ms = measure(cand)
        cands.append(cand)
        seen.append(ms)
        if ms <= good_ms:
            break
    keep = min(range(len(seen)), key=lambda i: seen[i])
    best = cands[keep]
    for i, c in enumerate(cands):   # (the losers are held until here: a freed range would be handed out again to the next trial)
        if i != keep:
            free(c)
    del cands
    return best, seen


def rgb24_leg(torch, dev, torch_alloc=False, tries=1):
    """north_star's first target: unscaled yuv420p -> rgb24, 3840x2160, 64-frame batch resident in HBM (4.5 B/pixel), HIP events around
    50 launches after 10 untimed ones, and this box's streaming probe at the kernel's own 1 : 2 read : write mix."""
    from ffmpeg_amd import swscale as S, _lib
    out = {}
    ev = lambda: torch.cuda.Event(enable_timing=True)
    n, w, h = 64, 3840, 2160
    bytes_ = n * w * h * 4.5

    def make():
        c = S.SwsContext(w, h, 0, w, h, 2, 4)      # (the context's launch tuner looks at THESE buffers)
        a, b = frame_batches(torch, dev, _lib, [(n, r, cc) for r, cc in S.plane_shapes(0, w, h)], [(n, h, 3 * w)], torch_alloc)
        for t in a:
            t.random_(0, 256)
        return c, a, b

    def measure(cand):
        c, a, b = cand
        for _ in range(12):
            c.scale_batch(a, b)
        torch.cuda.synchronize()      # the warm-up has finished: the context's launch tuner (ffhip_sws_tuned_numbering) can read its eight timed launches
        c.scale_batch(a, b)
        x0, x1 = ev(), ev()
        x0.record()
        for _ in range(10):
            c.scale_batch(a, b)
        x1.record()
        torch.cuda.synchronize()
        return x0.elapsed_time(x1) / 10

    def free(cand):
        cand[0].close()
        free_batches(torch, cand[1], cand[2])
    (ctx, src, dst), trial_ms = place_best(torch, make, measure, free, 1 if torch_alloc else tries, bytes_ / (0.75 * HBM_PEAK_GBS * 1e9) * 1e3)
    e0, e1 = ev(), ev()
    reps = 50
    e0.record()
    for _ in range(reps):
        ctx.scale_batch(src, dst)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    gbs = n * w * h * 4.5 / (ms * 1e-3) / 1e9
    out["yuv420p_rgb24_4k"] = {"Mpixels/s": round(n * w * h / (ms * 1e-3) / 1e6, 1), "GB/s": round(gbs, 1),
                               "hbm_frac": round(gbs / HBM_PEAK_GBS, 4), "frames": n, "ms": round(ms, 4)}
    out["yuv420p_rgb24_4k"]["frac_of_guide_achievable_6290"] = round(gbs / HBM_GUIDE_ACHIEVABLE_GBS, 4)
    # this box's streaming probe at the kernel's own mix (read n / 2, write n: 1.5 B read and 3 B written per pixel) — north_star's 0.70 of
    # the 8 TB/s peak is 5.6 TB/s of a 1 : 2 mix; the probe is the best of 24 ways to issue that traffic with no arithmetic at all
    # (ffhip_membw_probe pattern 4): what the box of this run gives the mix
    g = C.c_double(0)
    if _lib.lib().ffhip_membw_probe(4, 2 << 30, 10, C.byref(g)) == 0:
        out["yuv420p_rgb24_4k"]["box_probe_read1_write2_GB/s"] = round(g.value, 1)
        out["yuv420p_rgb24_4k"]["frac_of_box_probe_read1_write2"] = round(gbs / g.value, 4)
    out["yuv420p_rgb24_4k"]["tuned_numbering"] = ctx.tuned_numbering     # 0 plain, 1 an eighth of the launch per XCD: what this box preferred
    out["yuv420p_rgb24_4k"]["placement_trials"] = len(trial_ms)
    out["yuv420p_rgb24_4k"]["placement_trial_fracs"] = [round(bytes_ / (m * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) for m in trial_ms]
    ctx.close()
    free_batches(torch, src, dst)
    del src, dst
    return out["yuv420p_rgb24_4k"]


def extras(torch, dev, torch_alloc=False):
    """Secondary hot-path kernels, short runs (rank 0, N=1)."""
    from ffmpeg_amd import swscale as S, h264, _lib
    out = {}
    ev = lambda: torch.cuda.Event(enable_timing=True)

    def sws_case(key, sf, sw, sh, df, dw, dh, n):
        c = S.SwsContext(sw, sh, sf, dw, dh, df, 4)
        # (frame batches in libffhip's frame memory, as the headline's: ffhip_frames_alloc)
        s_, d_ = frame_batches(torch, dev, _lib, [(n, r, cc) for r, cc in S.plane_shapes(sf, sw, sh)], [(n, r, cc) for r, cc in S.plane_shapes(df, dw, dh)], torch_alloc)
        for t_ in s_:
            t_.random_(0, 256)
        if sf in (62, 158):   # 10-bit samples: valid ones (planar: the low 10 bits of the word, P010: the high 10)
            for t_ in s_:
                w16 = t_.view(torch.int16)
                w16.bitwise_and_(0x03FF if sf == 62 else -64)
        for _ in range(2):
            c.scale_batch(s_, d_)
        a, b = ev(), ev()
        a.record()
        for _ in range(5):
            c.scale_batch(s_, d_)
        b.record()
        torch.cuda.synchronize()
        t = a.elapsed_time(b) / 5
        byt = n * (S.frame_bytes(sf, sw, sh) + S.frame_bytes(df, dw, dh))
        out[key] = {"Mpixels/s": round(n * dw * dh / (t * 1e-3) / 1e6, 1), "GB/s": round(byt / (t * 1e-3) / 1e9, 1),
                    "hbm_frac": round(byt / (t * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "frames": n, "ms": round(t, 4)}
        c.close()

    # the host-pointer face of the bench conversion (SwsFunc shape: pageable host planes in and out, staging and PCIe included):
    # SURVEY.md §8d's end-to-end figure, reported beside — never as — the resident-frame rate
    hc = S.SwsContext(SRC_W, SRC_H, NV12, DST_W, DST_H, NV12, 4)
    hs = [np.random.default_rng(5).integers(0, 256, (r, c), dtype=np.uint8) for r, c in S.plane_shapes(NV12, SRC_W, SRC_H)]
    hd = [np.zeros((r, c), np.uint8) for r, c in S.plane_shapes(NV12, DST_W, DST_H)]
    hc.scale(hs, hd)
    t0 = time.perf_counter()
    for _ in range(5):
        hc.scale(hs, hd)
    t = (time.perf_counter() - t0) / 5
    out["sws_host_pointer_end_to_end"] = {"Mpixels/s": round(DST_W * DST_H / t / 1e6, 1), "ms_per_frame": round(t * 1e3, 3),
                                          "note": "ffhip_sws_scale on pageable host memory, one frame per call: H2D + kernel + D2H"}
    hc.close()
    # down-scaling (8 x 8-tap banks at exact 2:1: k_sws_down2) and scaled packed-RGB output (column walker + yuv2rgb)
    sws_case("sws_nv12_4k_to_1080p_bicubic", 23, 3840, 2160, 23, 1920, 1080, 64)
    sws_case("sws_yuv420p_1080p_to_rgb24_4k_bicubic", 0, 1920, 1080, 2, 3840, 2160, 32)
    # down-scaling INTO packed RGB (round 5: two stages — the wide-bank walker with an int16 luma plane, then the tables' closed form;
    # was the LDS-tiled kernel at 0.05), and the exact-2x RGB writer at 32 bits per pixel
    sws_case("sws_nv12_4k_to_rgb24_1080p_bicubic", 23, 3840, 2160, 2, 1920, 1080, 32)
    # NV12 into rgb24 at the source's size: what sws_scale() runs for a decoder's frame (no table converter for semi-planar sources: the
    # scaler with one-tap luma and the 4-tap vertical chroma bank; round 5: k_sws_eq_rgb, was the column walker at 0.355)
    sws_case("sws_nv12_1080p_to_rgb24_1080p_bicubic", 23, 1920, 1080, 2, 1920, 1080, 64)
    sws_case("sws_yuv420p_1080p_to_bgra_4k_bicubic", 0, 1920, 1080, 28, 3840, 2160, 32)
    # a J (full-range) source: range conversion between the passes of the exact-2x kernel (round 4; AV_PIX_FMT_YUVJ420P = 12)
    sws_case("sws_yuvj420p_1080p_to_yuv420p_4k_bicubic", 12, 1920, 1080, 0, 3840, 2160, 64)
    # 4:4:4 planar through the exact-2x kernel: three planes of the luma's size (AV_PIX_FMT_YUV444P = 5)
    sws_case("sws_yuv444p_1080p_to_4k_bicubic", 5, 1920, 1080, 5, 3840, 2160, 32)
    # above 8 bits: p010 / yuv420p10 1080p -> 4K (3.75 B per output pixel) through the exact-2x kernel's 16-bit twin, and a ratio that
    # takes the tiled 16-bit scaler (k_sws_scale16)
    sws_case("sws_p010_1080p_to_4k_bicubic", 158, 1920, 1080, 158, 3840, 2160, 64)
    sws_case("sws_yuv420p10_1080p_to_4k_bicubic", 62, 1920, 1080, 62, 3840, 2160, 64)
    sws_case("sws_p010_4k_to_1080p_bicubic", 158, 3840, 2160, 158, 1920, 1080, 16)
    # a 10-bit decoder's frame for an 8-bit consumer (round 5: the 16-bit column walker with the dithered 8-bit output stage; was the tiled
    # k_sws_scale16 at 0.05)
    sws_case("sws_p010_4k_to_nv12_1080p_bicubic", 158, 3840, 2160, 23, 1920, 1080, 16)
    # ... and the ratios that are not exactly 2: the 16-bit column walker (round 4, sws_walk16.hip; was k_sws_scale16)
    sws_case("sws_p010_720p_to_1080p_bicubic", 158, 1280, 720, 158, 1920, 1080, 64)
    sws_case("sws_p010_4k_to_1440p_bicubic", 158, 3840, 2160, 158, 2560, 1440, 16)
    sws_case("sws_yuv420p10_1080p_to_1440p_bicubic", 62, 1920, 1080, 62, 2560, 1440, 32)
    # round 6: the other static periods — 4:3 down above 8 bits (k_sws_down32h's second period) and the 8-bit twin of the 3:2 up-scaler
    sws_case("sws_p010_1440p_to_1080p_bicubic", 158, 2560, 1440, 158, 1920, 1080, 32)
    sws_case("sws_nv12_720p_to_1080p_bicubic", 23, 1280, 720, 23, 1920, 1080, 64)
    # round 5's second survey: conversions at the source's size that sat on general kernels — planar 4:4:4 into RGB (no table converter:
    # the full-chroma writer on one-tap banks, sws_full444.hip), 4:2:0 between its layouts (sws_copy420.hip), 4:4:4 -> 4:2:0 (the luma
    # copied, the chroma on the exact-2:1 kernel) — and the commonest down-scale on the wide walker
    sws_case("sws_yuv444p_1080p_to_rgb24_1080p", 5, 1920, 1080, 2, 1920, 1080, 64)
    sws_case("sws_nv12_1080p_to_yuv420p_1080p", 23, 1920, 1080, 0, 1920, 1080, 64)
    sws_case("sws_yuv444p_1080p_to_yuv420p_1080p", 5, 1920, 1080, 0, 1920, 1080, 64)
    sws_case("sws_nv12_1080p_to_720p_bicubic", 23, 1920, 1080, 23, 1280, 720, 64)
    # round 6: a packed RGB source for an encoder (bgra 1080p -> nv12 at the source's size: the input converters as a kernel, then the 16-bit
    # walker on their 14-bit lines: 5.5 algorithmic B / px, 13.5 moved) and 10-bit video for a display (p010 -> bgra: the walker's first stage
    # into the context's intermediate, then k_y16_rgb)
    sws_case("sws_bgra_1080p_to_nv12_1080p_bicubic", 28, 1920, 1080, 23, 1920, 1080, 32)
    # ... and into a planar 4:4:4 target: every bank the identity, the converter pass writes the target's planes itself (7 B / px)
    sws_case("sws_bgra_1080p_to_yuv444p_1080p", 28, 1920, 1080, 5, 1920, 1080, 32)
    sws_case("sws_p010_1080p_to_bgra_1080p_bicubic", 158, 1920, 1080, 28, 1920, 1080, 32)
    # ... and an 8-bit source for a 10-bit encoder (planes widened to words, then the exact-2x kernel's 16-bit twin; was the tiled kernel at 0.05)
    sws_case("sws_nv12_1080p_to_p010_4k_bicubic", 23, 1920, 1080, 158, 3840, 2160, 32)
    # H.264 8x8 IDCT + add over 32 4K luma planes (129,600 blocks each, 384 B/block)
    planes, stride = 32, 3840
    nb = planes * 129600
    plane = torch.randint(0, 256, (planes * 2160, stride), dtype=torch.uint8, device=dev)
    by, bx = torch.meshgrid(torch.arange(planes * 270, device=dev), torch.arange(480, device=dev), indexing="ij")
    offs = (by * 8 * stride + bx * 8).to(torch.int32).reshape(-1).contiguous()
    coefs0 = torch.randint(-512, 512, (nb, 64), dtype=torch.int16, device=dev)
    coefs = coefs0.clone()
    h264.idct_add_batch(h264.IDCT8, plane, stride, offs, coefs)
    tot = 0.0
    reps = 5
    for _ in range(reps):
        coefs.copy_(coefs0)
        e0, e1 = ev(), ev()
        e0.record()
        h264.idct_add_batch(h264.IDCT8, plane, stride, offs, coefs)
        e1.record()
        torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    ms = tot / reps
    gbs = nb * 384 / (ms * 1e-3) / 1e9
    out["h264_idct8_add"] = {"Gblocks/s": round(nb / (ms * 1e-3) / 1e9, 3), "GB/s": round(gbs, 1),
                             "hbm_frac": round(gbs / HBM_PEAK_GBS, 4), "blocks": nb, "ms": round(ms, 4)}
    del plane, coefs, coefs0, offs
    # HEVC 32x32 (matrix cores), 16x16 and 8x8 inverse transform + add_residual over 4K luma planes (coefficients read + residual written in
    # place + picture read + written: 6 B per sample)
    from ffmpeg_amd import hevc
    for lg, planes in ((5, 16), (4, 16), (3, 8)):
        nsz = 1 << lg
        bw, bh = 3840 // nsz, 2160 // nsz
        ntu = planes * bw * bh
        tus = np.zeros(ntu, hevc.TU_DTYPE)
        idx = np.arange(ntu)
        pl, rem = idx // (bw * bh), idx % (bw * bh)
        tus["coeff_offset"] = idx * nsz * nsz
        tus["dst_offset"] = pl * 3840 * 2160 + (rem // bw) * nsz * 3840 + (rem % bw) * nsz
        tus["col_limit"] = nsz
        d_t = torch.from_numpy(tus.view(np.uint8).reshape(ntu, 12).copy()).to(dev)
        c0 = torch.randint(-512, 512, (ntu, nsz * nsz), dtype=torch.int16, device=dev)
        pic = torch.randint(0, 256, (planes * 2160, 3840), dtype=torch.uint8, device=dev)
        cc = c0.clone()
        hevc.idct_batch(hevc.IDCT, lg, cc, pic, 3840, d_t, ntu)
        tot = 0.0
        for _ in range(5):
            cc.copy_(c0)
            e0, e1 = ev(), ev()
            e0.record()
            hevc.idct_batch(hevc.IDCT, lg, cc, pic, 3840, d_t, ntu)
            e1.record()
            torch.cuda.synchronize()
            tot += e0.elapsed_time(e1)
        ms = tot / 5
        gbs = ntu * nsz * nsz * 6 / (ms * 1e-3) / 1e9
        out["hevc_idct%d_add" % nsz] = {"Mblocks/s": round(ntu / (ms * 1e-3) / 1e6, 1), "GB/s": round(gbs, 1),
                                        "hbm_frac": round(gbs / HBM_PEAK_GBS, 4), "blocks": ntu, "ms": round(ms, 4)}
        del cc, c0, pic, d_t
    # complex FFT-1024 forward, 65,536 transforms: 16,384 B per transform.  The default context (the register-resident radix kernel,
    # within the float tolerance of the C codelets) and the FFHIP_TX_BITEXACT one (the split-radix network in the reference's order)
    from ffmpeg_amd import tx as _tx
    fin = torch.rand((65536, 2048), dtype=torch.float32, device=dev)
    fout = torch.empty((65536, 2048), dtype=torch.float32, device=dev)
    for key, flags in (("fft1024_fwd", 0), ("fft1024_fwd_bitexact", _tx.BITEXACT)):
        fctx = _tx.TxContext(_tx.FLOAT_FFT, 0, 1024, 1.0, flags=flags)
        for _ in range(2):
            fctx.batch(fout, fin)
        e0, e1 = ev(), ev()
        e0.record()
        for _ in range(10):
            fctx.batch(fout, fin)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        gbs = 65536 * 16384 / (ms * 1e-3) / 1e9
        out[key] = {"Mtransforms/s": round(65536 / (ms * 1e-3) / 1e6, 2), "GB/s": round(gbs, 1), "hbm_frac": round(gbs / HBM_PEAK_GBS, 4),
                    "transforms": 65536, "ms": round(ms, 4)}
        fctx.close()
    del fin, fout
    # vector_fmul_window (the windowing + overlap-add after an IMDCT): 65,536 frames of len 1024 (16,384 B moved each)
    from ffmpeg_amd import fdsp
    nv, ln = 65536, 1024
    a0 = torch.rand((nv, ln), dtype=torch.float32, device=dev)
    a1 = torch.rand((nv, ln), dtype=torch.float32, device=dev)
    win = torch.rand((2 * ln,), dtype=torch.float32, device=dev)
    o = torch.empty((nv, 2 * ln), dtype=torch.float32, device=dev)
    for _ in range(2):
        fdsp.batch(fdsp.FMUL_WINDOW, o, a0, a1, win, 0.0, ln)
    e0, e1 = ev(), ev()
    e0.record()
    for _ in range(10):
        fdsp.batch(fdsp.FMUL_WINDOW, o, a0, a1, win, 0.0, ln)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    gbs = nv * 16384 / (ms * 1e-3) / 1e9
    out["fdsp_vector_fmul_window_1024"] = {"Mvectors/s": round(nv / (ms * 1e-3) / 1e6, 1), "GB/s": round(gbs, 1),
                                           "hbm_frac": round(gbs / HBM_PEAK_GBS, 4), "vectors": nv, "ms": round(ms, 4)}
    del a0, a1, win, o
    # float MDCT-1024 forward, 65,536 transforms (BASELINE configs[3]): 12,288 B per transform
    from ffmpeg_amd import tx, me
    nt, ln = 65536, 1024
    for inv, key, per in ((0, "mdct1024_fwd", 12288), (1, "mdct1024_inv", 8192)):   # inverse: 1024 coefficients in, 1024 samples out
        f = tx.TxContext(tx.FLOAT_MDCT, inv, ln, 1.0)
        tin = torch.rand((nt, ln if inv else 2 * ln), dtype=torch.float32, device=dev)
        tout = torch.empty((nt, ln), dtype=torch.float32, device=dev)
        for _ in range(2):
            f.batch(tout, tin)
        e0, e1 = ev(), ev()
        e0.record()
        for _ in range(10):
            f.batch(tout, tin)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        gbs = nt * per / (ms * 1e-3) / 1e9
        out[key] = {"Mtransforms/s": round(nt / (ms * 1e-3) / 1e6, 2), "GB/s": round(gbs, 1),
                    "hbm_frac": round(gbs / HBM_PEAK_GBS, 4), "transforms": nt, "ms": round(ms, 4)}
        f.close()
        del tin, tout
    # the av_tx tails: complex FFTs of 960 (fft15 x fft64, the prime-factor kernel) and 16384 points (one workgroup per transform): 16 B per point
    for ln, nt, key in ((960, 32768, "fft960_pfa"), (16384, 4096, "fft16384")):
        f = tx.TxContext(tx.FLOAT_FFT, 0, ln, 1.0)
        tin = torch.rand((nt, 2 * ln), dtype=torch.float32, device=dev)
        tout = torch.empty((nt, 2 * ln), dtype=torch.float32, device=dev)
        for _ in range(2):
            f.batch(tout, tin)
        e0, e1 = ev(), ev()
        e0.record()
        for _ in range(5):
            f.batch(tout, tin)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        gbs = nt * ln * 16 / (ms * 1e-3) / 1e9
        out[key] = {"Mtransforms/s": round(nt / (ms * 1e-3) / 1e6, 3), "GB/s": round(gbs, 1), "hbm_frac": round(gbs / HBM_PEAK_GBS, 4),
                    "transforms": nt, "ms": round(ms, 4)}
        f.close()
        del tin, tout
    # exhaustive SAD search, 16x16 blocks, R = 7, 8 pairs of 3840x2160 luma planes (BASELINE configs[4] shape)
    nf, w, h = 8, 3840, 2160
    cur = torch.randint(0, 256, (nf, h, w), dtype=torch.uint8, device=dev)
    ref = torch.roll(cur, shifts=(3, -2), dims=(1, 2)).contiguous()
    mv = torch.empty((nf, (w // 16) * (h // 16) * 2), dtype=torch.int16, device=dev)
    cost = torch.empty((nf, (w // 16) * (h // 16)), dtype=torch.int32, device=dev)
    for kind, name in ((me.SAD, "me_esa_sad_r7"), (me.SATD, "me_esa_satd_r7")):
        me.esa_batch(cur, ref, w, h, w, w * h, nf, 16, 7, kind, mv, cost)
        e0, e1 = ev(), ev()
        e0.record()
        for _ in range(3):
            me.esa_batch(cur, ref, w, h, w, w * h, nf, 16, 7, kind, mv, cost)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 3
        nmb = nf * (w // 16) * (h // 16)
        # roof of this path: packed-SAD issue — v_sad_u8 scores 4 abs-diffs per lane at 0.547 wave-instructions
        # per ns per SIMD (profiles/r01_valu_rate_ubench.txt) x 1024 SIMDs x 64 lanes = 1.43e14 abs-diff/s
        ad = nmb * 225 * 256 / (ms * 1e-3)
        out[name] = {"MB-searches/s": round(nmb / (ms * 1e-3), 1), "candidates/s": round(nmb * 225 / (ms * 1e-3), 1),
                     "abs_diff/s": float("%.4g" % ad), "sad_issue_roof_frac": round(ad / 1.434e14, 4),
                     "hbm_frac": round(nf * 2 * w * h / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "frame_pairs": nf, "ms": round(ms, 4)}
        if kind == me.SATD:
            # SATD is priced on its own roof too: VALU issue.  Round 5 (me_satd.hip: the 2-D Hadamard as a dense int8 product on the matrix
            # cores, one v_sad_u32 per coefficient): 1,673 VALU wave-instructions per macroblock incl. 253 MFMAs (profiles/r05_esa_satd_mx_pmc.txt;
            # r04's packed-int16 butterflies: 3,466) against 1024 SIMDs x 0.6 wave-instructions / ns (a wave64 instruction occupies its SIMD
            # for 4 cycles at 2.4 GHz; the part runs this kernel at ~1.8 GHz)
            out[name]["valu_issue_roof_frac"] = round(nmb * 1673 / (ms * 1e-3) / (1024 * 0.6e9), 4)
            out[name]["note"] = ("a candidate is 4 blocks x 64 coefficients: 16 MFMAs + 64 v_sad_u32 per 16 candidates and lane; PMC: VALU busy 82.7 % "
                                 "at 1.77 GHz, an MFMA costs the VALU port two issue slots (profiles/r05_mfma_i8_rate.txt)")
    # the SATD search once more on 32 pairs: the launch-size dependence of a kernel whose clock the part lowers (profiles/r05_esa_satd_mx_pmc.txt)
    del cur, ref
    nf = 32
    cur = torch.randint(0, 256, (nf, h, w), dtype=torch.uint8, device=dev)
    ref = torch.roll(cur, shifts=(3, -2), dims=(1, 2)).contiguous()
    mv = torch.empty((nf, (w // 16) * (h // 16) * 2), dtype=torch.int16, device=dev)
    cost = torch.empty((nf, (w // 16) * (h // 16)), dtype=torch.int32, device=dev)
    me.esa_batch(cur, ref, w, h, w, w * h, nf, 16, 7, me.SATD, mv, cost)
    e0, e1 = ev(), ev()
    e0.record()
    for _ in range(3):
        me.esa_batch(cur, ref, w, h, w, w * h, nf, 16, 7, me.SATD, mv, cost)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    nmb = nf * (w // 16) * (h // 16)
    out["me_esa_satd_r7_32_pairs"] = {"MB-searches/s": round(nmb / (ms * 1e-3), 1), "candidates/s": round(nmb * 225 / (ms * 1e-3), 1),
                                      "valu_issue_roof_frac": round(nmb * 1673 / (ms * 1e-3) / (1024 * 0.6e9), 4), "frame_pairs": nf, "ms": round(ms, 4)}
    del cur, ref, mv, cost
    # H.264 luma qpel: every 16x16 macroblock of 8 4K planes, mixed mcXY, put (BASELINE configs[2]): 2 B / sample
    for nf, key in ((8, "h264_qpel16_mixed"), (32, "h264_qpel16_mixed_32_planes")):
        w, h, P = 3840, 2160, 32
        stride = w + 2 * P
        refp = torch.randint(0, 256, (nf * (h + 2 * P), stride), dtype=torch.uint8, device=dev)
        dstp = torch.zeros_like(refp)
        rng = np.random.default_rng(3)
        my, mx = np.meshgrid(np.arange(h // 16), np.arange(w // 16), indexing="ij")
        blk = np.zeros(nf * my.size, dtype=np.dtype([("d", np.int32), ("s", np.int32), ("mc", np.uint8), ("sz", np.uint8),
                                                         ("avg", np.uint8), ("flags", np.uint8), ("sx", np.int16), ("sy", np.int16)]))
        for fi in range(nf):
            base = fi * (h + 2 * P) * stride
            d = base + (P + my.ravel() * 16) * stride + P + mx.ravel() * 16
            dy, dx = rng.integers(-24, 25, (2, my.size))
            sl = slice(fi * my.size, (fi + 1) * my.size)
            blk["d"][sl] = d
            blk["s"][sl] = d + dy * stride + dx
            blk["mc"][sl] = rng.integers(0, 16, my.size)
        dblk = torch.from_numpy(blk.view(np.uint8).reshape(-1, 16)).to(dev)
        h264.qpel_batch(dstp, refp, stride, dblk, blk.size)
        e0, e1 = ev(), ev()
        e0.record()
        for _ in range(5):
            h264.qpel_batch(dstp, refp, stride, dblk, blk.size)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        px = blk.size * 256
        out[key] = {"Mpixels/s": round(px / (ms * 1e-3) / 1e6, 1), "GB/s": round(2 * px / (ms * 1e-3) / 1e9, 1),
                    "hbm_frac": round(2 * px / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "blocks": int(blk.size), "planes": nf,
                    "ms": round(ms, 4)}
        del refp, dstp, dblk
    # function-level luma loop filter (h264dsp.h_loop_filter_luma / v_loop_filter_luma as checkasm calls them): one vertical and
    # one horizontal edge per 16x16 tile of 8 4K planes, pairwise disjoint; a call reads and writes the 8 x 16 samples across its edge
    nf, w, h = 8, 3840, 2160
    plane = torch.randint(96, 160, (nf * h, w), dtype=torch.uint8, device=dev)
    ty, tx_ = np.meshgrid(np.arange(nf * h // 16), np.arange(w // 16), indexing="ij")
    edt = np.dtype([("offset", "<i4"), ("kind", "u1"), ("alpha", "u1"), ("beta", "u1"), ("pad", "u1"), ("tc0", "i1", (4,))])
    for kind, key, eo in ((1, "h264_h_loop_filter_luma", 8), (0, "h264_v_loop_filter_luma", 8 * w)):
        ed = np.zeros(ty.size, edt)
        ed["offset"] = (ty.ravel() * 16 * w + tx_.ravel() * 16 + eo).astype(np.int32)
        ed["kind"], ed["alpha"], ed["beta"] = kind, 40, 12
        ed["tc0"] = 2
        ded = torch.from_numpy(ed.view(np.uint8).reshape(-1, 12)).to(dev)
        h264.loop_filter_batch(plane, w, ded, ed.size)
        e0, e1 = ev(), ev()
        e0.record()
        for _ in range(5):
            h264.loop_filter_batch(plane, w, ded, ed.size)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        byt = ed.size * (2 * 128 + 12)
        out[key] = {"Medges/s": round(ed.size / (ms * 1e-3) / 1e6, 1), "GB/s": round(byt / (ms * 1e-3) / 1e9, 1),
                    "hbm_frac": round(byt / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "edges": int(ed.size), "ms": round(ms, 4)}
    del plane
    # frame-order luma deblocking of 8 independent 4K planes (240x135 MBs each), launched back to back
    mbw, mbh = w // 16, h // 16
    planes = [torch.randint(100, 140, (h, w), dtype=torch.uint8, device=dev) for _ in range(8)]
    ed = np.zeros(mbw * mbh * 8, dtype=np.dtype([("o", np.int32), ("k", np.uint8), ("a", np.uint8), ("b", np.uint8),
                                                     ("p", np.uint8), ("tc", np.int8, 4)]))
    ed["a"], ed["b"] = 40, 9
    kk = np.zeros((mbw * mbh, 2, 4), np.uint8)   # bS = 4 exists on the MACROBLOCK edges (edge 0) of intra macroblocks only: a quarter of them
    kk[rng.random(mbw * mbh) < .25, :, 0] = 4
    ed["k"] = kk.ravel()
    ed["tc"] = rng.integers(0, 4, (ed.size, 4))
    ded = torch.from_numpy(ed.view(np.uint8).reshape(-1, 12)).to(dev)
    h264.deblock_frame(planes[0], w, mbw, mbh, ded)
    torch.cuda.synchronize()
    e0, e1 = ev(), ev()
    e0.record()
    for pl in planes:
        h264.deblock_frame(pl, w, mbw, mbh, ded)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / len(planes)
    # independent frames in ONE launch (ffhip_h264_deblock_frames_dev): only the order inside a frame is serial
    per = {}
    for nfb in (8, 32):
        batch = torch.stack([planes[i % 8] for i in range(nfb)])
        dedn = ded.repeat(nfb, 1)
        h264.deblock_frames(batch, w * h, nfb, w, mbw, mbh, dedn)
        e0, e1 = ev(), ev()
        e0.record()
        h264.deblock_frames(batch, w * h, nfb, w, mbw, mbh, dedn)
        e1.record()
        torch.cuda.synchronize()
        per[nfb] = e0.elapsed_time(e1) / nfb
        del batch, dedn
    out["h264_deblock_frame_4k"] = {"Mpixels/s": round(w * h / (per[32] * 1e-3) / 1e6, 1), "ms_per_frame_one_stream": round(ms, 4),
                                    "ms_per_frame_batch_of_8": round(per[8], 4), "ms_per_frame_batch_of_32": round(per[32], 4),
                                    "note": "decoder order (2-D wavefront inside a frame); a batch runs its frames side by side"}
    # the same at 10 bits (uint16 samples, ffhip_h264_deblock_frames_dev_hbd): a lone plane and 32 in one launch
    per10 = {}
    for nfb in (1, 32):
        batch = (torch.randint(100, 140, (nfb, h, w), dtype=torch.int32, device=dev) << 2).to(torch.int16)
        dedn = ded.repeat(nfb, 1)
        h264.deblock_frames_hbd(10, batch, 2 * w * h, nfb, 2 * w, mbw, mbh, dedn)
        e0, e1 = ev(), ev()
        e0.record()
        h264.deblock_frames_hbd(10, batch, 2 * w * h, nfb, 2 * w, mbw, mbh, dedn)
        e1.record()
        torch.cuda.synchronize()
        per10[nfb] = e0.elapsed_time(e1) / nfb
        del batch, dedn
    out["h264_deblock_frame_4k_10bit"] = {"Mpixels/s": round(w * h / (per10[32] * 1e-3) / 1e6, 1), "ms_per_frame_alone": round(per10[1], 4),
                                          "ms_per_frame_batch_of_32": round(per10[32], 4)}
    del planes, ded
    # 15xM prime-factor MDCT, the CELT / AAC-960 frame size: inverse, len 960 (7,680 B moved per transform)
    nt, ln = 65536, 960
    f = tx.TxContext(tx.FLOAT_MDCT, 1, ln, 1.0 / ln)
    tin = torch.rand((nt, ln), dtype=torch.float32, device=dev)
    tout = torch.empty((nt, ln), dtype=torch.float32, device=dev)
    f.batch(tout, tin)
    e0, e1 = ev(), ev()
    e0.record()
    for _ in range(10):
        f.batch(tout, tin)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    gbs = nt * ln * 8 / (ms * 1e-3) / 1e9
    out["imdct960_pfa15"] = {"Mtransforms/s": round(nt / (ms * 1e-3) / 1e6, 2), "GB/s": round(gbs, 1),
                             "hbm_frac": round(gbs / HBM_PEAK_GBS, 4), "transforms": nt, "ms": round(ms, 4)}
    f.close()
    del tin, tout
    # HEVC put_hevc_qpel_uni: every 16x16 block of 8 4K planes, mixed quarter-sample positions: 2 B / sample
    from ffmpeg_amd import hevc
    P = 16
    refp = torch.randint(0, 256, (nf * h + 2 * P, w + 2 * P), dtype=torch.uint8, device=dev)
    pic = torch.zeros((nf * h, w), dtype=torch.uint8, device=dev)
    by, bx = np.meshgrid(np.arange(0, nf * h, 16), np.arange(0, w, 16), indexing="ij")
    mc = np.zeros(by.size, hevc.MC_DTYPE)
    mc["dst_offset"] = (by * w + bx).reshape(-1)
    mc["src_offset"] = ((by + P + rng.integers(-8, 9, by.shape)) * (w + 2 * P) + bx + P + rng.integers(-8, 9, by.shape)).reshape(-1)
    mc["width"] = mc["height"] = 16
    mc["mx"], mc["my"] = rng.integers(0, 4, by.size), rng.integers(0, 4, by.size)
    dmc = torch.from_numpy(mc.view(np.uint8).reshape(-1, 12)).to(dev)
    hevc.mc_batch(0, 1, pic, w, refp, w + 2 * P, dmc, by.size)
    e0, e1 = ev(), ev()
    e0.record()
    for _ in range(5):
        hevc.mc_batch(0, 1, pic, w, refp, w + 2 * P, dmc, by.size)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    px = by.size * 256
    out["hevc_qpel_uni16_mixed"] = {"Mpixels/s": round(px / (ms * 1e-3) / 1e6, 1), "GB/s": round(2 * px / (ms * 1e-3) / 1e9, 1),
                                    "hbm_frac": round(2 * px / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "blocks": int(by.size),
                                    "ms": round(ms, 4)}
    # VP9 motion compensation (round 6: the 16 x 16 blocks on the matrix cores): the same block grid, the three 8-tap sets mixed, all (mx, my), put
    from ffmpeg_amd import vp9 as _vp9
    vmc = np.zeros(by.size, _vp9.MC_DTYPE)
    vmc["dst_offset"], vmc["src_offset"] = mc["dst_offset"], mc["src_offset"]
    vmc["width"] = vmc["height"] = 16
    vmc["filter"] = rng.integers(0, 3, by.size)
    vmc["mx"], vmc["my"] = rng.integers(0, 16, by.size), rng.integers(0, 16, by.size)
    dvmc = torch.from_numpy(vmc.view(np.uint8).reshape(-1, 16)).to(dev)
    _vp9.mc_batch(pic, w, refp, w + 2 * P, dvmc, by.size)
    e0, e1 = ev(), ev()
    e0.record()
    for _ in range(5):
        _vp9.mc_batch(pic, w, refp, w + 2 * P, dvmc, by.size)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    out["vp9_mc16_8tap_mixed"] = {"Mpixels/s": round(px / (ms * 1e-3) / 1e6, 1), "GB/s": round(2 * px / (ms * 1e-3) / 1e9, 1),
                                  "hbm_frac": round(2 * px / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "blocks": int(by.size), "ms": round(ms, 4)}
    # H.264 chroma MC (h264chroma_template.c:28: the bilinear 8 x 8 blocks behind every 16 x 16 luma block of a 4:2:0 picture): one chroma
    # plane per luma plane, mixed eighth-sample fractions, displacements +-4
    from ffmpeg_amd import h264 as _h264
    ch, cw = nf * h // 2, w // 2
    cref = torch.randint(0, 256, (ch + 2 * P, cw + 2 * P), dtype=torch.uint8, device=dev)
    cpic = torch.zeros((ch + 2 * P, cw + 2 * P), dtype=torch.uint8, device=dev)
    cy, cx = np.meshgrid(np.arange(0, ch, 8), np.arange(0, cw, 8), indexing="ij")
    cmc = np.zeros(cy.size, _h264.CHROMA_DTYPE)
    cst = cw + 2 * P
    cmc["dst_offset"] = ((cy + P) * cst + cx + P).reshape(-1)
    cmc["src_offset"] = ((cy + P + rng.integers(-4, 5, cy.shape)) * cst + cx + P + rng.integers(-4, 5, cy.shape)).reshape(-1)
    cmc["w_idx"], cmc["h"] = 0, 8
    cmc["x"], cmc["y"] = rng.integers(0, 8, cy.size), rng.integers(0, 8, cy.size)
    dcmc = torch.from_numpy(cmc.view(np.uint8).reshape(-1, 20)).to(dev)
    _h264.chroma_mc_batch(cpic, cref, cst, dcmc, cy.size)
    e0, e1 = ev(), ev()
    e0.record()
    for _ in range(5):
        _h264.chroma_mc_batch(cpic, cref, cst, dcmc, cy.size)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    cpx = cy.size * 64
    out["h264_chroma_mc8_mixed"] = {"Mpixels/s": round(cpx / (ms * 1e-3) / 1e6, 1), "GB/s": round(2 * cpx / (ms * 1e-3) / 1e9, 1),
                                    "hbm_frac": round(2 * cpx / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "blocks": int(cy.size), "ms": round(ms, 4)}
    del refp, pic, cref, cpic
    # AV_TX_INT32_MDCT (round 6, kernels/tx_wide.hip: the fixed-point codecs' transform), 1024 coefficients forward, 16,384 transforms:
    # 12,288 B each (8 KiB in, 4 KiB out), bit-exact fixed-point arithmetic
    from ffmpeg_amd import tx as _tx
    nti = 16384
    ctx_i = _tx.TxContext(_tx.INT32_MDCT, 0, 1024, 1.0)
    ti = torch.randint(-2 ** 24, 2 ** 24, (nti, 2048), dtype=torch.int32, device=dev)
    to = torch.zeros((nti, 1024), dtype=torch.int32, device=dev)
    ctx_i.batch(to, ti)
    e0, e1 = ev(), ev()
    e0.record()
    for _ in range(5):
        ctx_i.batch(to, ti)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    gbs = nti * 12288 / (ms * 1e-3) / 1e9
    out["mdct1024_int32_fwd"] = {"Mtransforms/s": round(nti / (ms * 1e-3) / 1e6, 2), "GB/s": round(gbs, 1), "hbm_frac": round(gbs / HBM_PEAK_GBS, 4),
                                 "transforms": nti, "ms": round(ms, 4)}
    ctx_i.close()
    del ti, to
    # DCT-I / DST-I at wmavoice's length (libavcodec/wmavoice.c:398-404: 64 reals, scale 1/64): the transform as its 64 x 64 matrix,
    # summed in double precision (kernels/tx_dcst1.hip) — 512 B and 4096 multiply-adds per transform, bound by the FP64 issue rate
    for nm, ty in (("dctI_64", _tx.FLOAT_DCT_I), ("dstI_64", _tx.FLOAT_DST_I)):
        ntd = 1 << 20
        ctx_d = _tx.TxContext(ty, 0, 64, 1.0 / 64)
        di = torch.randn((ntd, 64), dtype=torch.float32, device=dev)
        do = torch.zeros_like(di)
        ctx_d.batch(do, di)
        e0, e1 = ev(), ev()
        e0.record()
        for _ in range(5):
            ctx_d.batch(do, di)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        gbs = ntd * 512 / (ms * 1e-3) / 1e9
        out[nm] = {"Mtransforms/s": round(ntd / (ms * 1e-3) / 1e6, 2), "GB/s": round(gbs, 1), "hbm_frac": round(gbs / HBM_PEAK_GBS, 4),
                   "fp64_TFLOP/s": round(ntd * 8192 / (ms * 1e-3) / 1e12, 2), "transforms": ntd, "ms": round(ms, 4)}
        ctx_d.close()
        del di, do
    # AAC imdct_and_windowing: 65,536 all-long channel-frames (2 channels), inverse MDCT -> window -> overlap-add resident in HBM:
    # 18,432 B per channel-frame (ffmpeg_amd/csrc/aac_api.hip).  Window tables: the decoder's own, from the committed fixture.
    from ffmpeg_amd import aac
    gd = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "golden", "aac.npz"))
    actx = aac.AacImdct([gd[k] for k in ("sine_1024", "sine_128", "kbd_long_1024", "kbd_short_128")])
    nch, nfr = 2, 32768
    aco = torch.randn((nfr, nch, 1024), dtype=torch.float32, device=dev) * 1000
    aout = torch.empty_like(aco)
    asv = torch.zeros((nch, 512), dtype=torch.float32, device=dev)
    aseq, az = np.zeros((nfr, nch), np.uint8), np.zeros(nch, np.uint8)
    for _ in range(2):
        actx.batch(aco, aout, asv, aseq, aseq + 1, az, az)
    e0, e1 = ev(), ev()
    e0.record()
    for _ in range(5):
        actx.batch(aco, aout, asv, aseq, aseq + 1, az, az)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    gbs = nch * nfr * 18432 / (ms * 1e-3) / 1e9
    out["aac_imdct_and_windowing_long"] = {"Mframes/s": round(nch * nfr / (ms * 1e-3) / 1e6, 1), "GB/s": round(gbs, 1),
                                           "hbm_frac": round(gbs / HBM_PEAK_GBS, 4), "channel_frames": nch * nfr, "ms": round(ms, 4)}
    actx.close()
    del aco, aout
    # float DCT-III (AV_TX_FLOAT_DCT inverse), N = 1024, 65,536 transforms: 8,192 B per transform
    nt, ln = 65536, 1024
    f = tx.TxContext(tx.FLOAT_DCT, 1, ln >> 1, 1.0)
    tin = torch.rand((nt, ln), dtype=torch.float32, device=dev)
    tout = torch.empty((nt, ln), dtype=torch.float32, device=dev)
    for _ in range(2):
        f.batch(tout, tin)
    e0, e1 = ev(), ev()
    e0.record()
    for _ in range(10):
        f.batch(tout, tin)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    gbs = nt * 8192 / (ms * 1e-3) / 1e9
    out["dct3_1024"] = {"Mtransforms/s": round(nt / (ms * 1e-3) / 1e6, 2), "GB/s": round(gbs, 1), "hbm_frac": round(gbs / HBM_PEAK_GBS, 4),
                        "transforms": nt, "ms": round(ms, 4)}
    f.close()
    out.update(round2_legs(torch, dev, ev))
    out.update(sws_ops_leg(torch, dev))
    return out


def round2_legs(torch, dev, ev):
    """rows added in round 2: the forward DCT-II, a 3xM prime-factor MDCT (768-sample AAC frames), VP9's 32x32 inverse
    transform, the VP9 loop filter of a 4K picture in decoder order"""
    from ffmpeg_amd import tx, vp9
    out = {}
    nt = 65536
    for key, typ, inv, ln, n_in, n_out in (("dct2_1024", tx.FLOAT_DCT, 0, 1024, 1024, 1024), ("mdct1536_pfa3_fwd", tx.FLOAT_MDCT, 0, 1536, 3072, 1536)):
        f = tx.TxContext(typ, inv, ln, 1.0)
        tin = torch.rand((nt, n_in), dtype=torch.float32, device=dev)
        tout = torch.empty((nt, n_out), dtype=torch.float32, device=dev)
        for _ in range(2):
            f.batch(tout, tin)
        e0, e1 = ev(), ev()
        e0.record()
        for _ in range(10):
            f.batch(tout, tin)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        gbs = nt * (n_in + n_out) * 4 / (ms * 1e-3) / 1e9
        out[key] = {"Mtransforms/s": round(nt / (ms * 1e-3) / 1e6, 2), "GB/s": round(gbs, 1), "hbm_frac": round(gbs / HBM_PEAK_GBS, 4),
                    "transforms": nt, "ms": round(ms, 4)}
        f.close()
        del tin, tout
    # VP9 itxfm_add 32x32 (DCT_DCT), every block of 8 4K luma planes: 6 B per sample
    planes, nsz = 8, 32
    bw, bh = 3840 // nsz, 2160 // nsz
    ntu = planes * bw * bh
    tus = np.zeros(ntu, vp9.TU_DTYPE)
    idx = np.arange(ntu)
    pl, rem = idx // (bw * bh), idx % (bw * bh)
    tus["coeff_offset"] = idx * nsz * nsz
    tus["dst_offset"] = pl * 3840 * 2160 + (rem // bw) * nsz * 3840 + (rem % bw) * nsz
    d_t = torch.from_numpy(tus.view(np.uint8).reshape(ntu, 12).copy()).to(dev)
    c0 = torch.randint(-512, 512, (ntu, nsz * nsz), dtype=torch.int16, device=dev)
    pic = torch.randint(0, 256, (planes * 2160, 3840), dtype=torch.uint8, device=dev)
    cc = c0.clone()
    vp9.itxfm_add_batch(3, cc, pic, 3840, d_t, ntu)
    tot = 0.0
    for _ in range(5):
        cc.copy_(c0)
        e0, e1 = ev(), ev()
        e0.record()
        vp9.itxfm_add_batch(3, cc, pic, 3840, d_t, ntu)
        e1.record()
        torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    ms = tot / 5
    gbs = ntu * nsz * nsz * 6 / (ms * 1e-3) / 1e9
    out["vp9_itxfm32_add"] = {"Mblocks/s": round(ntu / (ms * 1e-3) / 1e6, 1), "GB/s": round(gbs, 1), "hbm_frac": round(gbs / HBM_PEAK_GBS, 4),
                              "blocks": ntu, "ms": round(ms, 4)}
    del cc, c0, pic, d_t
    # the VP9 loop filter of a 4K 4:2:0 picture in the decoder's (superblock) order: one launch, a 2-D wavefront — latency, not bytes
    sbc, sbr = 60, 34
    rng = np.random.default_rng(9)
    tabs = np.zeros((sbr * sbc, 320), np.uint32)
    # every 8-sample luma segment on the 8x8 grid filtered 8 wide at level 32, every second chroma one (a dense, regular picture)
    ent = np.uint32(0x80000000 | 1 << 24 | (32 >> 4) << 16 | 6 << 8 | (2 * (32 + 2) + 6))
    y = tabs[:, :256].reshape(-1, 2, 16, 8)
    y[:, :, 0::2, :] = ent
    uv = tabs[:, 256:].reshape(-1, 2, 8, 4)
    uv[:, :, 0::2, :] = ent
    t3 = tabs.reshape(sbr, sbc, 320)
    t3[:, 0, 0:8] = 0            # no column edge at the picture's left border (luma position 0 / chroma position 0 of column edges)
    t3[:, 0, 256:260] = 0
    t3[0, :, 128:136] = 0        # nor a row edge at its top
    t3[0, :, 288:292] = 0
    d_tabs = torch.from_numpy(tabs.view(np.int32)).to(dev)
    yy = torch.from_numpy(np.clip(np.cumsum(rng.integers(-2, 3, (64 * sbr, 64 * sbc)), axis=1) + 128, 0, 255).astype(np.uint8)).to(dev)
    uu = yy[::2, ::2].contiguous()
    vv = uu.clone()
    for _ in range(2):
        vp9.loopfilter_frame(yy, uu, vv, 64 * sbc, 32 * sbc, 8 * sbc, 8 * sbr, d_tabs)
    e0, e1 = ev(), ev()
    e0.record()
    for _ in range(5):
        vp9.loopfilter_frame(yy, uu, vv, 64 * sbc, 32 * sbc, 8 * sbc, 8 * sbr, d_tabs)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    # N pictures per launch (round 4, ffhip_vp9_loopfilter_frames_dev): each picture its own planes, the same tables
    lfb = {}
    for npl in (8, 16, 32):
        pics = [(yy.clone(), uu.clone(), vv.clone(), d_tabs) for _ in range(npl)]
        for _ in range(2):
            vp9.loopfilter_frames(pics, 64 * sbc, 32 * sbc, 8 * sbc, 8 * sbr)
        b0, b1 = ev(), ev()
        b0.record()
        for _ in range(3):
            vp9.loopfilter_frames(pics, 64 * sbc, 32 * sbc, 8 * sbc, 8 * sbr)
        b1.record()
        torch.cuda.synchronize()
        lfb[npl] = b0.elapsed_time(b1) / 3
        del pics
    out["vp9_loopfilter_frame_4k"] = {"ms_per_picture_one_stream": round(ms, 4), "pictures_per_s": round(1e3 / ms, 1),
                                      "ms_per_launch_of_8_16_32_pictures": [round(lfb[8], 3), round(lfb[16], 3), round(lfb[32], 3)],
                                      "pictures_per_s_32_per_launch": round(32e3 / lfb[32], 1),
                                      "Mpixels/s": round(64 * sbc * 64 * sbr / (ms * 1e-3) / 1e6, 1),
                                      "note": "decoder order (superblock wavefront, luma and chroma chains side by side), every 8x8-grid edge 8 wide"}
    out.update(h264_picture_leg(torch, dev, ev))
    return out


def h264_intra_picture_leg(torch, dev, ev, h264):
    """a 1080p I-picture (every macroblock intra: Intra16x16 / 4x4 / 8x8-transform / I_PCM mixed with residuals, the decoder-side
    state from tests/h264_intra_gen.py — an input generator, not the oracle) through ffhip_h264_picture_flush: the reconstruction
    wavefront (kernels/h264_intra.hip) + the decoder-order deblocking with bS = 3 / 4 edges."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests"))
    import h264_intra_gen as G
    EDGE_DT = np.dtype([("offset", np.int32), ("kind", np.uint8), ("alpha", np.uint8), ("beta", np.uint8), ("pad", np.uint8), ("tc0", np.int8, 4)])
    mb_w, mb_h = 120, 68
    rng = np.random.default_rng(6)
    sy, sc = mb_w * 16, mb_w * 8
    dst = [torch.zeros((mb_h * 16, sy), dtype=torch.uint8, device=dev), torch.zeros((mb_h * 8, sc), dtype=torch.uint8, device=dev),
           torch.zeros((mb_h * 8, sc), dtype=torch.uint8, device=dev)]
    ed8, ed4 = np.zeros(8, EDGE_DT), np.zeros(4, EDGE_DT)
    for e in (ed8, ed4):
        e["alpha"], e["beta"], e["kind"] = 40, 9, 4
    ed4["kind"] = 6
    states = [(mx, my, G.make_intra_mb(rng, mx, my, mb_w, mb_h)) for my in range(mb_h) for mx in range(mb_w)]
    states = [(mx, my, d, G.to_record(d)) for mx, my, d in states]

    def record():
        pic = h264.Picture(mb_w, mb_h)
        pic.begin()
        for mx, my, d, rec in states:
            pic.intra_mb(rec, d["nnzc"], d["mb"].copy(), d["luma_dc"], d["pcm"])      # the record call consumes sl->mb as the decoder's dsp calls do
            for pl, ed in ((0, ed8), (1, ed4), (2, ed4)):
                pic.deblock_mb(pl, mx, my, ed)
        return pic
    pic = record()
    pic.flush(dst, [sy, sc, sc], dst)
    torch.cuda.synchronize()
    reps = 10
    e0, e1 = ev(), ev()
    e0.record()
    for _ in range(reps):
        pic.flush(dst, [sy, sc, sc], dst)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    # I-pictures IN FLIGHT (round 4): a lone wavefront is a latency chain that leaves the chip idle; a decoder with frame threads (or
    # several streams) has several pictures reconstructing at once.  NP pictures, each its own object, stream and host thread.
    import threading
    NP = 16
    pics = [pic] + [record() for _ in range(NP - 1)]
    dsts = [dst] + [[torch.zeros_like(t) for t in dst] for _ in range(NP - 1)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(NP)]
    flight = None
    for rep in range(2):
        torch.cuda.synchronize()
        f0, f1 = ev(), ev()
        f0.record()
        for s_ in streams:
            s_.wait_event(f0)
        rounds = 4

        def work(i):
            for _ in range(rounds):
                pics[i].flush(dsts[i], [sy, sc, sc], dsts[i], stream=streams[i].cuda_stream)
        th = [threading.Thread(target=work, args=(i,)) for i in range(NP)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        cur = torch.cuda.current_stream()
        for s_ in streams:
            cur.wait_stream(s_)
        f1.record()
        torch.cuda.synchronize()
        flight = f0.elapsed_time(f1) / rounds
    # ... and the same NP I-pictures flushed together (ffhip_h264_pictures_flush): one launch of NP wavefronts, then the in-loop filter of
    # all planes side by side
    flush_batch = None
    for rep in range(2):
        torch.cuda.synchronize()
        g0, g1 = ev(), ev()
        g0.record()
        for _ in range(4):
            h264.pictures_flush(pics, dsts, [sy, sc, sc], dsts)
        g1.record()
        torch.cuda.synchronize()
        flush_batch = g0.elapsed_time(g1) / 4
    for p_ in pics:
        p_.close()
    # N pictures' wavefronts in ONE launch (round 4, ffhip_h264_intra_frames_dev): the reconstruction alone (no deblocking), the same
    # records for every picture of the batch, each picture its own planes
    import ctypes as C
    from ffmpeg_amd import _lib
    L = _lib.lib()

    class IntraPic(C.Structure):
        _fields_ = [("y", C.c_void_p), ("cb", C.c_void_p), ("cr", C.c_void_p), ("recs", C.c_void_p), ("row_start", C.c_void_p), ("coefs", C.c_void_p)]
    L.ffhip_h264_intra_pack.restype = C.c_int
    coefs, ncoef, recs = np.zeros(mb_w * mb_h * 400, np.int16), 0, []
    for mx, my, d, rec in states:
        rec = rec.copy()
        mbc = d["mb"].copy()
        nn = C.c_int32(ncoef)
        assert L.ffhip_h264_intra_pack(rec.ctypes.data, d["nnzc"].ctypes.data, mbc.ctypes.data, d["luma_dc"].ctypes.data,
                                       G._p(d["pcm"], C.c_uint8), G._p(coefs, C.c_int16), C.byref(nn), C.c_int32(coefs.size)) == 0
        ncoef = nn.value
        recs.append(rec)
    rows = np.arange(mb_h + 1, dtype=np.int32) * mb_w
    d_rec = torch.from_numpy(np.concatenate(recs).view(np.uint8).reshape(-1, 108).copy()).to(dev)
    d_rows, d_coef = torch.from_numpy(rows).to(dev), torch.from_numpy(coefs[:ncoef].copy()).to(dev)
    batch = {}
    for npl in (1, 16, 32, 64):
        planes = [[torch.zeros_like(t) for t in dst] for _ in range(npl)]
        arr = (IntraPic * npl)(*[IntraPic(pp[0].data_ptr(), pp[1].data_ptr(), pp[2].data_ptr(), d_rec.data_ptr(), d_rows.data_ptr(), d_coef.data_ptr())
                                 for pp in planes])
        st_ = torch.cuda.current_stream().cuda_stream
        for _ in range(2):
            _lib.check(L.ffhip_h264_intra_frames_dev(8, npl, C.cast(arr, C.c_void_p), sy, sc, mb_w, mb_h, C.c_void_p(st_)), "ffhip_h264_intra_frames_dev")
        b0, b1 = ev(), ev()
        b0.record()
        for _ in range(4):
            _lib.check(L.ffhip_h264_intra_frames_dev(8, npl, C.cast(arr, C.c_void_p), sy, sc, mb_w, mb_h, C.c_void_p(st_)), "ffhip_h264_intra_frames_dev")
        b1.record()
        torch.cuda.synchronize()
        batch[npl] = b0.elapsed_time(b1) / 4
        del planes
    return {"h264_intra_picture_1080p": {"ms_per_picture": round(ms, 3), "pictures_per_s": round(1e3 / ms, 1), "intra_macroblocks": mb_w * mb_h,
                                         "ms_per_batched_flush_of_%d" % NP: round(flush_batch, 3),
                                         "pictures_per_s_batched_flush_of_%d" % NP: round(1e3 * NP / flush_batch, 1),
                                         "wavefront_alone_ms": round(batch[1], 3),
                                         "wavefront_ms_per_launch_of_16_32_64_pictures": [round(batch[16], 3), round(batch[32], 3), round(batch[64], 3)],
                                         "wavefront_pictures_per_s_32_per_launch": round(32e3 / batch[32], 1),
                                         "wavefront_pictures_per_s_64_in_two_launches": round(64e3 / batch[64], 1),
                                         "us_per_wavefront_step": round(1e3 * ms / (mb_w + 2 * mb_h), 2),
                                         "ms_per_picture_%d_in_flight" % NP: round(flight / NP, 3),
                                         "pictures_per_s_%d_in_flight" % NP: round(1e3 * NP / flight, 1),
                                         "note": "a dependency chain of mb_w + 2 mb_h macroblock steps (intra prediction reads the left / upper / upper-right "
                                                 "neighbours' reconstructed samples): latency-bound by construction, one wave per macroblock row; in flight: "
                                                 "one object, stream and host thread per picture; per launch: ffhip_h264_intra_frames_dev, the "
                                                 "reconstruction wavefronts of N pictures side by side (no deblocking)"}}


def h264_intra_formats_leg(torch, dev, ev, h264):
    """a 1080p I-picture at the other chroma formats of the picture layer (round 4): 4:2:2 — the luma wavefront beside the 8 x 16 chroma
    planes' own (kernels/h264_c422.hip) — and 4:4:4 — three luma-only wavefronts side by side (hl_decode_mb_444) —, reconstruction only."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests"))
    import h264_intra_gen as G
    mb_w, mb_h = 120, 68
    out = {}
    for cfmt, name in ((2, "422"), (3, "444")):
        rng = np.random.default_rng(60 + cfmt)
        sy = mb_w * 16
        sc, hc = (sy, mb_h * 16) if cfmt == 3 else (mb_w * 8, mb_h * 16)
        dst = [torch.zeros((mb_h * 16, sy), dtype=torch.uint8, device=dev), torch.zeros((hc, sc), dtype=torch.uint8, device=dev),
               torch.zeros((hc, sc), dtype=torch.uint8, device=dev)]
        pic = h264.Picture(mb_w, mb_h, chroma_format=cfmt)
        pic.begin()
        for my in range(mb_h):
            for mx in range(mb_w):
                d = G.make_intra_mb(rng, mx, my, mb_w, mb_h, cfmt=cfmt)
                dc = np.zeros(96, np.int16)                # sl->mb_luma_dc as the decoder holds it: [3][16 * 2] int16
                for p_ in range(len(d["luma_dc"]) // 16):
                    dc[32 * p_:32 * p_ + 16] = d["luma_dc"][16 * p_:16 * p_ + 16]
                pic.intra_mb(G.to_record(d), d["nnzc"], d["mb"].copy(), dc, d["pcm"])
        for _ in range(2):
            pic.flush(dst, [sy, sc, sc], dst)
        torch.cuda.synchronize()
        e0, e1 = ev(), ev()
        e0.record()
        for _ in range(10):
            pic.flush(dst, [sy, sc, sc], dst)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        out["h264_intra_picture_1080p_" + name] = {"ms_per_picture": round(ms, 3), "pictures_per_s": round(1e3 / ms, 1),
                                                   "note": "reconstruction wavefronts only (no deblocking records)"}
        pic.close()
    return out


def h264_picture_leg(torch, dev, ev):
    """the picture layer: synthetic 1080p P-pictures (tools/h264_synth.py) through ffhip_h264_picture_flush — MC,
    residual add and the in-loop filter in decoder order.  A lone picture is a chain of latency-bound kernels; 16 pictures in flight,
    each on its own stream and host thread, fill the hardware queues (GPU_MAX_HW_QUEUES, see main())."""
    import threading
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools"))
    from h264_synth import record_p_picture
    from ffmpeg_amd import h264
    mb_w, mb_h, P, npic = 120, 68, 32, 16
    W, H = mb_w * 16, mb_h * 16
    sy, sc = W + 2 * P, W // 2 + P
    rng = np.random.default_rng(5)
    shp = ((H + 2 * P, sy), (H // 2 + P, sc), (H // 2 + P, sc))
    refs = [torch.randint(0, 256, s_, dtype=torch.uint8, device=dev) for s_ in shp]
    pics, dsts, streams = [], [], []
    for _ in range(npic):
        p = h264.Picture(mb_w, mb_h)
        record_p_picture(p, h264, mb_w, mb_h, sy, sc, P, rng)
        pics.append(p)
        dsts.append([torch.zeros(s_, dtype=torch.uint8, device=dev) for s_ in shp])
        streams.append(torch.cuda.Stream(device=dev))
    res = {}
    for n in (1, npic):
        for rep in range(2):                                  # the first round warms the pools up
            torch.cuda.synchronize()
            e0, e1 = ev(), ev()
            e0.record()
            for s_ in streams[:n]:
                s_.wait_event(e0)
            rounds = 4

            def work(i):
                for _ in range(rounds):
                    pics[i].flush(dsts[i], [sy, sc, sc], refs, stream=streams[i].cuda_stream)
            th = [threading.Thread(target=work, args=(i,)) for i in range(n)]
            for t in th:
                t.start()
            for t in th:
                t.join()
            cur = torch.cuda.current_stream()
            for s_ in streams[:n]:
                cur.wait_stream(s_)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / rounds
        res[n] = ms
    # the same pictures flushed TOGETHER (round 4, ffhip_h264_pictures_flush): each picture's prediction and residual launches, then the
    # in-loop filter of all their planes in one launch per plane kind; one host thread, one stream
    for rep in range(2):
        torch.cuda.synchronize()
        e0, e1 = ev(), ev()
        e0.record()
        for _ in range(4):
            h264.pictures_flush(pics, dsts, [sy, sc, sc], [refs] * npic)
        e1.record()
        torch.cuda.synchronize()
        res["batch"] = e0.elapsed_time(e1) / 4
    for p in pics:
        p.close()
    intra = h264_intra_picture_leg(torch, dev, ev, h264)
    try:
        intra = {**intra, **h264_intra_formats_leg(torch, dev, ev, h264)}
    except Exception as e:  # an extras leg must not cost the bench line
        intra = {**intra, "h264_intra_picture_formats": {"error": repr(e)[:200]}}
    return {**intra, "h264_picture_pipeline_1080p": {"ms_per_picture_alone": round(res[1], 3), "ms_per_picture_16_in_flight": round(res[npic] / npic, 3),
                                           "pictures_per_s_16_in_flight": round(1e3 * npic / res[npic], 1),
                                           "ms_per_batched_flush_of_16": round(res["batch"], 3),
                                           "pictures_per_s_batched_flush_of_16": round(1e3 * npic / res["batch"], 1),
                                           "GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES", "runtime default (4)"),
                                           "note": "P-pictures: qpel + chroma MC, idct_add on ~half of the blocks, deblocking in decoder order; one stream and host thread per picture"}}


def sws_ops_leg(torch, dev):
    """SwsOpBackend `hip`: micro-op lists of real conversions (tests/golden/sws_uops.npz, as the reference's graph
    cut them) on device-resident 4K pictures, 16 per launch.  Algorithmic bytes: what the read and the write of the list move."""
    from ffmpeg_amd import swsops as O
    res = {}
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "golden", "sws_uops.npz"))
    want = {"yuv444p rgb24": "sws_ops_yuv444p_rgb24_4k", "rgb24 yuv444p": "sws_ops_rgb24_yuv444p_4k", "rgba argb": "sws_ops_rgba_argb_4k",
            "yuv444p10le rgb48le": "sws_ops_yuv444p10_rgb48_4k"}
    w, h, nf = 3840, 2160, 16
    for j in range(int(z["ncases"][0])):
        name = bytes(z["c%d_name" % j]).decode()
        size = tuple(int(v) for v in z["c%d_size" % j])
        if name not in want or size[0] != size[2] or size[1] != size[3] or want[name] in res:
            continue
        lst = O.load_lists(z, "c%d_" % j)[0]
        cu = O.CompiledUOps(lst.uops, lst.n)
        pin, bin_ = O.rw_geometry(lst.read)
        pout, bout = O.rw_geometry(lst.write)
        ls_in, ls_out = w * bin_ // 8, w * bout // 8
        src = [torch.randint(0, 256, (nf, h, ls_in), dtype=torch.uint8, device=dev) for c in range(4) if pin >> c & 1]
        dst = [torch.empty((nf, h, ls_out), dtype=torch.uint8, device=dev) for c in range(4) if pout >> c & 1]
        e = O.plain_exec(lst, [t.data_ptr() for t in src], [ls_in] * len(src), [t.data_ptr() for t in dst], [ls_out] * len(dst), w, h,
                         cu.block_size)
        ip, op = [h * ls_in] * len(src) + [0] * (4 - len(src)), [h * ls_out] * len(dst) + [0] * (4 - len(dst))
        for _ in range(2):
            cu.run_dev(e, w, h, nf, ip, op)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            cu.run_dev(e, w, h, nf, ip, op)
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 10
        byt = nf * h * (ls_in * len(src) + ls_out * len(dst))
        res[want[name]] = {"Mpixels/s": round(nf * w * h / (ms * 1e-3) / 1e6, 1), "GB/s": round(byt / (ms * 1e-3) / 1e9, 1),
                           "hbm_frac": round(byt / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "frames": nf, "ms": round(ms, 4),
                           "uops": lst.n}
        cu.close()
        del src, dst
    return res


def idct_leg(torch, dist, dev, world, reps=5, torch_alloc=False, tries=1):
    """BASELINE's second metric at every N: h264 idct8_add over 32 4K luma planes per rank (129,600 blocks each, 384 B per
    block), ranks independent (blocks shard with no collective), barrier + synchronize on both sides, MAX over ranks."""
    from ffmpeg_amd import h264
    planes, stride = 32, 3840
    nb = planes * 129600
    # (plain allocations here: in frame memory this leg measured 0.669 and 0.565 of HBM on two runs of one box, in torch's 0.63 - 0.70: placed
    # by trial like the frame batches — a new allocation while the last one is held lands on other pages)
    by, bx = torch.meshgrid(torch.arange(planes * 270, device=dev), torch.arange(480, device=dev), indexing="ij")
    offs = (by * 8 * stride + bx * 8).to(torch.int32).reshape(-1).contiguous()
    del by, bx

    def make():
        pl_ = torch.randint(0, 256, (planes * 2160, stride), dtype=torch.uint8, device=dev)
        c0 = torch.randint(-512, 512, (nb, 64), dtype=torch.int16, device=dev)
        return pl_, [c0.clone() for _ in range(reps + 1)], c0

    def measure(cand):
        pl_, bf, c0 = cand
        h264.idct_add_batch(h264.IDCT8, pl_, stride, offs, bf[reps])
        bf[reps].copy_(c0)
        x0, x1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        x0.record()
        h264.idct_add_batch(h264.IDCT8, pl_, stride, offs, bf[reps])
        x1.record()
        torch.cuda.synchronize()
        bf[reps].copy_(c0)
        return x0.elapsed_time(x1)
    (plane, bufs, coefs0), trial_ms = place_best(torch, make, measure, lambda c: None, tries, nb * 384 / (0.70 * HBM_PEAK_GBS * 1e9) * 1e3)
    h264.idct_add_batch(h264.IDCT8, plane, stride, offs, bufs[reps])
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(reps):
        h264.idct_add_batch(h264.IDCT8, plane, stride, offs, bufs[i])
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([el], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    ms = el / reps * 1e3
    gbs = nb * 384 / (ms * 1e-3) / 1e9
    return {"metric": "h264_idct8_add_Gblocks_per_s", "value": round(world * nb / (ms * 1e-3) / 1e9, 3), "unit": "Gblocks/s",
            "blocks_per_gpu": nb, "ms_per_pass": round(ms, 4), "hbm_frac_per_gpu": round(gbs / HBM_PEAK_GBS, 4), "planes_per_gpu": planes,
            "placement_trials": len(trial_ms), "placement_frac_first": round(nb * 384 / (trial_ms[0] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}


def strong_leg(torch, dist, dev, ctx, S, rank, world, n_total, reps=3):
    """The batch starts and ends on rank 0: scatter the source planes over RCCL (point-to-point sends, xGMI), convert the
    shard, gather the converted planes back.  Phases separated by barriers and timed on their own."""
    from ffmpeg_amd import dist as D
    full = None
    if rank == 0:
        g = torch.Generator(device=dev)
        g.manual_seed(0xF0F00002)
        full = [torch.randint(0, 256, (n_total, r, c), dtype=torch.uint8, device=dev, generator=g)
                for r, c in S.plane_shapes(NV12, SRC_W, SRC_H)]
    tmpl = [torch.empty((0, r, c), dtype=torch.uint8, device=dev) for r, c in S.plane_shapes(NV12, SRC_W, SRC_H)]
    lo, hi = D.shard_range(n_total, rank, world)
    dst = [torch.empty((hi - lo, r, c), dtype=torch.uint8, device=dev) for r, c in S.plane_shapes(NV12, DST_W, DST_H)]
    ts = [0.0, 0.0, 0.0]

    def sync():
        dist.barrier()
        torch.cuda.synchronize()

    for it in range(reps + 1):
        sync()
        t0 = time.perf_counter()
        shard = [D.scatter_batch(full[p] if rank == 0 else tmpl[p], n_total) for p in range(2)]
        sync()
        t1 = time.perf_counter()
        if hi > lo:
            ctx.scale_batch(shard, dst)
        sync()
        t2 = time.perf_counter()
        out = [D.gather_batch(dst[p], n_total) for p in range(2)]
        sync()
        t3 = time.perf_counter()
        if it:                                                    # first round warms the RCCL channels up
            ts = [ts[0] + t1 - t0, ts[1] + t2 - t1, ts[2] + t3 - t2]
        del shard, out
    ts = [x / reps * 1e3 for x in ts]
    px = n_total * DST_W * DST_H
    in_b = n_total * SRC_W * SRC_H * 3 // 2
    out_b = n_total * DST_W * DST_H * 3 // 2
    return {"frames_total": n_total, "scatter_ms": round(ts[0], 3), "convert_ms": round(ts[1], 3), "gather_ms": round(ts[2], 3),
            "scatter_GB/s": round(in_b * (world - 1) / world / ts[0] / 1e6, 1), "gather_GB/s": round(out_b * (world - 1) / world / ts[2] / 1e6, 1),
            "Mpixels/s_convert_only": round(px / ts[1] / 1e3, 1), "Mpixels/s_end_to_end": round(px / sum(ts) / 1e3, 1),
            "note": "rank 0 holds the batch; ceil(n/world) contiguous frames per rank; p2p isend/recv over RCCL, no reduction"}


def single_process(args, torch, S, _lib):
    """--single-process: the reference's own execution model — one process, worker threads (libavcodec/pthread_frame.c,
    libswscale/swscale.c:1645-1679) — over N GPUs.  One host thread per member of an FFHipDeviceSet; every thread binds to its
    device, creates its context there and queues the steps on the member's stream; a barrier + device synchronisation on both
    sides of the timed region, one clock.  Same workload, same sharding (256 frames per GPU) and the same JSON line as the
    torchrun form; the `strong` leg scatters a root batch with ffhip_batch_scatter (hipMemcpyPeerAsync over xGMI), converts and
    gathers it back."""
    import threading
    L = _lib.lib()
    n, world = args.frames, args.gpus
    if not torch.cuda.is_available() or L.ffhip_device_count() < world:
        raise SystemExit("--single-process --gpus %d needs %d visible HIP devices (%d here)" % (world, world, L.ffhip_device_count()))
    ds = C.c_void_p()
    _lib.check(L.ffhip_device_set_create(C.byref(ds), (C.c_int * world)(*range(world)), world), "ffhip_device_set_create")
    bar = threading.Barrier(world + 1)
    state = [None] * world
    errs = []

    def sync_all():
        _lib.check(L.ffhip_device_set_synchronize(ds), "ffhip_device_set_synchronize")

    def worker(i):
        try:
            _lib.check(L.ffhip_device_set_bind(ds, i), "ffhip_device_set_bind")
            dev = torch.device("cuda", i)
            st = L.ffhip_device_set_stream(ds, i)
            ctx = S.SwsContext(SRC_W, SRC_H, NV12, DST_W, DST_H, NV12, S.SWS_BICUBIC)
            gen = torch.Generator(device=dev)
            gen.manual_seed(0xF0F00002 + i)
            src = [torch.randint(0, 256, (n, r, c), dtype=torch.uint8, device=dev, generator=gen) for r, c in S.plane_shapes(NV12, SRC_W, SRC_H)]
            dst = [torch.empty((n, r, c), dtype=torch.uint8, device=dev) for r, c in S.plane_shapes(NV12, DST_W, DST_H)]
            torch.cuda.synchronize(dev)
            state[i] = (ctx, src, dst)
            for _ in range(args.warmup):
                ctx.scale_batch(src, dst, st)
            _lib.check(L.ffhip_stream_synchronize(st), "sync")
            bar.wait()            # everyone warm
            bar.wait()            # clock started
            for _ in range(args.steps):
                ctx.scale_batch(src, dst, st)
            _lib.check(L.ffhip_stream_synchronize(st), "sync")
            bar.wait()            # everyone done
        except Exception as e:  # noqa: BLE001
            errs.append(repr(e))
            bar.abort()
    ths = [threading.Thread(target=worker, args=(i,)) for i in range(world)]
    [t.start() for t in ths]
    try:
        bar.wait()
        t0 = time.perf_counter()
        bar.wait()
        bar.wait()
        elapsed = time.perf_counter() - t0
    except threading.BrokenBarrierError:
        [t.join() for t in ths]
        raise SystemExit("single-process bench failed: %s" % errs)
    [t.join() for t in ths]
    sync_all()
    assert all(float(st_[2][0][:1].to(torch.float64).sum().item()) > 0 for st_ in state)

    # strong leg: `n` frames on member 0, scattered over xGMI, converted on every member, gathered back
    strong = None
    if world > 1 and not args.no_strong:
        lo, hi = C.c_int64(), C.c_int64()
        ctx0, src0, dst0 = state[0]
        shards = []
        for i in range(world):
            L.ffhip_shard_range(n, i, world, C.byref(lo), C.byref(hi))
            k = hi.value - lo.value
            dev = torch.device("cuda", i)
            shards.append(([src0[0][:k], src0[1][:k]] if i == 0 else
                           [torch.empty((k,) + tuple(t.shape[1:]), dtype=torch.uint8, device=dev) for t in src0],
                           [dst0[0][:k], dst0[1][:k]] if i == 0 else
                           [torch.empty((k,) + tuple(t.shape[1:]), dtype=torch.uint8, device=dev) for t in dst0]))
        out_full = [torch.empty_like(t) for t in dst0]
        for i in range(world):
            torch.cuda.synchronize(torch.device("cuda", i))

        def ptrs(which, pl):
            return (C.c_void_p * world)(*[sh[which][pl].data_ptr() for sh in shards])
        phases = {"scatter_ms": 0.0, "convert_ms": 0.0, "gather_ms": 0.0}
        reps = 3
        for rep in range(reps + 1):
            t0 = time.perf_counter()
            for pl in range(2):
                _lib.check(L.ffhip_batch_scatter(ds, 0, src0[pl].data_ptr(), src0[pl].stride(0), n, ptrs(0, pl)), "scatter")
            sync_all()
            t1 = time.perf_counter()
            for i in range(world):
                if shards[i][0][0].shape[0]:
                    state[i][0].scale_batch(shards[i][0], shards[i][1], L.ffhip_device_set_stream(ds, i))
            sync_all()
            t2 = time.perf_counter()
            for pl in range(2):
                _lib.check(L.ffhip_batch_gather(ds, 0, out_full[pl].data_ptr(), out_full[pl].stride(0), n, ptrs(1, pl)), "gather")
            sync_all()
            t3 = time.perf_counter()
            if rep:   # the first round is the warm-up (peer mappings, first touches)
                phases["scatter_ms"] += (t1 - t0) * 1e3 / reps
                phases["convert_ms"] += (t2 - t1) * 1e3 / reps
                phases["gather_ms"] += (t3 - t2) * 1e3 / reps
        tot = sum(phases.values())
        strong = {k: round(v, 3) for k, v in phases.items()}
        strong.update({"frames": n, "total_ms": round(tot, 3), "Mpixels/s": round(n * DST_W * DST_H / (tot * 1e-3) / 1e6, 1),
                       "transport": "hipMemcpyPeerAsync over xGMI, one copy per member on the member's stream (ffhip_batch_scatter/gather)"})
    ms_per_step = elapsed / args.steps * 1e3
    value = world * n * DST_W * DST_H * args.steps / elapsed / 1e6
    achieved = n * BYTES_PER_FRAME / (ms_per_step * 1e-3) / 1e9
    line = {
        "metric": "swscale_nv12_1080p_to_4k_bicubic_Mpixels_per_s", "value": round(value, 1), "unit": "Mpixels/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": "swscale bicubic nv12 1920x1080 -> nv12 3840x2160, %d-frame batch per GPU, "
                               "frames resident in HBM (BASELINE.json configs[1])" % n,
                   "frames_per_gpu": n, "flags": "SWS_BICUBIC", "sharding": "frames/GPU, no data-path collective",
                   "launch": "single process: one host thread + FFHipDeviceSet member per GPU through the C ABI"},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
                     "note": "per GPU, from the wall clock of the timed region (launch overhead included), not kernel events"},
    }
    if strong is not None:
        line["strong"] = strong
    print(json.dumps(line), flush=True)
    for st_ in state:
        st_[0].close()
    L.ffhip_device_set_free(C.byref(ds))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)  # a step is < 1 ms: 200 steady-state steps, not an 18 ms glimpse
    ap.add_argument("--warmup", type=int, default=50)  # ~50 ms: the clocks have settled (20 / 3 measured 3-4 % slower)
    ap.add_argument("--frames", type=int, default=256, help="frames per GPU per step (BASELINE configs[1]: 256)")
    ap.add_argument("--settle-ms", type=int, default=300,
                    help="device settle time BEFORE the W warmup steps: the same launches, untimed, until the clocks and the page tables "
                         "have reached the steady state a resident converter runs in (the driver's --steps 20 --warmup 5 is a 23 ms job "
                         "on a cold device otherwise); reported in config.settle_ms, 0 switches it off")
    ap.add_argument("--sustain-ms", type=int, default=1200,
                    help="after the timed steps: back-to-back launches for this long, reported as roofline.frac_sustained (0 = skip)")
    ap.add_argument("--placement-trials", type=int, default=4,
                    help="the frame batches are placed by trial before the timed region: up to this many allocations, the fastest kept "
                         "(1: the first allocation as it comes)")
    ap.add_argument("--placement-good-frac", type=float, default=0.63,
                    help="a placement whose trial reaches this fraction of the HBM peak is taken without further trials (the slow layout reads 0.56 - 0.58 "
                         "in a trial, the others 0.62 - 0.66)")
    ap.add_argument("--torch-alloc", action="store_true",
                    help="frame batches from torch's allocator (one hipMalloc each) instead of libffhip's frame memory (ffhip_frames_alloc)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true")
    ap.add_argument("--no-strong", action="store_true", help="skip the RCCL scatter/convert/gather leg at N>1")
    ap.add_argument("--no-pmc", action="store_true", help="do not spawn the two rocprofv3 PMC passes that measure roofline.traffic")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)   # what measure_traffic() profiles
    ap.add_argument("--single-process", action="store_true",
                    help="ONE process drives --gpus N devices through the C ABI (one host thread + FFHipDeviceSet member per GPU) "
                         "instead of one torchrun rank per GPU")
    args = ap.parse_args()

    # one rank, many streams (the picture-layer leg of the extras): the HIP runtime folds all streams of a process onto
    # GPU_MAX_HW_QUEUES hardware queues, 4 by default; 16 is the measured optimum there.  Read at runtime
    # initialisation, hence before torch is imported; the timed headline runs on one stream and does not depend on it.
    if int(os.environ.get("WORLD_SIZE", "1")) == 1:
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
    import torch
    import torch.distributed as dist
    from ffmpeg_amd import swscale as S, _lib

    if args.single_process:
        return single_process(args, torch, S, _lib)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if args.gpus > 1 and world == 1:
        raise SystemExit("launch N>1 with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...  "
                         "(or --single-process: one process, one host thread per GPU)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (libffhip has no CPU path)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    _lib.check(_lib.lib().ffhip_set_device(local_rank), "ffhip_set_device")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    n = args.frames
    ctx = S.SwsContext(SRC_W, SRC_H, NV12, DST_W, DST_H, NV12, S.SWS_BICUBIC)
    gen = torch.Generator(device=dev)
    gen.manual_seed(0xF0F00002 + rank)                       # SURVEY.md §8d seed, one shard per rank
    # the frame batches live in libffhip's frame memory (ffhip_frames_alloc, include/ffhip.h: physical chunks of 16 MiB in shuffled
    # order); --torch-alloc takes torch's allocator instead (one hipMalloc per plane batch, whose
    # physical layout decides between 0.58 and 0.65 for the same launch, profiles/r06_alloc_vmm_sweep_*.txt)
    stream = torch.cuda.current_stream()

    def make_batches():
        a, b = frame_batches(torch, dev, _lib, [(n, r, c) for r, c in S.plane_shapes(NV12, SRC_W, SRC_H)],
                             [(n, r, c) for r, c in S.plane_shapes(NV12, DST_W, DST_H)], args.torch_alloc)
        gen.manual_seed(0xF0F00002 + rank)                   # (every placement holds the same frames)
        for t in a:
            t.random_(0, 256, generator=gen)
        return a, b

    def trial_ms(cand, warm=24, k=8):
        a, b = cand
        for _ in range(warm):
            ctx.scale_batch(a, b, stream.cuda_stream)
        x0, x1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        x0.record(stream)
        for _ in range(k):
            ctx.scale_batch(a, b, stream.cuda_stream)
        x1.record(stream)
        torch.cuda.synchronize()
        return x0.elapsed_time(x1) / k
    # the batches are PLACED before anything is timed (place_best: the physical pages an allocation gets decide between two speeds of the same
    # launch; --placement-trials 1 takes the first allocation as it comes); not part of the W warmup / K timed steps
    tries = 1 if (args.torch_alloc or args.pmc_child) else args.placement_trials
    (src, dst), placement_ms = place_best(torch, make_batches, trial_ms if tries > 1 else (lambda c: 0.0), lambda c: free_batches(torch, c[0], c[1]), tries,
                                          n * BYTES_PER_FRAME / (args.placement_good_frac * HBM_PEAK_GBS * 1e9) * 1e3)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if args.pmc_child:   # a few launches of the headline kernel and nothing else, for the PMC passes
        for _ in range(6):
            ctx.scale_batch(src, dst, stream.cuda_stream)
        torch.cuda.synchronize()
        return
    def timed(k):
        """k steps, one HIP event pair per step on the launch stream: (wall seconds incl. the closing barrier, mean kernel ms)"""
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(k)]
        t0 = time.perf_counter()
        for e0, e1 in evs:
            e0.record(stream)                                    # HIP events on the launch stream
            ctx.scale_batch(src, dst, stream.cuda_stream)
            e1.record(stream)
        barrier()
        el = time.perf_counter() - t0
        return el, sum(e0.elapsed_time(e1) for e0, e1 in evs) / max(k, 1)

    # (1) COLD: the driver's protocol on the device as this process found it — W untimed steps, K timed ones, nothing before them.
    # Reported as roofline.frac_cold (round-over-round comparable with rounds 1-4); the headline below is taken after the settle phase.
    cold_ms = None
    if args.settle_ms > 0:
        for _ in range(args.warmup):
            ctx.scale_batch(src, dst, stream.cuda_stream)
        barrier()
        _, cold_ms = timed(args.steps)
    # (2) SETTLED: settle_ms of the same launches untimed, then the W warmup steps and EXACTLY K timed steps — the headline
    t_settle = time.perf_counter()
    while (time.perf_counter() - t_settle) * 1e3 < args.settle_ms:
        for _ in range(16):
            ctx.scale_batch(src, dst, stream.cuda_stream)
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        ctx.scale_batch(src, dst, stream.cuda_stream)
    barrier()
    elapsed, kernel_ms = timed(args.steps)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    # (3) SUSTAINED: back-to-back launches for >= sustain_ms right after, one event pair around each block of 64 (rank 0 reports its own)
    sust_ms, sust_n, sust_wall = None, 0, 0.0
    if args.sustain_ms > 0:
        tot, t_s = 0.0, time.perf_counter()
        while (time.perf_counter() - t_s) * 1e3 < args.sustain_ms:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(stream)
            for _ in range(64):
                ctx.scale_batch(src, dst, stream.cuda_stream)
            b.record(stream)
            torch.cuda.synchronize()
            tot += a.elapsed_time(b)
            sust_n += 64
        sust_wall = time.perf_counter() - t_s
        sust_ms = tot / sust_n

    # sanity outside the timed region: every rank produced non-trivial output
    chk = torch.stack([d[:2].to(torch.float64).sum() for d in dst]).sum().reshape(1)
    if world > 1:
        dist.all_reduce(chk)
    assert float(chk.item()) > 0

    # BASELINE's second metric, every N; then (N>1) the scatter -> convert -> gather path over RCCL
    idct = idct_leg(torch, dist, dev, world, torch_alloc=args.torch_alloc, tries=args.placement_trials)
    strong = None
    if world > 1 and not args.no_strong:
        del src, dst
        torch.cuda.empty_cache()
        strong = strong_leg(torch, dist, dev, ctx, S, rank, world, n)

    kname = "k_sws_up2<6, 1, 0, 0, 1>" if ctx.up2_path else "k_sws_colwalk<1,6,false,true,true>" if ctx.fast_path else "k_sws_scale_yuv<4,4>"
    traffic, traffic_source = None, None
    if rank == 0 and world == 1 and not args.no_pmc:
        traffic, traffic_source = measure_traffic(kname, n)
    if traffic is None:   # no rocprofv3 here (or it failed): the committed PMC passes of the same kernel and batch, labelled as such
        try:
            pm = json.load(open(os.path.join(ROOT, "profiles", "r02_bench_pmc.json")))
            for k, v in pm.items():
                if k.split("<")[0] == kname.split("<")[0] and n == 256:
                    traffic = round(v["traffic_bytes_per_launch"])
                    traffic_source = "committed profile profiles/r02_bench_pmc.json (not measured in this run)"
        except (OSError, ValueError, KeyError):
            pass

    # streaming PROBES of this box: per traffic mix the BEST of a 24-variant sweep of plain streaming kernels (1 / 2 / 8 16-byte accesses in
    # flight per lane, plain or non-temporal, two grid sizes, grid-stride or XCD-adjacent private slices; ffhip_membw_probe, round 5 —
    # tools/ubench/membw2.hip is the long form, profiles/r05_membw2.txt its 146 variants on one box), and the runtime's own
    # hipMemcpyDtoDAsync.  They say what this box gives a kernel with no arithmetic at that mix; the achievable yardstick stays the larger
    # of the guide's measured 6.29 TB/s and the best probe
    probes, yard = None, HBM_GUIDE_ACHIEVABLE_GBS
    if rank == 0:
        probes = {}
        for pat, name in ((2, "copy"), (1, "write"), (3, "read1_write4"), (4, "read1_write2"), (0, "read"), (5, "hipMemcpyDtoDAsync")):
            g = C.c_double(0)
            if _lib.lib().ffhip_membw_probe(pat, 2 << 30, 10, C.byref(g)) == 0:
                probes[name] = round(g.value, 1)
        yard = max([HBM_GUIDE_ACHIEVABLE_GBS] + list(probes.values()))

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        total_px = world * n * DST_W * DST_H * args.steps
        value = total_px / elapsed / 1e6
        alg = n * BYTES_PER_FRAME
        achieved = alg / (kernel_ms * 1e-3) / 1e9
        roof = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                "kernel": kname, "kernel_ms": round(kernel_ms, 4),
                "algorithmic_bytes_per_launch": alg, "traffic_source": traffic_source,
                "achievable_GB/s": yard, "frac_of_achievable": round(achieved / yard, 4)}
        # the protocol beside the headline: the same K steps on the device as found (no settle), and >= sustain_ms of back-to-back launches
        if cold_ms:
            roof["frac_cold"] = round(alg / (cold_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
            roof["kernel_ms_cold"] = round(cold_ms, 4)
        if sust_ms:
            roof["frac_sustained"] = round(alg / (sust_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
            roof["kernel_ms_sustained"] = round(sust_ms, 4)
            roof["sustained_launches"] = sust_n
            roof["sustained_wall_s"] = round(sust_wall, 3)
        # how the batches came to lie where they do: the trials of place_best() (fractions of the peak, 8 launches each, before the timed region)
        if len(placement_ms) > 1 or (placement_ms and placement_ms[0]):
            fr = [round(alg / (m * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) for m in placement_ms if m]
            roof["placement_trials"] = len(fr)
            roof["placement_frac_first"] = fr[0]
            roof["placement_frac_min"] = min(fr)
            roof["placement_frac_kept"] = max(fr)
            for i, f in enumerate(fr):
                roof["placement_frac_%d" % i] = f
        for k, v in (probes or {}).items():
            roof["probe_%s_GBs" % k] = v
        if probes and probes.get("read1_write4"):
            roof["frac_of_probe_read1_write4"] = round(achieved / probes["read1_write4"], 4)
        line = {
            "metric": "swscale_nv12_1080p_to_4k_bicubic_Mpixels_per_s", "value": round(value, 1), "unit": "Mpixels/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "swscale bicubic nv12 1920x1080 -> nv12 3840x2160, %d-frame batch per GPU, "
                                   "frames resident in HBM (BASELINE.json configs[1])" % n,
                       "frames_per_gpu": n, "flags": "SWS_BICUBIC", "sharding": "frames/rank, no data-path collective",
                       "settle_ms": args.settle_ms, "sustain_ms": args.sustain_ms,
                       "frames_alloc": "torch (hipMalloc)" if args.torch_alloc else "ffhip_frames_alloc: 16 MiB physical chunks, shuffled",
                       "placement_trials_max": tries},
        }
        ex, cpu = None, None
        if world == 1:
            # north_star's first target, measured in every N = 1 run (with or without the extras)
            try:
                rgb = rgb24_leg(torch, dev, args.torch_alloc, args.placement_trials)
                roof.update({"rgb24_4k_frac": rgb["hbm_frac"], "rgb24_4k_ms": rgb["ms"], "rgb24_4k_Mpix": rgb["Mpixels/s"], "rgb24_4k_frames": rgb["frames"],
                             "rgb24_4k_tuned_numbering": rgb["tuned_numbering"], "rgb24_4k_placement_trials": rgb["placement_trials"],
                             "rgb24_4k_placement_frac_first": rgb["placement_trial_fracs"][0]})
                if "frac_of_box_probe_read1_write2" in rgb:
                    roof["rgb24_4k_frac_of_probe_read1_write2"] = rgb["frac_of_box_probe_read1_write2"]
            except Exception as e:  # never costs the headline line
                rgb = {"error": repr(e)[:200]}
            if not args.no_extras:
                try:
                    ex = extras(torch, dev, args.torch_alloc)
                except Exception as e:  # extras never invalidate the headline line
                    ex = {"error": repr(e)[:300]}
                ex["yuv420p_rgb24_4k"] = rgb
            if not args.no_cpu_baseline:
                cpu = cpu_baseline()
        roof["idct8_Gblocks"] = idct["value"]
        roof["idct8_frac"] = idct["hbm_frac_per_gpu"]
        roof["idct8_ms"] = idct["ms_per_pass"]
        if ex is not None:
            # BASELINE configs[2..4] (and their neighbours) as flat scalars: fraction of the roof that bounds each
            for key, fld, name in (("h264_qpel16_mixed", "hbm_frac", "qpel16_mixed_frac"), ("h264_qpel16_mixed", "Mpixels/s", "qpel16_mixed_Mpix"),
                                   ("h264_v_loop_filter_luma", "Medges/s", "h264_v_loop_filter_luma_Medges"),
                                   ("h264_h_loop_filter_luma", "Medges/s", "h264_h_loop_filter_luma_Medges"),
                                   ("h264_deblock_frame_4k", "Mpixels/s", "h264_deblock_frame_4k_Mpix"),
                                   ("mdct1024_fwd", "hbm_frac", "mdct1024_fwd_frac"), ("mdct1024_fwd", "Mtransforms/s", "mdct1024_fwd_Mtransforms"),
                                   ("mdct1024_inv", "hbm_frac", "mdct1024_inv_frac"), ("mdct1024_inv", "Mtransforms/s", "mdct1024_inv_Mtransforms"),
                                   ("me_esa_sad_r7", "MB-searches/s", "me_esa_sad_r7_MBsearches"), ("me_esa_sad_r7", "sad_issue_roof_frac", "me_esa_sad_r7_issue_frac"),
                                   ("me_esa_satd_r7", "MB-searches/s", "me_esa_satd_r7_MBsearches"),
                                   ("me_esa_satd_r7", "valu_issue_roof_frac", "me_esa_satd_r7_issue_frac"),
                                   ("hevc_qpel_uni16_mixed", "hbm_frac", "hevc_qpel_uni16_mixed_frac"), ("vp9_mc16_8tap_mixed", "hbm_frac", "vp9_mc16_8tap_mixed_frac"),
                                   ("h264_chroma_mc8_mixed", "hbm_frac", "h264_chroma_mc8_mixed_frac"), ("mdct1024_int32_fwd", "hbm_frac", "mdct1024_int32_fwd_frac"),
                                   ("sws_p010_720p_to_1080p_bicubic", "hbm_frac", "sws_p010_720p_to_1080p_frac"),
                                   ("sws_p010_4k_to_1440p_bicubic", "hbm_frac", "sws_p010_4k_to_1440p_frac"),
                                   ("sws_yuv420p10_1080p_to_1440p_bicubic", "hbm_frac", "sws_yuv420p10_1080p_to_1440p_frac"),
                                   ("sws_nv12_1080p_to_720p_bicubic", "hbm_frac", "sws_nv12_1080p_to_720p_frac"),
                                   ("sws_p010_1080p_to_4k_bicubic", "hbm_frac", "sws_p010_1080p_to_4k_frac"),
                                   ("sws_p010_4k_to_1080p_bicubic", "hbm_frac", "sws_p010_4k_to_1080p_frac"),
                                   ("dctI_64", "fp64_TFLOP/s", "dctI_64_fp64_TFLOPs"),
                                   ("sws_host_pointer_end_to_end", "ms_per_frame", "sws_host_pointer_ms_per_frame")):
                if isinstance(ex.get(key), dict) and fld in ex[key]:
                    roof[name] = ex[key][fld]
        if cpu is not None:
            # GPU : CPU per configuration (reported, not the target): the GPU rate over the reference's C path on every usable host core
            def ratio(name, gpu, cpu_key):
                c = cpu.get(cpu_key)
                if gpu and c:
                    roof[name] = round(gpu / c, 1)
            ratio("headline_x_cpu", value, "sws_slice_threads_Mpix" if "sws_slice_threads_Mpix" in cpu else "sws_1_thread_Mpix")
            ratio("rgb24_4k_x_cpu_all_cores", roof.get("rgb24_4k_Mpix"), "rgb24_4k_all_cores_Mpix")
            ratio("idct8_x_cpu_all_cores", roof.get("idct8_Gblocks"), "idct8_all_cores_Gblocks")
            ratio("qpel16_mixed_x_cpu_all_cores", roof.get("qpel16_mixed_Mpix"), "h264_qpel16_mixed_all_cores_Mpix")
            ratio("h264_v_loop_filter_luma_x_cpu_all_cores", roof.get("h264_v_loop_filter_luma_Medges"), "h264_v_loop_filter_luma_all_cores_Medges")
            ratio("h264_h_loop_filter_luma_x_cpu_all_cores", roof.get("h264_h_loop_filter_luma_Medges"), "h264_h_loop_filter_luma_all_cores_Medges")
            ratio("mdct1024_fwd_x_cpu_all_cores", roof.get("mdct1024_fwd_Mtransforms"), "mdct1024_fwd_all_cores_Mtransforms")
            ratio("mdct1024_inv_x_cpu_all_cores", roof.get("mdct1024_inv_Mtransforms"), "mdct1024_inv_all_cores_Mtransforms")
            ratio("me_esa_sad_r7_x_cpu_all_cores", roof.get("me_esa_sad_r7_MBsearches"), "me_esa_sad_r7_all_cores_MBsearches")
            ratio("me_esa_satd_r7_x_cpu_all_cores", roof.get("me_esa_satd_r7_MBsearches"), "me_esa_satd_r7_all_cores_MBsearches")
        # order: the long tail first, what BASELINE names last (the driver's record keeps the END of this line)
        if ex is not None:
            last = ["sws_host_pointer_end_to_end", "h264_deblock_frame_4k", "h264_h_loop_filter_luma", "h264_v_loop_filter_luma", "h264_qpel16_mixed",
                    "mdct1024_inv", "mdct1024_fwd", "me_esa_sad_r7", "me_esa_satd_r7", "h264_idct8_add", "yuv420p_rgb24_4k"]
            for v in ex.values():
                if isinstance(v, dict):
                    v.pop("note", None)     # prose lives in DESIGN.md 5, not in the line
            line["extras"] = {**{k: v for k, v in ex.items() if k not in last}, **{k: ex[k] for k in last if k in ex}}
        if strong is not None:
            line["strong"] = strong
        line["idct"] = idct
        if cpu is not None:
            line["cpu_baseline"] = cpu
        line["roofline"] = roof
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
