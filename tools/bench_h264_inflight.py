#!/usr/bin/env python3
"""tools/bench_h264_inflight.py — several H.264 pictures in flight through the caller-side batching layer: N FFHipH264Picture
objects (1080p P-pictures: every macroblock 16x16 uni-predicted at a random quarter-sample position, residuals on about half of the
8x8 luma / a third of the 4x4 chroma blocks, every edge filtered), each flushed on its own stream by its own host thread.  A lone picture is a chain of
mb_w + 2 mb_h dependent wavefront steps and leaves the device mostly idle; pictures of different streams (a transcoding farm) or
of one stream's independent frames fill it.  HIP-event time from the first flush to the last stream's completion."""
import json
import os
import sys
import threading

import numpy as np

# HIP maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4): pictures on streams that share a queue run one after the
# other.  16 is the measured optimum for this layer (8: 2,300, 16: 3,600 pictures/s with 32 in flight; 32 and more collapse);
# the variable is read when the runtime initialises, so it is set before torch is imported.  Override from the environment.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402
from ffmpeg_amd import h264  # noqa: E402
from h264_synth import record_p_picture  # noqa: E402

dev = torch.device("cuda", 0)
mb_w, mb_h, P = 120, 68, 32
W, H = mb_w * 16, mb_h * 16
sy, sc = W + 2 * P, W // 2 + P
NMAX = int(sys.argv[1]) if len(sys.argv) > 1 else 16
rng = np.random.default_rng(5)


def planes(zero):
    mk = torch.zeros if zero else (lambda s, **k: torch.randint(0, 256, s, **k))
    return [mk((H + 2 * P, sy), dtype=torch.uint8, device=dev), mk((H // 2 + P, sc), dtype=torch.uint8, device=dev),
            mk((H // 2 + P, sc), dtype=torch.uint8, device=dev)]


def record(pic):
    record_p_picture(pic, h264, mb_w, mb_h, sy, sc, P, rng)


refs = planes(False)
pics, dsts, streams = [], [], []
for i in range(NMAX):
    p = h264.Picture(mb_w, mb_h)
    record(p)
    pics.append(p)
    dsts.append(planes(True))
    streams.append(torch.cuda.Stream(device=dev))
strides = [sy, sc, sc]
n = 1
while n <= NMAX:
    for rep in range(2):                                  # the first round warms the pools up
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for s in streams[:n]:
            s.wait_event(e0)
        rounds = 4

        def work(i):                                       # one host thread per stream, as a decoder has (ctypes drops the GIL)
            for _ in range(rounds):
                pics[i].flush(dsts[i], strides, refs, stream=streams[i].cuda_stream)
        th = [threading.Thread(target=work, args=(i,)) for i in range(n)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        cur = torch.cuda.current_stream()
        for s in streams[:n]:
            cur.wait_stream(s)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / rounds
    print(json.dumps({"case": "h264 1080p P-pictures through ffhip_h264_picture_flush, %d in flight (one stream each)" % n,
                      "GPU_MAX_HW_QUEUES": os.environ["GPU_MAX_HW_QUEUES"],
                      "ms_per_round": round(ms, 3), "ms_per_picture": round(ms / n, 3), "pictures_per_s": round(1e3 * n / ms, 1)}), flush=True)
    n *= 2
for p in pics:
    p.close()
