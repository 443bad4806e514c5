"""What an MBAFF frame costs in the H.264 picture layer (round 6): the reference's whole decoder (oracle/_ref/libffref_h264dec.so) decodes a
1920 x 1088 MBAFF stream from tests/h264_bitstream.py with the recorder installed; every frame's four objects are flushed on the device
mirror of the picture arena and timed — the three inter objects (ffhip_h264_picture_flush) and the chains (ffhip_h264_mbaff_flush: intra
reconstruction + the recorded loop-filter calls, one wave per macroblock-pair row).  Prints one JSON line; the pictures are compared with
the plain decode first.  Usage: python tools/bench_h264_mbaff.py [frames]"""
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch

import h264_stream_driver as D
from ffmpeg_amd import _lib


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    mb_w, mb_h = 120, 68
    aus, ws = D.stream_mbaff_p(seed=47, mb_w=mb_w, mb_h=mb_h, n=n)
    plain, st0, _ = D.decode(aus, arena_bytes=256 << 20)
    L = _lib.lib()
    state, t_inter, t_chain, calls, intra = {}, [], [], [], []

    def make(base, size):
        dev = torch.full((size,), 0x55, dtype=torch.uint8, device="cuda:0")
        state["dev"] = dev
        stream = torch.cuda.current_stream().cuda_stream
        acc = {"t": 0.0}

        def flush(opaque, pic, off, stride, w, h, field):
            d = dev.data_ptr()
            dp = (C.c_void_p * 3)(*[d + off[i] for i in range(3)])
            rp = (C.c_void_p * 3)(d, d, d)
            st = (C.c_int * 3)(stride[0], stride[1], stride[2])
            torch.cuda.synchronize()
            t = time.perf_counter()
            r = L.ffhip_h264_picture_flush(pic, dp, st, rp, stream)
            torch.cuda.synchronize()
            acc["t"] += time.perf_counter() - t
            return r

        def flush_mbaff(opaque, chains, off, stride, w, h):
            d = dev.data_ptr()
            dp = (C.c_void_p * 3)(*[d + off[i] for i in range(3)])
            st = (C.c_int * 3)(stride[0], stride[1], stride[2])
            ml = D.MbaffLists()
            L.ffhip_h264_mbaff_lists(chains, C.byref(ml))
            calls.append(sum(ml.ncalls[k] for k in range(3)))
            intra.append(ml.nrecs)
            torch.cuda.synchronize()
            t = time.perf_counter()
            r = L.ffhip_h264_mbaff_flush(chains, dp, st, stream)
            torch.cuda.synchronize()
            t_chain.append(time.perf_counter() - t)
            t_inter.append(acc["t"])
            acc["t"] = 0.0
            return r
        flush.mbaff = flush_mbaff
        return flush, {}

    def read_back(base, used):
        host = state["dev"][:used].cpu().numpy()
        C.memmove(base, host.ctypes.data, used)
    got, st, _ = D.decode(aus, make_flush=make, read_back=read_back, arena_bytes=256 << 20)
    same = len(got) == len(plain) and all(np.array_equal(a[pl], b[pl]) for a, b in zip(plain, got) for pl in range(3))
    print(json.dumps({"tool": "bench_h264_mbaff", "picture": "1920x1088 MBAFF, 8 bits 4:2:0", "frames": n, "identical_to_plain_decode": bool(same),
                      "errors": st["errors"], "field_macroblocks": st["mbs_field"], "macroblocks": st["mbs_hl"],
                      "intra_macroblocks_per_frame": intra, "filter_calls_per_frame": calls,
                      "inter_objects_ms_per_frame": [round(1e3 * t, 3) for t in t_inter],
                      "chains_ms_per_frame": [round(1e3 * t, 3) for t in t_chain]}))


if __name__ == "__main__":
    main()
