"""Times the H.264 picture layer on 4:4:4 pictures (hl_decode_mb_444 recorded by the reference's own macroblock loop, see
tests/test_gpu_h264_decoder.py): a 1080p I-picture (three luma-only wavefronts side by side) and a P/B picture, 8 and 10 bits, beside the
4:2:0 picture of the same size.  Usage: python tools/bench_h264_444.py [reps [depth [4k]]]"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ffi  # noqa: E402
import h264_intra_gen as G  # noqa: E402
import h264_inter_gen as I  # noqa: E402


def run(depth, cfmt, p_intra, reps, mb_w=120, mb_h=68, nref=2):
    import torch
    from ffmpeg_amd import h264, _lib
    _lib.lib()
    RH = C.CDLL(I.REF_HIP_SO)
    rng = np.random.default_rng(7)
    px, dt, top = (2, np.uint16, 1 << depth) if depth > 8 else (1, np.uint8, 256)
    W, H = mb_w * 16, mb_h * 16
    sy = W
    sc, HC = (sy, H) if cfmt == 3 else (W // 2, H if cfmt == 2 else H // 2)
    strides = [sy * px, sc * px, sc * px]
    dev = lambda a: torch.from_numpy(a.view(np.uint8).reshape(a.shape[0], -1).copy()).cuda()
    d_refs = [dev(rng.integers(0, top, (nref * r, s), dtype=dt)) for r, s in ((H, sy), (HC, sc), (HC, sc))]
    d_dst = [dev(rng.integers(0, top, (r, s), dtype=dt)) for r, s in ((H, sy), (HC, sc), (HC, sc))]
    gpu = I.Dec(RH, "ffrefhip_", depth, mb_w, mb_h, strides[0], strides[1], 1, cfmt=cfmt)
    rows = [H, HC, HC]
    for lst in (0, 1):
        for i in range(nref):
            gpu.set_ref(lst, i, [d_refs[pl].data_ptr() + i * rows[pl] * strides[pl] for pl in range(3)])
    gpu.set_pwt(I.make_pwt(rng, 0, depth, nref))
    gpu.set_cur([t.data_ptr() for t in d_dst])
    pic = h264.Picture(mb_w, mb_h, bit_depth=depth, chroma_format=cfmt)
    pic.begin()
    RH.ffrefhip_h264dec_record_begin.argtypes = [C.c_void_p] * 5
    RH.ffrefhip_h264dec_record_begin.restype = None
    RH.ffrefhip_h264dec_record_begin(gpu.d, pic._p, *[t.data_ptr() for t in d_refs])
    t0 = time.time()
    for my in range(mb_h):
        for mx in range(mb_w):
            if rng.random() < p_intra:
                gpu.decode_intra(G.make_intra_mb(rng, mx, my, mb_w, mb_h, depth=depth, cfmt=cfmt))
            else:
                gpu.decode_inter(I.make_inter_mb(rng, gpu.bits, mx, my, nref, 64, depth=depth, cfmt=cfmt))
    rec_s = time.time() - t0
    for _ in range(3):
        pic.flush(d_dst, strides, d_refs)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        pic.flush(d_dst, strides, d_refs)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print("%2d bits %s %s %dx%d: flush %.3f ms = %.0f pictures/s (recording with the test generator: %.1f s)" % (
        depth, {1: "4:2:0", 2: "4:2:2", 3: "4:4:4"}[cfmt], "I-picture" if p_intra >= 1 else "P/B-picture (%.0f %% intra)" % (100 * p_intra), 16 * mb_w, 16 * mb_h, ms, 1e3 / ms,
        rec_s), flush=True)
    pic.close()
    gpu.close()


if __name__ == "__main__":
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    if not (ffi.have_ref() and I.have_ref_hip()):
        sys.exit("oracle/_ref not built")
    for depth in ((8, 10) if len(sys.argv) < 3 else (int(sys.argv[2]),)):
        for cfmt in ((1, 2, 3) if len(sys.argv) < 5 else (int(sys.argv[4]),)):
            for p_intra in (1.0, .05):
                if len(sys.argv) > 3 and sys.argv[3] == "4k":
                    run(depth, cfmt, p_intra, reps, mb_w=240, mb_h=135)
                else:
                    run(depth, cfmt, p_intra, reps)
