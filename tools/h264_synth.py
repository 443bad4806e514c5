"""tools/h264_synth.py — a synthetic 1080p-class H.264 P-picture recorded into an ffmpeg_amd.h264.Picture, for the picture-layer benches
(bench.py's extras leg, tools/bench_h264_inflight.py): every macroblock 16x16 uni-predicted at a random quarter-sample position
(luma qpel + 8x8 chroma MC), residuals on about half of the 8x8 luma and a third of the 4x4 chroma blocks, every edge filtered."""
import numpy as np

QPEL_DT = np.dtype([("dst_offset", np.int32), ("src_offset", np.int32), ("mcxy", np.uint8), ("size_idx", np.uint8), ("avg", np.uint8),
                    ("flags", np.uint8), ("src_x", np.int16), ("src_y", np.int16)])                                          # FFHipQpelBlock
CHROMA_DT = np.dtype([("dst_offset", np.int32), ("src_offset", np.int32), ("w_idx", np.uint8), ("h", np.uint8), ("x", np.uint8),
                      ("y", np.uint8), ("avg", np.uint8), ("flags", np.uint8), ("src_x", np.int16), ("src_y", np.int16), ("pad", np.int16)])  # FFHipChromaBlock
EDGE_DT = np.dtype([("offset", np.int32), ("kind", np.uint8), ("alpha", np.uint8), ("beta", np.uint8), ("pad", np.uint8),
                    ("tc0", np.int8, 4)])                                         # FFHipH264Edge


def record_p_picture(pic, h264, mb_w, mb_h, sy, sc, P, rng):
    """sy / sc: luma / chroma strides of planes with a P-sample (chroma P/2) margin for the reference reads"""
    pic.begin()
    q, c = np.zeros(1, QPEL_DT), np.zeros(1, CHROMA_DT)
    ed8, ed4 = np.zeros(8, EDGE_DT), np.zeros(4, EDGE_DT)
    for e in (ed8, ed4):
        e["alpha"], e["beta"] = 40, 9
        e["tc0"] = 1
    ed4["kind"] = 2
    blk8, blk4 = np.zeros(64, np.int16), np.zeros(16, np.int16)
    for my in range(mb_h):
        for mx in range(mb_w):
            x, y = mx * 16, my * 16
            dy, dx = (int(v) for v in rng.integers(-16, 17, 2))
            q[0] = (y * sy + x, (P + y + dy) * sy + P + x + dx, int(rng.integers(0, 16)), 0, 0, 0, 0, 0)
            pic.mc_luma(h264.MC_PUT, q)
            for pl in (1, 2):
                c[0] = ((y // 2) * sc + x // 2, (P // 2 + y // 2 + dy // 2) * sc + P // 2 + x // 2 + dx // 2, 0, 8, int(rng.integers(0, 8)),
                        int(rng.integers(0, 8)), 0, 0, 0, 0, 0)
                pic.mc_chroma(pl, h264.MC_PUT, c)
            for by in (0, 8):
                for bx in (0, 8):
                    if rng.random() < .5:
                        blk8[:] = 0
                        blk8[:6] = rng.integers(-80, 81, 6)
                        pic.idct_add(0, 1, (y + by) * sy + x + bx, blk8)
            for pl in (1, 2):
                for by in (0, 4):
                    for bx in (0, 4):
                        if rng.random() < .3:
                            blk4[:] = 0
                            blk4[:3] = rng.integers(-80, 81, 3)
                            pic.idct_add(pl, 0, (y // 2 + by) * sc + x // 2 + bx, blk4)
            pic.deblock_mb(0, mx, my, ed8)
            pic.deblock_mb(1, mx, my, ed4)
            pic.deblock_mb(2, mx, my, ed4)
