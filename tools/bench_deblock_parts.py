"""tools/db_exp.py — where a macroblock's time goes in the frame-order deblocking kernel (k_h264_deblock_band): timing
experiments through FFHIP_DEBLOCK_FAULT (2: no picture stores, 4: no filters, 8: rows do not wait for each other — wrong output,
measurement only) and FFHIP_DEBLOCK_BAND (rows per workgroup; 0 = the launcher's own choice), 4K luma planes."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from ffmpeg_amd import _lib as _fflib  # noqa: E402
_fflib.select("measure")  # the FFHIP_* knobs this tool sets exist only in libffhip_measure.so
from ffmpeg_amd import h264
dev = torch.device("cuda", 0)
w, h = 3840, 2160
mbw, mbh = w // 16, h // 16
rng = np.random.default_rng(3)
ed = np.zeros(mbw * mbh * 8, dtype=np.dtype([("o", np.int32), ("k", np.uint8), ("a", np.uint8), ("b", np.uint8), ("p", np.uint8), ("tc", np.int8, 4)]))
ed["a"], ed["b"] = 40, 9
ed["k"] = np.where(rng.random(ed.size) < .25, 4, 0)
ed["tc"] = rng.integers(0, 4, (ed.size, 4))
ded = torch.from_numpy(ed.view(np.uint8).reshape(-1, 12)).to(dev)
intra = float(os.environ.get("DB_INTRA", "0.25"))   # share of bS = 4 edges (0: the skewed-rows kernel never takes its intra branch)
mb_intra = rng.random(mbw * mbh) < intra      # bS = 4 on the macroblock edges (edge 0) of intra macroblocks, as in a stream
k = np.zeros((mbw * mbh, 2, 4), np.uint8)
k[mb_intra, :, 0] = 4
ed["k"] = k.ravel()
ded = torch.from_numpy(ed.view(np.uint8).reshape(-1, 12)).to(dev)
# old: "0" the skewed-rows kernel (default), "2" the band kernel
for old, bw, flt in () if os.environ.get("DB_SKIP_PARTS") == "1" else (("0", "0", "0"), ("0", "0", "4"), ("0", "0", "8"), ("0", "0", "12"), ("0", "0", "14"), ("2", "0", "0"), ("2", "4", "12")):
    os.environ["FFHIP_DEBLOCK_FAULT"] = flt
    os.environ["FFHIP_DEBLOCK_BAND"] = bw
    os.environ["FFHIP_DEBLOCK_OLD"] = old
    for nf in (1, 8, 32, 64):
        batch = torch.randint(100, 140, (nf, h, w), dtype=torch.uint8, device=dev)
        dd = ded.repeat(nf, 1)
        h264.deblock_frames(batch, w * h, nf, w, mbw, mbh, dd)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        h264.deblock_frames(batch, w * h, nf, w, mbw, mbh, dd)
        e1.record()
        torch.cuda.synchronize()
        print(json.dumps({"kernel": "skew" if old == "0" else "band", "band": bw, "flags(2 nostore,4 nofilter,8 nowait)": flt, "frames": nf, "ms": round(e0.elapsed_time(e1), 3)}), flush=True)
# waves per picture (FFHIP_DEBLOCK_WAVES; 0 = the launcher's choice)
os.environ["FFHIP_DEBLOCK_FAULT"] = "0"
os.environ["FFHIP_DEBLOCK_OLD"] = "0"
for waves, pad in (("0", "4"), ("0", "1"), ("0", "2"), ("0", "3"), ("20", "4"), ("36", "4")):
    os.environ["FFHIP_DEBLOCK_WAVES"] = waves
    os.environ["FFHIP_DEBLOCK_WPB"] = pad
    for nf in (1, 32, 64, 128):
        batch = torch.randint(100, 140, (nf, h, w), dtype=torch.uint8, device=dev)
        dd = ded.repeat(nf, 1)
        h264.deblock_frames(batch, w * h, nf, w, mbw, mbh, dd)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        h264.deblock_frames(batch, w * h, nf, w, mbw, mbh, dd)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        print(json.dumps({"kernel": "skew", "waves_per_picture": waves, "waves_per_block": pad, "frames": nf, "ms": round(ms, 3), "Gpixel/s": round(nf * w * h / ms / 1e6, 1)}), flush=True)
        del batch, dd
