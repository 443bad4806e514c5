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
for bw, flt in (("0", "0"), ("4", "0"), ("16", "0"), ("4", "4"), ("4", "8"), ("4", "12")):
    os.environ["FFHIP_DEBLOCK_FAULT"] = flt
    os.environ["FFHIP_DEBLOCK_BAND"] = bw
    for nf in (1, 8, 16, 32, 64):
        batch = torch.randint(100, 140, (nf, h, w), dtype=torch.uint8, device=dev)
        dd = ded.repeat(nf, 1)
        h264.deblock_frames(batch, w * h, nf, w, mbw, mbh, dd)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        h264.deblock_frames(batch, w * h, nf, w, mbw, mbh, dd)
        e1.record()
        torch.cuda.synchronize()
        print(json.dumps({"band": bw, "flags(2 nostore,4 nofilter,8 nowait)": flt, "frames": nf, "ms": round(e0.elapsed_time(e1), 3)}), flush=True)
