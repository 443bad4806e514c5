import json, os, sys
sys.path.insert(0, "/root/repo")
import torch
from ffmpeg_amd import _lib as _fflib  # noqa: E402
_fflib.select("measure")  # the FFHIP_* knobs this tool sets exist only in libffhip_measure.so
from ffmpeg_amd import tx
nt = 65536
for n in (1024, 256, 4096):
    for inv in (0, 1):
        tin = torch.rand((nt, n), dtype=torch.float32, device="cuda:0")
        tout = torch.empty((nt, n), dtype=torch.float32, device="cuda:0")
        ctx = tx.TxContext(tx.FLOAT_DCT, inv, n >> inv, 1.0)
        for _ in range(2):
            ctx.batch(tout, tin)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ctx.batch(tout, tin)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        byt = nt * 2 * n * 4
        print(json.dumps({"dct": n, "inv": inv, "wpb": os.environ.get("FFHIP_DCT_WPB"), "ms": round(ms, 4), "Mtx/s": round(nt / ms / 1e3, 1), "hbm_frac": round(byt / ms / 1e6 / 8000, 4)}), flush=True)
        ctx.close()
