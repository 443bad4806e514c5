"""DCT-I / DST-I batches (kernels/tx_dcst1.hip): transforms/s at a few lengths.  python tools/bench_dcst1.py"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ffmpeg_amd import _lib, tx  # noqa: E402

_lib.select("measure")   # FFHIP_DCST1_VALU=1: the first form of the kernel

dev = torch.device("cuda:0")
for typ, name, valu in ((tx.FLOAT_DCT_I, "dctI", 0), (tx.FLOAT_DCT_I, "dctI", 1), (tx.FLOAT_DST_I, "dstI", 0)):
    os.environ.pop("FFHIP_DCST1_VALU", None)
    if valu:
        os.environ["FFHIP_DCST1_VALU"] = "1"
    for n in (16, 64, 66, 128, 256, 1024):
        nt = (1 << 26) // (n * n) * 16 if n > 64 else 1 << 20
        ctx = tx.TxContext(typ, 0, n, 1.0 / n)
        x = torch.randn((nt, n), dtype=torch.float32, device=dev)
        y = torch.zeros_like(x)
        for _ in range(2):
            ctx.batch(y, x)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5):
            ctx.batch(y, x)
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 5
        print(json.dumps({"transform": name, "kernel": "k_dcst1 (VALU)" if valu else "k_dcst1_m", "n": n, "transforms": nt, "ms": round(ms, 4), "Mtransforms/s": round(nt / ms / 1e3, 2),
                          "GB/s": round(nt * n * 8 / ms / 1e6, 1), "fp64_TFLOP/s": round(nt * 2.0 * n * n / ms / 1e9, 2)}), flush=True)
        ctx.close()
