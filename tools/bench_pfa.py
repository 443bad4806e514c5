#!/usr/bin/env python3
"""tools/bench_pfa.py — the prime-factor MDCT lengths (15xM: CELT / AAC-960; 3xM: 96- / 768-sample AAC frames; 5xM: Siren;
7xM / 9xM) on one GPU, HIP events; 65,536 transforms each."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from ffmpeg_amd import tx  # noqa: E402

nt = 65536
LENS = (120, 240, 480, 960, 1920, 192, 1536, 640, 896, 1152, 4608)
for ln in LENS:
    for inv in (0, 1):
        tin = torch.rand((nt, ln if inv else 2 * ln), dtype=torch.float32, device="cuda:0")
        tout = torch.empty((nt, ln), dtype=torch.float32, device="cuda:0")
        ctx = tx.TxContext(tx.FLOAT_MDCT, inv, ln, 1.0 / ln if inv else 1.0)
        for _ in range(2):
            ctx.batch(tout, tin)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ctx.batch(tout, tin)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        byt = nt * ln * 4 * (2 if inv else 3)
        print(json.dumps({"len": ln, "inv": inv, "ms": round(ms, 4), "Mtx/s": round(nt / ms / 1e3, 1), "GB/s": round(byt / ms / 1e6, 1),
                          "hbm_frac": round(byt / ms / 1e6 / 8000, 4)}), flush=True)
        ctx.close()
# AV_TX_FLOAT_RDFT: r2c / c2r, 65,536 transforms
for ln in (1024, 4096):
    for inv in (0, 1):
        n_in, n_out = (ln + 2, ln) if inv else (ln, ln + 2)
        tin = torch.rand((nt, n_in), dtype=torch.float32, device="cuda:0")
        tout = torch.empty((nt, n_out), dtype=torch.float32, device="cuda:0")
        ctx = tx.TxContext(tx.FLOAT_RDFT, inv, ln, 1.0)
        for _ in range(2):
            ctx.batch(tout, tin)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ctx.batch(tout, tin)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        byt = nt * (n_in + n_out) * 4
        print(json.dumps({"rdft": ln, "inv": inv, "ms": round(ms, 4), "Mtx/s": round(nt / ms / 1e3, 1), "GB/s": round(byt / ms / 1e6, 1),
                          "hbm_frac": round(byt / ms / 1e6 / 8000, 4)}), flush=True)
        ctx.close()
# AV_TX_FLOAT_DCT: DCT-II / DCT-III of n reals, 65,536 transforms
for n in (1024,):
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
        print(json.dumps({"dct": n, "inv": inv, "ms": round(ms, 4), "Mtx/s": round(nt / ms / 1e3, 1), "GB/s": round(byt / ms / 1e6, 1),
                          "hbm_frac": round(byt / ms / 1e6 / 8000, 4)}), flush=True)
        ctx.close()
