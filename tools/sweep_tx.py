#!/usr/bin/env python3
"""tools/sweep_tx.py — A/B timing of the MDCT kernel variants (65,536 x N=1024, BASELINE configs[3])."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from ffmpeg_amd import _lib as _fflib  # noqa: E402
_fflib.select("measure")  # the FFHIP_* knobs this tool sets exist only in libffhip_measure.so
from ffmpeg_amd import tx  # noqa: E402

nt, ln = 65536, 1024
for inv in (0, 1):
    tin = torch.rand((nt, ln if inv else 2 * ln), dtype=torch.float32, device="cuda:0")
    tout = torch.empty((nt, ln), dtype=torch.float32, device="cuda:0")
    ref = None
    for env in ({}, {"FFHIP_TX_WPB": "8"}, {"FFHIP_TX_WPB": "4"}, {"FFHIP_TX_AHEAD": "1"}, {"FFHIP_TX_Z": "0"},
                {"FFHIP_TX_Z": "0", "FFHIP_TX_PERSISTENT": "0"},
                {"FFHIP_TX_Z": "0", "FFHIP_TX_PERSISTENT": "0", "FFHIP_TX_LDSTAB": "0"}):
        for k in ("FFHIP_TX_PERSISTENT", "FFHIP_TX_LDSTAB", "FFHIP_TX_AHEAD", "FFHIP_TX_Z", "FFHIP_TX_WPB"):
            os.environ.pop(k, None)
        os.environ.update(env)
        ctx = tx.TxContext(tx.FLOAT_MDCT, inv, ln, 1.0 if not inv else 1.0 / ln)  # FFHIP_TX_AHEAD is read at init
        for _ in range(2):
            ctx.batch(tout, tin)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ctx.batch(tout, tin)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        chk = float(tout.double().sum())
        ref = chk if ref is None else ref
        byt = nt * (8192 if inv else 12288)
        print(json.dumps({"inv": inv, "env": env, "ms": round(ms, 4), "Mtx/s": round(nt / ms / 1e3, 1), "GB/s": round(byt / ms / 1e6, 1),
                          "hbm_frac": round(byt / ms / 1e6 / 8000, 4), "same_output": chk == ref}), flush=True)
        ctx.close()
