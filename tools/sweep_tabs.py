#!/usr/bin/env python3
"""tools/sweep_tabs.py — tables in LDS vs in L2 for the larger av_tx transforms (FFHIP_TX_TABLDS), 65,536 / 16,384 transforms."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from ffmpeg_amd import _lib as _fflib  # noqa: E402
_fflib.select("measure")  # the FFHIP_* knobs this tool sets exist only in libffhip_measure.so
from ffmpeg_amd import tx  # noqa: E402

cases = [(tx.FLOAT_MDCT, 1024, 0), (tx.FLOAT_MDCT, 2048, 0), (tx.FLOAT_MDCT, 2048, 1), (tx.FLOAT_MDCT, 4096, 0), (tx.FLOAT_MDCT, 4096, 1),
         (tx.FLOAT_FFT, 1024, 0), (tx.FLOAT_FFT, 2048, 0), (tx.FLOAT_RDFT, 2048, 0), (tx.FLOAT_RDFT, 4096, 0), (tx.FLOAT_RDFT, 4096, 1)]
for typ, ln, inv in cases:
    nt = 65536 if ln <= 1024 else 16384
    if typ == tx.FLOAT_RDFT:
        n_in, n_out = (ln + 2, ln) if inv else (ln, ln + 2)
    elif typ == tx.FLOAT_FFT:
        n_in = n_out = 2 * ln
    else:
        n_in, n_out = (ln, ln) if inv else (2 * ln, ln)
    tin = torch.rand((nt, n_in), dtype=torch.float32, device="cuda:0")
    tout = torch.empty((nt, n_out), dtype=torch.float32, device="cuda:0")
    ref = None
    for mode in ("1", "0"):
        os.environ["FFHIP_TX_TABLDS"] = mode
        ctx = tx.TxContext(typ, inv, ln, 1.0)
        try:
            for _ in range(2):
                ctx.batch(tout, tin)
        except RuntimeError as e:
            print(json.dumps({"type": typ, "len": ln, "inv": inv, "tables_in_lds": mode, "error": str(e)[:80]}), flush=True)
            ctx.close()
            continue
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ctx.batch(tout, tin)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        chk = float(tout.double().sum())
        ref = chk if ref is None else ref
        byt = nt * (n_in + n_out) * 4
        print(json.dumps({"type": typ, "len": ln, "inv": inv, "tables_in_lds": mode, "ms": round(ms, 4), "Mtx/s": round(nt / ms / 1e3, 1),
                          "hbm_frac": round(byt / ms / 1e6 / 8000, 4), "same_output": chk == ref}), flush=True)
        ctx.close()
os.environ.pop("FFHIP_TX_TABLDS", None)
