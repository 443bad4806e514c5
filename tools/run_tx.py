#!/usr/bin/env python3
"""tools/run_tx.py <len> <inv> [type] — 6 launches of one av_tx batch (65,536 transforms) for profiler passes"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from ffmpeg_amd import tx  # noqa: E402

ln, inv = int(sys.argv[1]), int(sys.argv[2])
typ = int(sys.argv[3]) if len(sys.argv) > 3 else tx.FLOAT_MDCT
nt = 65536
if typ == tx.FLOAT_RDFT:
    n_in, n_out = (ln + 2, ln) if inv else (ln, ln + 2)
elif typ == tx.FLOAT_FFT:
    n_in = n_out = 2 * ln
elif typ == tx.FLOAT_DCT:
    n_in = n_out = ln          # <len> = number of samples; the inverse is initialised with half of it, as av_tx_init is
else:
    n_in, n_out = (ln, ln) if inv else (2 * ln, ln)
tin = torch.rand((nt, n_in), dtype=torch.float32, device="cuda:0")
tout = torch.empty((nt, n_out), dtype=torch.float32, device="cuda:0")
ctx = tx.TxContext(typ, inv, ln >> inv if typ == tx.FLOAT_DCT else ln, 1.0)
for _ in range(6):
    ctx.batch(tout, tin)
torch.cuda.synchronize()
print("ok", ln, inv, typ)
