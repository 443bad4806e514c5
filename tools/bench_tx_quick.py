#!/usr/bin/env python3
"""tools/bench_tx_quick.py — HIP-event times of the av_tx batch legs bench.py quotes (65,536 transforms each; fft16384: 4096)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from ffmpeg_amd import tx  # noqa: E402


def run(name, typ, ln, inv, nt, n_in, n_out, init_len=None, byt=None):
    tin = torch.rand((nt, n_in), dtype=torch.float32, device="cuda:0")
    tout = torch.empty((nt, n_out), dtype=torch.float32, device="cuda:0")
    ctx = tx.TxContext(typ, inv, init_len or ln, 1.0)
    for _ in range(2):
        ctx.batch(tout, tin)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ctx.batch(tout, tin)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    b = byt or 4 * (n_in + n_out) * nt
    print(json.dumps({"case": name, "ms": round(ms, 4), "hbm_frac": round(b / ms / 8e9, 4)}), flush=True)


run("fft1024_fwd", tx.FLOAT_FFT, 1024, 0, 65536, 2048, 2048)
run("fft512_fwd", tx.FLOAT_FFT, 512, 0, 65536, 1024, 1024)
run("mdct1024_fwd", tx.FLOAT_MDCT, 1024, 0, 65536, 2048, 1024)
run("mdct1024_inv", tx.FLOAT_MDCT, 1024, 1, 65536, 1024, 1024)
run("dct2_1024", tx.FLOAT_DCT, 1024, 0, 65536, 1024, 1024)
run("dct3_1024", tx.FLOAT_DCT, 1024, 1, 65536, 1024, 1024, init_len=512)
run("rdft1024_r2c", tx.FLOAT_RDFT, 1024, 0, 65536, 1024, 1026)
run("fft16384", tx.FLOAT_FFT, 16384, 0, 4096, 32768, 32768)
