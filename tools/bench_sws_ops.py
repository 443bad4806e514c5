"""tools/bench_sws_ops.py — the SwsOpBackend leg of bench.py on its own (device-resident 4K pictures through committed micro-op lists)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.chdir(ROOT)
import torch  # noqa: E402
from ffmpeg_amd import _lib as _fflib  # noqa: E402
_fflib.select("measure")  # the FFHIP_* knobs this tool sets exist only in libffhip_measure.so
import bench  # noqa: E402

r = bench.sws_ops_leg(torch, "cuda:0")
print(os.environ.get("FFHIP_UOPS_ROWS", "auto"), json.dumps({k: (v["Mpixels/s"], v["GB/s"]) for k, v in r.items()}))
