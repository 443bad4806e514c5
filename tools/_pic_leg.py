import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import torch
import bench
dev = torch.device("cuda", 0)
ev = lambda: torch.cuda.Event(enable_timing=True)
d = bench.h264_picture_leg(torch, dev, ev)
for k, v in d.items():
    print(k, {a: b for a, b in v.items() if a != "note"})
