import os, sys, json
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, bench
dev = torch.device("cuda", 0)
ev = lambda: torch.cuda.Event(enable_timing=True)
for i in range(3):
    print(json.dumps(bench.h264_picture_leg(torch, dev, ev)), flush=True)
