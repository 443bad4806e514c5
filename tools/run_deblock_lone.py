import os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from ffmpeg_amd import h264
dev = torch.device("cuda", 0)
w, h = 3840, 2160
mbw, mbh = w // 16, h // 16
rng = np.random.default_rng(3)
ed = np.zeros(mbw * mbh * 8, dtype=np.dtype([("o", np.int32), ("k", np.uint8), ("a", np.uint8), ("b", np.uint8), ("p", np.uint8), ("tc", np.int8, 4)]))
ed["a"], ed["b"] = 40, 9
mb_intra = rng.random(mbw * mbh) < .25
k = np.zeros((mbw * mbh, 2, 4), np.uint8); k[mb_intra, :, 0] = 4
ed["k"] = k.ravel(); ed["tc"] = rng.integers(0, 4, (ed.size, 4))
ded = torch.from_numpy(ed.view(np.uint8).reshape(-1, 12)).to(dev)
batch = torch.randint(100, 140, (1, h, w), dtype=torch.uint8, device=dev)
for _ in range(4):
    h264.deblock_frames(batch, w * h, 1, w, mbw, mbh, ded)
torch.cuda.synchronize()
