"""A/B of two builds of libffhip.so on the headline conversion (nv12 1080p -> 4K bicubic, 256 frames) and a few neighbours, alternating in
subprocesses of one run:  python tools/ab_headline.py tools/ab/libffhip_old.so ffmpeg_amd/libffhip.so"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 2 and sys.argv[1] != "--child":
    for p in range(3):
        for so in sys.argv[1:]:
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", os.path.abspath(so)], capture_output=True, text=True)
            print(out.stdout.strip() or out.stderr[-400:], flush=True)
    sys.exit(0)
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from ffmpeg_amd import _lib  # noqa: E402
_lib.SO = sys.argv[2]
from ffmpeg_amd import swscale as S  # noqa: E402

dev = torch.device("cuda:0")
res = {"so": os.path.basename(sys.argv[2])}
for name, sf, sw, sh, df, dw, dh, n in (("nv12 1080p->4K", 23, 1920, 1080, 23, 3840, 2160, 256), ("yuv420p 1080p->4K", 0, 1920, 1080, 0, 3840, 2160, 128),
                                        ("nv12 4K->1080p", 23, 3840, 2160, 23, 1920, 1080, 64)):
    src = [torch.randint(0, 256, (n, r, c), dtype=torch.uint8, device=dev) for r, c in S.plane_shapes(sf, sw, sh)]
    dst = [torch.zeros((n, r, c), dtype=torch.uint8, device=dev) for r, c in S.plane_shapes(df, dw, dh)]
    byt = n * (S.frame_bytes(sf, sw, sh) + S.frame_bytes(df, dw, dh))
    c = S.SwsContext(sw, sh, sf, dw, dh, df, 4)
    for _ in range(20):
        c.scale_batch(src, dst)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(60):
        c.scale_batch(src, dst)
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 60
    res[name] = [round(ms, 4), round(byt / ms / 1e6 / 8000, 4)]
    c.close()
    del src, dst
print(json.dumps(res))
