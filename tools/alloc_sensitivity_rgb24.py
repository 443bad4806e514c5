"""tools/alloc_sensitivity_rgb24.py — yuv420p -> rgb24 4K x 64 on six different allocations inside ONE process: the product kernel against the
XCD-contiguous workgroup numbering (FFHIP_YUV2RGB_VARIANT=xcd, measure build).  The numbering wins 5 % on some buffers and loses 5 % on others
(profiles/r05_rgb24_variants.txt); the product's row-pair order does not move.  Why the variant is not the product."""
import json, os, sys
sys.path.insert(0, os.getcwd())
import torch
from ffmpeg_amd import _lib
_lib.select("measure")
from ffmpeg_amd import swscale as S
dev = torch.device("cuda:0")
n, w, h = 64, 3840, 2160
ctx = S.SwsContext(w, h, 0, w, h, 2, 4)
keep = []
for trial in range(6):
    pad = torch.empty(((trial * 37 + 5) << 20,), dtype=torch.uint8, device=dev); keep.append(pad)
    src = [torch.randint(0, 256, (n, r, c), dtype=torch.uint8, device=dev) for r, c in S.plane_shapes(0, w, h)]
    dst = [torch.empty((n, h, 3 * w), dtype=torch.uint8, device=dev)]
    keep += src + dst
    res = {}
    for var in ("", "xcd", "", "xcd"):
        if var: os.environ["FFHIP_YUV2RGB_VARIANT"] = var
        else: os.environ.pop("FFHIP_YUV2RGB_VARIANT", None)
        for _ in range(10): ctx.scale_batch(src, dst)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(60): ctx.scale_batch(src, dst)
        b.record(); torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 60
        res.setdefault("x" if var else "product", []).append(round(n * w * h * 4.5 / (ms * 1e-3) / 8e12, 4))
    print(trial, hex(dst[0].data_ptr()), hex(src[0].data_ptr()), res)
