import os, sys
import ctypes as C
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
import ffi
from ffi import PIX
from ffmpeg_amd import swscale as S

def run(case, seed, fast):
    sf, sw, sh, df, dw, dh, flags = case
    if fast: os.environ.pop("FFHIP_SWS_FAST", None)
    else: os.environ["FFHIP_SWS_FAST"] = "0"
    ht = S.HostTables(sw, sh, PIX[sf], dw, dh, PIX[df], flags)
    banks = ht.banks()
    t = ffi.make_otables(sw, sh, PIX[sf], dw, dh, PIX[df], flags, banks, ht.coeffs())
    ctx = S.SwsContext(sw, sh, PIX[sf], dw, dh, PIX[df], flags)
    rng = np.random.default_rng(seed)
    src = ffi.alloc_frame(PIX[sf], sw, sh, rng)
    dsrc = []
    for a in src:
        pitch = (a.shape[1] + 63) // 64 * 64
        h = rng.integers(0, 256, (3, a.shape[0], pitch), dtype=np.uint8); h[0, :, :a.shape[1]] = a
        dsrc.append(torch.from_numpy(h).cuda())
    ddst = [torch.full((3, dh, (3 * dw + 63) // 64 * 64), 0xA5, dtype=torch.uint8, device="cuda:0")]
    ctx.scale_batch(dsrc, ddst); torch.cuda.synchronize()
    want = ffi.alloc_frame(PIX[df], dw, dh)
    sp, ss = ffi.planes(src); dp, ds = ffi.planes(want)
    assert ffi.oracle().ffo_sws_scale_frame(C.byref(t), sp, ss, dp, ds) == dh
    got = ddst[0][0].cpu().numpy()[:, :3 * dw]
    d = got != want[0]
    print(case, "fast" if fast else "tiled", "seed", seed, "mismatches", int(d.sum()))
    for y, x in np.argwhere(d)[:12]:
        print("   y", y, "byte", x, "px", x // 3, "ch", x % 3, "got", got[y, x], "want", want[0][y, x],
              "vLumPos", banks["vLum"][1][y], "vChrPos", banks["vChr"][1][y], "hLumPos", banks["hLum"][1][x // 3], "hChrPos", banks["hChr"][1][x // 6])
    ctx.close()

case = ("yuv420p", 96, 64, "rgb24", 136, 200, ffi.SWS_BICUBIC)
for seed in (1, 2, 3):
    run(case, seed, True)
    run(case, seed, False)
