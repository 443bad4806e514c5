#!/usr/bin/env python3
"""tools/sweep_sws.py — A/B timing of the measured variants of the swscale kernels on one GPU.

One process, HIP events on the launch stream, same workloads as bench.py (configs[1] and the 4K
yuv420p->rgb24 conversion).  Variants are selected through the FFHIP_* environment switches that
sws_api.hip / sws_yuv2rgb.hip read at launch time.  Output: a table on stdout and gpurun_out/sweep.json."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
from ffmpeg_amd import _lib as _fflib  # noqa: E402
_fflib.select("measure")  # the FFHIP_* knobs this tool sets exist only in libffhip_measure.so
from ffmpeg_amd import swscale as S  # noqa: E402

KEYS = ("FFHIP_UP2_XCD", "FFHIP_UP2_VAR", "FFHIP_SWS_UP2", "FFHIP_UP2_FSHIFT", "FFHIP_UP2_STRIP", "FFHIP_UP2_DEPTH", "FFHIP_UP2_HIPK", "FFHIP_SWS_MFMA", "FFHIP_MF_STRIP", "FFHIP_SWS_FAST", "FFHIP_CW_LUMA_GROUPS", "FFHIP_CW_DEPTH", "FFHIP_CW_PLAIN", "FFHIP_CW_STRIP", "FFHIP_CW_OPT",
        "FFHIP_YUV2RGB_VARIANT")


def setenv(env):
    for k in KEYS:
        os.environ.pop(k, None)
    os.environ.update(env)


def timed(fn, reps=30, warm=3):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    dev = torch.device("cuda", 0)
    out = {"scale_nv12_1080p_4k": [], "yuv420p_rgb24_4k": []}
    n = int(os.environ.get("SWEEP_FRAMES", "256"))
    ctx = S.SwsContext(1920, 1080, 23, 3840, 2160, 23, S.SWS_BICUBIC)
    src = [torch.randint(0, 256, (n, r, c), dtype=torch.uint8, device=dev) for r, c in S.plane_shapes(23, 1920, 1080)]
    dst = [torch.empty((n, r, c), dtype=torch.uint8, device=dev) for r, c in S.plane_shapes(23, 3840, 2160)]
    byt = n * 15552000
    variants = [{}]
    if ctx.up2_path:
        for spec in os.environ.get("SWEEP_VARS", "1,2,4,6").split(","):   # VAR[/FSHIFT[/STRIP[/XCD[/DEPTH]]]]
            f = spec.split("/")
            e = {"FFHIP_UP2_VAR": f[0]}
            for key, val in zip(("FFHIP_UP2_FSHIFT", "FFHIP_UP2_STRIP", "FFHIP_UP2_XCD", "FFHIP_UP2_DEPTH"), f[1:]):
                if val != "":
                    e[key] = val
            variants.append(e)
    if ctx.up2_path and not os.environ.get("SWEEP_ONLY_VARS"):
        variants += [{"FFHIP_UP2_XCD": "0"}, {"FFHIP_UP2_FSHIFT": "0"}, {"FFHIP_UP2_FSHIFT": "0", "FFHIP_UP2_VAR": "4"},
                     {"FFHIP_UP2_FSHIFT": "0", "FFHIP_UP2_VAR": "6"}, {"FFHIP_UP2_FSHIFT": "0", "FFHIP_UP2_XCD": "0"}]
        variants += [{"FFHIP_UP2_DEPTH": "3"}, {"FFHIP_UP2_STRIP": "120"}, {"FFHIP_UP2_STRIP": "120", "FFHIP_UP2_FSHIFT": "0"},
                     {"FFHIP_SWS_UP2": "0"}]
    if os.environ.get("SWEEP_FULL"):
        for o in ("1", "0"):
            for g in ("1", "2"):
                for d in ("3", "6"):
                    variants.append({"FFHIP_CW_OPT": o, "FFHIP_CW_LUMA_GROUPS": g, "FFHIP_CW_DEPTH": d})
    variants += [{}]
    if os.environ.get("SWEEP_OLD"):
        old = {"FFHIP_SWS_UP2": "0"}
        variants += [dict(old, FFHIP_SWS_MFMA="1"), dict(old, FFHIP_CW_STRIP="60"), dict(old, FFHIP_CW_STRIP="128"),
                     dict(old, FFHIP_CW_PLAIN="1"), dict(old, FFHIP_SWS_FAST="0")]
    print("fast path eligible:", ctx.fast_path, "exact-2x kernel:", ctx.up2_path)
    ref = None
    rounds = int(os.environ.get("SWEEP_ROUNDS", "4"))
    best = [1e9] * len(variants)
    same = [True] * len(variants)
    for rd in range(rounds):                      # interleaved: clock / thermal drift hits every variant alike
        for i, env in enumerate(variants):
            setenv(env)
            if rd == 0:
                for d in dst:
                    d.zero_()
            ms = timed(lambda: ctx.scale_batch(src, dst), reps=12, warm=2)
            best[i] = min(best[i], ms)
            if rd == 0:
                chk = [int(d.to(torch.int64).sum().item()) for d in dst]
                if ref is None:
                    ref = chk
                same[i] = chk == ref
    for i, env in enumerate(variants):
        ms = best[i]
        row = {"env": env, "ms": round(ms, 4), "GB/s": round(byt / ms / 1e6, 1), "hbm_frac": round(byt / ms / 1e6 / 8000, 4),
               "Mpix/s": round(n * 3840 * 2160 / ms / 1e3, 1), "same_output": same[i]}
        out["scale_nv12_1080p_4k"].append(row)
        print(json.dumps(row), flush=True)
    ctx.close()
    del src, dst

    n, w, h = 64, 3840, 2160
    ctx = S.SwsContext(w, h, 0, w, h, 2, 4)
    src = [torch.randint(0, 256, (n, r, c), dtype=torch.uint8, device=dev) for r, c in S.plane_shapes(0, w, h)]
    dst = [torch.empty((n, h, 3 * w), dtype=torch.uint8, device=dev)]
    byt = n * w * h * 4.5
    ref = None
    for env in ({}, {"FFHIP_YUV2RGB_VARIANT": "plain"}, {"FFHIP_YUV2RGB_VARIANT": "old"}):
        setenv(env)
        dst[0].zero_()
        ms = timed(lambda: ctx.scale_batch(src, dst))
        chk = int(dst[0].to(torch.int64).sum().item())
        if ref is None:
            ref = chk
        row = {"env": env, "ms": round(ms, 4), "GB/s": round(byt / ms / 1e6, 1), "hbm_frac": round(byt / ms / 1e6 / 8000, 4),
               "Mpix/s": round(n * w * h / ms / 1e3, 1), "same_output": chk == ref}
        out["yuv420p_rgb24_4k"].append(row)
        print(json.dumps(row), flush=True)
    # a plain device-to-device copy of the same byte count: the achievable-bandwidth yardstick
    a = torch.empty(int(byt) // 2, dtype=torch.uint8, device=dev)
    b = torch.empty_like(a)
    ms = timed(lambda: b.copy_(a))
    out["copy_yardstick"] = {"bytes_moved": int(byt) // 2 * 2, "ms": round(ms, 4), "GB/s": round(int(byt) // 2 * 2 / ms / 1e6, 1)}
    print(json.dumps(out["copy_yardstick"]), flush=True)
    setenv({})
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "sweep.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
