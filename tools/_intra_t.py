import ctypes as C, subprocess, sys, os, runpy
sys.argv = ["run_intra_batch.py", "1", "1"]
sys.path.insert(0, "/root/repo")
from ffmpeg_amd import _lib
L = _lib.lib()
runpy.run_path("tools/run_intra_batch.py", run_name="__main__")
out = (C.c_ulonglong * 32)()
L.ffhip_debug_intra_t(out, 0)
v = list(out)
names = ["loop-top/publish", "wait", "gather+sync", "prefetch-issue", "reconstruct(total minus inner)", "store+park", "chroma phase", "i16 luma", "i4 resid+info", "i4 ten steps", "i8 2nd pass", "i8 four blocks"]
launches = 2  # warm + 1
nmb = sum(v[20:24])
print("MBs by type (16x16,4x4,8x8,pcm):", v[20:24])
tot = sum(v[:12])
for i, n in enumerate(names):
    print("%-34s %12d cycles  %7.1f per MB  %5.1f %%" % (n, v[i], v[i] / max(nmb, 1), 100.0 * v[i] / tot))
print("total per MB", tot / nmb)
for t, (a, b) in enumerate([(7, 20), (9, 21), (11, 22)]):
    pass
print("i16 luma per i16 MB", v[7] / max(v[20], 1))
print("i4 resid per i4 MB", v[8] / max(v[21], 1), " ten steps per i4 MB", v[9] / max(v[21], 1))
print("i8 2nd per i8 MB", v[10] / max(v[22], 1), " four blocks per i8 MB", v[11] / max(v[22], 1))
