"""SwsOpBackend `hip`, CPU side: the oracle's restatement of the micro-op semantics pinned to backend_c — per
instance on checkasm's shapes (tests/checkasm/sw_ops.c) and as the backend of the reference's own graph — and the generator of
libffhip checked without a GPU (every program it writes for those lists compiles with hiprtc)."""
import ctypes as C
import os

import numpy as np
import pytest

import ffi
import swsops as S

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.skipif(not ffi.have_ref(), reason="oracle/_ref/libffref.so not built")


def run_c(R, case, src):
    dst = np.zeros_like(src)
    e = case.execute(src, dst)
    assert R.ffref_sws_uops_run_c(case.uops, len(case.uops), C.byref(e), 0, 0, S.PIXELS, S.LINES) == 0, case.name
    return dst


def run_backend(g, case, src, x0=0, x1=S.PIXELS, off=0):
    h = C.c_void_p()
    assert g("compile")(case.uops, len(case.uops), C.byref(h)) == 0, case.name
    bs = g("block_size")(h)
    dst = np.zeros_like(src)
    e = case.execute(src, dst, pixels=x1 - x0, x0=x0, block=bs, off=off)
    g("func")(C.byref(e), h, x0 // bs, 0, x1 // bs, S.LINES)
    g("free")(C.byref(h))
    return dst


def test_every_micro_op_instance_oracle_is_backend_c():
    R = S.declare_ref(ffi.ref())
    g = S.declare(ffi.oracle(), "ffo_sws_uops_")
    rng = np.random.default_rng(2025)
    inst = S.instances(R)
    assert len(inst) > 300
    n = 0
    for name, u in inst:
        for case in S.case_of(rng, name, u, R):
            src = case.planes(rng)
            err = case.compare(run_c(R, case, src), run_backend(g, case, src))
            assert err is None, (case.name, err)
            n += 1
    assert n > 400


PAIRS = [("yuv444p", "rgb24"), ("rgb24", "yuv444p"), ("rgb24", "bgra"), ("gbrp", "rgb24"), ("gray", "rgb24"), ("rgb24", "gray"),
         ("yuv444p", "gbrp"), ("rgb565le", "rgb24"), ("rgb24", "rgb565le"), ("yuv444p10le", "rgb48le"), ("rgba", "yuva444p"),
         ("monow", "gray"), ("gray", "monob"), ("pal8", "rgb24"), ("rgb4", "rgb24"), ("rgb24", "rgb4_byte"), ("yuv444p16be", "yuv444p"),
         ("gbrpf32le", "rgb24"), ("rgb24", "gbrpf32le"), ("ya8", "rgba"), ("x2rgb10le", "rgb24"), ("rgb24", "x2bgr10le"),
         ("yuv444p", "yuv444p10le"), ("gray16le", "gray"), ("bgr8", "rgb24")]
# sizes every backend block size agrees on: no scaling, or an output width that is whole 32-pixel blocks; RAGGED are the others
# (see test_reference_tail_path_defect)
SIZES = [(64, 16, 64, 16), (70, 9, 70, 9), (64, 16, 128, 32), (70, 18, 64, 9), (33, 7, 96, 21), (70, 18, 64, 18), (64, 18, 64, 9)]
RAGGED = [(70, 18, 35, 9), (33, 7, 100, 21), (70, 18, 35, 18), (35, 18, 35, 9)]


def convert_pair(R, sf, df, size, backends, rng_seed, slack=0, **kw):
    sw, sh, dw, dh = size
    rng = np.random.default_rng(rng_seed)
    src = S.Picture(R, sf, sw, sh, slack=slack)
    for rows in src.payload(R):                 # the same picture whatever the line padding
        rows[:] = rng.integers(0, 256, rows.shape, dtype=np.uint8)
    if sf == "pal8":
        src.planes[1][:] = rng.integers(0, 256, 1024, dtype=np.uint8)
    dst = S.Picture(R, df, dw, dh, slack=slack)
    r = S.convert(R, backends, src, dst, **kw)
    out = dst.payload(R)
    bits = {"monow": 1, "monob": 1, "rgb4": 4, "bgr4": 4}.get(df, 8)
    if dw * bits % 8:                            # the bits of the last byte of a line right of the picture are nobody's
        out[0][:, -1] &= (0xFF << (8 - dw * bits % 8)) & 0xFF
    return r, out


def graph_parity(R, L, prefix, sf, df, sizes):
    """sws_scale_frame() with backend_hip bound to L == with backend_c; returns the number of lists the bound backend compiled"""
    took = 0
    for size in sizes:
        for kw in ({}, {"scaler": 1}):          # default scaler (bicubic), bilinear
            if kw and size[0] == size[2] and size[1] == size[3]:
                continue
            S.unbind(R)
            rc, want = convert_pair(R, sf, df, size, S.BACKEND_C | S.BACKEND_MEMCPY, 7, **kw)
            if rc < 0:
                continue                        # the op layer does not take the pair at this size: nothing to pin
            rc, again = convert_pair(R, sf, df, size, S.BACKEND_C | S.BACKEND_MEMCPY, 7, **kw)
            if not all(np.array_equal(a, b) for a, b in zip(want, again)):
                continue                        # backend_c disagrees with itself (its tail path read uninitialised memory: seen for
                                                # gray -> monob, 70 -> 35 bilinear): there is no reference answer to compare with
            S.bind(R, L, prefix)
            R.ffref_sws_hip_count(-1)
            try:
                rh, got = convert_pair(R, sf, df, size, S.BACKEND_HIP | S.BACKEND_MEMCPY, 7, **kw)
            finally:
                S.unbind(R)
            assert rh >= 0, (sf, df, size, rh)
            took += R.ffref_sws_hip_count(0)
            for i, (a, b) in enumerate(zip(want, got)):
                assert np.array_equal(a, b), (sf, df, size, kw, "plane %d: %d bytes differ" % (i, (a != b).sum()))
    return took


@pytest.mark.parametrize("sf,df", PAIRS)
def test_graph_with_oracle_as_backend(sf, df):
    """through the reference's own op-list generation, optimizer, filter split and dispatch: byte-exact.  Once with the oracle's
    natural block (1 pixel; 8 / 2 for 1- and 4-bit formats), once reporting backend_c's 32-pixel blocks, which drives the dispatcher
    exactly as backend_c drives it and makes every size comparable."""
    R = S.declare_ref(ffi.ref())
    O = ffi.oracle()
    S.declare(O, "ffo_sws_uops_")
    took = graph_parity(R, O, "ffo_sws_uops_", sf, df, SIZES)
    O.ffo_sws_uops_force_block(32)
    try:
        # gray -> monob at 70 -> 35 is left out: backend_c does not reproduce its own answer there from run to run (its tail path
        # reads memory nobody wrote), so there is nothing to be equal to
        took += graph_parity(R, O, "ffo_sws_uops_", sf, df, SIZES + (RAGGED if df != "monob" else []))
    finally:
        O.ffo_sws_uops_force_block(0)
    assert took > 0, "backend_hip compiled no list for %s -> %s" % (sf, df)


def test_reference_tail_path_defect():
    """Why RAGGED is compared with 32-pixel blocks only.  A scaled conversion runs as several passes (the horizontal filter into a
    float plane, the vertical filter out of it, ops_dispatch.c:744-766; a packed source is split into planes first).  A backend with 32-pixel blocks sends
    the partial last block of every line through the dispatcher's padded-copy tail (ops_dispatch.c:437-500); for a vertically
    filtered read that copy takes get_lines_in() lines (ops_dispatch.c:166-176: up to the FIRST tap row of the slice's last line, not
    its last tap row), so the lower taps of the last lines read the zero-filled buffer.  The answer of the reference therefore depends
    on the block size of the backend: the same arithmetic (the oracle) with 1-pixel and with 32-pixel blocks differs exactly in the
    partial block, the 32-pixel answer is backend_c's, and the 1-pixel answer — no tail, every tap read from the picture — is the
    one this library produces.  Reference and oracle only; no product code involved."""
    R = S.declare_ref(ffi.ref())
    O = ffi.oracle()
    S.declare(O, "ffo_sws_uops_")
    size = RAGGED[0]
    S.unbind(R)
    rc, c32 = convert_pair(R, "yuv444p", "rgb24", size, S.BACKEND_C | S.BACKEND_MEMCPY, 7)
    out = {}
    for blk in (1, 32):
        O.ffo_sws_uops_force_block(blk)
        S.bind(R, O, "ffo_sws_uops_")
        try:
            r, out[blk] = convert_pair(R, "yuv444p", "rgb24", size, S.BACKEND_HIP | S.BACKEND_MEMCPY, 7)
        finally:
            S.unbind(R)
            O.ffo_sws_uops_force_block(0)
        assert r >= 0 and rc >= 0
    assert np.array_equal(out[32][0], c32[0])
    bad = np.argwhere(out[1][0] != c32[0])
    if bad.size:                                 # a fixed reference makes this vacuous, not red
        assert bad[:, 1].min() >= 32 * 3, "block sizes disagree left of the partial block"


def test_generated_programs_compile():
    """libffhip's generator: the program of every instance (and of whole conversion lists, through the graph) is valid HIP — hiprtc
    compiles it for gfx950 here, without a device"""
    from ffmpeg_amd import _lib
    L = _lib.lib()
    R = S.declare_ref(ffi.ref())
    L.ffhip_sws_uops_check.argtypes = [C.POINTER(S.UOp), C.c_int]
    rng = np.random.default_rng(5)
    seen, n = set(), 0
    for name, u in S.instances(R):
        key = (u.type, u.uop)
        if key in seen and u.uop not in (S.LINEAR, S.READ_PACKED, S.WRITE_PACKED, S.PERMUTE):
            continue                        # one instance per (type, micro-op) is enough for syntax; the GPU test runs all of them
        seen.add(key)
        for case in S.case_of(rng, name, u, None)[:1]:
            r = L.ffhip_sws_uops_check(case.uops, len(case.uops))
            assert r == 0, (case.name, r, L.ffhip_last_error())
            n += 1
    assert n > 100
    lut = [u for name, u in S.instances(R) if u.uop == S.LUT_3D]
    if lut:
        lst = (S.UOp * 3)(S._mk(S.U32, S.READ_PLANAR), lut[0], S._mk(S.U32, S.WRITE_PLANAR))
        assert L.ffhip_sws_uops_check(lst, 3) == -95


def test_code_objects_are_cached_on_disk(tmp_path):
    """The on-disk cache of compiled op lists (include/ffhip.h, ffhip_sws_uops_cache_stats): a second PROCESS finds the code objects
    the first one compiled (ffhip_sws_uops_set_cache_dir), a damaged file is a miss that is rewritten, no directory = every list compiled."""
    import subprocess
    import sys
    prog = r"""
import ctypes as C, sys, os
sys.path.insert(0, %r); sys.path.insert(0, %r)
import numpy as np
import ffi, swsops as S
from ffmpeg_amd import _lib
L = _lib.lib()
R = S.declare_ref(ffi.ref())
L.ffhip_sws_uops_check.argtypes = [C.POINTER(S.UOp), C.c_int]
if sys.argv[1]:
    assert L.ffhip_sws_uops_set_cache_dir(sys.argv[1].encode()) == 0
rng = np.random.default_rng(5)
n = 0
for name, u in S.instances(R):
    if u.uop in (S.LINEAR, S.PERMUTE) or n >= 6:
        continue
    case = S.case_of(rng, name, u, None)[0]
    assert L.ffhip_sws_uops_check(case.uops, len(case.uops)) == 0
    n += 1
a, b = C.c_long(), C.c_long()
L.ffhip_sws_uops_cache_stats(C.byref(a), C.byref(b))
print("STATS", n, a.value, b.value)
""" % (ROOT, os.path.join(ROOT, "tests"))

    def run(cache):
        r = subprocess.run([sys.executable, "-c", prog, cache], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout + r.stderr
        return [int(x) for x in [ln for ln in r.stdout.splitlines() if ln.startswith("STATS")][0].split()[1:]]
    d = str(tmp_path / "cache" / "ffhip")
    n, comp, hits = run(d)
    assert n == 6 and comp == 6 and hits == 0
    files = sorted(os.listdir(d))
    assert len(files) == 6 and all(f.endswith(".hsaco") for f in files)
    assert run(d) == [6, 0, 6]                                   # another process: nothing compiled
    with open(os.path.join(d, files[0]), "r+b") as f:            # a truncated file is a miss ...
        f.truncate(100)
    with open(os.path.join(d, files[1]), "r+b") as f:            # ... and so is one whose key was tampered with
        f.seek(20)
        f.write(b"X")
    assert run(d) == [6, 2, 4]
    assert run(d) == [6, 0, 6]                                   # ... rewritten whole
    assert run("") == [6, 6, 0]                                  # off
    assert sorted(os.listdir(d)) == files
    # trust (round 5): a flipped bit in the CODE (the key intact) fails the header's hash -> a miss, recompiled and rewritten
    with open(os.path.join(d, files[2]), "r+b") as f:
        f.seek(-7, os.SEEK_END)
        b = f.read(1)
        f.seek(-7, os.SEEK_END)
        f.write(bytes([b[0] ^ 0x40]))
    assert run(d) == [6, 1, 5]
    assert run(d) == [6, 0, 6]
    # a directory others may write to is not used at all: nothing is loaded from it, nothing stored into it
    os.chmod(d, 0o777)
    assert run(d) == [6, 6, 0]
    os.chmod(d, 0o700)
    assert run(d) == [6, 0, 6]
    # a symlink where the directory should be is not followed; a symlink where a code object should be is not opened
    link = str(tmp_path / "cache" / "link")
    os.symlink(d, link)
    assert run(link) == [6, 6, 0]
    victim = tmp_path / "victim"
    victim.write_bytes(b"precious")
    os.remove(os.path.join(d, files[3]))
    os.symlink(str(victim), os.path.join(d, files[3]))
    assert run(d) == [6, 1, 5]                                   # compiled; the planted link was replaced by rename, not written through
    assert victim.read_bytes() == b"precious" and not os.path.islink(os.path.join(d, files[3]))
    assert not [f for f in os.listdir(d) if not f.endswith(".hsaco")]   # no temporary left behind
