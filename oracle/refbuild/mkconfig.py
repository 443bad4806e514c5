#!/usr/bin/env python3
"""Write the handful of headers FFmpeg's C sources expect from a build tree.

TEST INFRASTRUCTURE ONLY (see oracle/README.md).  This is *our* recipe, not the
reference's `configure`: it scans the reference sources we compile for
HAVE_*/CONFIG_*/ARCH_* tokens and defines every one of them to 0 except the short
allow-list below (plain-C, little-endian, glibc/Linux, pthreads, no SIMD, no
inline asm).  The result is a pure-C, generic-arch build of exactly the hot-path
files named in oracle/refbuild/Makefile, compiled where they lie under
/root/reference.  Nothing from the reference is copied into this repository.

usage: mkconfig.py <reference_root> <out_dir> <src.c> [<src.c> ...]
"""
import os
import re
import sys

ONES = {
    # compiler / platform facts for gcc 11 on x86-64 Linux, expressed arch-neutrally
    "HAVE_FAST_64BIT", "HAVE_FAST_CLZ", "HAVE_FAST_UNALIGNED", "HAVE_ALIGNED_STACK",
    "HAVE_INT128", "HAVE_PRAGMA_DEPRECATED", "HAVE_SIMD_ALIGN_16", "HAVE_SIMD_ALIGN_32",
    "HAVE_SIMD_ALIGN_64", "HAVE_SECTION_DATA_REL_RO",
    # libc / libm
    "HAVE_MALLOC_H", "HAVE_UNISTD_H", "HAVE_SYS_PARAM_H", "HAVE_SYS_TIME_H", "HAVE_SYS_RESOURCE_H",
    "HAVE_ATANF", "HAVE_ATAN2F", "HAVE_CBRT", "HAVE_CBRTF", "HAVE_COPYSIGN", "HAVE_COSF", "HAVE_ERF",
    "HAVE_EXP2", "HAVE_EXP2F", "HAVE_EXPF", "HAVE_HYPOT", "HAVE_ISFINITE", "HAVE_ISINF", "HAVE_ISNAN",
    "HAVE_LDEXPF", "HAVE_LLRINT", "HAVE_LLRINTF", "HAVE_LOG2", "HAVE_LOG2F", "HAVE_LOG10F",
    "HAVE_LRINT", "HAVE_LRINTF", "HAVE_POWF", "HAVE_RINT", "HAVE_ROUND", "HAVE_ROUNDF", "HAVE_SINF",
    "HAVE_TRUNC", "HAVE_TRUNCF",
    "HAVE_ACCESS", "HAVE_CLOCK_GETTIME", "HAVE_FCNTL", "HAVE_GETENV", "HAVE_GETTIMEOFDAY",
    "HAVE_GMTIME_R", "HAVE_LOCALTIME_R", "HAVE_ISATTY", "HAVE_LSTAT", "HAVE_MEMALIGN",
    "HAVE_POSIX_MEMALIGN", "HAVE_MKSTEMP", "HAVE_MMAP", "HAVE_NANOSLEEP", "HAVE_STRERROR_R",
    "HAVE_SYSCONF", "HAVE_USLEEP", "HAVE_SCHED_GETAFFINITY", "HAVE_GETRUSAGE",
    "HAVE_STRUCT_STAT_ST_MTIM_TV_NSEC", "HAVE_STRUCT_RUSAGE_RU_MAXRSS",
    # threads: swscale's own slice threading is part of the CPU baseline we time
    "HAVE_THREADS", "HAVE_PTHREADS", "HAVE_PTHREAD_CANCEL", "HAVE_SEM_TIMEDWAIT",
    # library switches for the components on the hot path
    "CONFIG_SWSCALE", "CONFIG_AVUTIL", "CONFIG_AVCODEC", "CONFIG_SWSCALE_ALPHA", "CONFIG_STATIC",
    "CONFIG_PIC", "CONFIG_GPL", "CONFIG_SAFE_BITSTREAM_READER", "CONFIG_H264DSP", "CONFIG_H264QPEL",
    "CONFIG_H264CHROMA", "CONFIG_ME_CMP", "CONFIG_VIDEODSP", "CONFIG_UNSTABLE", "CONFIG_AAC_DECODER",
}

TOK = re.compile(r"\b((?:HAVE|CONFIG|ARCH)_[A-Z0-9_]+)\b")


def scan(path, seen, toks, roots):
    """Collect tokens from `path` and every quoted include that resolves under roots."""
    path = os.path.realpath(path)
    if path in seen or not os.path.isfile(path):
        return
    seen.add(path)
    try:
        text = open(path, errors="replace").read()
    except OSError:
        return
    toks.update(TOK.findall(text))
    # HAVE_<EXT><suffix> tokens are pasted together by libavutil/cpu_internal.h's CPUEXT macros
    for ext in set(re.findall(r"\bAV_CPU_FLAG_([A-Z0-9_]+)\b", text)) | set(
            re.findall(r"\bCPUEXT\w*\([^)]*?,\s*([A-Z0-9_]+)\s*\)", text)):
        for suf in ("", "_EXTERNAL", "_INLINE"):
            toks.add("HAVE_" + ext + suf)
    here = os.path.dirname(path)
    for inc in re.findall(r'#\s*include\s+"([^"]+)"', text):
        for base in [here] + roots:
            cand = os.path.join(base, inc)
            if os.path.isfile(cand):
                scan(cand, seen, toks, roots)
                break


def main():
    ref, out = sys.argv[1], sys.argv[2]
    srcs = sys.argv[3:]
    # a second configuration from the same recipe (the checkasm build switches a few more components on): space-separated tokens
    ONES.update(os.environ.get("MKCONFIG_ONES", "").split())
    ONES.difference_update(os.environ.get("MKCONFIG_ZEROS", "").split())
    toks, seen = set(), set()
    for s in srcs:
        scan(s, seen, toks, [ref])
    # headers reached only through macros (template includes) — scan the three lib dirs' headers too
    for lib in ("libavutil", "libswscale", "libavcodec"):
        d = os.path.join(ref, lib)
        for fn in os.listdir(d):
            if fn.endswith(".h"):
                scan(os.path.join(d, fn), seen, toks, [ref])
    os.makedirs(os.path.join(out, "libavutil"), exist_ok=True)
    # every token goes into config.h; config_components.h just forwards to it
    cfg = sorted(toks)
    comp = []
    with open(os.path.join(out, "config.h"), "w") as f:
        f.write("/* written by oracle/refbuild/mkconfig.py (ours) - pure C, generic arch */\n"
                "#ifndef FFMPEG_CONFIG_H\n#define FFMPEG_CONFIG_H\n"
                '#define FFMPEG_CONFIGURATION "ffhip-oracle pure-C"\n#define FFMPEG_LICENSE "GPL version 2 or later"\n'
                '#define CC_IDENT "gcc"\n#define OS_NAME linux\n#define EXTERN_PREFIX ""\n#define EXTERN_ASM\n'
                '#define BUILDSUF ""\n#define SLIBSUF ".so"\n#define SWS_MAX_FILTER_SIZE 256\n'
                '#define FFMPEG_DATADIR "/nonexistent"\n#define AVCONV_DATADIR "/nonexistent"\n')
        for t in cfg:
            f.write("#define %s %d\n" % (t, 1 if t in ONES else 0))
        f.write("#endif\n")
    with open(os.path.join(out, "config_components.h"), "w") as f:
        f.write("#ifndef FFMPEG_CONFIG_COMPONENTS_H\n#define FFMPEG_CONFIG_COMPONENTS_H\n#include \"config.h\"\n")
        for t in comp:
            f.write("#define %s %d\n" % (t, 1 if t in ONES else 0))
        f.write("#endif\n")
    with open(os.path.join(out, "libavutil", "avconfig.h"), "w") as f:
        f.write("#ifndef AVUTIL_AVCONFIG_H\n#define AVUTIL_AVCONFIG_H\n#define AV_HAVE_BIGENDIAN 0\n"
                "#define AV_HAVE_FAST_UNALIGNED 1\n#endif\n")
    with open(os.path.join(out, "libavutil", "ffversion.h"), "w") as f:
        f.write('#ifndef AVUTIL_FFVERSION_H\n#define AVUTIL_FFVERSION_H\n#define FFMPEG_VERSION "ffhip-oracle"\n#endif\n')
    print("mkconfig: %d config tokens, %d component tokens, %d files scanned" % (len(cfg), len(comp), len(seen)))


if __name__ == "__main__":
    main()
