/*
 * ffref_stubs.c — ours.  The reference files we compile reference a few symbols from parts of
 * FFmpeg that are not on the hot path (codec-level DCT helpers used by me_cmp's RD metrics, etc.).
 * They are never reached by the functions the shim exposes; define them as traps so the shared
 * object links with --no-undefined.  Filled in from the linker's complaint list.
 */
#include <stdlib.h>
#include <stdint.h>
#define TRAP(name) void name(void) { abort(); }
/* the AAC decoder around AACDecDSP (bitstream parsing, SBR): aacdec_float.c's proc functions name them */
TRAP(ff_aac_decode_ics) TRAP(ff_aac_sbr_ctx_alloc_init) TRAP(ff_aac_sbr_ctx_close) TRAP(ff_aac_sbr_decode_extension) TRAP(ff_aac_sbr_apply)
/* libavcodec/h264_mb.c (the macroblock reconstruction the picture-pipeline tests pin against): frame threading's wait on a
 * reference row — the shim decodes intra macroblocks only — and h264_ps.c's table of bytes per PCM macroblock */
void ff_thread_await_progress(const void *f, int progress, int field) { (void)f; (void)progress; (void)field; }
const uint16_t ff_h264_mb_sizes[4] = { 256, 384, 512, 768 };
