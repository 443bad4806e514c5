/*
 * ffref_stubs.c — ours.  The reference files we compile reference a few symbols from parts of
 * FFmpeg that are not on the hot path (codec-level DCT helpers used by me_cmp's RD metrics, etc.).
 * They are never reached by the functions the shim exposes; define them as traps so the shared
 * object links with --no-undefined.  Filled in from the linker's complaint list.
 */
#include <stdlib.h>
#include <stdint.h>
#define TRAP(name) void name(void) { abort(); }
