/*
 * ffref_stubs_h264dec.c — ours.  libavcodec's generic layer (avcodec.c, decode.c) names a few entry points of parts that are not compiled
 * into oracle/_ref/libffref_h264dec.so (the encoder side of avcodec.c, Dolby Vision RPU parsing of decode.c); an H.264 decode never
 * reaches them.  Traps, so that the object links with --no-undefined.  From the linker's complaint list.
 */
#include <stdlib.h>
#define TRAP(name) void name(void) { abort(); }
TRAP(ff_encode_receive_frame) TRAP(ff_encode_preinit) TRAP(ff_encode_internal_alloc) TRAP(ff_encode_flush_buffers)
TRAP(avcodec_default_get_encode_buffer) TRAP(ff_dovi_rpu_parse) TRAP(ff_dovi_get_metadata)
