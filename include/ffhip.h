/*
 * ffhip.h — C-ABI of libffhip.so, the MI355X (gfx950) "hip" arch for FFmpeg's data-parallel DSP loops.
 *
 * This is the drop-in boundary.  Every entry point below names the reference
 * interface it stands in for (file:line relative to the FFmpeg tree).  Plain C: pointers, sizes,
 * ints.  No torch / C++ types.  INTEGRATION.md shows the few-line `ff_*_init_hip()` stubs a
 * maintainer adds on the FFmpeg side to bind these.
 *
 * Two faces per component (SURVEY.md §7 "granularity mismatch"):
 *   - signature-exact single-call shims taking HOST pointers: the reference's own function-pointer
 *     types, installable into SwsInternal / H264DSPContext / H264QpelContext / MECmpContext /
 *     FFTXCodelet and exercisable by checkasm.  They stage through device scratch and synchronise.
 *   - batched `_dev` entry points taking DEVICE pointers + a hipStream_t (as void *): the only form
 *     that can reach the HBM roofline.  Asynchronous on the given stream.
 *
 * Return convention: 0 (or a line count where the reference returns one) on success, negative
 * errno-style (== AVERROR(e) on POSIX) on failure.  FFHIP_ENOSYS when no HIP device is usable —
 * the caller keeps its C function pointer.  Nothing here silently falls back to a CPU path.
 */
#ifndef FFHIP_H
#define FFHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FFHIP_EINVAL (-22)
#define FFHIP_ENOMEM (-12)
#define FFHIP_ENOSYS (-38)
#define FFHIP_EIO    (-5)    /* a HIP runtime call failed; ffhip_last_error() has the text */

/* ------------------------------------------------------------------------------------------ */
/* runtime                                                                                    */
/* ------------------------------------------------------------------------------------------ */
/** Number of usable HIP devices (0 when none / no driver).  Mirrors the role of av_get_cpu_flags()
 *  & AV_CPU_FLAG_* gating in every ff_*_init_<arch>() (libavutil/cpu.h:32-62). */
int         ffhip_device_count(void);
/** Bind the calling thread to a device.  HIP's current device is per thread; context-free entry points (the batched `_dev`
 *  faces, the host-pointer shims) run on the calling thread's device, and their shared resources (staging arena, progress
 *  counters, coefficient tables, compiled op lists) live in per-device tables, so one process may drive every GPU of the node from
 *  as many threads as it likes — the reference's own model of one process with frame / slice threads
 *  (libavcodec/pthread_frame.c, libswscale/swscale.c:1645-1679).  May be called at any time, any number of times.
 *  The FIRST call of the process also sets the process default: a thread that never calls ffhip_set_device() (an FFmpeg worker
 *  thread calling a shim face) is bound to that default at its first entry instead of HIP's device 0.
 *  Contexts (FFHipSwsContext, FFHipTXContext, FFHipH264Picture, FFHipAacImdct, FFHipAacLd, FFHipSwsUOps) are bound to the device
 *  that was current when they were created: their calls make that device current for their own duration, whatever the calling
 *  thread is bound to.  The stream handed to a call must belong to the device the call runs on. */
int         ffhip_set_device(int device);
/** The calling thread's device (after the default binding described above), or FFHIP_ENOSYS. */
int         ffhip_get_device(void);
/** ffhip_set_device() for the duration of a callback that must leave the calling thread bound as it found it (AVBuffer pool
 *  callbacks run on whichever thread drops the last reference): push makes `device` current and stores in *prev what pop restores. */
int         ffhip_device_push(int device, int *prev);
void        ffhip_device_pop(int prev);
/** A stream on the calling thread's device (hipStreamNonBlocking), for callers without the HIP headers. */
int         ffhip_stream_create(void **stream);
int         ffhip_stream_destroy(void *stream);
/** Everything queued on `first` so far happens before anything queued on `then` from now on (an event recorded on one, waited for on
 *  the other).  Either may be NULL, the legacy default stream — which a non-blocking stream is not ordered against by itself. */
int         ffhip_stream_order(void *first, void *then);
/** Streaming-bandwidth probe of the current device (measurement aid: bench.py reports the box's achievable roofs beside
 *  the 8 TB/s spec).  pattern 0 read, 1 write, 2 copy, 3 read n/4 + write n (the 1080p->4K scaler's mix), 4 read n/2 + write n
 *  (yuv420p->rgb24's), 5 the runtime's hipMemcpyDtoDAsync.  Patterns 0-4 are a SWEEP of 24 ways to issue the same traffic (1 / 2 / 8
 *  16-byte accesses in flight per lane, plain or non-temporal, 256 x 4 or 256 x 16 workgroups, grid-stride or private XCD-adjacent
 *  slices) and answer with the best one.  `bytes` = the larger side's buffer; *gbps = bytes moved per second / 1e9 over `reps`
 *  launches of that variant (HIP events). */
int         ffhip_membw_probe(int pattern, size_t bytes, int reps, double *gbps);
/**
 * Device memory for frame batches whose rate does not depend on where the allocator happened to find it (round 6).  What a streaming
 * kernel over several gigabytes gets out of HBM depends on the PHYSICAL layout of its buffers: the same launch of the bench kernel runs at
 * 0.575 – 0.585 of HBM on one plain allocation and at 0.63 – 0.66 on the next, process by process and allocation by allocation — two
 * modes, the slow one most often on large physically contiguous blocks (the first big hipMalloc of a fresh process; 1 GiB physical chunks:
 * 7 of 12), whatever the virtual addresses (profiles/r06_alloc_vmm_sweep_*.txt, r06_arena_offset_sweep.txt).  ffhip_frames_alloc() builds
 * the range from physical chunks of `chunk` bytes (0: 16 MiB; a power of two, at least the device's allocation granularity) mapped
 * into one virtual range in a fixed pseudo-random order (HIP virtual memory management: hipMemCreate / hipMemMap), so that no large
 * piece of the range is physically contiguous: measured 0.60 – 0.645 on every one of 12 allocations where plain ones gave 0.575 – 0.66
 * (mean 0.628 against 0.610), and +3.6 % on average for the table converter.  *ptr: the range, ordinary device memory for every kernel,
 * copy and torch view of the current device; the tail of the last chunk is mapped too.  bench.py holds its frame batches in it; a caller
 * that keeps hundreds of frames resident (a transcoder's look-ahead, an inference batch) should too.
 * Returns 0, FFHIP_EINVAL, FFHIP_ENOMEM or FFHIP_ENOSYS (no virtual memory management on this device / runtime).
 */
int         ffhip_frames_alloc(void **ptr, size_t bytes, size_t chunk);
/** Unmaps and releases a range ffhip_frames_alloc() returned (NULL: nothing).  0 or FFHIP_EINVAL (not such a range). */
int         ffhip_frames_free(void *ptr);
const char *ffhip_last_error(void);
const char *ffhip_version(void);
/** Device memory helpers for callers that do not bring their own allocator. */
int         ffhip_malloc(void **dev_ptr, size_t bytes);
int         ffhip_free(void *dev_ptr);
int         ffhip_memcpy_h2d(void *dev_dst, const void *host_src, size_t bytes);
int         ffhip_memcpy_d2h(void *host_dst, const void *dev_src, size_t bytes);
/** The device ordinal `p` lives on, or FFHIP_EINVAL for host / unknown memory. */
int         ffhip_pointer_device(const void *p);
/** Pitched plane copies, asynchronous on `stream` (the transfer_data_to / _from of an AVHWFramesContext: integration/avutil_hwcontext_hip.c). */
int         ffhip_memcpy2d_h2d_async(void *dev_dst, size_t dpitch, const void *host_src, size_t spitch, size_t width_bytes, size_t rows, void *stream);
int         ffhip_memcpy2d_d2h_async(void *host_dst, size_t dpitch, const void *dev_src, size_t spitch, size_t width_bytes, size_t rows, void *stream);
/** hipStreamSynchronize; also the point where a row-ordered launch OF THIS STREAM (frame-order deblocking, intra wavefront, VP9
 *  frame loop filter) that timed out on a row hand-off (never in a correct run) is reported: FFHIP_EIO once, ffhip_last_error()
 *  has the text — its picture is only partly processed.  Other streams' pictures are not affected and not reported here. */
int         ffhip_stream_synchronize(void *stream);

/* ------------------------------------------------------------------------------------------ */
/* several GPUs in one process: device set, frame-batch scatter / gather over xGMI              */
/* ------------------------------------------------------------------------------------------ */
/* The path shards embarrassingly: a batch is cut into contiguous ranges, one per device, with no data-path
 * collective.  What the reference does with threads inside one process (frame threads, libavcodec/pthread_frame.c; slice threads,
 * libswscale/swscale.c:1645-1679; avfilter slice threading for the motion search, libavfilter/vf_minterpolate.c) a caller does here
 * with one host thread (or one loop) per member of an FFHipDeviceSet.  ffmpeg_amd/dist.py holds the one-process-per-GPU form of
 * the same partition (torch.distributed / RCCL). */
typedef struct FFHipDeviceSet FFHipDeviceSet;
/** ceil(n/world) contiguous items per rank; the last ranks may get fewer or none. */
void ffhip_shard_range(int64_t n_items, int rank, int world, int64_t *lo, int64_t *hi);
/** Motion search over a sequence: pair p searches frame p+1 in frame p.  Pairs [plo, phi) shard like everything else and read
 *  frames [flo, fhi) = [plo, phi] — the rank's own range plus ONE halo frame (empty when the rank has no pair). */
void ffhip_shard_frame_pairs(int64_t n_frames, int rank, int world, int64_t *plo, int64_t *phi, int64_t *flo, int64_t *fhi);
/** `n` devices (devices == NULL or n <= 0: all of them), one non-blocking stream each, peer access enabled between every pair
 *  that has a direct xGMI path.  A device may appear more than once (a member is a (device, stream) pair). */
int   ffhip_device_set_create(FFHipDeviceSet **s, const int *devices, int n);
void  ffhip_device_set_free(FFHipDeviceSet **s);
int   ffhip_device_set_size(const FFHipDeviceSet *s);
int   ffhip_device_set_device(const FFHipDeviceSet *s, int member);
void *ffhip_device_set_stream(const FFHipDeviceSet *s, int member);
/** ffhip_set_device(member's device) for the calling thread: a worker thread's first call. */
int   ffhip_device_set_bind(const FFHipDeviceSet *s, int member);
/** Waits for every member's stream (ffhip_stream_synchronize semantics); the first error wins. */
int   ffhip_device_set_synchronize(FFHipDeviceSet *s);
/** full[lo_i, hi_i) (items of item_bytes, on the root member's device) -> shards[i] on member i, hipMemcpyPeerAsync on member
 *  i's stream behind what the root's stream has queued.  _scatter uses ffhip_shard_range(); _ranges takes explicit, possibly
 *  overlapping ranges (halos); _frames_for_pairs the motion search's ranges.  shards[root] may point into `full` (no copy). */
int   ffhip_batch_scatter(FFHipDeviceSet *s, int root, const void *full, size_t item_bytes, int64_t n_items, void *const *shards);
int   ffhip_batch_scatter_ranges(FFHipDeviceSet *s, int root, const void *full, size_t item_bytes, int64_t n_items, const int64_t *lo,
                                 const int64_t *hi, void *const *shards);
int   ffhip_batch_scatter_frames_for_pairs(FFHipDeviceSet *s, int root, const void *frames, size_t frame_bytes, int64_t n_frames,
                                           void *const *shards);
/** The inverse: shards[i] -> full[lo_i, hi_i) on the root, each copy on member i's stream behind the work that produced the shard;
 *  the root's stream then waits for all copies (so work queued on it afterwards sees the whole batch). */
int   ffhip_batch_gather(FFHipDeviceSet *s, int root, void *full, size_t item_bytes, int64_t n_items, const void *const *shards);
int   ffhip_batch_gather_ranges(FFHipDeviceSet *s, int root, void *full, size_t item_bytes, int64_t n_items, const int64_t *lo,
                                const int64_t *hi, const void *const *shards);

/* ------------------------------------------------------------------------------------------ */
/* libswscale: hscale / vscale / yuv2rgb                                                      */
/* ------------------------------------------------------------------------------------------ */
/* Pixel formats: numeric values are AVPixelFormat's (libavutil/pixfmt.h). */
#define FFHIP_PIX_FMT_YUV420P 0
#define FFHIP_PIX_FMT_RGB24   2
#define FFHIP_PIX_FMT_BGR24   3
#define FFHIP_PIX_FMT_YUV422P 4    /* planar 4:2:2 and 4:4:4, 8 bits (== AV_PIX_FMT_YUV422P / _YUV444P): sources and targets of the scaler;
                                    * to packed RGB 4:4:4 sources run with SWS_FULL_CHR_H_INT, as in the reference (utils.c:1276-1285) */
#define FFHIP_PIX_FMT_YUV444P 5
#define FFHIP_PIX_FMT_YUVA420P 33  /* planar YUV with an alpha plane, 8 bits (== AV_PIX_FMT_YUVA420P / _YUVA422P / _YUVA444P).  As sources their
                                    * alpha plane is not read when the target has none (c->needAlpha = 0, utils.c:1398); as targets of sources
                                    * without alpha the plane dst[3] is filled with 255, as ff_swscale() fills it (swscale.c:536-553).  Alpha on
                                    * both sides: the alpha plane is scaled by the luma banks (FFHipSwsTables.dst_alpha_fill == 2) */
#define FFHIP_PIX_FMT_YUVA422P 78
#define FFHIP_PIX_FMT_YUVA444P 79
#define FFHIP_PIX_FMT_YUVJ420P 12  /* the full-range "J" twins (== AV_PIX_FMT_YUVJ420P / 422P / 444P): taken when BOTH sides are J formats —
                                    * equal ranges need no range conversion, the conversion is the base formats' (handle_jpeg(),
                                    * libswscale/utils.c:1019-1050); a J format on one side only, or J to packed RGB, is not on the hip path */
#define FFHIP_PIX_FMT_YUVJ422P 13
#define FFHIP_PIX_FMT_YUVJ444P 14
/* Above 8 bits (== AV_PIX_FMT_*; little-endian): planar yuv4xxp at 9 / 10 / 12 / 14 / 16 bits with the samples in the low bits, and
 * the semi-planar P010 / P012 / P016 with the samples in the HIGH bits.  Sources and targets of SCALED contexts on the legacy path
 * (hScale16To15_c / hScale16To19_c / hScale8To19_c, yuv2plane1 / yuv2planeX at the target depth, the P01x readers and writers:
 * libswscale/swscale.c:69-160, output.c:150-360, input.c p010LEToY_c / p010LEToUV_c), freely mixed with the 8-bit YUV formats above
 * (an 8-bit target fed from a deeper source is dithered with ff_dither_8x8_128, as swscale does: swscale.c:291,519-522).  Equal-size
 * YUV conversions take the reference's special converters (swscale_unscaled.c) and are not on the hip path.  Round 6: 9 .. 14-bit sources into
 * the packed 8-bit RGB targets (yuv2rgb_X / _2 / _1 from 16-bit lines, vscale.c:126-170) — even widths, subsampled sources, banks of at
 * most 16 taps; not the full-chroma writers.
 * Packed 8-bit RGB as a SOURCE (round 6): rgb24, bgr24, argb, rgba, abgr, bgra into any YUV target above — the input converters
 * (libswscale/input.c:264-400,1068-1190) run as a kernel and the context is the one of their 14-bit lines (ffhip_sws_from_tables_rgb_source);
 * not into full-range (J) or alpha-carrying targets from an alpha source, not RGB -> RGB, and bgr24 -> yuv420p at the source's size is the
 * reference's own special converter. */
#define FFHIP_PIX_FMT_YUV420P16LE 45
#define FFHIP_PIX_FMT_YUV422P16LE 47
#define FFHIP_PIX_FMT_YUV444P16LE 49
#define FFHIP_PIX_FMT_YUV420P9LE  60
#define FFHIP_PIX_FMT_YUV420P10LE 62
#define FFHIP_PIX_FMT_YUV422P10LE 64
#define FFHIP_PIX_FMT_YUV444P9LE  66
#define FFHIP_PIX_FMT_YUV444P10LE 68
#define FFHIP_PIX_FMT_YUV422P9LE  70
#define FFHIP_PIX_FMT_YUV420P12LE 123
#define FFHIP_PIX_FMT_YUV420P14LE 125
#define FFHIP_PIX_FMT_YUV422P12LE 127
#define FFHIP_PIX_FMT_YUV422P14LE 129
#define FFHIP_PIX_FMT_YUV444P12LE 131
#define FFHIP_PIX_FMT_YUV444P14LE 133
#define FFHIP_PIX_FMT_P010LE      158
#define FFHIP_PIX_FMT_P016LE      169
#define FFHIP_PIX_FMT_P012LE      209
#define FFHIP_PIX_FMT_NV12    23
#define FFHIP_PIX_FMT_NV21    24
#define FFHIP_PIX_FMT_ARGB    25   /* packed 8:8:8:8; alpha = 255 (the sources on this path carry none) */
#define FFHIP_PIX_FMT_RGBA    26
#define FFHIP_PIX_FMT_ABGR    27
#define FFHIP_PIX_FMT_BGRA    28
#define FFHIP_PIX_FMT_GBRP    71   /* planar G, B, R (== AV_PIX_FMT_GBRP): a target of the equal-size table converter only (yuv420p_gbrp_c /
                                    * yuv422p_gbrp_c, libswscale/yuv2rgb.c:533,553) */
/* The equal-size table converter (ff_yuv2rgb_get_func_ptr(), libswscale/yuv2rgb.c:562-676; chosen at swscale_unscaled.c:2425-2431 when
 * the sizes are equal, SWS_ACCURATE_RND is off and the height is even) takes yuv420p, yuv422p (each luma row with the chroma row of its
 * own, YUV422FUNC) and yuva420p sources; the alpha plane of the latter drives the alpha byte of the four 32-bit targets (yuva2rgba_c /
 * yuva2argb_c) and is not read for the others.  Targets: rgb24, bgr24, argb, rgba, abgr, bgra, gbrp; even widths (an odd width is the C
 * converter's tail case and stays there). */
/* Flags: numeric values are SwsFlags' (libswscale/swscale.h:130-153). */
#define FFHIP_SWS_FULL_CHR_H_INT 0x2000 /* full chroma interpolation for packed RGB targets (swscale.h:147) */
#define FFHIP_SWS_FULL_CHR_H_INP 0x4000 /* full chroma input from a packed RGB source (swscale.h:153): no horizontal 2:1 in the converters */
#define FFHIP_SWS_FAST_BILINEAR 0x1
#define FFHIP_SWS_BILINEAR      0x2
#define FFHIP_SWS_BICUBIC       0x4
#define FFHIP_SWS_POINT         0x10
#define FFHIP_SWS_AREA          0x20
#define FFHIP_SWS_BICUBLIN      0x40
#define FFHIP_SWS_GAUSS         0x80
#define FFHIP_SWS_SINC          0x100
#define FFHIP_SWS_LANCZOS       0x200
#define FFHIP_SWS_ACCURATE_RND  0x40000
#define FFHIP_SWS_BITEXACT      0x80000

/**
 * One separable filter bank as the reference's initFilter() produces it
 * (libswscale/utils.c:197-612): `filter[n][size]` int16 coefficients, `pos[n]` first source sample.
 * Horizontal banks sum to ~1<<14, vertical to 1<<12.
 */
typedef struct FFHipSwsFilter {
    const int16_t *filter;
    const int32_t *pos;
    int            size;   /* taps per output sample  (SwsInternal.hLumFilterSize ...) */
    int            n;      /* number of output samples (dstW, chrDstW, dstH, chrDstH)  */
} FFHipSwsFilter;

/**
 * What ff_sws_init_swscale_hip(SwsInternal *c) hands over: the context fields the hot path reads
 * (libswscale/swscale_internal.h:330-700).  All arrays are copied by ffhip_sws_from_tables().
 */
typedef struct FFHipSwsTables {
    int srcW, srcH, srcFormat;
    int dstW, dstH, dstFormat;
    int flags;
    FFHipSwsFilter hLum, hChr, vLum, vChr;   /* c->{h,v}{Lum,Chr}Filter[Pos|Size]           */
    /* yuv2rgb: the already cy-scaled coefficients the LUTs are built from
     * (libswscale/yuv2rgb.c:717-800): cy, oy, crv', cbu', cgu', cgv' and the ramp offset yoffs. */
    int64_t yuv2rgb_cy, yuv2rgb_oy, yuv2rgb_crv, yuv2rgb_cbu, yuv2rgb_cgu, yuv2rgb_cgv;
    int     yuv2rgb_yoffs;
    /* Range conversion between YUV formats (round 3): c->opts.src_range / dst_range (0 limited, 1 full) and, when they differ,
     * the constants of c->lumConvertRange / chrConvertRange (ff_sws_init_range_convert + init_range_convert_constants,
     * libswscale/swscale.c:591-660), which the scaler applies to its horizontal intermediates: lumRangeToJpeg_c & co
     * (swscale.c:160-255).  Packed RGB targets take the range through the yuv2rgb coefficients instead (not these fields). */
    int      src_range, dst_range;
    uint32_t lumConvertRange_coeff, chrConvertRange_coeff;
    int64_t  lumConvertRange_offset, chrConvertRange_offset;
    /* SWS_FULL_CHR_H_INT on a packed RGB target — asked for, or forced by an odd width or a 4:4:4 source (libswscale/utils.c:1270-1290):
     * hChr has dstW entries and the yuv2rgb_full_{1,2,X} writers run (output.c:1998-2310) with c->yuv2rgb_{y_coeff, y_offset,
     * v2r_coeff, v2g_coeff, u2g_coeff, u2b_coeff} (yuv2rgb.c:786-791), in this order */
    int      full_chr_h_int;
    int      yuv2rgb_full[6];
    /* 1: the target has an alpha plane the source does not drive: dst[3] is filled with 255 (fillPlane, swscale.c:536-553).
     * 2: both sides are planar YUVA: the alpha plane is scaled by the LUMA banks (lum_h_scale / lum_planar_vscale on plane 3,
     *    hscale.c:63-79, vscale.c:57-70) — src[3] -> dst[3] as the luma of a second pass of the context in which the planners enumerate
     *    the luma job only; not together with a range conversion (the reference converts plane 0 only).
     *    With the equal-size table converter to a 32-bit packed RGB target: src[3] drives the alpha byte (yuva2rgba_c / yuva2argb_c).
     * srcFormat / dstFormat above are then the formats without the alpha plane */
    int      dst_alpha_fill;
} FFHipSwsTables;

typedef struct FFHipSwsContext FFHipSwsContext;

/** Stand-alone construction; same argument meaning as sws_getContext() (libswscale/swscale.h,
 *  libswscale/utils.c:2043) for the formats above; srcFilter/dstFilter/param are the defaults.
 *  Filter banks come from our host restatement of initFilter() (cpu_flags == 0 => filterAlign 1).
 *  Returns NULL (and sets ffhip_last_error) for an unsupported conversion or when no device. */
FFHipSwsContext *ffhip_sws_getContext(int srcW, int srcH, int srcFormat,
                                      int dstW, int dstH, int dstFormat, int flags);
/** Drop-in construction from FFmpeg's own tables (no filter generation on our side).  A context the reference gave a special
 *  converter has no banks (ff_sws_init_single_context() returns before initFilter(), libswscale/utils.c:1625-1637): for the
 *  equal-size yuv420p -> packed RGB table converter (swscale_unscaled.c:2425-2431) the four banks may be left NULL. */
FFHipSwsContext *ffhip_sws_from_tables(const FFHipSwsTables *t);
/** The same for a packed 8-bit RGB source (rgb24 / bgr24 / rgba / bgra / argb / abgr) in front of a YUV target (round 6; ffhip_sws_getContext()
 *  takes these formats directly).  In the reference such a source is its input converters' int16 lines (lumToYV12 / chrToYV12: rgb24ToY_c,
 *  rgb24ToUV_c, rgb24ToUV_half_c and their twins, libswscale/input.c:264-400,1068-1190) scaled by hScale16To15_c at sh = 13
 *  (swscale.c:100-128) — the 16-bit path of a 14-bit planar source, undithered (swscale.c:291).  `t` describes that conversion:
 *  t->srcFormat = FFHIP_PIX_FMT_YUV422P14LE when the chroma converters read pixel pairs (c->chrSrcHSubSample == 1, utils.c:1340-1352),
 *  FFHIP_PIX_FMT_YUV444P14LE otherwise; the banks, ranges (equal) and flags are the context's own.  rgbFormat: the caller's source
 *  format; rgb2yuv: c->input_rgb2yuv_table[RY_IDX .. BV_IDX] (swscale_internal.h:468-477; sws_setColorspaceDetails() fills it).
 *  The context's source is then ONE packed plane, on every face. */
FFHipSwsContext *ffhip_sws_from_tables_rgb_source(const FFHipSwsTables *t, int rgbFormat, const int32_t rgb2yuv[9]);
/** sws_setColorspaceDetails() on a live RGB-source context (fill_rgb2yuv_table(), libswscale/utils.c:614-705,1002): the converter table
 *  replaces the context's.  0, or FFHIP_EINVAL for a context that has no RGB source. */
int ffhip_sws_set_rgb2yuv(FFHipSwsContext *c, const int32_t rgb2yuv[9]);
/** The yuv2rgb fields of the tables (yuv2rgb_cy .. yuv2rgb_yoffs, yuv2rgb_full[]) as ff_yuv2rgb_c_init_tables() derives them
 *  (libswscale/yuv2rgb.c:750-797) from what the context stores: inv_table = c->srcColorspaceTable, fullRange = sws->src_range,
 *  c->brightness / c->contrast / c->saturation (sws_setColorspaceDetails(), utils.c:848-905).  No device needed.  0 or FFHIP_EINVAL. */
int              ffhip_sws_yuv2rgb_coeffs(FFHipSwsTables *t, const int inv_table[4], int fullRange, int brightness, int contrast,
                                          int saturation);
/** sws_setColorspaceDetails() after the context was made (utils.c:990-996 re-runs ff_yuv2rgb_c_init_tables()): the yuv2rgb fields of
 *  `t` replace the context's; every other field of `t` is ignored.  0, FFHIP_EINVAL (coefficients outside the closed form's range). */
int              ffhip_sws_set_yuv2rgb(FFHipSwsContext *c, const FFHipSwsTables *t);
void             ffhip_sws_freeContext(FFHipSwsContext *c);
/** 1 when the context's banks run on the register-resident column-walking kernel (4-tap x 4-tap banks
 *  whose 4-column groups read one 8-byte source span; sws_colwalk.hip), 0 when they take the general
 *  LDS-tiled kernel; bit 1 set when the matrix-core variant (k_sws_mfma) is available too; bit 2 set when the
 *  banks (5..32 taps: down-scaling, long kernels) run on the LDS-backed wide-bank walker (sws_lwalk.hip); bit 3 set when
 *  the conversion is an exact 2x up-scale served by the static-schedule kernel (sws_up2.hip); bit 4: an exact 2:1 down-scale
 *  (sws_down2.hip); bit 5: the 16-bit column walker (sws_walk16.hip); bit 6: exact 2x of 4:2:0 (yuv420p, NV12, NV21) into packed RGB, the
 *  static-schedule kernel with the yuv2rgb_X writer fused (sws_up2rgb.hip); bit 7: the same sources into packed RGB at the source's size
 *  (sws_eqrgb.hip); bit 8: planar 4:4:4 into packed RGB at the source's size, the full-chroma writer on one-tap banks (sws_full444.hip); bit 9: 4:2:0
 *  between its planar and semi-planar layouts at the same size (sws_copy420.hip); bits 10 / 11: yuv444p -> yuv420p / yuv420p -> yuv444p at
 *  the same size: the luma plane copied, the chroma planes on the exact-2:1 / exact-2x static-schedule kernels; bit 12: an exact 3:2
 *  down-scale (sws_down32.hip); bit 13: an exact 3:2 up-scale above 8 bits (sws_up32.hip).
 *  Diagnostic only: results are identical. */
int              ffhip_sws_fast_path(const FFHipSwsContext *c);
/** Diagnostic: the workgroup numbering the context's launch tuner settled on for large launches of the table converter
 *  (yuv2rgb_c_24_rgb's replacement, libswscale/yuv2rgb.c:530) — -1 undecided (fewer than four large launches so far), 0 plain,
 *  1 an eighth of the launch per XCD.  Which one is faster is a property of the box; results are identical. */
int              ffhip_sws_tuned_numbering(const FFHipSwsContext *c);
/** Host-side preparation of the matrix-core horizontal pass (no device needed): turns one 4-tap horizontal
 *  bank (hLumFilter/hLumFilterPos or the chroma pair's) into per-tile MFMA operand records of 2320 bytes —
 *  B_hi[64 lanes][16], B_lo[64][16] (coefficient = 256*hi + lo, zero outside the band), bias[64] = 128*sum(f),
 *  window base — for tiles of 32 samples (pair != 0: 16 U + 16 V columns of a byte-interleaved plane,
 *  src_swap: V first).  Returns the tile count (records written to `out` when non-NULL) or FFHIP_EINVAL when
 *  the bank does not fit the tiling.  Exposed for the CPU test-suite. */
int              ffhip_sws_mfma_tiles_host(const int16_t *filter, const int32_t *pos, int n, int srcW, int pair, int src_swap,
                                           uint8_t *out, size_t out_size);

/** Host-side preparation of the exact-2x kernel (no device needed): re-expresses a 4-tap bank of a 2x up-scale (n_dst ==
 *  2 n_src) as four coefficients per output on the REGULAR window start (x >> 1) - 2 + (x & 1) of the edge-replicated row —
 *  what initFilter()'s border fold (libswscale/utils.c:519-561) amounts to.  out: n_dst x 2 dwords, (c0, c1) (c2, c3) as
 *  int16 pairs.  Returns 1, or 0 when some tap of the bank does not sit on its regular window (the kernel is then not
 *  used).  Exposed for the CPU test-suite. */
int              ffhip_sws_up2_virtual_bank_host(const int16_t *filter, const int32_t *pos, int n_dst, int n_src, uint32_t *out);
/** The same at `ratio` 2 or 4 (round 5; sws_up2rgb.hip: a packed-RGB target has a chroma line per output line, so 4:2:0 chroma goes up
 *  four times vertically — output y on the regular window start ((y + 2) >> 2) - 2).  ratio 2 == the call above. */
int              ffhip_sws_upn_virtual_bank_host(const int16_t *filter, const int32_t *pos, int n_dst, int n_src, int ratio, uint32_t *out);
/** The two horizontal virtual banks of the exact-2x RGB kernel (luma: n_hl columns, chroma: n_hc; 2 dwords each, as the calls above
 *  write them) folded into the 32 dwords the kernel reads with scalar loads: [0..3] luma even c01, c23, odd c01, c23; [4..7] chroma;
 *  [8..13] / [14..19] the luma bank's first / last three columns; [20..25] / [26..31] the chroma bank's.  Returns 1, or 0 when a bank
 *  does not repeat with period 2 between its ends (the kernel is then not used). */
int              ffhip_sws_up2rgb_hco_host(const uint32_t *hl, int n_hl, const uint32_t *hc, int n_hc, uint32_t out[32]);
/** The same for the exact-2:1 kernel (sws_down2.hip): a bank of `fsize` <= 16 taps of a 2:1 down-scale (n_src == 2 n_dst) as
 *  eight coefficients per output on the regular window 2x - 3 .. 2x + 4 of the edge-replicated row.  out: n_dst x 4 dwords,
 *  (c0, c1) .. (c6, c7).  Returns 1, or 0 when some tap does not sit on its regular window. */
int              ffhip_sws_down2_virtual_bank_host(const int16_t *filter, const int32_t *pos, int fsize, int n_dst, int n_src, uint32_t *out);
/** The same for the exact-3:2 kernel (sws_down32.hip): a bank of `fsize` <= 12 taps of a 3:2 down-scale (2 n_src == 3 n_dst, n_dst
 *  even) as six coefficients per output on the regular window 3 (x >> 1) - 2 + (x & 1) .. + 5 of the edge-replicated row.  out: n_dst x 3
 *  dwords, (c0, c1) (c2, c3) (c4, c5).  Returns 1, or 0 when some tap does not sit on its regular window. */
int              ffhip_sws_d32_virtual_bank_host(const int16_t *filter, const int32_t *pos, int fsize, int n_dst, int n_src, uint32_t *out);

/** Host-table generation alone (no device needed): our initFilter().  `which`: 0 hLum 1 hChr 2 vLum
 *  3 vChr.  Returns filter size or <0; pointers stay valid until ffhip_sws_tables_free().  Used by
 *  the CPU test-suite to pin the host logic against the reference's tables. */
typedef struct FFHipSwsHostTables FFHipSwsHostTables;
FFHipSwsHostTables *ffhip_sws_tables_create(int srcW, int srcH, int srcFormat,
                                            int dstW, int dstH, int dstFormat, int flags);
int  ffhip_sws_tables_get(const FFHipSwsHostTables *t, FFHipSwsTables *out);
/** 1 when the (src,dst,flags) triple takes the reference's table-driven unscaled converter
 *  `yuv2rgb_c_24_rgb` (rule at libswscale/swscale_unscaled.c:2425-2431), 0 for ff_swscale(). */
int  ffhip_sws_tables_is_unscaled_yuv2rgb(const FFHipSwsHostTables *t);
/** The srcRange / dstRange arguments of sws_setColorspaceDetails() (libswscale/utils.c:848-1000) for YUV targets: 0 limited
 *  (MPEG), 1 full (JPEG).  When they differ the tables carry the constants of c->lumConvertRange / chrConvertRange and the
 *  scaler applies them to its horizontal intermediates (swscale.c:160-255); a J format on one side of
 *  ffhip_sws_tables_create() sets the same.  Call before ffhip_sws_tables_get(). */
int  ffhip_sws_tables_set_ranges(FFHipSwsHostTables *t, int src_range, int dst_range);
void ffhip_sws_tables_free(FFHipSwsHostTables *t);

/**
 * SwsFunc-shaped frame call with HOST pointers — the pointer installed as `c->convert_unscaled`
 * (typedef at libswscale/swscale_internal.h:99-101; called at libswscale/swscale.c:1185).
 * Stages src -> device, runs the kernels, copies the written lines back; returns the number of
 * output lines written (srcSliceH for unscaled, dstH for a whole-frame scaled call) or <0.
 * Scaled contexts may be fed in source slices the way sws_scale() allows (in order, top to bottom, starting on even lines;
 * src[] points at the slice's first rows): the slices are collected on the device, the calls before the last return 0 lines
 * and the last one writes the whole picture and returns dstH — the pixels do not depend on the slicing
 * (libswscale's tools/scale_slice_test.c property); a slice out of order is FFHIP_EINVAL.
 */
int ffhip_sws_scale(FFHipSwsContext *c, const uint8_t *const src[], const int srcStride[],
                    int srcSliceY, int srcSliceH, uint8_t *const dst[], const int dstStride[]);

/**
 * Batched, device-resident face.  Frame f of plane p starts at src[p] + f*srcFramePitch[p]
 * (bytes), rows are srcStride[p] bytes apart; likewise dst.  Planes: yuv420p {Y,U,V}, nv12 {Y,UV},
 * rgb24 {RGB}.  Asynchronous on `stream` (hipStream_t).  Returns 0 or <0.
 */
int ffhip_sws_scale_batch_dev(FFHipSwsContext *c, int nframes,
                              const void *const src[4], const int srcStride[4], const size_t srcFramePitch[4],
                              void *const dst[4], const int dstStride[4], const size_t dstFramePitch[4],
                              void *stream);

/** Per-line parity face (device pointers): hyScale/hcScale == hScale8To15_c
 *  (libswscale/swscale.c:128-142; pointer type swscale_internal.h:648-653), batched over `nlines`
 *  lines that share one filter bank. */
int ffhip_sws_hscale8to15_dev(int16_t *dst, int dstW, ptrdiff_t dstPitch, const uint8_t *src, ptrdiff_t srcPitch,
                              int nlines, const int16_t *filter, const int32_t *filterPos, int filterSize,
                              void *stream);
/** yuv2planeX_8_c / yuv2plane1_8_c (libswscale/output.c:468-493; type swscale_internal.h:128-144):
 *  `nsrc` int16 lines at src + j*srcPitch, one output line. dither is 8 bytes. */
int ffhip_sws_yuv2planeX8_dev(const int16_t *filter, int filterSize, const int16_t *src, ptrdiff_t srcPitch,
                              uint8_t *dest, int dstW, const uint8_t *dither8, int offset, void *stream);

/**
 * The per-line members an ff_sws_init_swscale_<arch>() installs (libswscale/swscale.c:697-714; template x86/swscale.c), with the
 * reference's exact signatures, HOST pointers — what tests/checkasm/sw_scale.c:109-458 exercises:
 *   hyScale / hcScale   swscale_internal.h:648-653 (8-bit sources: hScale8To15_c, swscale.c:128-142); `c` is passed through untouched
 *   yuv2plane1          :128 (yuv2plane1_8_c, output.c:482-493)        yuv2planeX   :144 (yuv2planeX_8_c, output.c:468-480)
 *   yuv2nv12cX          :164 (yuv2nv12cX_c, output.c:500-529)
 * One call = one launch through device scratch; lines may be over-read exactly as far as the reference's may (filterPos + filterSize).
 */
typedef struct FFHipSwsLineContext {
    void (*hyScale)(void *c, int16_t *dst, int dstW, const uint8_t *src, const int16_t *filter, const int32_t *filterPos, int filterSize);
    void (*hcScale)(void *c, int16_t *dst, int dstW, const uint8_t *src, const int16_t *filter, const int32_t *filterPos, int filterSize);
    void (*yuv2plane1)(const int16_t *src, uint8_t *dest, int dstW, const uint8_t *dither, int offset);
    void (*yuv2planeX)(const int16_t *filter, int filterSize, const int16_t **src, uint8_t *dest, int dstW, const uint8_t *dither, int offset);
    void (*yuv2nv12cX)(int dstFormat, const uint8_t *chrDither, const int16_t *chrFilter, int chrFilterSize, const int16_t **chrUSrc,
                       const int16_t **chrVSrc, uint8_t *dest, int dstW);
} FFHipSwsLineContext;
/** Fills the members that apply to the format pair (8-bit yuv420p / nv12 / nv21 sources; planar, NV or packed-RGB targets) and
 *  remembers the ones it displaces as fallbacks (see ff_h264dsp_init_hip).  Returns 0, FFHIP_EINVAL or FFHIP_ENOSYS. */
int ff_sws_init_swscale_hip(FFHipSwsLineContext *c, int srcFormat, int dstFormat);
/** yuv2packed1 / yuv2packed2 / yuv2packedX (swscale_internal.h:201-266) for the packed RGB targets (yuv2rgb_{1,2,X}_c_template,
 *  output.c:1789-1939): the reference's argument lists, except that the first argument is this library's context (the members
 *  read the yuv2rgb tables of the context) and that they return 0 or a negative FFHIP_E* — the caller then runs the C member.
 *  dstW even; alpSrc / y are accepted and unused (no alpha plane among the supported sources, no dithered target). */
int ffhip_sws_yuv2packed1(FFHipSwsContext *c, const int16_t *lumSrc, const int16_t *chrUSrc[2], const int16_t *chrVSrc[2],
                          const int16_t *alpSrc, uint8_t *dest, int dstW, int uvalpha, int y);
int ffhip_sws_yuv2packed2(FFHipSwsContext *c, const int16_t *lumSrc[2], const int16_t *chrUSrc[2], const int16_t *chrVSrc[2],
                          const int16_t *alpSrc[2], uint8_t *dest, int dstW, int yalpha, int uvalpha, int y);
int ffhip_sws_yuv2packedX(FFHipSwsContext *c, const int16_t *lumFilter, const int16_t **lumSrc, int lumFilterSize,
                          const int16_t *chrFilter, const int16_t **chrUSrc, const int16_t **chrVSrc, int chrFilterSize,
                          const int16_t **alpSrc, uint8_t *dest, int dstW, int y);

/* ------------------------------------------------------------------------------------------ */
/* libavcodec: h264dsp                                                                        */
/* ------------------------------------------------------------------------------------------ */
/**
 * H264DSPContext subset (libavcodec/h264dsp.h:42-117), same member names and signatures, HOST
 * pointers.  ff_h264dsp_init_hip() below fills it the way ff_h264dsp_init_<arch>() would
 * (libavcodec/h264dsp.c:155-169).
 */
typedef struct FFHipH264DSPContext {
    void (*v_loop_filter_luma)(uint8_t *pix, ptrdiff_t stride, int alpha, int beta, int8_t *tc0);
    void (*h_loop_filter_luma)(uint8_t *pix, ptrdiff_t stride, int alpha, int beta, int8_t *tc0);
    void (*v_loop_filter_luma_intra)(uint8_t *pix, ptrdiff_t stride, int alpha, int beta);
    void (*h_loop_filter_luma_intra)(uint8_t *pix, ptrdiff_t stride, int alpha, int beta);
    void (*v_loop_filter_chroma)(uint8_t *pix, ptrdiff_t stride, int alpha, int beta, int8_t *tc0);
    void (*h_loop_filter_chroma)(uint8_t *pix, ptrdiff_t stride, int alpha, int beta, int8_t *tc0);
    void (*v_loop_filter_chroma_intra)(uint8_t *pix, ptrdiff_t stride, int alpha, int beta);
    void (*h_loop_filter_chroma_intra)(uint8_t *pix, ptrdiff_t stride, int alpha, int beta);
    void (*idct_add)(uint8_t *dst, int16_t *block, ptrdiff_t stride);
    void (*idct8_add)(uint8_t *dst, int16_t *block, ptrdiff_t stride);
    void (*idct_dc_add)(uint8_t *dst, int16_t *block, ptrdiff_t stride);
    void (*idct8_dc_add)(uint8_t *dst, int16_t *block, ptrdiff_t stride);
    void (*idct_add16)(uint8_t *dst, const int *blockoffset, int16_t *block, ptrdiff_t stride,
                       const uint8_t nnzc[5 * 8]);
    void (*idct8_add4)(uint8_t *dst, const int *blockoffset, int16_t *block, ptrdiff_t stride,
                       const uint8_t nnzc[5 * 8]);
    void (*idct_add16intra)(uint8_t *dst, const int *blockoffset, int16_t *block, ptrdiff_t stride,
                            const uint8_t nnzc[5 * 8]);
    void (*idct_add8)(uint8_t **dst, const int *blockoffset, int16_t *block, ptrdiff_t stride,
                      const uint8_t nnzc[15 * 8]);                                   /* h264dsp.h:96-98 (4:2:0) */
    void (*luma_dc_dequant_idct)(int16_t *output, int16_t *input, int qmul);         /* h264dsp.h:102-103 */
    void (*chroma_dc_dequant_idct)(int16_t *block, int qmul);                        /* h264dsp.h:104 (4:2:0) */
    void (*add_pixels8_clear)(uint8_t *dst, int16_t *block, ptrdiff_t stride);      /* h264dsp.h:107 */
    void (*add_pixels4_clear)(uint8_t *dst, int16_t *block, ptrdiff_t stride);      /* h264dsp.h:108 */
    /* the MBAFF forms of the vertical-edge filters (h264dsp.h:50-51,57-58,63-65,70-71): half as many lines per tc0 entry */
    void (*h_loop_filter_luma_mbaff)(uint8_t *pix, ptrdiff_t stride, int alpha, int beta, int8_t *tc0);
    void (*h_loop_filter_luma_mbaff_intra)(uint8_t *pix, ptrdiff_t stride, int alpha, int beta);
    void (*h_loop_filter_chroma_mbaff)(uint8_t *pix, ptrdiff_t stride, int alpha, int beta, int8_t *tc0);
    void (*h_loop_filter_chroma_mbaff_intra)(uint8_t *pix, ptrdiff_t stride, int alpha, int beta);
} FFHipH264DSPContext;
/**
 * Fills every member, the way ff_h264dsp_init_<arch>() does at the end of ff_h264dsp_init() (libavcodec/h264dsp.c:155-169):
 * *c arrives holding the C functions (or NULL members) — the hip faces REMEMBER them, and a call that cannot run on the device
 * (a HIP error, an argument outside the staged range, or FFHIP_FAULT=1 in the environment: the test hook) is answered by the
 * displaced C function instead of returning with dst untouched; ffhip_shim_fallbacks() counts such calls.  The same holds for
 * every ff_*_init_hip() below.  Returns 0, or FFHIP_ENOSYS / FFHIP_EINVAL leaving *c untouched.
 * bit_depth 8 / 9 / 10 / 12 / 14 (every depth the reference instantiates, h264dsp.c:135-147: above 8 bits samples are uint16_t and
 * coefficients int32_t, the depth is baked into the installed functions) and chroma_format_idc 0..2: for 4:2:2 the members the
 * reference switches are the 4:2:2 ones (h_loop_filter_chroma422[_intra / _mbaff], idct_add8_422, chroma422_dc_dequant_idct,
 * h264dsp.c:74-78,98-101,121-132).  4:4:4 (chroma_format_idc 3) uses the 4:2:0 table in the reference too.
 */
int ff_h264dsp_init_hip(FFHipH264DSPContext *c, int bit_depth, int chroma_format_idc);
/** Calls a host-pointer face answered through the displaced C pointer so far (and, when there was none to call, left undone:
 *  ffhip_last_error() has the member's name). */
long ffhip_shim_fallbacks(void);

/** IDCT kinds for the batch face; one kernel per kind == one reference function
 *  (libavcodec/h264idct_template.c:33,69,145,161). */
#define FFHIP_H264_IDCT4     0
#define FFHIP_H264_IDCT8     1
#define FFHIP_H264_IDCT4_DC  2
#define FFHIP_H264_IDCT8_DC  3
#define FFHIP_H264_ADD_PIXELS4_CLEAR 4   /* add_pixels4_clear / add_pixels8_clear: dst += block (8-bit wrap-around), block cleared */
#define FFHIP_H264_ADD_PIXELS8_CLEAR 5   /* (h264addpx_template.c:28-74; the lossless transform bypass) */
/**
 * n independent blocks: block i adds into dst_base + dst_offset[i] (bytes, rows `stride` apart) from
 * coefficients blocks + i*(16|64) int16 (transposed storage as the decoder leaves them), then the
 * coefficients are zeroed — exactly ff_h264_idct{,8}{,_dc}_add_8_c.  Destinations must not overlap.
 */
int ffhip_h264_idct_add_batch_dev(int kind, uint8_t *dst_base, ptrdiff_t stride, const int32_t *dst_offset,
                                  int16_t *blocks, int n, void *stream);
/**
 * Macroblock-level dispatchers over a batch of `nmb` macroblocks: idct_add16 / idct8_add4 /
 * idct_add16intra (libavcodec/h264idct_template.c:177-214).  MB m uses dst_base + mb_offset[m],
 * blocks + m*256 int16, nnzc + m*40 (indexed through scan8, libavcodec/h264_parse.h:40),
 * blockoffset[16] shared.  which: 0 add16, 1 idct8_add4, 2 add16intra.
 */
int ffhip_h264_idct_add_mb_batch_dev(int which, uint8_t *dst_base, ptrdiff_t stride, const int32_t *mb_offset,
                                     const int32_t *blockoffset16, int16_t *blocks, const uint8_t *nnzc,
                                     int nmb, void *stream);

/** idct_add8 (4:2:0, h264idct_template.c:216-228) over `nmb` macroblocks: blocks + m*768 int16 are the decoder's sl->mb (3 x 256),
 *  of which blocks 16..19 (Cb) and 32..35 (Cr) are used; nnzc + m*120 is the 15 x 8 non-zero-count cache (scan8 indexing);
 *  blockoffset48[i] as the decoder's block_offset[]; Cb / Cr block i of MB m lands at c?_base + mb_offset[m] + blockoffset48[i]. */
int ffhip_h264_idct_add8_batch_dev(uint8_t *cb_base, uint8_t *cr_base, ptrdiff_t stride, const int32_t *mb_offset,
                                   const int32_t *blockoffset48, int16_t *blocks, const uint8_t *nnzc, int nmb, void *stream);
/** luma_dc_dequant_idct (h264idct_template.c:259-293) for n macroblocks: input + m*in_pitch (16 DC values), output + m*out_pitch
 *  (the macroblock's 256 coefficients: the 16 results land on the blocks' DC positions), qmul[m].  Pitches in int16 units. */
int ffhip_h264_luma_dc_dequant_idct_batch_dev(int16_t *output, size_t out_pitch, const int16_t *input, size_t in_pitch,
                                              const int32_t *qmul, int n, void *stream);
/** chroma_dc_dequant_idct (4:2:0, h264idct_template.c:323-345), in place on blocks[block_offset[m] + {0,16,32,48}], qmul[m]. */
int ffhip_h264_chroma_dc_dequant_idct_batch_dev(int16_t *blocks, const int32_t *block_offset, const int32_t *qmul, int n, void *stream);

/** Loop-filter kinds == H264DSPContext members (libavcodec/h264dsp_template.c:104-330). */
#define FFHIP_H264_LF_V_LUMA          0
#define FFHIP_H264_LF_H_LUMA          1
#define FFHIP_H264_LF_V_CHROMA        2
#define FFHIP_H264_LF_H_CHROMA        3
#define FFHIP_H264_LF_V_LUMA_INTRA    4
#define FFHIP_H264_LF_H_LUMA_INTRA    5
#define FFHIP_H264_LF_V_CHROMA_INTRA  6
#define FFHIP_H264_LF_H_CHROMA_INTRA  7
/** One edge descriptor of the batch face: everything the reference call takes besides pix/stride. */
typedef struct FFHipH264Edge {
    int32_t offset;      /* pix = base + offset                                    */
    uint8_t kind;        /* FFHIP_H264_LF_*                                         */
    uint8_t alpha, beta; /* 8-bit tables top out at 255 / 18 (h264_loopfilter.c)   */
    uint8_t pad;
    int8_t  tc0[4];
} FFHipH264Edge;
/**
 * n edges whose touched pixels are pairwise DISJOINT (function-level batch, as checkasm's tiles).
 * Order-dependent frame deblocking is ffhip_h264_deblock_frame_dev().
 */
int ffhip_h264_loop_filter_batch_dev(uint8_t *base, ptrdiff_t stride, const FFHipH264Edge *edges, int n,
                                     void *stream);
/**
 * True frame-order luma deblocking of a whole picture (mb_w x mb_h macroblocks): for every MB in
 * raster order, vertical edges 0..3 left-to-right then horizontal edges 0..3 top-to-bottom, as
 * ff_h264_filter_mb() orders the h264dsp calls (libavcodec/h264_loopfilter.c:716).
 * edges[(mb*2 + dir)*4 + e] describes MB mb, dir 0 = vertical edges (h_loop_filter_*), 1 =
 * horizontal; `kind` selects normal (tc0 used) vs intra, alpha == 0 skips the edge; `offset` ignored.
 * Implemented as a 2-D wavefront over MBs — bit-exact with the serial order.
 */
int ffhip_h264_deblock_frame_dev(uint8_t *luma, ptrdiff_t stride, int mb_w, int mb_h,
                                 const FFHipH264Edge *edges, void *stream);
/** The same over `nframes` independent pictures in one launch (frame f at luma + f*frame_pitch, its edges at
 *  edges + f*mb_w*mb_h*8): the order inside a picture is serial, pictures run side by side. */
int ffhip_h264_deblock_frames_dev(uint8_t *luma, size_t frame_pitch, int nframes, ptrdiff_t stride, int mb_w, int mb_h,
                                  const FFHipH264Edge *edges, void *stream);

/**
 * The same for one 4:2:0 CHROMA plane (8x8 samples per macroblock; call once for Cb and once for Cr — their alpha / beta / tc0
 * come from different QPs): per MB the vertical edges at x = 0 and 4, then the horizontal ones at y = 0 and 4, as
 * filter_mb_dir() filters chroma on the even luma edges (libavcodec/h264_loopfilter.c:644-700).
 * edges[((f * mb_h * mb_w + mb) * 2 + dir) * 2 + e]; kinds FFHIP_H264_LF_*_CHROMA[_INTRA]; alpha == 0 skips an edge.
 * plane, stride and frame_pitch must be 4-byte aligned.
 */
int ffhip_h264_deblock_frames_chroma_dev(uint8_t *plane, size_t frame_pitch, int nframes, ptrdiff_t stride, int mb_w, int mb_h,
                                         const FFHipH264Edge *edges, void *stream);

/**
 * Both of the above at the depths H264DSPContext is instantiated for (libavcodec/h264dsp.c:135-147; bit_depth 8, 9, 10, 12 or 14): above
 * 8 bits the samples are uint16_t (stride and frame_pitch stay in BYTES), alpha / beta / tc0 of the edge records stay in the 8-bit
 * units the decoder passes and are scaled inside as h264dsp_template.c:104-330 scales them (alpha, beta << (depth - 8); luma tc0 *
 * (1 << (depth - 8)); chroma ((tc0 - 1) << (depth - 8)) + 1).  chroma != 0: one 4:2:0 chroma plane per frame.  Above 8 bits plane,
 * stride, frame_pitch and edges must be 16-byte aligned (the byte / dword paths are 8-bit kernels): FFHIP_EINVAL otherwise.
 */
int ffhip_h264_deblock_frames_dev_hbd(int bit_depth, int chroma, uint8_t *plane, size_t frame_pitch, int nframes, ptrdiff_t stride, int mb_w,
                                      int mb_h, const FFHipH264Edge *edges, void *stream);

/**
 * The batched faces above at ANY depth the reference instantiates (bit_depth 8 / 9 / 10 / 12 / 14), plus the members that exist
 * only here: MBAFF and 4:2:2.  Above 8 bits samples are uint16_t and coefficients int32_t (libavcodec/bit_depth_template.c);
 * strides and offsets stay in BYTES, coefficient pitches in coefficients.  Bit-exact restatement of the reference's templates
 * (h264idct_template.c, h264addpx_template.c, h264dsp_template.c, h264qpel_template.c, h264chroma_template.c) with the depth as a
 * kernel argument; the 8-bit kernels above remain the fast path of 8-bit 4:2:0 streams.
 *   idct_add:     kinds FFHIP_H264_IDCT4 .. ADD_PIXELS8_CLEAR, n x (16 | 64) coefficients
 *   idct_mb:      which 0 idct_add16, 1 idct8_add4, 2 idct_add16intra (one plane: dst2 unused), 3 idct_add8 (4:2:0: Cb = dst_base,
 *                 Cr = dst2, 768 coefficients and a 15 x 8 cache per macroblock), 4 idct_add8_422
 *   dc_dequant:   which 0 luma_dc_dequant_idct (input + m*in_pitch -> the DC positions of output + m*out_pitch), 1 chroma_dc_dequant_idct,
 *                 2 chroma422_dc_dequant_idct (in place on output + block_offset[m])
 *   loop_filter:  FFHipH264Edge.pad = lines per tc0 entry (0: the plain member: luma 4, chroma 2; MBAFF: luma 2, chroma 1;
 *                 4:2:2 h_loop_filter_chroma422: 4, its MBAFF form 2); alpha / beta at the 8-bit scale, as the decoder's tables hold them
 */
int ffhip_h264_idct_add_batch_dev_hbd(int bit_depth, int kind, uint8_t *dst_base, ptrdiff_t stride, const int32_t *dst_offset, int16_t *blocks,
                                      int n, void *stream);
int ffhip_h264_idct_mb_batch_dev_hbd(int bit_depth, int which, uint8_t *dst_base, uint8_t *dst2, ptrdiff_t stride, const int32_t *mb_offset,
                                     const int32_t *blockoffset, int16_t *blocks, const uint8_t *nnzc, int nmb, void *stream);
int ffhip_h264_dc_dequant_batch_dev_hbd(int bit_depth, int which, int16_t *output, size_t out_pitch, const int16_t *input, size_t in_pitch,
                                        const int32_t *block_offset, const int32_t *qmul, int n, void *stream);
int ffhip_h264_loop_filter_batch_dev_hbd(int bit_depth, uint8_t *base, ptrdiff_t stride, const FFHipH264Edge *edges, int n, void *stream);

/* ------------------------------------------------------------------------------------------ */
/* libavcodec: h264qpel                                                                       */
/* ------------------------------------------------------------------------------------------ */
/** qpel_mc_func (libavcodec/qpeldsp.h:65-67) and H264QpelContext (libavcodec/h264qpel.h:27-30). */
typedef void (*ffhip_qpel_mc_func)(uint8_t *dst, const uint8_t *src, ptrdiff_t stride);
typedef struct FFHipH264QpelContext {
    ffhip_qpel_mc_func put_h264_qpel_pixels_tab[3][16];
    ffhip_qpel_mc_func avg_h264_qpel_pixels_tab[3][16];
} FFHipH264QpelContext;
int ff_h264qpel_init_hip(FFHipH264QpelContext *c, int bit_depth);

/** One motion-compensated block of the batch face (what mc_dir_part() passes, h264_mb.c:206-250).
 *
 *  Edge emulation (round 4).  The reference allocates no border around a picture: when the 6-tap footprint of a block leaves the
 *  reference picture, mc_dir_part() copies it through h->vdsp.emulated_edge_mc() (libavcodec/videodsp_template.c:24-100: samples
 *  outside the picture repeat the nearest one inside) and filters from that copy (h264_mb.c:229-247 luma, :297-317 chroma).  A record
 *  says the same with FFHIP_MC_EMU in `flags`: src_offset then addresses sample (0, 0) of the reference PICTURE (pic->data[plane] -
 *  ref base), (src_x, src_y) is the integer-sample position of the block's origin in that picture — any value, also far outside —
 *  and the kernel reads footprint sample (x, y) at row clamp(y, 0, pic_h - 1), column clamp(x, 0, pic_w - 1): emulated_edge_mc's
 *  result, sample for sample (its whole-block-outside cases included, :37-56), for the 21 x 21 window the decoder copies as for any
 *  other.  pic_w / pic_h are those of the plane (luma: 16 * mb_width, 16 * mb_height; 4:2:0 chroma: half), handed to the *_pic
 *  entry points (the picture object knows them).  A record without the flag is read unclamped, as before. */
#define FFHIP_MC_EMU 1
typedef struct FFHipQpelBlock {
    int32_t dst_offset;  /* into dst plane                                            */
    int32_t src_offset;  /* into ref plane: integer-pel position of the block origin; FFHIP_MC_EMU: of the reference picture's (0, 0) */
    uint8_t mcxy;        /* luma_xy = (mx&3) + ((my&3)<<2)                            */
    uint8_t size_idx;    /* 0: 16x16, 1: 8x8, 2: 4x4                                   */
    uint8_t avg;         /* 0 put, 1 avg                                              */
    uint8_t flags;       /* FFHIP_MC_EMU                                              */
    int16_t src_x, src_y;/* FFHIP_MC_EMU: block origin in the reference picture, integer samples (else unused, 0) */
} FFHipQpelBlock;        /* sizeof == 16 */
/** n blocks, one stride for src & dst (as qpel_mc_func); dst blocks must not overlap. src must be
 *  readable 2 px left/up and 3 px right/down of each block (records flagged FFHIP_MC_EMU are not honoured here: use the _pic form). */
int ffhip_h264_qpel_batch_dev(uint8_t *dst, const uint8_t *src, ptrdiff_t stride,
                              const FFHipQpelBlock *blocks, int n, void *stream);
/** The same with the reference pictures' dimensions (samples of this plane): records flagged FFHIP_MC_EMU read their footprint
 *  through clamped coordinates, and only samples inside a reference picture are ever touched for them. */
int ffhip_h264_qpel_batch_dev_pic(uint8_t *dst, const uint8_t *src, ptrdiff_t stride, int pic_w, int pic_h,
                                  const FFHipQpelBlock *blocks, int n, void *stream);

/* ------------------------------------------------------------------------------------------ */
/* libavcodec: h264chroma + explicit weighted prediction (the first "next" row) */
/* ------------------------------------------------------------------------------------------ */
/** h264_chroma_mc_func and H264ChromaContext (libavcodec/h264chroma.h:25-32): tab index 0 = 8 wide, 1 = 4, 2 = 2
 *  (ff_h264chroma_init, libavcodec/h264chroma.c:38-52); x, y are the eighth-pel fractions. */
typedef void (*ffhip_h264_chroma_mc_func)(uint8_t *dst, const uint8_t *src, ptrdiff_t srcStride, int h, int x, int y);
typedef struct FFHipH264ChromaContext {
    ffhip_h264_chroma_mc_func put_h264_chroma_pixels_tab[4];
    ffhip_h264_chroma_mc_func avg_h264_chroma_pixels_tab[4];
} FFHipH264ChromaContext;
/** Fills entries 0..2 (entry 3, the 1-wide VP9 helper, is left untouched).  8-bit only. */
int ff_h264chroma_init_hip(FFHipH264ChromaContext *c, int bit_depth);

/** One chroma MC call of the batch face (what mc_dir_part() passes to chroma_op, h264_mb.c:270-300).  FFHIP_MC_EMU as for
 *  FFHipQpelBlock: the (w + 1) x (h + 1) samples are read at clamped coordinates of the reference picture's chroma plane
 *  (emulated_edge_mc(…, 9, 8 * chroma_idc + 1, mx >> 3, my >> ysh, pic_width >> 1, pic_height >> 1), h264_mb.c:297-317). */
typedef struct FFHipChromaBlock {
    int32_t dst_offset, src_offset;
    uint8_t w_idx;       /* 0: 8 wide, 1: 4, 2: 2 */
    uint8_t h;           /* rows, <= 16            */
    uint8_t x, y;        /* eighth-pel fractions   */
    uint8_t avg, flags;  /* flags: FFHIP_MC_EMU    */
    int16_t src_x, src_y;/* FFHIP_MC_EMU: block origin in the reference picture's plane, integer samples */
    int16_t pad;
} FFHipChromaBlock;      /* sizeof == 20 */
int ffhip_h264_chroma_mc_batch_dev(uint8_t *dst, const uint8_t *src, ptrdiff_t stride, const FFHipChromaBlock *blocks, int n,
                                   void *stream);
int ffhip_h264_chroma_mc_batch_dev_pic(uint8_t *dst, const uint8_t *src, ptrdiff_t stride, int pic_w, int pic_h,
                                       const FFHipChromaBlock *blocks, int n, void *stream);

/** h264_weight_func / h264_biweight_func (libavcodec/h264dsp.h:31-37) and the two tables of H264DSPContext
 *  (:44-45): index 0 = 16 wide, 1 = 8, 2 = 4, 3 = 2 (libavcodec/h264dsp.c:100-107). */
typedef void (*ffhip_h264_weight_func)(uint8_t *block, ptrdiff_t stride, int height, int log2_denom, int weight, int offset);
typedef void (*ffhip_h264_biweight_func)(uint8_t *dst, uint8_t *src, ptrdiff_t stride, int height, int log2_denom, int weightd,
                                         int weights, int offset);
typedef struct FFHipH264WeightContext {
    ffhip_h264_weight_func   weight_pixels_tab[4];
    ffhip_h264_biweight_func biweight_pixels_tab[4];
} FFHipH264WeightContext;
int ff_h264dsp_weight_init_hip(FFHipH264WeightContext *c, int bit_depth);

/** One weight (bi == 0: dst only, weightd = the weight) or biweight call of the batch face. */
typedef struct FFHipWeightBlock {
    int32_t dst_offset, src_offset;
    uint8_t w_idx;       /* 0: 16 wide, 1: 8, 2: 4, 3: 2 */
    uint8_t height, log2_denom, bi;
    int16_t weightd, weights, offset, pad;
} FFHipWeightBlock;
int ffhip_h264_weight_batch_dev(uint8_t *dst, const uint8_t *src, ptrdiff_t stride, const FFHipWeightBlock *blocks, int n,
                                void *stream);
/** qpel / chroma MC / weighted prediction at bit_depth 8 / 9 / 10 / 12 / 14 (16-bit samples above 8; offsets and stride in bytes):
 *  h264qpel.c:87-103, h264chroma.c:38-52, h264dsp.c:102-109 at the depth. */
int ffhip_h264_qpel_batch_dev_hbd(int bit_depth, uint8_t *dst, const uint8_t *src, ptrdiff_t stride, const FFHipQpelBlock *blocks, int n,
                                  void *stream);
int ffhip_h264_chroma_mc_batch_dev_hbd(int bit_depth, uint8_t *dst, const uint8_t *src, ptrdiff_t stride, const FFHipChromaBlock *blocks, int n,
                                       void *stream);
int ffhip_h264_weight_batch_dev_hbd(int bit_depth, uint8_t *dst, const uint8_t *src, ptrdiff_t stride, const FFHipWeightBlock *blocks, int n,
                                    void *stream);
/** ... and their FFHIP_MC_EMU forms: pic_w / pic_h in samples of the plane. */
int ffhip_h264_qpel_batch_dev_hbd_pic(int bit_depth, uint8_t *dst, const uint8_t *src, ptrdiff_t stride, int pic_w, int pic_h,
                                      const FFHipQpelBlock *blocks, int n, void *stream);
int ffhip_h264_chroma_mc_batch_dev_hbd_pic(int bit_depth, uint8_t *dst, const uint8_t *src, ptrdiff_t stride, int pic_w, int pic_h,
                                           const FFHipChromaBlock *blocks, int n, void *stream);

/* ------------------------------------------------------------------------------------------ */
/* libavcodec: caller-side batching for the H.264 macroblock loop           */
/* ------------------------------------------------------------------------------------------ */
/**
 * A picture's worth of per-block dsp calls, recorded on the host while the decoder parses the picture and run as a handful of
 * launches.  The record functions take the operands hl_decode_mb() passes to the dsp pointers (libavcodec/h264_mb_template.c:41-270,
 * h264_mb.c:206-420,612-800) and ff_h264_filter_mb() computes (h264_loopfilter.c:716); flush() runs, per plane,
 *     MC put -> picture | MC put -> bi-prediction scratch | MC avg -> picture | weight / biweight | IDCT + add |
 *     intra macroblocks (reconstruction wavefront, all three planes) | deblock (decoder order)
 * ffhip_h264_picture_create(): 4:2:0, 8 bits; the other depths and chroma formats: ffhip_h264_picture_create_hbd / _fmt below.  All planes
 * of the picture, of the references and the scratch share one stride per plane (qpel_mc_func has one).
 * ref[pl] is the base the blocks' src_offset counts from — typically the decoded-picture-buffer allocation, so that one base
 * reaches every reference picture.  One object serves one stream; begin() starts the next picture (flush() does not clear).
 */
typedef struct FFHipH264Picture FFHipH264Picture;
#define FFHIP_H264_MC_PUT 0   /* put into the picture                                              */
#define FFHIP_H264_MC_TMP 1   /* put into the bi-prediction scratch plane (sl->bipred_scratchpad)  */
#define FFHIP_H264_MC_AVG 2   /* avg onto the picture: the second list of an unweighted bi-prediction */
int  ffhip_h264_picture_create(FFHipH264Picture **p, int mb_w, int mb_h);
/** The same object for a High 10 / High 4:2:0 picture of bit_depth 9 / 10 / 12 / 14 (8: == ffhip_h264_picture_create): planes hold
 *  uint16_t samples, offsets and strides stay in BYTES (as the decoder's linesize / block_offset << pixel_shift), the blocks handed to
 *  idct_add() hold int32_t coefficients (dctcoef, libavcodec/bit_depth_template.c:39-50; sl->mb as the decoder keeps it), edge records
 *  keep alpha / beta / tc0 at the 8-bit scale (the kernels scale them as h264dsp_template.c:108-110 does), intra macroblocks hand
 *  over sl->mb / sl->mb_luma_dc / sl->intra_pcm_ptr as ffhip_h264_intra_pack_hbd() describes.  Planes and strides 8-byte aligned. */
int  ffhip_h264_picture_create_hbd(FFHipH264Picture **p, int mb_w, int mb_h, int bit_depth);
/** The same object for a picture of sps->chroma_format_idc 1 (4:2:0: == ffhip_h264_picture_create_hbd), 2 (4:2:2) or 3 (4:4:4) (round 4),
 *  at any of the depths.
 *  A 4:2:2 picture (hl_decode_mb() with block_h = 16, hl_motion_422: h264_mb_template.c:41-262,172) keeps the 4:2:0 record calls with 8 x 16
 *  chroma: ffhip_h264_picture_mc_chroma() blocks of up to 16 rows (the decoder's `height`, y fraction (my << 1) & 7: h264_mb.c:289-317), the
 *  8-wide weights over 16 rows, ffhip_h264_picture_idct_mb() which 3 = ff_h264_idct_add8_422 (eight blocks per plane after the decoder's own
 *  chroma422_dc_dequant_idct), SIX edge records per macroblock and chroma plane (the vertical edges x = 0, 4 — h_loop_filter_chroma422, 16
 *  lines, tc0 per 4 — then the horizontal ones y = 0, 4, 8, 12); ffhip_h264_picture_intra_mb() is the same call (qmul[1], qmul[2] =
 *  dequant4_coeff[1 + p][chroma_qp[p] + 3][0]; 512 I_PCM fields) and becomes a luma-only record plus an FFHipH264IntraC422.  flush() runs the
 *  luma plane through the luma kernels and the chroma planes' two dependency chains through plain kernels of their own, beside the luma
 *  ones on the object's second stream (ffhip_h264_pictures_flush(): the chroma planes of all pictures side by side in one launch); Cb and Cr
 *  share a stride when the picture has intra macroblocks.
 *  A 4:4:4 picture is what hl_decode_mb_444() (libavcodec/h264_mb_template.c:256-362) makes of it: three planes of luma geometry,
 *  all reconstructed by the LUMA members —
 *    prediction: qpix_op[luma_xy] on dest_cb / dest_cr with the luma vector (mc_dir_part(), h264_mb.c:262-288), weights of the luma width
 *                (mc_part_weighted(), :362-366): ffhip_h264_picture_mc_luma_plane() / ffhip_h264_picture_weight() with plane 1 / 2;
 *                ffhip_h264_picture_mc_chroma() is refused;
 *    residual:   idct_add16 / idct8_add4 per plane on sl->mb + 256 p with the cache rows of plane p (hl_decode_mb_idct_luma(..., p),
 *                h264_mb.c:735-800): ffhip_h264_picture_idct_mb() which 0 / 1 with plane p; which 3 (idct_add8) is refused;
 *    intra:      hl_decode_mb_predict_luma(..., p) for p = 0, 1, 2 with ONE set of prediction modes (h264_mb.c:614-733):
 *                ffhip_h264_picture_intra_mb() takes the decoder's arrays as for 4:2:0 — mb_luma_dc = sl->mb_luma_dc[0], the [3][16 * 2]
 *                int16 array as it stands (plane p's DCs 32 int16 further at every depth, libavcodec/h264dec.h), pcm = the 768 fields —
 *                and splits the macroblock into three luma-only records; flush() runs the three planes' wavefronts side by side in one
 *                launch;
 *    deblocking: filter_mb_edgev / filter_mb_edgeh on img_cb / img_cr (h264_loopfilter.c:601-703): ffhip_h264_picture_deblock_mb() takes 8
 *                luma-kind edge records for every plane.
 *  The three planes share one stride when the picture carries intra macroblocks (the decoder's linesize == uvlinesize there).
 *  chroma_format_idc 0 (monochrome) makes the 4:2:0 object: the decoder reconstructs such a picture as 4:2:0 with mid-grey chroma through the
 *  ordinary members (h264_mb_template.c:112-148); an I_PCM macroblock's record carries mid-grey chroma fields (the FFmpeg-side recorder
 *  appends them). */
int  ffhip_h264_picture_create_fmt(FFHipH264Picture **p, int mb_w, int mb_h, int bit_depth, int chroma_format_idc);
void ffhip_h264_picture_free(FFHipH264Picture **p);
void ffhip_h264_picture_begin(FFHipH264Picture *p);
/** mc_dir_part(): qpix_op[luma_xy] and chroma_op (h264_mb.c:206-300); blk->avg is set from `stage`. */
int  ffhip_h264_picture_mc_luma(FFHipH264Picture *p, int stage, const FFHipQpelBlock *blk);
/** ... on plane 0 (== mc_luma), or on Cb / Cr of a 4:4:4 picture (offsets into that plane and that plane of the references). */
int  ffhip_h264_picture_mc_luma_plane(FFHipH264Picture *p, int plane, int stage, const FFHipQpelBlock *blk);
int  ffhip_h264_picture_mc_chroma(FFHipH264Picture *p, int plane /* 1 Cb, 2 Cr */, int stage, const FFHipChromaBlock *blk);
/** mc_part_weighted(): weight_op / biweight_op (h264_mb.c:340-420); a biweight's src_offset addresses the scratch plane. */
int  ffhip_h264_picture_weight(FFHipH264Picture *p, int plane, const FFHipWeightBlock *blk);
/** idct_add / idct8_add / idct_dc_add / idct8_dc_add: the 16 / 64 coefficients are copied and the caller's block is consumed
 *  exactly as the dsp function does (zeroed; dc forms: block[0] = 0). */
int  ffhip_h264_picture_idct_add(FFHipH264Picture *p, int plane, int kind, int32_t dst_offset, int16_t *block);
/** The macroblock-level residual members of an INTER macroblock as hl_decode_mb() calls them (libavcodec/h264_mb.c:780-797,
 *  h264_mb_template.c:254-257): which 0 = idct_add16, 1 = idct8_add4 (plane's dst_offset[0]), 3 = idct_add8 (4:2:0; dst_offset[0] Cb,
 *  [1] Cr, `plane` ignored; block = sl->mb).  Expanded on the host into idct_add records the way the dsp functions dispatch
 *  (h264idct_template.c:176-228: nnz == 1 with a DC -> the dc form, …); block_offset is h->block_offset (bytes), nnzc the pointer the
 *  member is handed (sl->non_zero_count_cache); `block` is consumed as those functions consume it. */
int  ffhip_h264_picture_idct_mb(FFHipH264Picture *p, int which, int plane, const int32_t dst_offset[2], const int *block_offset, int16_t *block,
                                const uint8_t *nnzc);
/** ff_h264_filter_mb(): the macroblock's edge records, 8 for luma ((dir * 4 + e)), 4 for a chroma plane ((dir * 2 + e)); 4:4:4: 8 for
 *  every plane, luma kinds. */
int  ffhip_h264_picture_deblock_mb(FFHipH264Picture *p, int plane, int mb_x, int mb_y, const FFHipH264Edge *edges);
/**
 * An INTRA macroblock: what hl_decode_mb() does for IS_INTRA(mb_type) (libavcodec/h264_mb_template.c:137-262 with
 * hl_decode_mb_predict_luma / hl_decode_mb_idct_luma, libavcodec/h264_mb.c:612-760) as ONE record.  Intra prediction reads the
 * reconstructed, not yet deblocked samples of the left / top-left / top / top-right macroblocks and chains through the residual
 * add from block to block, so these records are not a batch of independent calls: flush() runs them as a reconstruction
 * WAVEFRONT (one wave per macroblock row; an intra macroblock starts when the row above has finished the macroblock up and to the
 * right) after the inter macroblocks' prediction and residual stages and before deblocking — the order of the reference's
 * data flow, whose in-loop filter runs behind reconstruction on samples intra prediction never sees (xchg_mb_border,
 * h264_mb.c:528-597).  Fields are the decoder's H264SliceContext state of the macroblock.  Frame macroblocks, or the field macroblocks of
 * a field picture whose object was made for the field (mb_y = the row inside the field); 4:2:2 / 4:4:4: ffhip_h264_picture_create_fmt();
 * the lossless transform bypass (a macroblock with qscale 0 in a stream with sps->transform_bypass; round 6, 8 bits, 4:2:0 and 4:4:4): the
 * caller sets FFHIP_H264_INTRA_BYPASS (and _DPCM under profile_idc 244) in `flags` — see below.
 */
#define FFHIP_H264_INTRA_16x16 0   /* IS_INTRA16x16(mb_type)                                                         */
#define FFHIP_H264_INTRA_4x4   1   /* IS_INTRA4x4(mb_type) && !IS_8x8DCT(mb_type)                                    */
#define FFHIP_H264_INTRA_8x8   2   /* IS_INTRA4x4(mb_type) &&  IS_8x8DCT(mb_type): pred8x8l + the 8x8 transform      */
#define FFHIP_H264_INTRA_PCM   3   /* IS_INTRA_PCM(mb_type): sl->intra_pcm_ptr's 384 bytes                            */
#define FFHIP_H264_INTRA_LUMA_DC 1 /* flags, set by ffhip_h264_picture_intra_mb() from the cache: nnz[scan8[LUMA_DC_BLOCK_INDEX]] */
#define FFHIP_H264_INTRA_CB_DC   2 /*   nnz[scan8[CHROMA_DC_BLOCK_INDEX + 0]] */
#define FFHIP_H264_INTRA_CR_DC   4 /*   nnz[scan8[CHROMA_DC_BLOCK_INDEX + 1]] */
/* flags the CALLER sets (every other bit of `flags` zero on entry): the macroblock is decoded with the transform bypassed
 * (hl_decode_mb()'s transform_bypass: sl->qscale == 0 && sps->transform_bypass, h264_mb_template.c:51) — its "coefficients" are residual
 * samples, added to the prediction as add_pixels4 / 8_clear add them (modulo 256, h264addpx_template.c:30-74), no DC transforms;
 * _DPCM (sps->profile_idc == 244): blocks predicted vertically / horizontally take the pred*_add forms (h264pred_template.c:1067-1330,
 * h264_mb.c:638-672,745-750, h264_mb_template.c:198-207): every sample is its neighbour along the direction plus its residual — the packer
 * turns those residuals into running sums along the direction, which makes them ordinary residuals of the V / H prediction. */
#define FFHIP_H264_INTRA_BYPASS  8
#define FFHIP_H264_INTRA_DPCM    16
typedef struct FFHipH264IntraMB {
    int16_t  mb_x, mb_y;
    uint8_t  type;            /* FFHIP_H264_INTRA_*                                                                  */
    uint8_t  pred16;          /* sl->intra16x16_pred_mode (after ff_h264_check_intra_pred_mode)                      */
    uint8_t  chroma_pred;     /* sl->chroma_pred_mode (likewise)                                                     */
    uint8_t  cbp;             /* sl->cbp & 0x3f                                                                      */
    uint16_t topleft_avail;   /* sl->topleft_samples_available  (h264_mvpred.h:599-639)                              */
    uint16_t topright_avail;  /* sl->topright_samples_available                                                      */
    uint8_t  pred4[16];       /* sl->intra4x4_pred_mode_cache[scan8[i]], i = 0..15 (8x8: entries 0, 4, 8, 12)        */
    int32_t  qmul[3];         /* pps->dequant4_coeff[0][qscale][0], [1][chroma_qp[0]][0], [2][chroma_qp[1]][0]       */
    /* ---- filled in by ffhip_h264_picture_intra_mb() ---- */
    uint8_t  flags;           /* in: 0, or FFHIP_H264_INTRA_BYPASS [| _DPCM]; out: + FFHIP_H264_INTRA_LUMA_DC | _CB_DC | _CR_DC */
    uint8_t  pad[3];          /* pad[0], out: how many regions of the macroblock (a luma block, the 16x16 block, a chroma plane) were DPCM-coded */
    uint8_t  nnz[24];         /* non_zero_count_cache[scan8[i]]: luma i = 0..15, Cb 16..19 at [16..19], Cr 32..35 at [20..23] */
    int32_t  coef;            /* the macroblock's run in the picture's packed coefficient array (int16 units)         */
    uint32_t blocks;          /* which blocks the run holds, in this order: bit i luma block i (8x8: bits 0, 4, 8, 12, 64
                                 coefficients each), bit 16 + k Cb block k, bit 20 + k Cr block k                     */
    int16_t  luma_dc[16];     /* sl->mb_luma_dc[0] (Intra16x16)                                                      */
} FFHipH264IntraMB;           /* sizeof == 108 */
/** The chroma planes of an intra macroblock of a 4:2:2 picture, as ffhip_h264_picture_intra_mb() records them on a picture made with
 *  chroma_format_idc 2 (its luma is a luma-only FFHipH264IntraMB): pred8x16 by chroma_pred_mode on both planes, chroma422_dc_dequant_idct
 *  + idct_add8_422 when cbp & 0x30 (libavcodec/h264_mb_template.c:151-262, h264idct_template.c:230-252,295-321).  Exported through
 *  ffhip_h264_picture_lists(). */
typedef struct FFHipH264IntraC422 {
    int16_t  mb_x, mb_y;
    uint8_t  type;        /* FFHIP_H264_INTRA_PCM: the run holds 2 x 128 samples; anything else: prediction + residual */
    uint8_t  chroma_pred; /* sl->chroma_pred_mode (after ff_h264_check_intra_pred_mode)                                */
    uint8_t  cbp;         /* sl->cbp & 0x30                                                                            */
    uint8_t  flags;       /* bit p: nnz[scan8[CHROMA_DC_BLOCK_INDEX + p]] — the plane's DC block is coded             */
    int32_t  qmul[2];     /* pps->dequant4_coeff[1 + p][sl->chroma_qp[p] + 3][0]                                       */
    uint16_t full;        /* bit 8 p + k: block k of plane p has a non-zero count (idct_add; else idct_dc_add if its DC != 0) */
    uint16_t blocks;      /* bit 8 p + k: block k of plane p travels in the run (16 coefficients each, in bit order)   */
    int32_t  coef;        /* the run's start in the packed coefficient array (int16 units)                             */
    int32_t  pad[2];
} FFHipH264IntraC422;     /* sizeof == 32 */
/** Records one intra macroblock: *mb with the fields above `flags` set.  non_zero_count_cache: the decoder's 15 x 8 cache
 *  (scan8 indexing, libavcodec/h264_parse.h:40-57).  mb: sl->mb (3 x 256 int16; Cb at 256, Cr at 512), CONSUMED the way the dsp
 *  functions hl_decode_mb() calls consume it (blocks zeroed after idct_add, [0] after idct_dc_add).  mb_luma_dc: sl->mb_luma_dc[0]
 *  (Intra16x16 with a coded DC block, else may be NULL).  pcm: sl->intra_pcm_ptr (FFHIP_H264_INTRA_PCM, else NULL). */
int  ffhip_h264_picture_intra_mb(FFHipH264Picture *p, const FFHipH264IntraMB *mb_desc, const uint8_t *non_zero_count_cache,
                                 int16_t *mb, const int16_t *mb_luma_dc, const uint8_t *pcm);
/** The host side of the record alone (no device involved; ffhip_h264_picture_intra_mb() is this + an append): fills the fields
 *  from `flags` down and appends the macroblock's coefficient run to coefs[*ncoefs ...] (capacity `cap` int16; a run is at most
 *  391 of them), advancing *ncoefs.  FFHIP_ENOMEM when the run does not fit. */
int  ffhip_h264_intra_pack(FFHipH264IntraMB *rec, const uint8_t *non_zero_count_cache, int16_t *mb, const int16_t *mb_luma_dc,
                           const uint8_t *pcm, int16_t *coefs, int32_t *ncoefs, int32_t cap);
/** The same at bit_depth 9 / 10 / 12 / 14 (8: == ffhip_h264_intra_pack; what ffhip_h264_picture_intra_mb() runs on a picture made by
 *  ffhip_h264_picture_create_hbd()): mb and mb_luma_dc are the decoder's arrays AS THEY STAND at that depth — 3 x 256 and 16 int32
 *  (dctcoef, libavcodec/bit_depth_template.c:39-50) behind the int16_t pointers sl->mb / sl->mb_luma_dc are declared with; pcm is
 *  sl->intra_pcm_ptr, 384 bit_depth-bit fields (h264_mb_template.c:100-131), unpacked here into uint16_t samples.  coefs, *ncoefs, cap
 *  and rec->coef keep counting int16 entries (a run is at most 807 of them); an Intra16x16 macroblock's sixteen luma DCs lead its run
 *  (rec->luma_dc stays zero: it cannot hold them). */
int  ffhip_h264_intra_pack_hbd(int bit_depth, FFHipH264IntraMB *rec, const uint8_t *non_zero_count_cache, int16_t *mb,
                               const int16_t *mb_luma_dc, const uint8_t *pcm, int16_t *coefs, int32_t *ncoefs, int32_t cap);
/** One PLANE of a 4:4:4 macroblock as a luma-only record (what ffhip_h264_picture_intra_mb() does three times on a 4:4:4 picture): the
 *  arguments are plane p's slices of the decoder's arrays — non_zero_count_cache + 5 * 8 * p (scan8[i + 16 p] == scan8[i] + 40 p, the DC
 *  entry scan8[LUMA_DC_BLOCK_INDEX + p] == 40 p), sl->mb + 256 p dctcoef, sl->mb_luma_dc[p], the plane's 256 I_PCM fields — and
 *  rec->qmul[0] = pps->dequant4_coeff[p][p ? sl->chroma_qp[p - 1] : sl->qscale][0] (hl_decode_mb_predict_luma, h264_mb.c:626,712).  The
 *  record's chroma fields are cleared; an I_PCM run keeps its 4:2:0 length (the last third zero). */
int  ffhip_h264_intra_pack_plane(int bit_depth, FFHipH264IntraMB *rec, const uint8_t *non_zero_count_cache, int16_t *mb, const int16_t *mb_luma_dc,
                                 const uint8_t *pcm, int16_t *coefs, int32_t *ncoefs, int32_t cap);
/** The intra reconstruction wavefront alone, on records already in device memory (what flush() launches): recs sorted by
 *  (mb_y, mb_x), row_start[mb_h + 1] indexes them by macroblock row, coefs is the packed coefficient array.  Planes and strides
 *  4-byte aligned.  Asynchronous on `stream`; a lost hand-off is reported by the next flush / ffhip_stream_synchronize. */
int  ffhip_h264_intra_frame_dev(uint8_t *y, uint8_t *cb, uint8_t *cr, ptrdiff_t stride_y, ptrdiff_t stride_c, int mb_w, int mb_h,
                                const FFHipH264IntraMB *recs, const int32_t *row_start, const int16_t *coefs, void *stream);
/** The same on uint16_t samples at bit_depth 9 / 10 / 12 / 14 (records and runs as ffhip_h264_intra_pack_hbd() leaves them; strides in
 *  bytes; planes and strides 8-byte aligned). */
int  ffhip_h264_intra_frame_dev_hbd(int bit_depth, uint8_t *y, uint8_t *cb, uint8_t *cr, ptrdiff_t stride_y, ptrdiff_t stride_c, int mb_w,
                                    int mb_h, const FFHipH264IntraMB *recs, const int32_t *row_start, const int16_t *coefs, void *stream);
/** N pictures' wavefronts in ONE launch (round 4): pictures of one geometry and depth, each with its own planes, sorted records, row
 *  starts and coefficient runs (all device pointers, as for the call above).  A wavefront's latency is its dependency chain (mb_w +
 *  2 mb_h macroblock steps: 2.6 ms for a 1080p I-picture) and one picture occupies 68 of the chip's 8,192 wave slots — a decoder that
 *  holds several pictures (frame threads, an all-intra stream) gets them reconstructed side by side for the latency of one.  Launches
 *  of 32 pictures; asynchronous on `stream`. */
typedef struct FFHipH264IntraPic {
    uint8_t *y, *cb, *cr;
    const FFHipH264IntraMB *recs;
    const int32_t *row_start;
    const int16_t *coefs;
} FFHipH264IntraPic;
int  ffhip_h264_intra_frames_dev(int bit_depth, int npics, const FFHipH264IntraPic *pics /* host array */, ptrdiff_t stride_y,
                                 ptrdiff_t stride_c, int mb_w, int mb_h, void *stream);
/** The same where every entry is ONE PLANE of a 4:4:4 picture: y = the plane, recs / row_start / coefs = that plane's luma-only records
 *  (ffhip_h264_intra_pack_plane); cb / cr are not touched (pass y). */
int  ffhip_h264_intra_planes_dev(int bit_depth, int nplanes, const FFHipH264IntraPic *planes /* host array */, ptrdiff_t stride, int mb_w, int mb_h,
                                 void *stream);
/** What has been recorded since begin(), as it stands: the lists flush() copies to the device, by the stage and plane it runs them in
 *  (pointers into the object, valid until the next record call or begin(); a list with no entry may be NULL).  Recording needs no HIP
 *  device — ffhip_h264_picture_create*() succeed without one and only flush() returns FFHIP_ENOSYS then — so a decoder's macroblock loop
 *  over the recording members can be checked on any machine (tests/test_h264_picture_cpu.py runs the lists through the oracle's dsp
 *  functions in flush()'s order and compares with the reference's own decode).
 *    qpel[plane][stage]  luma-table MC (4:2:0: plane 0 only), stage FFHIP_H264_MC_PUT / _TMP / _AVG
 *    cmc[plane - 1][stage]  chroma MC of Cb / Cr (4:2:0)
 *    wt[plane]           weight / biweight calls (a biweight's src_offset addresses the plane's bi-prediction scratch)
 *    idct_off / idct_coef[plane][kind]  residual blocks by FFHIP_H264_IDCT4 / IDCT8 / IDCT4_DC / IDCT8_DC: nidct offsets, 16 or 64 dctcoef each
 *    intra[set] / intra_coef[set]  intra macroblock records in recording order and their packed runs (int16 units): set 0 = whole
 *                        macroblocks (4:2:0), or one luma-only set per plane (4:4:4)
 *    edges[plane]        mb_w * mb_h * (8 | 4 | 6) edge records (zero records = not filtered), NULL when the plane has none; a 4:2:2
 *                        chroma plane: 6 per macroblock — the vertical edges at x = 0, 4 (16 lines, tc0 per 4 lines), then the horizontal
 *                        ones at y = 0, 4, 8, 12 */
typedef struct FFHipH264PictureLists {
    int mb_w, mb_h, bit_depth, chroma_format_idc;
    const FFHipQpelBlock *qpel[3][3];
    int nqpel[3][3];
    const FFHipChromaBlock *cmc[2][3];
    int ncmc[2][3];
    const FFHipWeightBlock *wt[3];
    int nwt[3];
    const int32_t *idct_off[3][4];
    const int16_t *idct_coef[3][4];
    int nidct[3][4];
    const FFHipH264IntraMB *intra[3];
    int nintra[3];
    const int16_t *intra_coef[3];
    int nintra_coef[3];
    const FFHipH264Edge *edges[3];
    /* 4:2:2: the chroma planes of the intra macroblocks (their luma: intra[0], luma-only) and their packed runs (int16 units) */
    const FFHipH264IntraC422 *intra_c422;
    int nintra_c422;
    const int16_t *intra_c422_coef;
    int nintra_c422_coef;
    /* the lossless bypass of inter macroblocks (round 6): add_pixels4_clear ([pl][0], 16 dctcoef each) / add_pixels8_clear ([pl][1], 64)
     * calls — ffhip_h264_picture_idct_add() kinds FFHIP_H264_ADD_PIXELS4_CLEAR / 8_CLEAR */
    const int32_t *addpx_off[3][2];
    const int16_t *addpx_coef[3][2];
    int naddpx[3][2];
} FFHipH264PictureLists;
int  ffhip_h264_picture_lists(const FFHipH264Picture *p, FFHipH264PictureLists *out);
/** One host-to-device copy of everything recorded since begin(), then the launches; asynchronous on `stream`. */
int  ffhip_h264_picture_flush(FFHipH264Picture *p, uint8_t *const dst[3], const int stride[3], const uint8_t *const ref[3],
                              void *stream);
/* ---- MBAFF frames (round 6): mb_adaptive_frame_field_flag, 4:2:0, 8 - 14 bits ------------------------------------------------------------
 * A frame whose macroblock pairs mix frame and field macroblocks (libavcodec/h264_mb_template.c:61-78, h264_loopfilter.c:494-560,716-760)
 * is recorded into FOUR objects over the same planes (integration/avcodec_h264_picture_hip.c does it):
 *   - three ordinary FFHipH264Picture objects for the INTER macroblocks' prediction, weight and residual lists: the frame macroblocks
 *     (mb_w x mb_h, the frame's line sizes), the top-field macroblocks and the bottom-field macroblocks (each mb_w x mb_h / 2 at TWICE the
 *     line sizes, the bottom one's planes one frame line further down) — a field macroblock of a pair is the field-picture case, which is
 *     how hl_decode_mb() addresses it; flush each with ffhip_h264_picture_flush() (they hold no intra macroblocks and no edges);
 *   - one FFHipH264Mbaff for the two chains whose order crosses macroblocks: the intra macroblocks (decoding order: pair rows, pairs
 *     left to right, top then bottom macroblock) and the in-loop filter's dsp calls, recorded call by call in the order
 *     ff_h264_filter_mb() issues them.  ffhip_h264_mbaff_flush() runs the intra reconstruction, then the filter, and goes LAST.
 */
typedef struct FFHipH264Mbaff FFHipH264Mbaff;
/** mb_w x mb_h: the frame's macroblocks (mb_h even).  FFHIP_EINVAL otherwise.  _fmt: bit_depth 8 / 9 / 10 / 12 / 14 (above 8: uint16_t samples,
 *  the decoder's arrays as they stand at that depth — see ffhip_h264_intra_pack_hbd()); 4:2:0 at every depth. */
int  ffhip_h264_mbaff_create(FFHipH264Mbaff **m, int mb_w, int mb_h);
int  ffhip_h264_mbaff_create_fmt(FFHipH264Mbaff **m, int mb_w, int mb_h, int bit_depth);
void ffhip_h264_mbaff_free(FFHipH264Mbaff **m);
void ffhip_h264_mbaff_begin(FFHipH264Mbaff *m);
/** One intra macroblock, as ffhip_h264_picture_intra_mb(): desc->mb_x, desc->mb_y = the macroblock's position in the FRAME (mb_y odd: the
 *  bottom macroblock of its pair); field != 0: a field macroblock (MB_FIELD(sl)) — its lines are every second frame line from the pair's
 *  line (mb_y & 1) on.  Macroblocks must arrive in decoding order. */
int  ffhip_h264_mbaff_intra_mb(FFHipH264Mbaff *m, const FFHipH264IntraMB *desc, int field, const uint8_t *non_zero_count_cache, int16_t *mb,
                               const int16_t *mb_luma_dc, const uint8_t *pcm);
/** One loop-filter dsp call of macroblock (mb_x, mb_y) on `plane` (0 luma, 1 Cb, 2 Cr), as the H264DSPContext member received it:
 *  call->offset = pix - the plane's first sample (a multiple of 4), call->kind = FFHIP_H264_LF_* of the member, alpha, beta, tc0, and in
 *  call->pad the flags below.  Calls arrive in the order ff_h264_filter_mb() issues them, macroblock by macroblock in decoding order. */
#define FFHIP_H264_LF_CALL_FIELD 1 /* the member was called with twice the plane's line size (a field macroblock's lines)              */
#define FFHIP_H264_LF_CALL_MBAFF 2 /* the _mbaff member: h264_h_loop_filter_luma_mbaff[_intra] (8 lines), _chroma_mbaff[_intra] (4 lines) */
int  ffhip_h264_mbaff_filter_call(FFHipH264Mbaff *m, int plane, int mb_x, int mb_y, const FFHipH264Edge *call);
/** What has been recorded since begin(), as flush() uploads it (host pointers, valid until the next record call): the CPU tier's list
 *  executor (oracle/emul_h264_mbaff.cpp) runs these. */
typedef struct FFHipH264MbaffLists {
    int mb_w, mb_h;
    const FFHipH264IntraMB *recs;   /* nrecs intra macroblocks in decoding order */
    const uint32_t *geo;            /* per record: mb_x | mb_y << 12 | field << 24 */
    const int16_t *coefs;           /* their packed runs (ncoefs int16) */
    const int32_t *intra_row;       /* mb_h / 2 + 1 starts into recs, by pair row */
    int32_t nrecs, ncoefs;
    const FFHipH264Edge *calls[3];  /* per plane: the calls in order */
    const int32_t *pair_end[3];     /* per plane and pair (row-major, mb_w x mb_h / 2): one past the pair's last call */
    int32_t ncalls[3];
    int bit_depth;
} FFHipH264MbaffLists;
int  ffhip_h264_mbaff_lists(FFHipH264Mbaff *m, FFHipH264MbaffLists *out);
/** The intra reconstruction (one wave per pair row, the tile and phases of every other picture's intra macroblocks at the
 *  macroblock's own line step), then the recorded filter calls in order (one wave per pair row and plane; pair x of row p after pair x + 1
 *  of row p - 1).  dst[] / stride[]: the FRAME's planes and line sizes (4-byte aligned; Cb and Cr share a line size).  Call after the
 *  three inter objects' flushes on the same stream.  Above 8 bits planes and line sizes are 8-byte aligned.  Asynchronous on `stream` once the
 *  lists are uploaded. */
int  ffhip_h264_mbaff_flush(FFHipH264Mbaff *m, uint8_t *const dst[3], const int stride[3], void *stream);

/** n picture objects (one geometry, depth and device; the pictures a decoder's frame threads hold at once) flushed TOGETHER (round 4):
 *  dst[3 i + pl] / ref[3 i + pl] are picture i's planes and reference bases, stride[] is shared.  Every picture's staging copy,
 *  prediction and residual launches are its own; the two stages that are a latency chain per picture — the intra reconstruction
 *  wavefront and the in-loop filter — run ONCE for all pictures, side by side (launches of up to 32 pictures; planes and strides
 *  16-byte aligned when any picture carries deblocking records).  The result per picture is flush()'s.  Asynchronous on `stream`. */
int  ffhip_h264_pictures_flush(FFHipH264Picture *const *pics, int n, uint8_t *const *dst, const int stride[3], const uint8_t *const *ref,
                               void *stream);
/** What the picture's last flush came to: 0, or the FFHIP_E* that left ITS planes incomplete.  When one picture of a
 *  ffhip_h264_pictures_flush() batch fails on the host side (staging memory, a malformed record) the others are still finished — their
 *  prediction and residual stages were queued against their planes already — and the call returns the first failure: this says which
 *  pictures it was.  A failure of a stage the pictures share marks every picture of the batch.
 *  A picture object takes records from ONE thread at a time (the record calls are plain appends): a decoder with several slice threads
 *  working on one picture gives each a recorder and serialises the calls into the shared object, or records slice by slice. */
int  ffhip_h264_picture_status(const FFHipH264Picture *p);

/* ------------------------------------------------------------------------------------------ */
/* libavcodec: AACDecDSP.imdct_and_windowing — float decoder, 1024-sample frames */
/* ------------------------------------------------------------------------------------------ */
/** AACDecDSP.imdct_and_windowing (libavcodec/aac/aacdec.h:488, body libavcodec/aac/aacdec_dsp_template.c:325-387): the inverse
 *  MDCT(s) of a channel's frame, the window and the overlap-add with the previous frame's tail.  The context owns the two inverse
 *  MDCTs (created with the scales ff_aac_decode_init() gives them, aacdec.c:1267-1285) and a device copy of the four window tables
 *  the decoder already holds (ff_sine_1024 / ff_sine_128, libavcodec/sinewin.h; ff_aac_kbd_long_1024 / ff_aac_kbd_short_128,
 *  libavcodec/aactab.h) — handed over by the caller, so that the tables in use are the decoder's own bit for bit. */
typedef struct FFHipAacImdct FFHipAacImdct;
int  ffhip_aac_imdct_create(FFHipAacImdct **c, const float *sine_1024, const float *sine_128, const float *kbd_long_1024,
                            const float *kbd_short_128, float scale_1024, float scale_128);
/** The same for the other frame lengths: 960 (AACDecDSP.imdct_and_windowing_960, aacdec_dsp_template.c:453-512; DAB+ / DRM) and 768
 *  (_768, :389-448); the tables are sine_<len>, sine_<len / 8>, the KBD windows of alpha 4 / 6 at those sizes, the scales those of
 *  mdct960 / mdct120 resp. mdct768 / mdct96 (aacdec.c:1278-1284).  Row pitches stay those of the 1024 case — coeffs / out rows 1024
 *  floats (sce->coeffs / sce->output are that wide whatever the frame length; a short window's coefficients sit 128 apart in a 960
 *  frame, 96 apart in a 768 one, as the reference reads them), saved 512 per channel — of which the first len / len / len / 2 are
 *  used.  Long-term prediction exists at 1024 only. */
int  ffhip_aac_imdct_create_len(FFHipAacImdct **c, int frame_len, const float *sine_long, const float *sine_short, const float *kbd_long,
                                const float *kbd_short, float scale_long, float scale_short);
void ffhip_aac_imdct_free(FFHipAacImdct **c);
/** One channel, one frame, host pointers: the member's effect on sce->coeffs / ics.window_sequence[2] / ics.use_kb_window[2]
 *  ([0] this frame, [1] the previous one) / sce->saved (512 floats in and out) / sce->output (1024 floats). */
int  ffhip_aac_imdct_and_windowing(FFHipAacImdct *c, const float *coeffs, const int window_sequence[2], const int use_kb_window[2],
                                   float *saved, float *out);
/** nframes consecutive frames of nch channels, device pointers, frame-major: coeffs / out [nframes][nch][1024], saved [nch][512]
 *  in and out.  window_sequence / use_kb_window [nframes][nch] and the state before the first frame, prev_* [nch], are HOST arrays
 *  (the decoder parses them; enum WindowSequence values, libavcodec/aac.h:63-68).  All frames are processed in parallel: the
 *  overlap state a frame leaves behind depends on that frame alone.  Asynchronous on `stream`; the host arrays may be released on
 *  return.  Not reentrant per context. */
int  ffhip_aac_imdct_and_windowing_batch_dev(FFHipAacImdct *c, const float *coeffs, float *out, float *saved,
                                             const uint8_t *window_sequence, const uint8_t *use_kb_window, const uint8_t *prev_sequence,
                                             const uint8_t *prev_kb_window, int nch, int nframes, void *stream);

/** AACDecDSP.apply_tns (libavcodec/aac/aacdec.h, body aacdec_dsp_template.c:164-223), float: one record per TNS filter with a
 *  non-empty range.  coef[] holds the transmitted reflection coefficients (TemporalNoiseShaping.coef[w][filt]); the LPC
 *  coefficients are derived on the device as the reference derives them (compute_lpc_coefs, lpc_functions.h:54-103). */
typedef struct FFHipAacTnsFilter {
    int32_t frame;      /* channel-frame the filter belongs to: coeffs + frame * 1024 */
    int16_t start;      /* first coefficient filtered (index into the frame, window offset included) */
    int16_t size;       /* number of coefficients */
    int8_t  inc;        /* +1 upwards, -1 downwards (TemporalNoiseShaping.direction) */
    uint8_t order;      /* 1..20 (TNS_MAX_ORDER) */
    uint8_t pad[2];
    float   coef[20];   /* sizeof == 92 */
} FFHipAacTnsFilter;
/** apply_tns's walk over windows and filters for one channel-frame (host side, from the parsed TemporalNoiseShaping and the
 *  IndividualChannelStream fields it reads); writes at most 32 records, returns their number. */
int ffhip_aac_tns_filters(FFHipAacTnsFilter *out, int frame, const int n_filt[8], const int length[8][4], const int direction[8][4],
                          const int order[8][4], const float coef[8][4][20], int num_windows, int num_swb, const uint16_t *swb_offset,
                          int tns_max_bands, int max_sfb);
/** The filters of a batch of channel-frames, in place on coeffs [nframes][1024] (device); filters is a device array.  decode != 0:
 *  the decoder's all-pole filter; 0: the moving-average form of the LTP path (apply_ltp, aacdec_dsp_template.c:252-282).  The
 *  filters of a frame cover disjoint ranges, so all of them run concurrently. */
int ffhip_aac_apply_tns_batch_dev(float *coeffs, const FFHipAacTnsFilter *filters, int nfilters, int decode, void *stream);

/** AACDecDSP.apply_prediction (AAC Main's backward-adaptive predictors; aacdec_dsp_template.c:636-664, aacdec_float_prediction.h):
 *  one record per channel-frame, one thread per predictor.  predictor_state: device array [channels][672] of PredictorState (8 floats:
 *  cor0 cor1 var0 var1 r0 r1 k1 x_est, libavcodec/aac_defines.h:130-139), carried from frame to frame by the caller — so a batch is
 *  one frame of many channels.  The host helper folds predictor_present / prediction_used[] / ff_aac_pred_sfb_max[] / swb_offset into
 *  the per-coefficient enable bits. */
enum { FFHIP_AAC_PRED_LONG = 1,          /* window_sequence[0] != EIGHT_SHORT_SEQUENCE (else: all predictors are reset) */
       FFHIP_AAC_PRED_RESET_FIRST = 2 }; /* ics.predictor_initialized was 0 */
typedef struct FFHipAacPrediction {
    int32_t  channel;      /* predictor_state + channel * 672 * 8 */
    int32_t  frame;        /* coeffs + frame * 1024 */
    int16_t  kmax;         /* swb_offset[pred_sfb_max]: predictors 0 .. kmax - 1 run */
    uint8_t  flags;
    uint8_t  reset_group;  /* ics.predictor_reset_group (0: none) */
    uint32_t enable[21];   /* bit k: coeffs[k] += prediction */
    uint32_t pad;          /* sizeof == 100 */
} FFHipAacPrediction;
int ffhip_aac_prediction_record(FFHipAacPrediction *out, int channel, int frame, int is_long, int initialized, int predictor_present,
                                const uint8_t *prediction_used, int pred_sfb_max, const uint16_t *swb_offset, int reset_group);
int ffhip_aac_apply_prediction_batch_dev(float *predictor_state, float *coeffs, const FFHipAacPrediction *recs, int n, void *stream);

/** AAC-LD and AAC-ELD: AACDecDSP.imdct_and_windowing_ld / _eld (aacdec_dsp_template.c:516-602), float.  eld == 0: 512-sample frames,
 *  w0 = ff_sine_512, w1 = ff_sine_128, scale = mdct512's ((1.0 / 512) / 32768, aacdec.c:1282).  eld != 0: frame_len 512 or 480, w0 =
 *  ff_aac_eld_window_512 / _480 (1920 / 1800 floats, libavcodec/aactab.h:61-63), w1 unused, scale that of mdct512 / mdct480.
 *  Batch: nframes consecutive frames of nch channels, frame-major, device pointers: coeffs / out rows of 1024 floats (the first
 *  frame_len used; the reference's ELD member shuffles sce->coeffs in place — this one leaves them alone), saved [nch][256] (LD) or
 *  [nch][3 * frame_len] (ELD: the three previous frames, newest first) in and out; kb_prev (LD only, HOST array [nframes][nch]) =
 *  ics->use_kb_window[1] of each frame, which selects the low-overlap window.  Asynchronous on `stream`. */
typedef struct FFHipAacLd FFHipAacLd;
int  ffhip_aac_ld_create(FFHipAacLd **c, int eld, int frame_len, const float *w0, const float *w1, float scale);
void ffhip_aac_ld_free(FFHipAacLd **c);
int  ffhip_aac_ld_batch_dev(FFHipAacLd *c, const float *coeffs, float *out, float *saved, const uint8_t *kb_prev, int nch, int nframes,
                            void *stream);

/** AACDecDSP.apply_mid_side_stereo / apply_intensity_stereo (aacdec_dsp_template.c:83-160), float, and the band-wise add that ends
 *  apply_ltp (:276-280): one record per range of coefficients.  The host walks (window groups, scalefactor bands, band types, the
 *  M/S mask — the fields decode_cpe / decode_ics leave in ChannelElement / IndividualChannelStream) produce the records; the ranges
 *  of one channel pair are disjoint (M/S bands have band_type < NOISE_BT, intensity bands 14 / 15), so a pair's M/S and intensity
 *  records — and any number of pairs — run in one launch. */
enum { FFHIP_AAC_BAND_MS = 0,         /* butterflies_float: a, b = a + b, a - b          (libavutil/float_dsp.c:112-122) */
       FFHIP_AAC_BAND_INTENSITY = 1,  /* vector_fmul_scalar: b = a * scale               (libavutil/float_dsp.c:45-51)   */
       FFHIP_AAC_BAND_ADD = 2,        /* a += b                                          (apply_ltp's last loop)         */
       FFHIP_AAC_BAND_FMAC = 3 };     /* a += scale * b   (product rounded, then the sum) (channel coupling)              */
typedef struct FFHipAacBandOp {
    int32_t frame0;     /* a = base_a + frame0 * 1024 + start */
    int32_t frame1;     /* b = base_b + frame1 * 1024 + start */
    int16_t start, len;
    float   scale;
    uint8_t kind;
    uint8_t pad[3];     /* sizeof == 20 */
} FFHipAacBandOp;
/** apply_mid_side_stereo's walk: cpe->ch[0].ics grouping, cpe->max_sfb_ste, cpe->ms_mask, both channels' band_type (enum BandType
 *  as int, 128 entries each).  Writes at most 64 records, returns their number. */
int ffhip_aac_ms_bands(FFHipAacBandOp *out, int frame0, int frame1, int num_window_groups, const uint8_t *group_len, int max_sfb_ste,
                       const uint8_t *ms_mask, const int *band_type0, const int *band_type1, const uint16_t *swb_offset);
/** apply_intensity_stereo's walk: cpe->ch[1].ics grouping and max_sfb, ms_present, cpe->ms_mask, ch[1].band_type and ch[1].sf.
 *  At most 128 records. */
int ffhip_aac_is_bands(FFHipAacBandOp *out, int frame0, int frame1, int num_window_groups, const uint8_t *group_len, int max_sfb,
                       int ms_present, const uint8_t *ms_mask, const int *band_type1, const float *sf1, const uint16_t *swb_offset);
/** apply_ltp's last loop: coeffs[frame] += predFreq[pred_frame] on ltp->used bands below min(max_sfb, MAX_LTP_LONG_SFB).  At most
 *  20 records. */
int ffhip_aac_ltp_bands(FFHipAacBandOp *out, int frame, int pred_frame, int max_sfb, const int8_t *used, const uint16_t *swb_offset);
/** The records of a batch on device arrays a / b of channel-frames (1024 floats each; the same array for the stereo tools,
 *  coeffs / predFreq for the LTP add); ops is a device array. */
int ffhip_aac_band_ops_batch_dev(float *a, float *b, const FFHipAacBandOp *ops, int n, void *stream);
/** AACDecDSP.apply_dependent_coupling (aacdec_float_coupling.h:42-71): the walk over the coupling channel's bands as FMAC records
 *  (dest_frame / src_frame index the same device array of channel-frames; run with ffhip_aac_band_ops_batch_dev(coeffs, coeffs, ..)).
 *  apply_independent_coupling (:78-88) is ONE such record over the output samples: { dest, src, 0, len, gain, FFHIP_AAC_BAND_FMAC }. */
int ffhip_aac_coupling_bands(FFHipAacBandOp *out, int dest_frame, int src_frame, int num_window_groups, const uint8_t *group_len, int max_sfb,
                             const int *band_type, const float *gain, const uint16_t *swb_offset);


/** AACDecDSP.apply_ltp (aacdec_dsp_template.c:252-282) as its three steps: (1) ffhip_aac_ltp_predict_batch_dev — the delayed state
 *  times ltp->coef, windowing_and_mdct_ltp (:225-247) and the forward 1024-point MDCT (created by ffhip_aac_ltp_init with the scale
 *  ff_aac_decode_init gives it, aacdec.c:1288-1291) -> predFreq; (2) ffhip_aac_apply_tns_batch_dev(predFreq, filters, n, 0) when
 *  the channel has TNS; (3) ffhip_aac_band_ops_batch_dev with ffhip_aac_ltp_bands' records.  A frame's prediction needs the
 *  previous frame's output, so a batch is one frame of many channels / streams, not many frames of one. */
typedef struct FFHipAacLtp {
    int32_t state;      /* the channel: ltp_state + state * 3072 */
    int16_t lag;        /* LongTermPrediction.lag (0..2047) */
    uint8_t seq0;       /* ics.window_sequence[0]; never EIGHT_SHORT_SEQUENCE (apply_ltp does nothing there: send no record) */
    uint8_t kb;         /* ics.use_kb_window[0] | use_kb_window[1] << 1 */
    float   coef;       /* LongTermPrediction.coef */
    int32_t pad;        /* sizeof == 16; record r writes pred_freq + r * 1024 */
} FFHipAacLtp;
int ffhip_aac_ltp_init(FFHipAacImdct *c, float scale_ltp);
int ffhip_aac_ltp_predict_batch_dev(FFHipAacImdct *c, const float *ltp_state, float *pred_freq, const FFHipAacLtp *recs, int n, void *stream);
/** AACDecDSP.update_ltp (aacdec_dsp_template.c:287-320) for the nch channels whose frame the preceding
 *  ffhip_aac_imdct_and_windowing_batch_dev call on this context ended with (its inverse-MDCT output — ac->buf_mdct — and overlap
 *  state are still in the context): ltp_state [nch][3072] in and out, out [nch][1024] = that frame's output samples. */
int ffhip_aac_update_ltp_batch_dev(FFHipAacImdct *c, float *ltp_state, const float *out, int nch, void *stream);

/* ------------------------------------------------------------------------------------------ */
/* libavcodec: H264PredContext — H.264 codec, 8 bits, chroma_format_idc <= 1 */
/* ------------------------------------------------------------------------------------------ */
/** H264PredContext (libavcodec/h264pred.h:92-116), the members ff_h264_pred_init() fills for AV_CODEC_ID_H264
 *  (libavcodec/h264pred.c:448-538): same signatures, host pointers, in place on the picture like the C functions.  Mode indices are
 *  the reference's (h264pred.h:35-48 for pred4x4 / pred8x8l, :67-82 for pred8x8 / pred16x16); the _add members are filled at
 *  [VERT_PRED] / [HOR_PRED] resp. [VERT_PRED8x8] / [HOR_PRED8x8] only, as in the reference.  A face reads exactly the neighbours its
 *  C counterpart reads (a DC_128 call touches none). */
typedef struct FFHipH264PredContext {
    void (*pred4x4[9 + 3 + 3])(uint8_t *src, const uint8_t *topright, ptrdiff_t stride);
    void (*pred8x8l[9 + 3])(uint8_t *src, int topleft, int topright, ptrdiff_t stride);
    void (*pred8x8[4 + 3 + 4])(uint8_t *src, ptrdiff_t stride);
    void (*pred16x16[4 + 3 + 2])(uint8_t *src, ptrdiff_t stride);
    void (*pred4x4_add[2])(uint8_t *pix, int16_t *block, ptrdiff_t stride);
    void (*pred8x8l_add[2])(uint8_t *pix, int16_t *block, ptrdiff_t stride);
    void (*pred8x8l_filter_add[2])(uint8_t *pix, int16_t *block, int topleft, int topright, ptrdiff_t stride);
    void (*pred8x8_add[3])(uint8_t *pix, const int *block_offset, int16_t *block, ptrdiff_t stride);
    void (*pred16x16_add[3])(uint8_t *pix, const int *block_offset, int16_t *block, ptrdiff_t stride);
} FFHipH264PredContext;
#define FFHIP_CODEC_ID_H264 27  /* AV_CODEC_ID_H264 (libavcodec/codec_id.h:77) */
#define FFHIP_CODEC_ID_SVQ3 23  /* AV_CODEC_ID_SVQ3, _RV40, _VP8, _VP7 (libavcodec/codec_id.h): the codecs whose decoders share H264PredContext, with their */
#define FFHIP_CODEC_ID_RV40 69  /* own forms of some members at 8 bits (libavcodec/h264pred.c:540-578)                                                    */
#define FFHIP_CODEC_ID_VP8  139
#define FFHIP_CODEC_ID_VP7  178
/** ff_h264_pred_init_<arch>(H264PredContext *, codec_id, bit_depth, chroma_format_idc) shape (libavcodec/h264pred.h:120-127).
 *  bit_depth 8 / 9 / 10 / 12 / 14 (16-bit samples and int32 coefficients of the _add members above 8; the depth is baked into the
 *  installed functions).  chroma_format_idc 0..3: from 2 on pred8x8[] / pred8x8_add[] are the 8 wide x 16 tall forms, as
 *  ff_h264_pred_init() installs them (h264pred.c:478-535).  codec_id: FFHIP_CODEC_ID_H264, or — 8 bits, chroma_format_idc <= 1 — SVQ3 /
 *  RV40 / VP7 / VP8: the table as ff_h264_pred_init() leaves it for that codec (their own forms of pred4x4[] / pred8x8[] / pred16x16[]
 *  members, h264pred.c:540-578; members the reference does not set for the codec — RV40 / VP7 / VP8: pred8x8[]'s four "mad cow" DC
 *  slots, VP8: pred4x4[DC_128_PRED] — are left as they were).  FFHIP_EINVAL for any other codec_id. */
int ff_h264_pred_init_hip(FFHipH264PredContext *h, int codec_id, int bit_depth, int chroma_format_idc);

#define FFHIP_H264_PRED4x4             0  /* pred4x4[mode]                                 */
#define FFHIP_H264_PRED8x8L            1  /* pred8x8l[mode]                                */
#define FFHIP_H264_PRED8x8             2  /* pred8x8[mode]   (chroma, 4:2:0)               */
#define FFHIP_H264_PRED16x16           3  /* pred16x16[mode]                               */
#define FFHIP_H264_PRED4x4_ADD         4  /* pred4x4_add[mode]: mode 0 VERT_PRED, 1 HOR_PRED */
#define FFHIP_H264_PRED8x8L_ADD        5  /* pred8x8l_add[mode]                            */
#define FFHIP_H264_PRED8x8L_FILTER_ADD 6  /* pred8x8l_filter_add[mode]                     */
#define FFHIP_H264_PRED8x16            7  /* pred8x8[mode] at chroma_format_idc 2 (4:2:2): the 8 wide x 16 tall forms */
#define FFHIP_H264_PRED_CODEC          8  /* the forms ff_h264_pred_init() installs for SVQ3 / RV40 / VP7 / VP8 (8 bits): mode = a
                                           * FFHIP_H264_PREDV_* code below, which also says the block size */
/* 4x4 (aux = bytes into the plane of topright[0..3], as for PRED4x4; the RV40 forms without _NODOWN also read the four rows below the
 * block in the left column: LOAD_DOWN_LEFT_EDGE, h264pred.c:141-190) */
#define FFHIP_H264_PREDV_127_DC          0   /* pred4x4_127_dc_c               VP7 / VP8 pred4x4[DC_127_PRED]            */
#define FFHIP_H264_PREDV_129_DC          1   /* pred4x4_129_dc_c                         pred4x4[DC_129_PRED]            */
#define FFHIP_H264_PREDV_VERT_VP8        2   /* pred4x4_vertical_vp8_c                   pred4x4[VERT_PRED]              */
#define FFHIP_H264_PREDV_HOR_VP8         3   /* pred4x4_horizontal_vp8_c                 pred4x4[HOR_PRED]               */
#define FFHIP_H264_PREDV_DL_SVQ3         4   /* pred4x4_down_left_svq3_c       SVQ3      pred4x4[DIAG_DOWN_LEFT_PRED]    */
#define FFHIP_H264_PREDV_DL_RV40         5   /* pred4x4_down_left_rv40_c       RV40      pred4x4[DIAG_DOWN_LEFT_PRED]    */
#define FFHIP_H264_PREDV_DL_RV40_NODOWN  6   /* ..._nodown_c                             pred4x4[DIAG_DOWN_LEFT_PRED_RV40_NODOWN] */
#define FFHIP_H264_PREDV_VL_RV40         7   /* pred4x4_vertical_left_rv40_c             pred4x4[VERT_LEFT_PRED]         */
#define FFHIP_H264_PREDV_VL_RV40_NODOWN  8   /* ..._nodown_c                             pred4x4[VERT_LEFT_PRED_RV40_NODOWN] */
#define FFHIP_H264_PREDV_VL_VP8          9   /* pred4x4_vertical_left_vp8_c    VP7 / VP8 pred4x4[VERT_LEFT_PRED]         */
#define FFHIP_H264_PREDV_HU_RV40         10  /* pred4x4_horizontal_up_rv40_c   RV40      pred4x4[HOR_UP_PRED]            */
#define FFHIP_H264_PREDV_HU_RV40_NODOWN  11  /* ..._nodown_c                             pred4x4[HOR_UP_PRED_RV40_NODOWN] */
#define FFHIP_H264_PREDV_TM_VP8          12  /* pred4x4_tm_vp8_c               VP7 / VP8 pred4x4[TM_VP8_PRED]            */
/* 8x8 (chroma) */
#define FFHIP_H264_PREDV8_DC_RV40        16  /* pred8x8_dc_rv40_c       RV40 / VP7 / VP8 pred8x8[DC_PRED8x8]             */
#define FFHIP_H264_PREDV8_LEFT_DC_RV40   17  /* pred8x8_left_dc_rv40_c                   pred8x8[LEFT_DC_PRED8x8]        */
#define FFHIP_H264_PREDV8_TOP_DC_RV40    18  /* pred8x8_top_dc_rv40_c                    pred8x8[TOP_DC_PRED8x8]         */
#define FFHIP_H264_PREDV8_TM_VP8         19  /* pred8x8_tm_vp8_c               VP7 / VP8 pred8x8[PLANE_PRED8x8]          */
#define FFHIP_H264_PREDV8_127_DC         20  /* pred8x8_127_dc_8_c                       pred8x8[DC_127_PRED8x8]         */
#define FFHIP_H264_PREDV8_129_DC         21  /* pred8x8_129_dc_8_c                       pred8x8[DC_129_PRED8x8]         */
/* 16x16 */
#define FFHIP_H264_PREDV16_PLANE_SVQ3    32  /* pred16x16_plane_svq3_c         SVQ3      pred16x16[PLANE_PRED8x8]        */
#define FFHIP_H264_PREDV16_PLANE_RV40    33  /* pred16x16_plane_rv40_c         RV40      pred16x16[PLANE_PRED8x8]        */
#define FFHIP_H264_PREDV16_TM_VP8        34  /* pred16x16_tm_vp8_c             VP7 / VP8 pred16x16[PLANE_PRED8x8]        */
#define FFHIP_H264_PREDV16_127_DC        35  /* pred16x16_127_dc_8_c                     pred16x16[DC_127_PRED8x8]       */
#define FFHIP_H264_PREDV16_129_DC        36  /* pred16x16_129_dc_8_c                     pred16x16[DC_129_PRED8x8]       */
#define FFHIP_H264_PRED_TOPLEFT   1  /* flags: pred8x8l's has_topleft                                                       */
#define FFHIP_H264_PRED_TOPRIGHT  2  /* flags: pred8x8l's has_topright                                                      */
#define FFHIP_H264_PRED_TR_SPLAT  4  /* flags: pred4x4's topright is src[3 - stride] four times (the decoder's substitute when the
                                        top-right block is unavailable, h264_mb_template.c hl_decode_mb_predict_luma)        */
/** One block of the batch face, predicted in place from its neighbours in the plane. */
typedef struct FFHipH264Pred {
    int32_t offset;   /* bytes into the plane: the block's top-left sample */
    int32_t aux;      /* PRED4x4: bytes into the plane of topright[0..3] (unless TR_SPLAT); the _ADD kinds: index into coeffs of
                         the block's first coefficient */
    uint8_t mode;
    uint8_t flags;
    uint8_t pad[2];   /* sizeof == 12 */
} FFHipH264Pred;
/** n blocks of one kind whose neighbours are final: intra prediction chains through the reconstruction, so a decoder batches
 *  what its wavefront allows (all blocks of a launch are read-before-write independent; a launch orders after the previous one
 *  on the stream).  pred8x8_add / pred16x16_add are 4 / 16 PRED4x4_ADD records at block_offset[], row of blocks by row (VERT) or
 *  column by column (HOR).  coeffs is used (and the blocks' coefficients cleared) by the _ADD kinds only. */
int ffhip_h264_pred_batch_dev(int kind, uint8_t *plane, ptrdiff_t stride, int16_t *coeffs, const FFHipH264Pred *blocks, int n,
                              void *stream);

/* ------------------------------------------------------------------------------------------ */
/* libavutil: AVFloatDSPContext vector operations around the MDCT           */
/* ------------------------------------------------------------------------------------------ */
/** The float members of AVFloatDSPContext an (I)MDCT pipeline uses (libavutil/float_dsp.h:31-175; windowing and
 *  overlap-add of AAC & co: imdct -> vector_fmul_window).  Same signatures, host pointers.  scalarproduct_float is not
 *  offered: its sequential summation order is the result. */
typedef struct FFHipFloatDSPContext {
    void (*vector_fmul)(float *dst, const float *src0, const float *src1, int len);
    void (*vector_fmac_scalar)(float *dst, const float *src, float mul, int len);
    void (*vector_fmul_scalar)(float *dst, const float *src, float mul, int len);
    void (*vector_fmul_window)(float *dst, const float *src0, const float *src1, const float *win, int len);
    void (*vector_fmul_add)(float *dst, const float *src0, const float *src1, const float *src2, int len);
    void (*vector_fmul_reverse)(float *dst, const float *src0, const float *src1, int len);
    void (*butterflies_float)(float *v1, float *v2, int len);
} FFHipFloatDSPContext;
/** ff_float_dsp_init_<arch>(AVFloatDSPContext *) shape (libavutil/float_dsp.c:153-165). */
int ff_float_dsp_init_hip(FFHipFloatDSPContext *c);

#define FFHIP_FDSP_FMUL          0   /* dst = src0 * src1                                                   */
#define FFHIP_FDSP_FMAC_SCALAR   1   /* dst += src0 * mul                                                   */
#define FFHIP_FDSP_FMUL_SCALAR   2   /* dst = src0 * mul                                                    */
#define FFHIP_FDSP_FMUL_WINDOW   3   /* dst[2 len] = window(src0[len], src1[len], win = src2[2 len])        */
#define FFHIP_FDSP_FMUL_ADD      4   /* dst = src0 * src1 + src2                                            */
#define FFHIP_FDSP_FMUL_REVERSE  5   /* dst[i] = src0[i] * src1[len - 1 - i]                                */
#define FFHIP_FDSP_BUTTERFLIES   6   /* (dst, src0) = (dst + src0, dst - src0): src0 is WRITTEN             */
/**
 * nvec independent vectors of len floats (device pointers): vector v of an operand starts pitch bytes after vector
 * v - 1; pitch 0 shares one vector across the batch (a window).  Operands an operation does not use may be NULL.
 * Results are bit-identical to the C reference (no fused multiply-add).
 */
int ffhip_fdsp_batch_dev(int op, float *dst, size_t dst_pitch, const float *src0, size_t pitch0, const float *src1, size_t pitch1,
                         const float *src2, size_t pitch2, float mul, int len, int nvec, void *stream);

/* ------------------------------------------------------------------------------------------ */
/* libavcodec: hevcdsp inverse transforms (north_star's "hevcdsp integer IDCT") */
/* ------------------------------------------------------------------------------------------ */
/** The transform members of HEVCDSPContext (libavcodec/hevc/dsp.h:46-61), 8-bit: index = log2_size - 2.
 *  idct leaves the residual IN PLACE in coeffs (libavcodec/hevc/dsp_template.c:261-284); col_limit bounds the
 *  non-zero coefficient columns/rows exactly as the reference's partial butterflies use it (coefficients beyond it are
 *  ignored, not assumed zero). */
typedef void (*ffhip_hevc_idct_func)(int16_t *coeffs, int col_limit);
typedef void (*ffhip_hevc_idct_dc_func)(int16_t *coeffs);
typedef void (*ffhip_hevc_add_residual_func)(uint8_t *dst, const int16_t *res, ptrdiff_t stride);
/** SAOParams (libavcodec/hevc/dsp.h:34-46), as sao_edge_restore takes it. */
typedef struct FFHipSAOParams {
    int offset_abs[3][4];
    int offset_sign[3][4];
    uint8_t band_position[3];
    int eo_class[3];
    int16_t offset_val[3][5];
    uint8_t type_idx[3];
} FFHipSAOParams;
/** hevc_{h,v}_loop_filter_luma / _chroma (hevc/dsp.h:103-124): 8 sample lines along an edge, two groups of 4 with their own
 *  tc / no_p / no_q.  h_: the edge is horizontal (samples of a line are `stride` apart). */
typedef void (*ffhip_hevc_lf_luma_func)(uint8_t *pix, ptrdiff_t stride, int beta, const int32_t *tc, const uint8_t *no_p,
                                        const uint8_t *no_q);
typedef void (*ffhip_hevc_lf_chroma_func)(uint8_t *pix, ptrdiff_t stride, const int32_t *tc, const uint8_t *no_p, const uint8_t *no_q);
typedef struct FFHipHEVCDSPContext {
    ffhip_hevc_add_residual_func add_residual[4];
    void (*transform_4x4_luma)(int16_t *coeffs);
    ffhip_hevc_idct_func    idct[4];
    ffhip_hevc_idct_dc_func idct_dc[4];
    ffhip_hevc_lf_luma_func   hevc_h_loop_filter_luma, hevc_v_loop_filter_luma;
    ffhip_hevc_lf_chroma_func hevc_h_loop_filter_chroma, hevc_v_loop_filter_chroma;
    ffhip_hevc_lf_luma_func   hevc_h_loop_filter_luma_c, hevc_v_loop_filter_luma_c;     /* the decoder's "_c" slots: same functions */
    ffhip_hevc_lf_chroma_func hevc_h_loop_filter_chroma_c, hevc_v_loop_filter_chroma_c;
    /* sample adaptive offset (hevc/dsp.h:63-69; index = width class 8/16/32/48/64, one function for all).  The edge filter
     * reads its source with the reference's fixed stride of 2*MAX_PB_SIZE + AV_INPUT_BUFFER_PADDING_SIZE = 192 bytes and needs
     * one sample of margin around the block, as in the reference. */
    void (*sao_band_filter[5])(uint8_t *dst, const uint8_t *src, ptrdiff_t stride_dst, ptrdiff_t stride_src, const int16_t *sao_offset_val,
                               int sao_left_class, int width, int height);
    void (*sao_edge_filter[5])(uint8_t *dst, const uint8_t *src, ptrdiff_t stride_dst, const int16_t *sao_offset_val, int eo, int width,
                               int height);
    /* uni-directional motion compensation (hevc/dsp.h:74-77,88-92): [width class][!!my][!!mx]; the plain forms write 14-bit
     * int16 intermediates with a row stride of MAX_PB_SIZE = 64, the _uni forms write pixels */
    void (*put_hevc_qpel[10][2][2])(int16_t *dst, const uint8_t *src, ptrdiff_t srcstride, int height, intptr_t mx, intptr_t my, int width);
    void (*put_hevc_qpel_uni[10][2][2])(uint8_t *dst, ptrdiff_t dststride, const uint8_t *src, ptrdiff_t srcstride, int height, intptr_t mx,
                                        intptr_t my, int width);
    void (*put_hevc_epel[10][2][2])(int16_t *dst, const uint8_t *src, ptrdiff_t srcstride, int height, intptr_t mx, intptr_t my, int width);
    void (*put_hevc_epel_uni[10][2][2])(uint8_t *dst, ptrdiff_t dststride, const uint8_t *src, ptrdiff_t srcstride, int height, intptr_t mx,
                                        intptr_t my, int width);
    /* the small members (hevc/dsp.h:52-54,70-72): transform-skip scaling, RDPCM running sums, SAO border fix-up.  put_pcm reads a
     * bitstream (GetBitContext) and stays with the decoder */
    void (*dequant)(int16_t *coeffs, int16_t log2_size);
    void (*transform_rdpcm)(int16_t *coeffs, int16_t log2_size, int mode);
    void (*sao_edge_restore[2])(uint8_t *dst, const uint8_t *src, ptrdiff_t stride_dst, ptrdiff_t stride_src, const FFHipSAOParams *sao,
                                const int *borders, int width, int height, int c_idx, const uint8_t *vert_edge,
                                const uint8_t *horiz_edge, const uint8_t *diag_edge);
    /* weighted and bi-directional prediction (hevc/dsp.h:78-87,93-101); src2 = the other list's put_hevc_* output (row stride 64) */
    void (*put_hevc_qpel_uni_w[10][2][2])(uint8_t *dst, ptrdiff_t dststride, const uint8_t *src, ptrdiff_t srcstride, int height, int denom,
                                          int wx, int ox, intptr_t mx, intptr_t my, int width);
    void (*put_hevc_qpel_bi[10][2][2])(uint8_t *dst, ptrdiff_t dststride, const uint8_t *src, ptrdiff_t srcstride, const int16_t *src2,
                                       int height, intptr_t mx, intptr_t my, int width);
    void (*put_hevc_qpel_bi_w[10][2][2])(uint8_t *dst, ptrdiff_t dststride, const uint8_t *src, ptrdiff_t srcstride, const int16_t *src2,
                                         int height, int denom, int wx0, int wx1, int ox, intptr_t mx, intptr_t my, int width);
    void (*put_hevc_epel_uni_w[10][2][2])(uint8_t *dst, ptrdiff_t dststride, const uint8_t *src, ptrdiff_t srcstride, int height, int denom,
                                          int wx, int ox, intptr_t mx, intptr_t my, int width);
    void (*put_hevc_epel_bi[10][2][2])(uint8_t *dst, ptrdiff_t dststride, const uint8_t *src, ptrdiff_t srcstride, const int16_t *src2,
                                       int height, intptr_t mx, intptr_t my, int width);
    void (*put_hevc_epel_bi_w[10][2][2])(uint8_t *dst, ptrdiff_t dststride, const uint8_t *src, ptrdiff_t srcstride, const int16_t *src2,
                                         int height, int denom, int wx0, int wx1, int ox, intptr_t mx, intptr_t my, int width);
} FFHipHEVCDSPContext;
/** ff_hevc_dsp_init_<arch> shape (libavcodec/hevc/dsp.h:127-140).  bit_depth 8, 10 or 12 (the depth is baked into the installed
 *  functions, as the reference's per-BIT_DEPTH instantiations are); FFHIP_EINVAL for 9 and above 12: those keep the C pointers. */
int ff_hevc_dsp_init_hip(FFHipHEVCDSPContext *c, int bit_depth);

#define FFHIP_HEVC_IDCT      0   /* idct[log2_size - 2](coeffs, col_limit)     */
#define FFHIP_HEVC_IDCT_DC   1   /* idct_dc[log2_size - 2](coeffs)             */
#define FFHIP_HEVC_DST_4X4   2   /* transform_4x4_luma(coeffs), log2_size == 2 */
#define FFHIP_HEVC_ADD_ONLY  3   /* coeffs already hold the residual           */
#define FFHIP_HEVC_DEQUANT   4   /* dequant(coeffs, log2_size): transform-skip scaling (hevc/dsp_template.c:127-143) */
#define FFHIP_HEVC_RDPCM_H   5   /* transform_rdpcm(coeffs, log2_size, 0): running sums along rows (:85-105)        */
#define FFHIP_HEVC_RDPCM_V   6   /* transform_rdpcm(coeffs, log2_size, 1): running sums down columns                */
/** One transform unit of the batch face: what hls_residual_coding / hls_transform_unit pass per TU
 *  (libavcodec/hevc/cabac.c, hevcdec.c). */
typedef struct FFHipHevcTU {
    int32_t coeff_offset; /* in int16 units into coeffs: the TU's size*size block, row-major      */
    int32_t dst_offset;   /* in bytes into dst; < 0: leave the picture alone                      */
    int32_t col_limit;    /* FFHIP_HEVC_IDCT only                                                 */
} FFHipHevcTU;
/**
 * n transform units of one size and kind: the inverse transform in place in coeffs (device), then, when dst is
 * non-NULL and the TU's dst_offset >= 0, add_residual[log2_size - 2](dst + dst_offset, residual, stride).
 * TUs of one call must not overlap in coeffs or dst.
 */
int ffhip_hevc_idct_batch_dev(int kind, int log2_size, int16_t *coeffs, uint8_t *dst, ptrdiff_t stride,
                              const FFHipHevcTU *tus, int n, void *stream);

#define FFHIP_HEVC_LF_H_LUMA    0
#define FFHIP_HEVC_LF_V_LUMA    1
#define FFHIP_HEVC_LF_H_CHROMA  2
#define FFHIP_HEVC_LF_V_CHROMA  3
/** One edge segment of the batch face: everything a hevc_*_loop_filter_* call takes besides pix/stride. */
typedef struct FFHipHevcEdge {
    int32_t offset;      /* pix = base + offset */
    uint8_t kind;        /* FFHIP_HEVC_LF_* */
    uint8_t beta;        /* luma only (betatable tops out at 64) */
    uint8_t no_p[2], no_q[2];
    int16_t tc[2];
    uint8_t pad[2];      /* sizeof == 16 */
} FFHipHevcEdge;
/**
 * n edge segments whose touched pixels are pairwise DISJOINT.  HEVC deblocking has no order dependency inside one
 * direction (edges lie on an 8x8 grid and change at most 3 samples on either side): a picture is one call with all its
 * vertical edges followed by one call with all its horizontal edges (libavcodec/hevc/filter.c ff_hevc_deblocking_boundary_strengths
 * / deblocking_filter_CTB).
 */
int ffhip_hevc_loop_filter_batch_dev(uint8_t *base, ptrdiff_t stride, const FFHipHevcEdge *edges, int n, void *stream);

/** One prediction block of the batch face: what luma_mc_uni / chroma_mc_uni pass per block (libavcodec/hevc/hevcdec.c). */
typedef struct FFHipHevcMcBlock {
    int32_t dst_offset;   /* uni: bytes into dst; plain: int16 elements into dst (rows are 64 elements apart) */
    int32_t src_offset;   /* bytes into src: the block's integer-sample origin */
    uint8_t width, height;/* 2..64 */
    uint8_t mx, my;       /* luma: quarter-sample 0..3; chroma: eighth-sample 0..7 */
} FFHipHevcMcBlock;
/**
 * n blocks: put_hevc_{qpel,epel}[..][!!my][!!mx] (uni == 0, dst = int16) or put_hevc_{qpel,epel}_uni (uni != 0, dst = pixels with
 * dststride).  src must be readable 3 (chroma: 1) samples left/up and 4 (2) right/down of each block.
 */
int ffhip_hevc_mc_batch_dev(int chroma, int uni, void *dst, ptrdiff_t dststride, const uint8_t *src, ptrdiff_t srcstride,
                            const FFHipHevcMcBlock *blocks, int n, void *stream);

/** Weighted / bi-directional prediction (luma_mc_uni with weights, luma_mc_bi, chroma_mc_bi: libavcodec/hevc/hevcdec.c:1745-1770,
 *  1795-1860, 1975-2040): one record = the operands of one put_hevc_{qpel,epel}_{uni_w,bi,bi_w} call (hevc/dsp.h:78-101). */
#define FFHIP_HEVC_MC_UNI_W 2   /* dst = clip(((v * wx0 + 2^(shift-1)) >> shift) + ox), shift = denom + 6          */
#define FFHIP_HEVC_MC_BI    3   /* dst = clip((v + src2 + 64) >> 7)                                                 */
#define FFHIP_HEVC_MC_BI_W  4   /* dst = clip((v * wx1 + src2 * wx0 + (ox + 1) << log2Wd) >> (log2Wd + 1))         */
typedef struct FFHipHevcMcWBlock {
    int32_t dst_offset;   /* bytes into dst */
    int32_t src_offset;   /* bytes into src: the block's integer-sample origin */
    int32_t src2_offset;  /* int16 elements into src2 (the other list's put_hevc_* output, rows 64 elements apart); unused by uni_w */
    uint8_t width, height;/* 2..64 */
    uint8_t mx, my;       /* luma: quarter-sample 0..3; chroma: eighth-sample 0..7 */
    int16_t wx0, wx1;     /* uni_w: wx0 = wx; bi_w: wx0 weights src2, wx1 weights this block (the reference's argument order) */
    int16_t ox;           /* uni_w: the offset; bi_w: o0 + o1 as the decoder passes it */
    uint8_t denom;        /* log2 weight denominator (0..7 from the slice header; checkasm goes to 12) */
    uint8_t pad;          /* sizeof == 24 */
} FFHipHevcMcWBlock;
int ffhip_hevc_mc_w_batch_dev(int chroma, int mode, uint8_t *dst, ptrdiff_t dststride, const uint8_t *src, ptrdiff_t srcstride,
                              const int16_t *src2, const FFHipHevcMcWBlock *blocks, int n, void *stream);

/** One SAO call of the batch face (a CTB plane or part of one). */
typedef struct FFHipHevcSao {
    int32_t dst_offset, src_offset;  /* into dst / src */
    int16_t offset_val[5];           /* sao_offset_val: [0] unused by the band filter */
    uint8_t edge;                    /* 0: sao_band_filter, 1: sao_edge_filter */
    uint8_t cls;                     /* band: sao_left_class (0..31); edge: eo = SAO_EO_* (0 horizontal, 1 vertical, 2 135 deg, 3 45 deg) */
    uint8_t width, height;           /* 1..64 */
    uint8_t pad[2];                  /* sizeof == 24 */
} FFHipHevcSao;
/** n SAO blocks: dst[..] = clip(src[..] + offset(class)); src and dst are different buffers (the decoder filters from a
 *  copy), blocks do not overlap in dst; the edge filter reads src one sample beyond the block on every side. */
int ffhip_hevc_sao_batch_dev(uint8_t *dst, ptrdiff_t stride_dst, const uint8_t *src, ptrdiff_t stride_src, const FFHipHevcSao *blocks,
                             int n, void *stream);

/** One sao_edge_restore[variant] call of the batch face (libavcodec/hevc/filter.c:440-470 builds the operands). */
typedef struct FFHipHevcSaoRestore {
    int32_t dst_offset, src_offset;
    int16_t offset0;      /* sao->offset_val[c_idx][0]                                              */
    uint8_t width, height;/* 2..64 (the reference indexes column width - 2)                         */
    uint8_t eo;           /* sao->eo_class[c_idx] = SAO_EO_*: 0 horizontal, 1 vertical, 2 135 degrees, 3 45 degrees */
    uint8_t variant;      /* 0: sao_edge_restore[0] (borders only), 1: [1] (+ the restore part)     */
    uint8_t borders;      /* bit i = borders[i] != 0: left, top, right, bottom                     */
    uint8_t vert_edge;    /* bits 0..1 = vert_edge[0..1]                                            */
    uint8_t horiz_edge;   /* bits 0..1                                                              */
    uint8_t diag_edge;    /* bits 0..3                                                              */
    uint8_t pad[2];       /* sizeof == 20                                                           */
} FFHipHevcSaoRestore;
/** n blocks, pairwise disjoint in dst; src and dst as for ffhip_hevc_sao_batch_dev. */
int ffhip_hevc_sao_restore_batch_dev(uint8_t *dst, ptrdiff_t stride_dst, const uint8_t *src, ptrdiff_t stride_src,
                                     const FFHipHevcSaoRestore *blocks, int n, void *stream);

/**
 * hevcdsp above 8 bits (Main10 / Main12 streams): the batch faces above with the BIT_DEPTH the reference instantiates its templates
 * for (libavcodec/hevc/dsp.c:133-196; bit_depth_template.c: pixel = uint16_t, av_clip_pixel to (1 << bit_depth) - 1).  bit_depth
 * 8, 10 or 12; 8 is the face without the suffix.  Records are unchanged: pixel offsets and strides stay in BYTES as in the
 * reference's signatures (16-bit planes 2-byte aligned); beta / tc / SAO offsets / weights are the values the decoder passes —
 * the scaling by << (bit_depth - 8) happens inside, where the reference's templates do it (hevc/dsp_template.c:845,862,907;
 * h2656_inter_template.c:72; dsp_template.c:408).  Coefficients and the put_hevc_* intermediates are int16 at every depth.
 */
int ffhip_hevc_idct_batch_dev_hbd(int bit_depth, int kind, int log2_size, int16_t *coeffs, uint8_t *dst, ptrdiff_t stride,
                                  const FFHipHevcTU *tus, int n, void *stream);
int ffhip_hevc_loop_filter_batch_dev_hbd(int bit_depth, uint8_t *base, ptrdiff_t stride, const FFHipHevcEdge *edges, int n, void *stream);
int ffhip_hevc_sao_batch_dev_hbd(int bit_depth, uint8_t *dst, ptrdiff_t stride_dst, const uint8_t *src, ptrdiff_t stride_src,
                                 const FFHipHevcSao *blocks, int n, void *stream);
int ffhip_hevc_sao_restore_batch_dev_hbd(int bit_depth, uint8_t *dst, ptrdiff_t stride_dst, const uint8_t *src, ptrdiff_t stride_src,
                                         const FFHipHevcSaoRestore *blocks, int n, void *stream);
int ffhip_hevc_mc_batch_dev_hbd(int bit_depth, int chroma, int uni, void *dst, ptrdiff_t dststride, const uint8_t *src, ptrdiff_t srcstride,
                                const FFHipHevcMcBlock *blocks, int n, void *stream);
int ffhip_hevc_mc_w_batch_dev_hbd(int bit_depth, int chroma, int mode, uint8_t *dst, ptrdiff_t dststride, const uint8_t *src,
                                  ptrdiff_t srcstride, const int16_t *src2, const FFHipHevcMcWBlock *blocks, int n, void *stream);

/* ------------------------------------------------------------------------------------------ */
/* libavcodec: vp9dsp inverse transforms                                    */
/* ------------------------------------------------------------------------------------------ */
/** VP9DSPContext.itxfm_add (libavcodec/vp9dsp.h:71-75): [tx][txtp](dst, stride, block, eob); tx 0..3 = TX_4X4..TX_32X32,
 *  4 = the lossless 4x4 Walsh-Hadamard transform; txtp = enum TxfmType (DCT_DCT 0, DCT_ADST 1, ADST_DCT 2, ADST_ADST 3,
 *  libavcodec/vp9.h).  The transform adds to dst and consumes the block (zeroed; eob == 1 on DCT_DCT: block[0] only).
 *  8 bits; host pointers. */
typedef void (*ffhip_vp9_itxfm_add_func)(uint8_t *dst, ptrdiff_t stride, int16_t *block, int eob);
typedef struct FFHipVP9ItxfmContext {
    ffhip_vp9_itxfm_add_func itxfm_add[5][4];
} FFHipVP9ItxfmContext;
/** ff_vp9dsp_init_<arch> shape for the itxfm_add table (libavcodec/vp9dsp.c:88-112).  bpp 8, 10 or 12 (baked into the installed functions; above 8 bits samples are uint16_t and blocks hold int32 coefficients). */
int ff_vp9dsp_itxfm_init_hip(FFHipVP9ItxfmContext *c, int bpp);

/** One transform block of the batch face. */
typedef struct FFHipVp9TU {
    int32_t coeff_offset; /* int16 elements into coeffs: size * size coefficients, the decoder's layout */
    int32_t dst_offset;   /* bytes into dst                                                           */
    uint8_t txtp;         /* enum TxfmType; ignored for 32x32 and the WHT                              */
    uint8_t dc_only;      /* eob == 1: DCT_DCT takes its dc-only shortcut                              */
    uint8_t pad[2];       /* sizeof == 12                                                              */
} FFHipVp9TU;
/** n blocks of one size (tx as above), pairwise disjoint in coeffs and dst: itxfm_add[tx][txtp] each. */
int ffhip_vp9_itxfm_add_batch_dev(int tx, int16_t *coeffs, uint8_t *dst, ptrdiff_t stride, const FFHipVp9TU *tus, int n,
                                  void *stream);

/** vp9_mc_func and VP9DSPContext.mc[size 64/32/16/8/4][filter][put/avg][!!mx][!!my] (libavcodec/vp9dsp.h:33-35,115;
 *  enum FilterMode, libavcodec/vp9.h:64-70: 0 smooth, 1 regular, 2 sharp, 3 bilinear); mx, my in sixteenths. */
typedef void (*ffhip_vp9_mc_func)(uint8_t *dst, ptrdiff_t dst_stride, const uint8_t *src, ptrdiff_t src_stride, int h, int mx, int my);
typedef struct FFHipVP9McContext {
    ffhip_vp9_mc_func mc[5][4][2][2][2];
} FFHipVP9McContext;
int ff_vp9dsp_mc_init_hip(FFHipVP9McContext *c, int bpp);
/** One prediction block of the batch face (what inter_pred / mc_luma_unscaled pass, libavcodec/vp9recon.c). */
typedef struct FFHipVp9McBlock {
    int32_t dst_offset, src_offset; /* bytes into dst / src: the block's integer-sample origin */
    uint8_t width;                  /* 4, 8, 16, 32 or 64 */
    uint8_t height;                 /* 1..64 */
    uint8_t filter;                 /* enum FilterMode 0..3 */
    uint8_t mx, my;                 /* 0..15 */
    uint8_t avg;                    /* 0 put, 1 avg (compound prediction's second reference) */
    uint8_t pad[2];                 /* sizeof == 16 */
} FFHipVp9McBlock;
/** n blocks, pairwise disjoint in dst; src must be readable 3 samples left/up and 4 right/down of each block (rows are read in
 *  whole dwords: 1 byte more on the right). */
int ffhip_vp9_mc_batch_dev(uint8_t *dst, ptrdiff_t dststride, const uint8_t *src, ptrdiff_t srcstride, const FFHipVp9McBlock *blocks,
                           int n, void *stream);

/** vp9_scaled_mc_func and VP9DSPContext.smc[size][filter][put/avg] (libavcodec/vp9dsp.h:36-38,121): prediction from a reference
 *  picture of another size; dx, dy = the step in sixteenths of a reference sample per output sample (1..32: 16x up to 2x down). */
typedef void (*ffhip_vp9_scaled_mc_func)(uint8_t *dst, ptrdiff_t dst_stride, const uint8_t *src, ptrdiff_t src_stride, int h, int mx, int my,
                                         int dx, int dy);
typedef struct FFHipVP9ScaledMcContext {
    ffhip_vp9_scaled_mc_func smc[5][4][2];
} FFHipVP9ScaledMcContext;
int ff_vp9dsp_scaled_mc_init_hip(FFHipVP9ScaledMcContext *c, int bpp);
typedef struct FFHipVp9ScaledBlock {
    int32_t dst_offset, src_offset;
    uint8_t width, height;          /* 4..64 (a power of two), 1..64 */
    uint8_t filter, mx, my, avg;    /* as FFHipVp9McBlock */
    uint8_t dx, dy;                 /* 1..32 */
} FFHipVp9ScaledBlock;
/** n blocks, pairwise disjoint in dst; src must be readable 3 samples left/up of the block origin and through column
 *  ((mx + (w - 1) dx) >> 4) + 4, row ((my + (h - 1) dy) >> 4) + 4. */
int ffhip_vp9_scaled_mc_batch_dev(uint8_t *dst, ptrdiff_t dststride, const uint8_t *src, ptrdiff_t srcstride,
                                  const FFHipVp9ScaledBlock *blocks, int n, void *stream);

/** The loop-filter tables of VP9DSPContext (libavcodec/vp9dsp.h:76-105): loop_filter_8[width 4/8/16][h col-edge / v row-edge],
 *  loop_filter_16[dir], loop_filter_mix2[wd1][wd2][dir] (two 8-sample halves, limits packed in the two low bytes). */
typedef void (*ffhip_vp9_lf_func)(uint8_t *dst, ptrdiff_t stride, int mb_lim, int lim, int hev_thr);
typedef struct FFHipVP9LoopFilterContext {
    ffhip_vp9_lf_func loop_filter_8[3][2];
    ffhip_vp9_lf_func loop_filter_16[2];
    ffhip_vp9_lf_func loop_filter_mix2[2][2][2];
} FFHipVP9LoopFilterContext;
int ff_vp9dsp_loopfilter_init_hip(FFHipVP9LoopFilterContext *c, int bpp);
/** One 8-sample segment of an edge: what one loop_filter() call of the reference covers (vp9dsp_template.c:1780-1889). */
typedef struct FFHipVp9Edge {
    int32_t offset;      /* bytes into the plane: the first q0 sample of the segment */
    uint8_t wd_idx;      /* 0: 4, 1: 8, 2: 16 */
    uint8_t dir;         /* 0: column edge (loop_filter_h_*), 1: row edge (loop_filter_v_*) */
    uint8_t E, I, H;     /* mb_lim, lim, hev_thr */
    uint8_t pad[3];      /* sizeof == 12 */
} FFHipVp9Edge;
/** n segments that share no sample (VP9 orders the overlapping edges of a superblock: column edges left to right, then row
 *  edges; a caller batches what is disjoint, e.g. every other 8-sample column of wd <= 8 edges). */
int ffhip_vp9_loop_filter_batch_dev(uint8_t *base, ptrdiff_t stride, const FFHipVp9Edge *edges, int n, void *stream);

/** VP9DSPContext.intra_pred[tx 4x4..32x32][enum IntraPredMode] (libavcodec/vp9dsp.h:52-55, libavcodec/vp9.h:45-62): left[] runs
 *  bottom to top, top[-1] is the corner; the 4x4 down-left / vert-left modes read top[0..7]. */
typedef void (*ffhip_vp9_intra_func)(uint8_t *dst, ptrdiff_t stride, const uint8_t *left, const uint8_t *top);
typedef struct FFHipVP9IntraContext {
    ffhip_vp9_intra_func intra_pred[4][15];
} FFHipVP9IntraContext;
int ff_vp9dsp_intrapred_init_hip(FFHipVP9IntraContext *c, int bpp);
/** One block of the batch face.  Its neighbours live in `edges` as the edge line: left[0..N-1], the corner, top[0..max(N,8)-1]
 *  (N + 1 + max(N, 8) bytes at edge_offset) — what the decoder's edge preparation (libavcodec/vp9recon.c, check_intra_mode)
 *  produces, concatenated. */
typedef struct FFHipVp9Intra {
    int32_t dst_offset;   /* bytes into dst */
    int32_t edge_offset;  /* bytes into edges */
    uint8_t mode;         /* enum IntraPredMode 0..14 */
    uint8_t pad[3];       /* sizeof == 12 */
} FFHipVp9Intra;
int ffhip_vp9_intra_pred_batch_dev(int tx, uint8_t *dst, ptrdiff_t stride, const uint8_t *edges, const FFHipVp9Intra *blocks, int n,
                                   void *stream);
/**
 * The VP9 loop filter in SUPERBLOCK (decoder) order: ff_vp9_loopfilter_sb() (libavcodec/vp9lpf.c:180-203) for a whole picture.
 * The function-level batch above takes segments that share no sample; a decoder's edges overlap (a 16-wide filter reaches 8
 * samples either way, and a superblock's left / top edges rewrite its neighbours' samples), so the picture is a dependency graph:
 * per plane all column edges of a superblock, then all its row edges, superblocks in raster order.  Here a superblock's VP9Filter
 * becomes a table of 8-line segments per edge position on the host (ffhip_vp9_lf_sb_tables, device-free) and one launch walks the
 * picture as a wavefront (one wave per superblock row; superblock (x, y) starts when row y - 1 has finished x + 1), all three
 * planes in the same step.  4:2:0; bit_depth 8, 10 or 12.
 */
typedef struct FFHipVp9Filter {          /* == VP9Filter (libavcodec/vp9dec.h:79-83) */
    uint8_t level[8 * 8];
    uint8_t mask[2 /* 0 = y, 1 = uv */][2 /* 0 = col, 1 = row */][8 /* rows */][4 /* 0 = 16, 1 = 8, 2 = 4, 3 = inner 4 */];
} FFHipVp9Filter;
typedef struct FFHipVp9LfSb {            /* entry: bit 31 valid, 24..25 width (0: 4, 1: 8, 2: 16), 16..23 H, 8..15 I (lim), 0..7 E (mblim) */
    uint32_t y[2][16][8];                /* [0 column edges / 1 row edges][position: 4 p samples along the filter axis][segment of 8 lines] */
    uint32_t uv[2][8][4];                /* the same for both chroma planes (32 x 32 samples) */
} FFHipVp9LfSb;
/** The chroma planes' table of one superblock where the two sub-sampling shifts differ — 4:2:2 (ss_h 1, ss_v 0: 32 x 64 chroma samples) and
 *  4:4:0 (ss_h 0, ss_v 1: 64 x 32): entries as in FFHipVp9LfSb; first the column edges [position][segment of 8 lines] (4:2:2: 8 x 8, 4:4:0:
 *  16 x 4), then the row edges [position][segment of 8 columns] (4:2:2: 16 x 4, 4:4:0: 8 x 8) — 128 words either way.  Both chroma planes
 *  share it (filter_plane_cols / _rows with uv_masks = lflvl->mask[1], libavcodec/vp9lpf.c:185-201). */
typedef struct FFHipVp9LfSbC {
    uint32_t t[128];
} FFHipVp9LfSbC;
int ffhip_vp9_lf_sb_ctables(FFHipVp9LfSbC *out, const FFHipVp9Filter *lflvl, int row, int col, int ss_h, int ss_v, const uint8_t *lim_lut,
                            const uint8_t *mblim_lut);
/** Host side: the tables of the superblock at (row, col) — in 8-sample units, as the reference passes them (superblock (r, c): row =
 *  8 r, col = 8 c; only "is it the first" matters) — from its VP9Filter and the frame's filter_lut (vp9.c:683-697).  ss_h = ss_v = 1
 *  (4:2:0) or 0 (4:4:4: y only, the chroma planes are filtered by the luma tables); 1, 0 / 0, 1 (4:2:2 / 4:4:0): y only as well, the chroma
 *  planes take ffhip_vp9_lf_sb_ctables(). */
int ffhip_vp9_lf_sb_tables(FFHipVp9LfSb *out, const FFHipVp9Filter *lflvl, int row, int col, int ss_h, int ss_v, const uint8_t *lim_lut,
                           const uint8_t *mblim_lut);
/** One picture of cols x rows 8x8 blocks (VP9Context.cols / .rows: (width + 7) >> 3, (height + 7) >> 3), i.e. (cols + 7) >> 3 by
 *  (rows + 7) >> 3 superblocks; tables[sb_rows * sb_cols] in device memory in raster order.  Nothing outside the picture's
 *  8 cols x 8 rows luma samples (half of that per chroma plane) is read or written — the reference touches no more, and the
 *  decoder's frame buffers are not padded to whole superblocks.  Planes and strides 4-byte aligned.  Asynchronous on `stream`; a
 *  lost hand-off is reported by ffhip_stream_synchronize. */
int ffhip_vp9_loopfilter_frame_dev(int bit_depth, uint8_t *y, uint8_t *u, uint8_t *v, ptrdiff_t stride_y, ptrdiff_t stride_uv, int cols,
                                   int rows, const FFHipVp9LfSb *tables, void *stream);
/** The same with the picture's chroma sub-sampling (VP9Context.ss_h / .ss_v).  1, 1: the call above.  0, 0 (4:4:4, profiles 1 / 3): the
 *  chroma planes are filtered exactly as luma — ff_vp9_loopfilter_sb passes luma's masks (uv_masks = lflvl->mask[ss_h | ss_v]) and
 *  levels with ss 0 (vp9lpf.c:185-201) — so all three planes use tables[].y and three luma chains run side by side; tables[].uv is not
 *  read.  4:4:0 / 4:2:2: FFHIP_EINVAL here (they need the chroma tables: ffhip_vp9_loopfilter_frame_ssc_dev). */
int ffhip_vp9_loopfilter_frame_ss_dev(int bit_depth, int ss_h, int ss_v, uint8_t *y, uint8_t *u, uint8_t *v, ptrdiff_t stride_y,
                                      ptrdiff_t stride_uv, int cols, int rows, const FFHipVp9LfSb *tables, void *stream);
/** 4:2:2 (ss_h 1, ss_v 0) and 4:4:0 (0, 1) — VP9 profiles 1 / 3 with rectangular chroma superblocks (round 4): luma by tables[].y, both
 *  chroma planes by ctables[] (one per superblock, raster order, device memory).  A plain kernel (one wave per superblock row and plane);
 *  one picture per launch. */
int ffhip_vp9_loopfilter_frame_ssc_dev(int bit_depth, int ss_h, int ss_v, uint8_t *y, uint8_t *u, uint8_t *v, ptrdiff_t stride_y,
                                       ptrdiff_t stride_uv, int cols, int rows, const FFHipVp9LfSb *tables, const FFHipVp9LfSbC *ctables,
                                       void *stream);
/** N pictures of one geometry in ONE launch (round 4; what a decoder's frame threads hold at once): each picture its own planes and
 *  tables (device pointers), strides shared.  A picture's filter is a dependency chain through it (0.9 ms for a 4K picture on a chip it
 *  cannot fill); the pictures of a batch are filtered side by side.  ss_h == ss_v (4:2:0 / 4:4:4).  `pics` is a host array. */
typedef struct FFHipVp9LfPic {
    uint8_t *y, *u, *v;
    const FFHipVp9LfSb *tables;
} FFHipVp9LfPic;
int ffhip_vp9_loopfilter_frames_dev(int bit_depth, int ss_h, int ss_v, int npics, const FFHipVp9LfPic *pics, ptrdiff_t stride_y,
                                    ptrdiff_t stride_uv, int cols, int rows, void *stream);
/** The same for the formats with rectangular chroma superblocks — 4:2:2 (ss_h 1, ss_v 0) and 4:4:0 (0, 1) (round 5): every picture brings
 *  its chroma tables as well (ffhip_vp9_lf_sb_ctables()); ffhip_vp9_loopfilter_frame_ssc_dev()'s plain kernel with the pictures of the batch
 *  side by side (libavcodec/vp9lpf.c:27-203).  ffhip_vp9_loopfilter_frames_dev() answers FFHIP_EINVAL for these formats. */
typedef struct FFHipVp9LfPicC {
    uint8_t *y, *u, *v;
    const FFHipVp9LfSb *tables;
    const FFHipVp9LfSbC *ctables;
} FFHipVp9LfPicC;
int ffhip_vp9_loopfilter_frames_ssc_dev(int bit_depth, int ss_h, int ss_v, int npics, const FFHipVp9LfPicC *pics, ptrdiff_t stride_y,
                                        ptrdiff_t stride_uv, int cols, int rows, void *stream);

/**
 * vp9dsp above 8 bits (profiles 2 / 3): the batch faces above at the bpp ff_vp9dsp_init(dsp, bpp, bitexact) instantiates its template
 * for (libavcodec/vp9dsp.c:88-112, vp9dsp_10bpp.c / vp9dsp_12bpp.c).  bit_depth 8, 10 or 12.  Samples are uint16_t above 8 bits,
 * itxfm_add's coefficients int32_t (the reference's dctcoef; FFHipVp9TU.coeff_offset counts coefficients) and its butterflies run
 * in 64 bits (dctint = int64_t); record offsets and strides stay in BYTES; E / I / H of the loop filter are the 8-bit-unit values
 * the decoder passes (scaled by << (bit_depth - 8) inside, vp9dsp_template.c:1784-1788); FFHipVp9Intra.edge_offset addresses an
 * edge line of uint16_t samples.
 */
int ffhip_vp9_itxfm_add_batch_dev_hbd(int bit_depth, int tx, void *coeffs, uint8_t *dst, ptrdiff_t stride, const FFHipVp9TU *tus, int n,
                                      void *stream);
int ffhip_vp9_mc_batch_dev_hbd(int bit_depth, uint8_t *dst, ptrdiff_t dststride, const uint8_t *src, ptrdiff_t srcstride,
                               const FFHipVp9McBlock *blocks, int n, void *stream);
int ffhip_vp9_scaled_mc_batch_dev_hbd(int bit_depth, uint8_t *dst, ptrdiff_t dststride, const uint8_t *src, ptrdiff_t srcstride,
                                      const FFHipVp9ScaledBlock *blocks, int n, void *stream);
int ffhip_vp9_loop_filter_batch_dev_hbd(int bit_depth, uint8_t *base, ptrdiff_t stride, const FFHipVp9Edge *edges, int n, void *stream);
int ffhip_vp9_intra_pred_batch_dev_hbd(int bit_depth, int tx, uint8_t *dst, ptrdiff_t stride, const uint8_t *edges,
                                       const FFHipVp9Intra *blocks, int n, void *stream);



/* ------------------------------------------------------------------------------------------ */
/* libavcodec: me_cmp + full search                                                           */
/* ------------------------------------------------------------------------------------------ */
/** me_cmp_func (libavcodec/me_cmp.h:45-48); the context argument is unused by these metrics
 *  (checkasm passes NULL, tests/checkasm/motion.c:75). */
typedef int (*ffhip_me_cmp_func)(void *c, const uint8_t *blk1, const uint8_t *blk2, ptrdiff_t stride, int h);
typedef struct FFHipMECmpContext {
    ffhip_me_cmp_func sad[2];             /* [0] 16 wide = pix_abs16_c, [1] 8 wide = pix_abs8_c */
    ffhip_me_cmp_func hadamard8_diff[2];  /* [0] hadamard8_diff16_c, [1] hadamard8_diff8x8_c     */
    ffhip_me_cmp_func pix_abs[2][1];      /* [w][0] full-pel                                     */
    /* what a motion search uses beside them (me_cmp.c:961-1012): the half-pel SADs pix_abs[w][1..3] = _x2 / _y2 / _xy2 (blk2 read
     * one column / row / both further), sse[0..1] and nsse[0..1] ([0] 16 wide, [1] 8 wide).  nsse weighs its second term with the
     * encoder's nsse_weight when the context argument is non-NULL: such calls go to the displaced C function, NULL means 8 */
    ffhip_me_cmp_func pix_abs_hpel[2][3];
    ffhip_me_cmp_func sse[2];
    ffhip_me_cmp_func nsse[2];
} FFHipMECmpContext;
int ff_me_cmp_init_hip(FFHipMECmpContext *c);

#define FFHIP_ME_SAD   0
#define FFHIP_ME_SATD  1
#define FFHIP_ME_SAD_X2  2   /* pix_abs*_x2_c: blk2 averaged with its right neighbour (w + 1 columns read) */
#define FFHIP_ME_SAD_Y2  3   /* pix_abs*_y2_c: ... with the row below (h + 1 rows read)                  */
#define FFHIP_ME_SAD_XY2 4   /* pix_abs*_xy2_c: ... with all three                                       */
#define FFHIP_ME_SSE     5   /* sse16_c / sse8_c                                                          */
#define FFHIP_ME_NSSE    6   /* nsse16_c / nsse8_c with the context-free weight 8                         */
/** n independent comparisons: out[i] = cmp(blk1 + off1[i], blk2 + off2[i], stride, h). width 16|8. */
int ffhip_me_cmp_batch_dev(int kind, int width, int h, const uint8_t *blk1, const int32_t *off1,
                           const uint8_t *blk2, const int32_t *off2, ptrdiff_t stride, int32_t *out, int n,
                           void *stream);
/**
 * Exhaustive search, semantics of ff_me_search_esa() driven as vf_mestimate does
 * (libavfilter/motion_estimation.c:78-95, COST_MV :32-40; libavfilter/vf_mestimate.c:101,119-129):
 * for every mb_size x mb_size block of `cur`, window [x_mb±R]∩[0,(b_w-1)*mb]×[y_mb±R]∩[0,(b_h-1)*mb],
 * zero-MV evaluated first (and returned at once when its cost is 0), then raster scan with strict <.
 * cost_kind FFHIP_ME_SAD reproduces the filter bit-exactly; FFHIP_ME_SATD swaps the metric for
 * hadamard8_diff (libavcodec/me_cmp.c:514-562,933-950) under the same search order.
 * Outputs per block (raster): mv_out[2*b+{0,1}] = absolute best (x,y) as int16, cost_out[b] u32.
 * nframes frame pairs, frame f at cur + f*frame_pitch / ref + f*frame_pitch.
 */
int ffhip_me_esa_batch_dev(const uint8_t *cur, const uint8_t *ref, int width, int height, ptrdiff_t stride,
                           size_t frame_pitch, int nframes, int mb_size, int search_param, int cost_kind,
                           int16_t *mv_out, uint32_t *cost_out, void *stream);

/* ------------------------------------------------------------------------------------------ */
/* libavutil: av_tx float MDCT                                                                */
/* ------------------------------------------------------------------------------------------ */
typedef struct FFHipTXContext FFHipTXContext;
#define FFHIP_TX_FLOAT_FFT  0   /* == AV_TX_FLOAT_FFT  (libavutil/tx.h:47-132) */
#define FFHIP_TX_FLOAT_MDCT 1   /* == AV_TX_FLOAT_MDCT                          */
#define FFHIP_TX_FLOAT_RDFT 6   /* == AV_TX_FLOAT_RDFT (r2c forward, c2r inverse; libavutil/tx.h:70-90) */
#define FFHIP_TX_FLOAT_DCT  9   /* == AV_TX_FLOAT_DCT: DCT-II forward, DCT-III inverse (libavutil/tx.h:95-104), power of two 8..4096; as with av_tx_init the
                                   inverse is initialised with half the number of samples it transforms */
#define FFHIP_TX_FLOAT_DCT_I 12 /* == AV_TX_FLOAT_DCT_I, AV_TX_FLOAT_DST_I (libavutil/tx.h:107-128; ff_tx_dctI / ff_tx_dstI, tx_template.c:2006-2075): forward, */
#define FFHIP_TX_FLOAT_DST_I 15 /*    even len 4..1024 (64 in libavcodec/wmavoice.c:398-404); len reals in — `stride` bytes apart — len reals out; outputs as the
                                 *    C code's, its two middle ones at *scale != 1 included (kernels/tx_dcst1.hip); inverse contexts: FFHIP_ENOSYS */
#define FFHIP_TX_DOUBLE_FFT  2  /* == AV_TX_DOUBLE_FFT,  AV_TX_DOUBLE_MDCT (libavutil/tx.h:48-58; tx_double.c): rows of double, *scale a double */
#define FFHIP_TX_DOUBLE_MDCT 3
#define FFHIP_TX_INT32_FFT   4  /* == AV_TX_INT32_FFT, AV_TX_INT32_MDCT (libavutil/tx.h:59-69; tx_int32.c): rows of int32_t, *scale a float; the */
#define FFHIP_TX_INT32_MDCT  5  /*    fixed-point arithmetic of tx_priv.h:117-147 (64-bit products rounded to nearest, wrapping sums, the MDCT's
                                 *    input folded with >> 6).  Powers of two only: FFT 4 .. 8192 (double) / 16384 (int32) complex points, MDCT
                                 *    len 16 .. twice that; contiguous rows aligned to a complex sample; bit-identical to the C codelets
                                 *    (kernels/tx_wide.hip).  AV_TX_FULL_IMDCT is float-only. */
/* Refused — ffhip_tx_init() returns FFHIP_ENOSYS and the caller keeps the C / SIMD codelets (the reference's convention for an arch
 * that does not offer a transform: its codelet list simply has no entry, libavutil/tx.c:593-650):
 *   the RDFT / DCT / DCT-I / DST-I forms of the double and int32 types (7, 8, 10, 11, 13, 14, 16, 17) and their prime-factor lengths;
 *   inverse contexts of AV_TX_FLOAT_DCT_I / _DST_I (ff_tx_dcstI_init doubles their length, tx_template.c:2017-2021), their len 2 and len > 1024. */
#define FFHIP_TX_FULL_IMDCT        (1ULL << 2)   /* == AV_TX_FULL_IMDCT: an inverse MDCT writes 2 * len outputs (ff_tx_mdct_inv_full,
                                                  * libavutil/tx_template.c:1391-1408); batches: 8-byte aligned rows of 2 * len floats */
#define FFHIP_TX_REAL_TO_REAL      (1ULL << 3)   /* == AV_TX_REAL_TO_REAL: a forward RDFT writes the len/2 + 1 real parts only (ff_tx_rdft_r2r,
                                                  * libavutil/tx_template.c:1718-1827)                                                  */
#define FFHIP_TX_REAL_TO_IMAGINARY (1ULL << 4)   /* == AV_TX_REAL_TO_IMAGINARY: ... the len/2 imaginary parts only (ff_tx_rdft_r2i, :1829); the last one
                                                  * is, as in the reference, the underlying FFT's own value.  Both forward-only (EINVAL otherwise) */
#define FFHIP_TX_BITEXACT          (1ULL << 32)  /* libffhip's own: a bit neither AVTXFlags (bits 0..4, libavutil/tx.h:134-166) nor the
                                                  * codelet-private FF_TX_* flags (bits 58..63, tx_priv.h:154-159) use; av_tx_init() rejects
                                                  * unknown bits, so the wrapper never forwards it: float FFT (256 … 16384 points) and MDCT / RDFT / DCT contexts on 256, 512 and 1024 complex points run, by
                                                  * default, a radix-16 / -8 / -4 factorisation held in registers (kernels/tx_radix.hip) whose
                                                  * results agree with the C codelets within 2^-18 of a transform's largest output — the
                                                  * position FFmpeg's own SIMD codelets are in (tests/checkasm/av_tx.c compares with an
                                                  * epsilon).  With this flag the context runs the split-radix network in the C reference's
                                                  * operation order instead: bit-identical to ff_tx_*_float_c, at about half the rate.  Every
                                                  * other length and type is bit-identical either way. */
/** av_tx_fn (libavutil/tx.h:151) with an opaque context of ours in place of AVTXContext. */
typedef void (*ffhip_tx_fn)(FFHipTXContext *s, void *out, void *in, ptrdiff_t stride);
/**
 * Same argument meaning as av_tx_init() (libavutil/tx.h:169-172, libavutil/tx.c:903): type, inv,
 * len (MDCT: number of output coefficients of the forward transform, power of two 16..32768 (above 4096: contiguous 8-byte aligned
 * rows, one workgroup per transform) or one of the prime-factor
 * lengths 2 * 15 * 2^k, k = 2..6 = 120 / 240 / 480 / 960 / 1920 (CELT, AAC-960) and 2 * F * 2^k, F = 3 / 5 / 7 / 9, k = 2..8 (96- and
 * 768-sample AAC frames, Siren's 320, ...): ff_tx_mdct_pfa_<F>xM, libavutil/tx_template.c:1425-1600 — their batches must be
 * contiguous 8-byte aligned rows); FFT: number of complex samples, power of two 4..16384 or F * 2^k for F = 3 / 5 / 7 / 9 (k = 2..8) and
 * 15 (k = 2..7: 60 .. 1920) — ff_tx_fft_pfa over fft<F>_ns, libavutil/tx_template.c:948-1101; RDFT: number of real samples,
 * power of two 8..4096 — forward (r2c) reads len floats and writes len/2 + 1 complex bins, inverse (c2r) the other way round,
 * ff_tx_rdft_r2c / _c2r, libavutil/tx_template.c:1601-1716), *scale (FFT: ignored, may be NULL).  This is what the FFTXCodelet.init of a `ff_tx_codelet_list_float_hip[]` entry runs
 * (libavutil/tx_priv.h:199-237).  *fn receives the single-transform host-pointer shim.
 */
int  ffhip_tx_init(FFHipTXContext **ctx, ffhip_tx_fn *fn, int type, int inv, int len, const void *scale,
                   uint64_t flags);
void ffhip_tx_uninit(FFHipTXContext **ctx);
/**
 * Batched device face: transform t reads in + t*in_pitch floats... pitches in BYTES; out stride is
 * the av_tx_fn `stride` (bytes between output coefficients; sizeof(float) for contiguous).
 * Forward MDCT: in = 2*len floats, out = len floats.  Inverse: in = len floats (read with
 * `in_stride` bytes between coefficients), out = len floats (half-window iMDCT as the reference's
 * default; libavutil/tx_template.c:1312-1342).  FFT (either direction): in = out = len complex (re, im) floats,
 * contiguous and 8-byte aligned, `stride` ignored as in the reference (libavutil/tx_template.c:735-749); unnormalised.
 */
int  ffhip_tx_batch_dev(FFHipTXContext *ctx, void *out, size_t out_pitch, const void *in, size_t in_pitch,
                        ptrdiff_t stride, int ntransforms, void *stream);

/* =====================================================================================================================
 * swscale SwsOpBackend "hip" (SURVEY.md §8 f-1): the micro-op lists libswscale's new format layer hands a backend
 * (SwsOpBackend.compile / .compile_uops, libswscale/ops_dispatch.h:135-156) become one generated gfx950 kernel each.
 *
 * The structs below are LAYOUT-IDENTICAL to the reference's (the FFmpeg-side stub passes its own pointers through a cast and
 * static_asserts the sizes; oracle/refbuild/ffref_shim_ops.c holds those asserts for this repository's tests):
 *   FFHipSwsPixel == SwsPixel (uops.h:81-88), FFHipSwsFilterWeights == SwsFilterWeights (filters.h:83-121),
 *   FFHipSwsUOp == SwsUOp (uops.h:262-281), FFHipSwsOpExec == SwsOpExec (ops_dispatch.h:36-85);
 *   FFHIP_SWS_PIXEL_* == SwsPixelType (uops.h:43-50), FFHIP_SWS_UOP_* == SwsUOpType (uops.h:129-184).
 * Semantics of every micro-op are the C template backend's (libswscale/uops_tmpl.c, compiled without FP contraction,
 * uops_backend.c:24-35): results are bit-identical to backend_c for every type, floats included.
 * ===================================================================================================================== */
#define FFHIP_ENOTSUP (-95)   /* == AVERROR(ENOTSUP): the caller tries its next backend / splits the list (ops_dispatch.c:744-766) */

enum { FFHIP_SWS_PIXEL_NONE = 0, FFHIP_SWS_PIXEL_U8, FFHIP_SWS_PIXEL_U16, FFHIP_SWS_PIXEL_U32, FFHIP_SWS_PIXEL_F32 };
enum {
    FFHIP_SWS_UOP_INVALID = 0,
    FFHIP_SWS_UOP_READ_PLANAR, FFHIP_SWS_UOP_READ_PLANAR_FH, FFHIP_SWS_UOP_READ_PLANAR_FV, FFHIP_SWS_UOP_READ_PLANAR_FV_FMA,
    FFHIP_SWS_UOP_READ_PACKED, FFHIP_SWS_UOP_READ_NIBBLE, FFHIP_SWS_UOP_READ_BIT, FFHIP_SWS_UOP_READ_PALETTE,
    FFHIP_SWS_UOP_WRITE_PLANAR, FFHIP_SWS_UOP_WRITE_PACKED, FFHIP_SWS_UOP_WRITE_NIBBLE, FFHIP_SWS_UOP_WRITE_BIT,
    FFHIP_SWS_UOP_RW_SHUFFLE, FFHIP_SWS_UOP_PERMUTE, FFHIP_SWS_UOP_COPY,
    FFHIP_SWS_UOP_SWAP_BYTES, FFHIP_SWS_UOP_EXPAND_BIT, FFHIP_SWS_UOP_EXPAND_PAIR, FFHIP_SWS_UOP_EXPAND_QUAD,
    FFHIP_SWS_UOP_TO_U8, FFHIP_SWS_UOP_TO_U16, FFHIP_SWS_UOP_TO_U32, FFHIP_SWS_UOP_TO_F32,
    FFHIP_SWS_UOP_SCALE, FFHIP_SWS_UOP_ADD, FFHIP_SWS_UOP_MIN, FFHIP_SWS_UOP_MAX,
    FFHIP_SWS_UOP_UNPACK, FFHIP_SWS_UOP_PACK, FFHIP_SWS_UOP_LSHIFT, FFHIP_SWS_UOP_RSHIFT, FFHIP_SWS_UOP_CLEAR,
    FFHIP_SWS_UOP_LINEAR, FFHIP_SWS_UOP_LINEAR_FMA, FFHIP_SWS_UOP_DITHER, FFHIP_SWS_UOP_LUT_3D,
    FFHIP_SWS_UOP_TYPE_NB
};
#define FFHIP_SWS_FILTER_SCALE (1 << 14)          /* SWS_FILTER_SCALE, filters.h:39 */

typedef union FFHipSwsPixel { char data[4]; uint8_t u8; uint16_t u16; uint32_t u32; float f32; } FFHipSwsPixel;

typedef struct FFHipSwsFilterWeights {
    int     filter_size;              /* taps per output sample */
    int    *weights;                  /* [dst_size][filter_size], scaled by FFHIP_SWS_FILTER_SCALE */
    size_t  num_weights;
    int    *offsets;                  /* first source sample of every output sample */
    int     src_size, dst_size;
    double  virtual_size, offset;
    char    name[16];
    int     sum_positive, sum_negative;
} FFHipSwsFilterWeights;

typedef union FFHipSwsUOpParams {
    struct { uint8_t clear_value, read_size, write_size; } shuffle;
    struct { int32_t type; } filter;                               /* READ_PLANAR_FH / _FV: type the result is stored as */
    struct { uint8_t amount; } shift;
    struct { int32_t num_moves; int8_t dst[6], src[6]; } move;     /* PERMUTE / COPY; register -1 is a temporary */
    struct { uint8_t pattern[4]; } pack;
    struct { uint8_t one, zero; } clear;
    struct { uint32_t one, zero, exact; } lin;                     /* bit 5 * row + column */
    struct { uint8_t y_offset[4]; uint8_t size_log2; } dither;
    struct { int32_t dynamic; } lut3d;
} FFHipSwsUOpParams;

typedef struct FFHipSwsUOp {
    int32_t type;                     /* FFHIP_SWS_PIXEL_* */
    int32_t uop;                      /* FFHIP_SWS_UOP_* */
    uint8_t mask;                     /* components, bit c */
    FFHipSwsUOpParams par;
    union {
        FFHipSwsFilterWeights *kernel;
        FFHipSwsPixel *ptr;           /* DITHER: (1 << size_log2) columns, (1 << size_log2) + max(y_offset) rows */
        FFHipSwsPixel scalar;
        FFHipSwsPixel vec4[4];
        FFHipSwsPixel mat4[4][5];
        struct { int8_t mask[16]; uint8_t pixels; } shuffle;
        const void *lut3d;
        void *opaque;
    } data;
} FFHipSwsUOp;

typedef struct FFHipSwsOpExec {
    const uint8_t *in[4];
    uint8_t *out[4];
    ptrdiff_t in_stride[4], out_stride[4];
    ptrdiff_t in_bump[4], out_bump[4];
    int32_t width, height, slice_y, slice_h;
    int32_t block_size_in[4], block_size_out[4];
    uint8_t in_sub_y[4], out_sub_y[4], in_sub_x[4], out_sub_x[4];
    int32_t *in_bump_y;               /* READ_PLANAR_FV: extra source lines after output line y (absolute y) */
    int32_t *in_offset_x;             /* READ_PLANAR_FH: byte offset of the first tap of output sample x (absolute x) */
} FFHipSwsOpExec;

/** SwsOpFunc (ops_dispatch.h:93-99). */
typedef void (*FFHipSwsOpFunc)(const FFHipSwsOpExec *exec, const void *priv, int bx_start, int y_start, int bx_end, int y_end);

typedef struct FFHipSwsUOps FFHipSwsUOps;   /* one compiled micro-op list: what SwsCompiledOp.priv holds */

/** SwsOpBackend.compile_uops (ops_dispatch.h:148): generates, compiles (hiprtc, cached by program text) and loads the kernel of
 *  the list.  FFHIP_ENOTSUP for lists this backend does not take (LUT_3D, RW_SHUFFLE and the _FMA variants — a translation with
 *  flags 0, which is what bit-exactness with backend_c needs, never emits the latter two), FFHIP_EINVAL for malformed ones,
 *  FFHIP_ENOSYS without a device. */
int  ffhip_sws_uops_compile(const FFHipSwsUOp *uops, int num_uops, FFHipSwsUOps **out);
void ffhip_sws_uops_free(FFHipSwsUOps **p);
/** SwsCompiledOp.block_size: pixels per block of bx_start / bx_end — 1, or what makes a block whole bytes (8 for the 1-bit,
 *  2 for the 4-bit reads and writes).  over_read / over_write are 0 for every list: no thread touches a byte outside the blocks. */
int  ffhip_sws_uops_block_size(const FFHipSwsUOps *p);
/** The program text the list compiles to (diagnostics, and the CPU-side test of the generator); returns its length or < 0. */
int  ffhip_sws_uops_source(const FFHipSwsUOp *uops, int num_uops, char *buf, size_t size);
/** Generates and compiles the program without loading it (no device needed): 0, FFHIP_ENOTSUP, FFHIP_EINVAL or FFHIP_EIO. */
int  ffhip_sws_uops_check(const FFHipSwsUOp *uops, int num_uops);
/** Compiled programs are kept per process (by program text) and — once the application names a directory — across processes as
 *  code objects on disk, so that the ~25 ms hiprtc compile of a new op list is paid once per machine.  NULL or "" turns the disk
 *  cache off, which is the default: the library reads no environment variable; the FFmpeg-side backend chooses
 *  $XDG_CACHE_HOME/ffhip or $HOME/.cache/ffhip (integration/swscale_hw_hip.c).  The directory (and its parent) is created on the first
 *  store.  A file carries its whole key (hiprtc version, architecture, program text): a stale, damaged or foreign file is a miss and
 *  is rewritten; files appear by rename, so concurrent processes are safe.  The directory is used only while it is a real directory
 *  (lstat: no symlink) OWNED BY THE EFFECTIVE USER and not writable by group or others — code objects loaded from it run with the
 *  process's access to its device memory — else the cache silently stays off; files are opened O_NOFOLLOW, must be the user's own
 *  regular files, carry a hash of the code object that is verified on load, and are written through mkstemp().  Process-wide; returns 0. */
int  ffhip_sws_uops_set_cache_dir(const char *dir);
/** hiprtc compiles and disk hits of this process so far (either pointer may be NULL). */
void ffhip_sws_uops_cache_stats(long *compiles, long *disk_hits);
/** What the void face below runs when the device fails under it (the same list compiled by the caller's C backend). */
void ffhip_sws_uops_set_fallback(FFHipSwsUOps *p, FFHipSwsOpFunc func, const void *priv);
/** SwsCompiledOp.func for software frames: HOST pointers in `exec`, priv = the FFHipSwsUOps.  Stages exactly the bytes the
 *  C backend would read, runs the kernel, commits exactly the bytes it would write. */
void ffhip_sws_uops_func(const FFHipSwsOpExec *exec, const void *priv, int bx_start, int y_start, int bx_end, int y_end);
/** The device-resident face (an AV_PIX_FMT_HIP frame pool, SwsPass.run of an opaque compiled op): `exec->in / out` are DEVICE
 *  pointers, the two small tables (in_bump_y, in_offset_x) stay host arrays as the dispatcher built them.  `nframes` pictures
 *  that share the geometry run in one launch, plane i of picture f at in[i] + f * in_frame_pitch[i] (NULL pitches: 1 picture).
 *  Asynchronous on `stream`. */
int  ffhip_sws_uops_run_dev(FFHipSwsUOps *p, const FFHipSwsOpExec *exec, int bx_start, int y_start, int bx_end, int y_end,
                            int nframes, const ptrdiff_t *in_frame_pitch, const ptrdiff_t *out_frame_pitch, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* FFHIP_H */
