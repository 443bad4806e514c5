/*
 * integration/swscale_graph_hip.c — libswscale/graph.c for hip frames: the intermediate frames of a multi-pass graph in DEVICE memory.
 *
 * A conversion the graph cuts in two (two-dimensional scaling: one pass per filter direction, ops_dispatch.c:745-766) hands its
 * first pass's output to the second through a frame pass_alloc_output() allocates (graph.c:130-175) — with av_buffer_alloc(), host
 * memory, for every device type but Vulkan (pass_alloc_output_hw(), graph.c:105-128).  The patch gives hip the same branch Vulkan
 * has; expressed without touching the reference file: graph.c is compiled unchanged, where it lies, and for the duration of the
 * include
 *   - av_frame_alloc(), called once, at the top of pass_alloc_output() with `pass` in scope, also notes whether the pass belongs to a
 *     graph between two hip frames (SwsGraph.src / .dst .hw_format), and
 *   - av_buffer_alloc(), called once, for a plane of that frame, then returns an AVBufferRef over ffhip_malloc()'ed memory of the
 *     calling thread's current device (the hwcontext made it current: integration/avutil_hwcontext_hip.c).
 * The recipe (oracle/refbuild/Makefile) checks that graph.c calls each of the two exactly once.
 */
#include "config.h"
#include <stddef.h>
#include <stdint.h>
#include "libavutil/buffer.h"
#include "libavutil/frame.h"
#include "libavutil/pixfmt.h"
#include "avutil_hwcontext_hip.h"
#include "ffhip.h"

struct SwsPass;
static AVBufferRef *ffhip_graph_buffer_alloc(size_t size);
static void ffhip_graph_note_pass(const struct SwsPass *pass);

#define av_buffer_alloc ffhip_graph_buffer_alloc
#define av_frame_alloc() (ffhip_graph_note_pass(pass), (av_frame_alloc)())
#include "libswscale/graph.c"
#undef av_frame_alloc
#undef av_buffer_alloc

static _Thread_local int hip_graph; /* the frame being allocated belongs to a graph between two hip frames */
static long device_intermediates;
long ffhip_integration_device_intermediates(void) { return device_intermediates; }

static void ffhip_graph_note_pass(const struct SwsPass *pass)
{
    const SwsGraph *graph = pass->graph;
    hip_graph = graph->src.hw_format == FFHIP_HW_PIX_FMT && graph->dst.hw_format == FFHIP_HW_PIX_FMT;
}

static void device_free(void *opaque, uint8_t *data)
{
    (void)opaque;
    ffhip_free(data);
}

static AVBufferRef *ffhip_graph_buffer_alloc(size_t size)
{
    void *p = NULL;
    AVBufferRef *ref;
    if (!hip_graph)
        return av_buffer_alloc(size);
    if (ffhip_malloc(&p, size) < 0 || !p)
        return NULL;
    ref = av_buffer_create(p, size, device_free, NULL, 0);
    if (!ref)
        ffhip_free(p);
    else
        __atomic_fetch_add(&device_intermediates, 1, __ATOMIC_RELAXED);
    return ref;
}
