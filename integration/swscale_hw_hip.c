#define _GNU_SOURCE /* secure_getenv */
/*
 * integration/swscale_hw_hip.c — libswscale/hip/ops_hw.c of the FFmpeg-side patch: the `hip` SwsOpBackend for HARDWARE frames.
 *
 * SwsOpBackend.hw_format (libswscale/ops_dispatch.h:150-155) says which hardware pixel format a backend takes; ff_sws_ops_compile
 * (ops_dispatch.c:106-127) only offers a list to the backends whose hw_format equals the frames'.  This backend declares the hip
 * frames of avutil_hwcontext_hip.c: sws_scale_frame(ctx, dst, src) on two frames of an AVHWFramesContext of the hip device reaches
 * op_pass_run (ops_dispatch.c:403-500) with AVFrame.data[] — device pointers — in SwsOpExec.in / .out, and the compiled function
 * below hands them to ffhip_sws_uops_run_dev: the conversion runs in HBM, nothing crosses PCIe.  over_read / over_write are 0 and
 * block_size is what ffhip_sws_uops_block_size says, so the dispatcher never takes its memcpy tail path on the device pointers.
 *
 * The same micro-op lowering as the software-frame backend (flags 0: backend_c's own lowering, bit-exact with it).  SwsOpFunc carries
 * no stream, so launches go to the legacy default stream; the device context's transfer stream is a NON-blocking one
 * (ffhip_stream_create) and therefore not ordered against it by itself — avutil_hwcontext_hip.c's hip_transfer() says the order with
 * events on both sides (ffhip_stream_order: a transfer starts behind what the default stream has queued; after an upload that does
 * not wait, the default stream continues behind it).
 *
 * This build's list of backends is { hip (hw), c }: the software-frame `hip` backend (oracle/refbuild/ffref_shim_ops.c in this
 * repository's tests) sits between them in the real patch.
 */
#include "libavutil/log.h"
#include "libavutil/mem.h"
#include "libswscale/swscale.h"
#include "libswscale/ops.h"
#include "libswscale/ops_dispatch.h"
#include "libswscale/ops_internal.h"
#include "libswscale/uops.h"

#include <stdio.h>
#include <stdlib.h>

#include "libavutil/thread.h"

#include "ffhip.h"
#include "avutil_hwcontext_hip.h"

_Static_assert(sizeof(FFHipSwsOpExec) == sizeof(SwsOpExec) && sizeof(FFHipSwsUOp) == sizeof(SwsUOp), "layout contract of include/ffhip.h");

extern const SwsOpBackend backend_c;
static long hw_launches;
long ffhip_integration_hw_launches(void) { return hw_launches; }

static long hw_refused;
long ffhip_integration_hw_refused(void) { return hw_refused; }

static void hip_hw_func(const SwsOpExec *exec, const void *priv, int bx_start, int y_start, int bx_end, int y_end)
{
    /* A conversion the graph split into several passes (a list this backend does not take whole, e.g. both scaling filters:
     * ops_dispatch.c:745-766) runs through an intermediate frame, and pass_alloc_output (graph.c:130-175) allocates those in HOST
     * memory for every device type but Vulkan: the hip twin of its pass_alloc_output_hw is the part of the patch this build does not
     * carry.  Such a pass is refused here (logged, counted) rather than launched on a host pointer. */
    if (ffhip_pointer_device(exec->in[0]) < 0 || ffhip_pointer_device(exec->out[0]) < 0) {
        if (!__atomic_fetch_add(&hw_refused, 1, __ATOMIC_RELAXED))
            av_log(NULL, AV_LOG_ERROR, "hip_hw: a pass of this conversion runs through a host-memory intermediate frame; not run\n");
        return;
    }
    /* SwsOpFunc returns void: a failed launch leaves the error in ffhip_last_error() and the destination untouched */
    if (ffhip_sws_uops_run_dev((FFHipSwsUOps *)priv, (const FFHipSwsOpExec *)exec, bx_start, y_start, bx_end, y_end, 1, NULL, NULL, NULL) >= 0)
        __atomic_add_fetch(&hw_launches, 1, __ATOMIC_RELAXED);
}

static void hip_hw_free(void *priv)
{
    FFHipSwsUOps *u = priv;
    ffhip_sws_uops_free(&u);
}

/* Where compiled op lists are kept between processes (ffhip_sws_uops_set_cache_dir): the user's cache directory, as the Vulkan
 * pipeline cache and the shader caches of the drivers do.  libffhip itself reads no environment variable; this is the FFmpeg side. */
static AVOnce hip_uops_cache_once = AV_ONCE_INIT;
static void hip_uops_cache_dir(void)
{
    /* secure_getenv: a set-uid / set-gid or otherwise privileged process does not take the location of executable code from its
     * caller's environment; libffhip additionally refuses a directory that is not the effective user's own and closed to others
     * (ffhip_sws_uops_set_cache_dir) */
    const char *x = secure_getenv("XDG_CACHE_HOME"), *h = secure_getenv("HOME");
    char dir[1024];
    if (x && x[0])
        snprintf(dir, sizeof(dir), "%s/ffhip", x);
    else if (h && h[0])
        snprintf(dir, sizeof(dir), "%s/.cache/ffhip", h);
    else
        return;
    ffhip_sws_uops_set_cache_dir(dir);
}

static int compile_uops_hip_hw(SwsContext *ctx, const SwsUOpList *uops, SwsCompiledOp *out)
{
    FFHipSwsUOps *u = NULL;
    ff_thread_once(&hip_uops_cache_once, hip_uops_cache_dir);
    const int ret = ffhip_sws_uops_compile((const FFHipSwsUOp *)uops->ops, uops->num_ops, &u);
    if (ret < 0)
        return ret == FFHIP_ENOTSUP ? AVERROR(ENOTSUP) : ret == FFHIP_ENOSYS ? AVERROR(ENOSYS) : AVERROR(EINVAL);
    *out = (SwsCompiledOp) {
        .func        = hip_hw_func,
        .priv        = u,
        .free        = hip_hw_free,
        .slice_align = 1,
        .block_size  = ffhip_sws_uops_block_size(u),
    };
    return 0;
}

static int compile_hip_hw(SwsContext *ctx, const SwsOpList *ops, SwsCompiledOp *out)
{
    SwsUOpList *uops = ff_sws_uop_list_alloc();
    int ret;
    if (!uops)
        return AVERROR(ENOMEM);
    ret = ff_sws_ops_translate(ctx, ops, 0, uops);
    if (ret >= 0)
        ret = compile_uops_hip_hw(ctx, uops, out);
    ff_sws_uop_list_free(&uops);
    return ret;
}

#define SWS_BACKEND_HIP (1 << 6) /* the next free SwsBackend bit (swscale.h:112-127) */

const SwsOpBackend backend_hip_hw = {
    .name         = "hip_hw",
    .flags        = SWS_BACKEND_HIP,
    .compile      = compile_hip_hw,
    .compile_uops = compile_uops_hip_hw,
    .hw_format    = FFHIP_HW_PIX_FMT,
};

const SwsOpBackend *const ff_sws_op_backends[] = { &backend_hip_hw, &backend_c, NULL };
