/*
 * integration/swscale_hip.c — libswscale/hip/swscale_hip.c of the FFmpeg-side patch: the per-line members of SwsInternal.
 *
 * ff_sws_init_scale() is sws_init_swscale() followed by the ARCH_* chain (libswscale/swscale.c:697-714); the reference file is
 * compiled where it lies with the function renamed to ff_sws_init_scale_c, the function below takes its name and adds the `hip`
 * call behind the chain.  What libffhip replaces are the 8-bit members sws_init_swscale() / ff_sws_init_output_funcs() choose by
 * bit depth, not by pixel format (swscale.c:608-611: srcBpc 8 and dstBpc <= 14 -> hScale8To15_c; output.c:3261-3275: 8-bit targets
 * -> yuv2plane1_8_c / yuv2planeX_8_c, semi-planar ones also yuv2nv12cX_c), so the hook asks for the planar (resp. NV12) set and
 * takes the members whose C twin is the one in place.  tests/checkasm/sw_scale.c then runs against the hip arch unchanged.
 * The ops backend ("hip" in ff_sws_op_backends[]) is oracle/refbuild/ffref_shim_ops.c.  Round 5: the function also installs the
 * frame-level SwsFunc of a SCALED context (ff_sws_hip_scaled_hook(), integration/swscale_unscaled_hip.c): this is where the banks are ready.
 */
#include "libavutil/attributes.h"
#include "libavutil/cpu.h"
#include "libavutil/pixdesc.h"
#include "libswscale/swscale_internal.h"

#include "ffhip.h"
#include "hip_cpu.h"

void ff_sws_init_scale_c(SwsInternal *c);
void ff_sws_hip_scaled_hook(SwsInternal *c); /* integration/swscale_unscaled_hip.c: the frame-level SwsFunc of a scaled context */

av_cold void ff_sws_init_scale(SwsInternal *c)
{
    FFHipSwsLineContext l;
    const enum AVPixelFormat dst = c->opts.dst_format;
    const int nv = dst != AV_PIX_FMT_NONE && isSemiPlanarYUV(dst);
    ff_sws_init_scale_c(c);
    if (!(av_get_cpu_flags() & AV_CPU_FLAG_HIP))
        return;
    l.hyScale    = (void *)c->hyScale;          /* in: the C functions become libffhip's fallbacks */
    l.hcScale    = (void *)c->hcScale;
    l.yuv2plane1 = c->yuv2plane1;
    l.yuv2planeX = c->yuv2planeX;
    l.yuv2nv12cX = (void *)c->yuv2nv12cX;
    ff_sws_hip_scaled_hook(c);   /* the frame hook first: a context that gets it never runs the line members (checkasm calls them directly) */
    if (ff_sws_init_swscale_hip(&l, FFHIP_PIX_FMT_YUV420P, nv ? FFHIP_PIX_FMT_NV12 : FFHIP_PIX_FMT_YUV420P) < 0)
        return;
    if (c->srcBpc == 8 && c->dstBpc <= 14) {     /* one function serves luma and chroma, as hScale8To15_c does (swscale.c:608-611) */
        c->hcScale = c->hyScale == c->hcScale ? (void *)l.hyScale : (void *)l.hcScale;
        c->hyScale = (void *)l.hyScale;
    }
    if (c->dstBpc == 8 && !c->use_mmx_vfilter) {
        c->yuv2plane1 = l.yuv2plane1;
        c->yuv2planeX = l.yuv2planeX;
        if (nv && c->yuv2nv12cX)
            c->yuv2nv12cX = (void *)l.yuv2nv12cX;
    }
}
