/*
 * integration/avutil_hip.c — libavutil/hip/{float_dsp,tx_float}_init.c of the FFmpeg-side patch.
 *
 * float_dsp: avpriv_float_dsp_alloc() ends with the ARCH_* chain (libavutil/float_dsp.c:153-165); the reference file is compiled
 * where it lies with that function renamed to avpriv_float_dsp_alloc_c, the function below takes its name and adds the `hip` call.
 *
 * av_tx needs no hook: its dispatch is a list of codelet lists (libavutil/tx.c:340-352) filtered by cpu flag and ordered by
 * priority (tx.c:367-400,447,778).  The `hip` arch is one more list, ff_tx_codelet_list_float_hip below; avutil_tx_list_hip.c puts
 * it into codelet_list[].  A codelet's init that does not take a length (FFHIP_EINVAL / ENOSYS from ffhip_tx_init) returns an error
 * and av_tx moves on to the next codelet in priority order — the C one — exactly as for an x86 codelet that rejects a length.
 */
#include "libavutil/attributes.h"
#include "libavutil/cpu.h"
#include "libavutil/error.h"
#include "libavutil/float_dsp.h"
#include "libavutil/mem.h"
#include "libavutil/tx_priv.h"

#include "ffhip.h"
#include "hip_cpu.h"

AVFloatDSPContext *avpriv_float_dsp_alloc_c(int bit_exact);

av_cold AVFloatDSPContext *avpriv_float_dsp_alloc(int bit_exact)
{
    AVFloatDSPContext *fdsp = avpriv_float_dsp_alloc_c(bit_exact);
    FFHipFloatDSPContext h;
    if (!fdsp || !(av_get_cpu_flags() & AV_CPU_FLAG_HIP))
        return fdsp;
#define FDSP(X) X(vector_fmul) X(vector_fmac_scalar) X(vector_fmul_scalar) X(vector_fmul_window) X(vector_fmul_add) \
    X(vector_fmul_reverse) X(butterflies_float)
#define GIVE(m) h.m = fdsp->m;
#define TAKE(m) fdsp->m = h.m;
    FDSP(GIVE)
    if (ff_float_dsp_init_hip(&h) >= 0) {       /* the double members and scalarproduct_float keep their C pointers */
        FDSP(TAKE)
    }
    return fdsp;
}

/* ---- av_tx ------------------------------------------------------------------------------------------------------------------ */
typedef struct HipTx { FFHipTXContext *h; ffhip_tx_fn fn; } HipTx;   /* AVTXContext.opaque ("free to use by implementations") */

static av_cold int hip_tx_init(AVTXContext *s, const FFTXCodelet *cd, uint64_t flags, FFTXCodeletOptions *opts, int len, int inv,
                               const void *scale)
{
    HipTx *t;
    ffhip_tx_fn fn;
    FFHipTXContext *h = NULL;
    /* the AVTXFlags libffhip knows by the same bit values (include/ffhip.h); in-place is not offered (FF_TX_OUT_OF_PLACE) */
    const uint64_t f = flags & (AV_TX_FULL_IMDCT | AV_TX_REAL_TO_REAL | AV_TX_REAL_TO_IMAGINARY);
    const float one = 1.0f;
    const double one_d = 1.0;   /* the double types read *scale as a double (SCALE_TYPE) */
    const int is_double = cd->type == AV_TX_DOUBLE_FFT || cd->type == AV_TX_DOUBLE_MDCT;
    (void)opts;
    if (ffhip_tx_init(&h, &fn, cd->type, inv, len, scale ? scale : is_double ? (const void *)&one_d : (const void *)&one, f) < 0)
        return AVERROR(ENOSYS);                 /* next codelet in priority order is tried (tx.c:837-895) */
    if (!(t = av_malloc(sizeof(*t)))) {
        ffhip_tx_uninit(&h);
        return AVERROR(ENOMEM);
    }
    t->h = h;
    t->fn = fn;
    s->opaque = t;
    return 0;
}
static av_cold int hip_tx_uninit(AVTXContext *s)
{
    HipTx *t = s->opaque;
    if (t) {
        ffhip_tx_uninit(&t->h);
        av_freep(&s->opaque);
    }
    return 0;
}
static void hip_tx(AVTXContext *s, void *out, void *in, ptrdiff_t stride) /* av_tx_fn (tx.h:151): one transform per call */
{
    const HipTx *t = s->opaque;
    t->fn(t->h, out, in, stride);
}

#define HIP_CODELET(NAME, TYPE, FLAGS, F0, F1, NF, MIN, MAX)                                                                    \
    static const FFTXCodelet NAME = {                                                                                           \
        .name = #NAME, .function = hip_tx, .type = TYPE, .flags = AV_TX_UNALIGNED | FF_TX_OUT_OF_PLACE | (FLAGS),              \
        .factors = { F0, F1 }, .nb_factors = NF, .min_len = MIN, .max_len = MAX, .init = hip_tx_init, .uninit = hip_tx_uninit, \
        .cpu_flags = AV_CPU_FLAG_HIP, .prio = FF_TX_PRIO_MAX, /* "custom implementations/ASICs", tx_priv.h:168 */               \
    }
HIP_CODELET(ff_tx_fft_float_hip_def,       AV_TX_FLOAT_FFT,  0, 2, 0, 1, 4, 16384);
HIP_CODELET(ff_tx_mdct_float_hip_def,      AV_TX_FLOAT_MDCT, 0, 2, 0, 1, 16, 32768);
/* the prime-factor FFT lengths F x 2^k (ff_tx_fft_pfa over fft<F>_ns, tx_template.c:948-1101): 120 / 240 / 480 / 960 / 1920 ... */
HIP_CODELET(ff_tx_fft_pfa_15_float_hip_def, AV_TX_FLOAT_FFT, 0, 15, 2, 2, 60, 1920);
HIP_CODELET(ff_tx_fft_pfa_3_float_hip_def,  AV_TX_FLOAT_FFT, 0, 3, 2, 2, 12, 768);
HIP_CODELET(ff_tx_fft_pfa_5_float_hip_def,  AV_TX_FLOAT_FFT, 0, 5, 2, 2, 20, 1280);
HIP_CODELET(ff_tx_fft_pfa_7_float_hip_def,  AV_TX_FLOAT_FFT, 0, 7, 2, 2, 28, 1792);
HIP_CODELET(ff_tx_fft_pfa_9_float_hip_def,  AV_TX_FLOAT_FFT, 0, 9, 2, 2, 36, 2304);
/* the prime-factor MDCT lengths F x 2^k (ff_tx_mdct_pfa_<F>xM, tx_template.c:1425-1600): CELT 120..960, AAC-960 / -768, ... */
HIP_CODELET(ff_tx_mdct_pfa_15_float_hip_def, AV_TX_FLOAT_MDCT, 0, 15, 2, 2, 120, 1920);
HIP_CODELET(ff_tx_mdct_pfa_3_float_hip_def,  AV_TX_FLOAT_MDCT, 0, 3, 2, 2, 24, 1536);
HIP_CODELET(ff_tx_mdct_pfa_5_float_hip_def,  AV_TX_FLOAT_MDCT, 0, 5, 2, 2, 40, 2560);
HIP_CODELET(ff_tx_mdct_pfa_7_float_hip_def,  AV_TX_FLOAT_MDCT, 0, 7, 2, 2, 56, 3584);
HIP_CODELET(ff_tx_mdct_pfa_9_float_hip_def,  AV_TX_FLOAT_MDCT, 0, 9, 2, 2, 72, 4608);
HIP_CODELET(ff_tx_rdft_float_hip_def,      AV_TX_FLOAT_RDFT, 0, 4, 2, 2, 8, 4096);
HIP_CODELET(ff_tx_dctII_float_hip_def,     AV_TX_FLOAT_DCT,  FF_TX_FORWARD_ONLY, 2, TX_FACTOR_ANY, 2, 8, 4096);
HIP_CODELET(ff_tx_dctIII_float_hip_def,    AV_TX_FLOAT_DCT,  FF_TX_INVERSE_ONLY, 2, TX_FACTOR_ANY, 2, 8, 4096);
/* DCT-I / DST-I (wmavoice's 64; tx_template.c:2006-2105): forward contexts only — the C codelets' inverse reads twice the length */
HIP_CODELET(ff_tx_dctI_float_hip_def,      AV_TX_FLOAT_DCT_I, FF_TX_FORWARD_ONLY, 2, TX_FACTOR_ANY, 2, 4, 1024);
HIP_CODELET(ff_tx_dstI_float_hip_def,      AV_TX_FLOAT_DST_I, FF_TX_FORWARD_ONLY, 2, TX_FACTOR_ANY, 2, 4, 1024);

/* the other two sample types at power-of-two lengths (libffhip's kernels/tx_wide.hip).  They ride in the same list: av_tx walks every
 * codelet of every list and filters by .type (libavutil/tx.c:367-400), the list's name is a convention of the per-type files */
HIP_CODELET(ff_tx_fft_double_hip_def,      AV_TX_DOUBLE_FFT,  0, 2, 0, 1, 4, 8192);
HIP_CODELET(ff_tx_mdct_double_hip_def,     AV_TX_DOUBLE_MDCT, 0, 2, 0, 1, 16, 16384);
HIP_CODELET(ff_tx_fft_int32_hip_def,       AV_TX_INT32_FFT,   0, 2, 0, 1, 4, 16384);
HIP_CODELET(ff_tx_mdct_int32_hip_def,      AV_TX_INT32_MDCT,  0, 2, 0, 1, 16, 32768);

const FFTXCodelet * const ff_tx_codelet_list_float_hip[] = {
    &ff_tx_fft_float_hip_def, &ff_tx_fft_pfa_15_float_hip_def, &ff_tx_fft_pfa_3_float_hip_def, &ff_tx_fft_pfa_5_float_hip_def,
    &ff_tx_fft_pfa_7_float_hip_def, &ff_tx_fft_pfa_9_float_hip_def, &ff_tx_mdct_float_hip_def, &ff_tx_mdct_pfa_15_float_hip_def, &ff_tx_mdct_pfa_3_float_hip_def,
    &ff_tx_mdct_pfa_5_float_hip_def, &ff_tx_mdct_pfa_7_float_hip_def, &ff_tx_mdct_pfa_9_float_hip_def, &ff_tx_rdft_float_hip_def,
    &ff_tx_dctII_float_hip_def, &ff_tx_dctIII_float_hip_def, &ff_tx_dctI_float_hip_def, &ff_tx_dstI_float_hip_def,
    &ff_tx_fft_double_hip_def, &ff_tx_mdct_double_hip_def, &ff_tx_fft_int32_hip_def, &ff_tx_mdct_int32_hip_def, NULL,
};
