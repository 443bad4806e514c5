/*
 * sws_api.hip — C-ABI entry points of the swscale part of libffhip (declared in include/ffhip.h).
 *
 * Mirrors, for the hot path only:
 *   sws_getContext()/sws_init_context()  -> ffhip_sws_getContext()      (libswscale/utils.c:1137,2043)
 *   SwsFunc c->convert_unscaled / whole-frame scale -> ffhip_sws_scale() (swscale_internal.h:99-101, swscale.c:1185)
 *   batched device-resident frames        -> ffhip_sws_scale_batch_dev() (no reference equivalent)
 */
#include <mutex>
#include <new>
#include <stdlib.h>
#include <string.h>
#include <vector>

#include "kernels/common.h"
#include "kernels/sws_kernels.h"
#include "kernels/shim_arena.h"

/* a packed 8-bit RGB source (round 6, kernels/sws_rgbin.hip): the context is that of the 14-bit planar source its converter lines make */
struct FFHipSwsRgbIn {
    int bpp = 0;            /* 0: none; 3 / 4 bytes per pixel */
    int half = 0;           /* the chroma converters average pixel pairs (4:2:2 lines) */
    int direct_c = 0;       /* ... the chroma banks the identity as well (a planar 4:2:2 / 4:4:4 target at the source's size): one elementwise pass */
    int fused420 = 0;       /* ... and the chroma too: RGB -> yuv420p / NV12 at the source's size in one kernel (k_sws_rgb420); vfv: its vertical chroma bank on the device */
    uint32_t *vfv = nullptr;
    int y_direct = 0;       /* identity luma banks into an 8-bit plane on the walker: the converter pass writes the target's luma, the walker the chroma */
    int fmt = 0;            /* the caller's source format */
    int ofs[3] = { 0, 0, 0 };
    int32_t table[9] = {};
    void *planes = nullptr; /* device: the converter lines of the batch in flight (Y, U, V, uint16) */
    size_t planes_sz = 0;
    void *stage = nullptr;  /* device: the host face's packed source rows */
    size_t stage_sz = 0;
};
static void rgb_in_plan_luma(struct FFHipSwsContext *c);
static thread_local int g_sws_create_flat_dither = 0; /* ffhip_sws_getContext -> ffhip_sws_from_tables: the context being made is an RGB source's */

struct FFHipSwsContext {
    int device = 0; /* the banks live on this device; every call of the context makes it current for its duration */
    FFHipSwsRgbIn rgb_in;
    int hrgb_seed0 = 0;  /* a deeper source into packed RGB whose rows all take yuv2rgb_2 (two-tap vertical banks): no rounding term in the sums */
    int widen8 = 0;      /* an 8-bit source into a 9..14-bit target on the 16-bit walker: its planes are widened to 16-bit samples first (k_sws_widen8) */
    void *widen_tmp = nullptr;
    size_t widen_tmp_sz = 0;
    bool rgb_in_luma_done = false; /* the call in flight (under mu): the converter pass wrote the target's luma plane, the walker skips its luma job */
    int flat_dither = 0; /* an 8-bit target's dither entries are all 64 (an RGB source is not dithered: swscale.c:291 looks at the source format) */
    int hbd_sw = 320, hbd_rows = 96; /* LDS shape the banks need: samples per staged source row, source rows per 32-row tile */
    int hbd = 0;    /* a side above 8 bits: the 16-bit scaler (sws_scale16.hip) serves the context, none of the 8-bit fast paths apply */
    FFHipSwsTables t;
    std::vector<int16_t> f[4];
    std::vector<int32_t> p[4];
    void *dev_tables = nullptr;
    FFHipDevFilter d[4];
    int unscaled_yuv2rgb = 0;
    FFHipYuv2RgbK k;
    int chrSrcW, chrSrcH;
    FFHipScalePlaneArgs lum, chr;
    FFHipScaleRgbArgs rgb;
    /* banks as the fast path sees them: sizes 1..3 zero-padded to 4 taps (same sums, positions kept in range) */
    std::vector<int16_t> nf[4];
    std::vector<int32_t> np[4];
    void *dev_ntables = nullptr;
    FFHipDevFilter dn[4];
    int cw_ok = 0; /* both bank pairs fit the column-walking fast path (sws_colwalk.hip) */
    int cw_opt = 0; /* ... and no horizontal sum can wrap int16: the hand-scheduled variant applies */
    int cw_dup = 0; /* ... and columns 1,2 of every group share a window (exact 2x): one unpack serves both */
    int cw_rgb = 0; /* packed-RGB target on the column walker (k_sws_colwalk_rgb) */
    int cw_vround = 1 << 18; /* its vertical rounding seed: yuv2rgb_X / _1: 1 << 18, yuv2rgb_2: 0 */
    /* wide-bank walker (sws_lwalk.hip): banks padded to 4*lw_ht x 2*lw_vt taps */
    int lw_ok = 0, lw_ht = 0, lw_vt = 0;
    std::vector<int16_t> wf[4];
    std::vector<int32_t> wp[4];
    void *dev_wtables = nullptr;
    FFHipDevFilter dw[4];
    /* exact-2x fast path (sws_up2.hip): virtual banks (regular windows of the edge-replicated rows) on the device */
    /* the column walker above 8 bits (sws_walk16.hip): banks padded to w16_ht x w16_vt taps on the device */
    int w16_ok = 0, w16_ht = 0, w16_vt = 0;
    void *w16_dev = nullptr;
    int w16_span[2][3] = {};           /* [luma, chroma][a plane job (256 columns), a pair job on planes (128), a pair job on an interleaved plane] */
    const int16_t *w16_f[4] = { nullptr, nullptr, nullptr, nullptr };
    const int32_t *w16_p[4] = { nullptr, nullptr, nullptr, nullptr };
    int up2_ok = 0;
    int up2_rc = 0; /* a range-converting context: the exact-2x kernel with the range stage is its only fast kernel */
    void *up2_dev = nullptr;
    const uint32_t *up2_h[2] = { nullptr, nullptr }, *up2_v[2] = { nullptr, nullptr };
    /* launch tuner of the table converter (round 6): which workgroup numbering of k_yuv420p_rgb24_t is faster is a property of the BOX
     * (profiles/r06_arena_offset_sweep.txt, r06_xcd_numbering_sweep.txt: eighth-per-XCD +1 .. +8 % on some, -4 % on others, whatever the
     * addresses), so the first eight large launches of a context alternate the two between events and the rest take the winner */
    hipEvent_t tune_ev[16] = {};
    int tune_n = 0, tune_choice = -1;
    uint32_t up2_hco[2][16] = {};      /* the horizontal banks as scalars (FFHipUp2Job.hco), when they have that shape */
    int up2_hco_ok[2] = { 0, 0 };
    /* 4:2:0 into packed RGB at the source's size through the scaler (sws_eqrgb.hip): the virtual vertical chroma bank on the device */
    int eqr_ok = 0;
    void *eqr_dev = nullptr;
    /* a scaled packed-RGB target in two stages (lw_ok on an RGB context; sws_lwalk.hip with an int16 luma plane, then sws_y16rgb.hip):
     * the intermediate planes, grown on demand — a context is used by one caller at a time, as an SwsContext is */
    void *rgb2_tmp = nullptr;
    size_t rgb2_tmp_sz = 0;
    hipEvent_t rgb2_done = nullptr; /* the last launch that read the intermediate: the next use waits for it, whatever its stream */
    std::mutex rgb2_mu;             /* (its own lock: the host face reaches this path holding `mu`) */
    hipStream_t rgb2_aux = nullptr; /* exact 2:1: the luma job (k_sws_down2) runs beside the chroma jobs (k_sws_lwalk), forked and joined with events */
    hipEvent_t rgb2_fork = nullptr, rgb2_join = nullptr;
    /* exact 2x of 4:2:0 (yuv420p, NV12, NV21) into packed RGB (sws_up2rgb.hip): virtual banks of all four axes, the vertical ones merged row by row */
    int u2r_ok = 0;
    void *u2r_dev = nullptr;
    const uint32_t *u2r_hco = nullptr, *u2r_vt = nullptr;
    /* exact-2:1 fast path (sws_down2.hip): the same for banks of up to 8 taps on the windows 2x - 3 .. 2x + 4 */
    int dn2_ok = 0;
    int mix_up2 = 0;  /* luma one tap on the sample itself, chroma planes exactly 2x both ways (yuv420p -> yuv444p at the same size): copy + k_sws_up2 */
    int mix_dn2 = 0;  /* luma one tap on the sample itself, chroma planes exactly 2:1 both ways (yuv444p -> yuv420p at the same size): copy + k_sws_down2 */
    int c420_ok = 0;  /* 4:2:0 between planar and semi-planar layouts at the same size, no range change: a copy (sws_copy420.hip) */
    int f444_ok = 0;  /* planar 4:4:4 into packed RGB at the source's size: four one-tap banks, the full-chroma writer (sws_full444.hip) */
    int dn2_luma = 0; /* an RGB context's luma banks alone (its first stage's luma job on k_sws_down2, the chroma on the wide walker); 2: the chroma planes there as well (no vertical filter: FFHipDn2Job.v1) */
    void *dn2_dev = nullptr;
    int d32_ok = 0;   /* exact 3:2 down in both directions: the static-schedule kernel of sws_down32.hip */
    void *d32_dev = nullptr;
    int u32_ok = 0;   /* exact 3:2 (1) or 4:3 (2) UP between 9..14-bit formats laid out alike: the static-schedule kernel of sws_up32.hip */
    void *u32_dev = nullptr;
    const uint32_t *u32_h[2] = { nullptr, nullptr }, *u32_v[2] = { nullptr, nullptr };
    const uint32_t *d32_h[2] = { nullptr, nullptr }, *d32_v[2] = { nullptr, nullptr };
    const uint32_t *dn2_h[2] = { nullptr, nullptr }, *dn2_v[2] = { nullptr, nullptr };
    /* MFMA-horizontal variant (k_sws_mfma): tile records + window-start index tables on the device */
    int mf_ok = 0, mf_chr_pair = 0, mf_ntiles[2] = { 0, 0 };
    void *mf_dev = nullptr;
    const uint8_t *mf_tiles[2] = { nullptr, nullptr };
    const int32_t *mf_ys[2] = { nullptr, nullptr };
    /* staging for the host-pointer face */
    void *stage = nullptr;
    size_t stage_sz = 0;
    bool luma_pass = false;        /* the alpha pass is running: the planners enumerate the luma job only */
    int slice_next = 0;   /* scaled contexts fed in slices: the next source line expected */
    std::mutex mu;
};

static bool em_forced()
{
    const char *em = FFHIP_KNOB("FFHIP_SWS_MFMA");
    return em && em[0] == '1';
}
static bool fmt_hbd(int f) { return ffhip_pixfmt_hbd(f, nullptr, nullptr, nullptr, nullptr) != 0; }
static bool fmt_yuv(int f)
{
    return f == FFHIP_PIX_FMT_YUV420P || f == FFHIP_PIX_FMT_NV12 || f == FFHIP_PIX_FMT_NV21 || f == FFHIP_PIX_FMT_YUV422P || f == FFHIP_PIX_FMT_YUV444P ||
           fmt_hbd(f);
}
/* chroma subsampling of the YUV formats on this path (av_pix_fmt_get_chroma_sub_sample) */
static int fmt_hsub(int f)
{
    int hs;
    return ffhip_pixfmt_hbd(f, nullptr, nullptr, &hs, nullptr) ? hs : f == FFHIP_PIX_FMT_YUV444P ? 0 : 1;
}
static int fmt_vsub(int f)
{
    int vs;
    return ffhip_pixfmt_hbd(f, nullptr, nullptr, nullptr, &vs) ? vs : f == FFHIP_PIX_FMT_YUV444P || f == FFHIP_PIX_FMT_YUV422P ? 0 : 1;
}
static bool fmt_nv(int f) { return f == FFHIP_PIX_FMT_NV12 || f == FFHIP_PIX_FMT_NV21; }
/* packed layout number of an RGB target (the kernels' `layout` / `bgr` argument): 0 rgb24, 1 bgr24, 2 argb, 3 rgba, 4 abgr, 5 bgra */
static int rgb_layout(int f)
{
    switch (f) {
    case FFHIP_PIX_FMT_RGB24: return 0;
    case FFHIP_PIX_FMT_BGR24: return 1;
    case FFHIP_PIX_FMT_ARGB:  return 2;
    case FFHIP_PIX_FMT_RGBA:  return 3;
    case FFHIP_PIX_FMT_ABGR:  return 4;
    case FFHIP_PIX_FMT_BGRA:  return 5;
    }
    return -1;
}
static bool fmt_rgb(int f) { return rgb_layout(f) >= 0; }
/* planar gbrp: a target of the equal-size table converter only (yuv420p_gbrp_c, yuv2rgb.c:533); the launcher's layout 6 */
static bool fmt_gbrp(int f) { return f == FFHIP_PIX_FMT_GBRP; }
/* the reference's table-driven converter takes the conversion (swscale_unscaled.c:2425-2431; yuva420p arrives as yuv420p + dst_alpha_fill) */
static bool unscaled_rule(const FFHipSwsTables *t)
{
    return t->srcW == t->dstW && t->srcH == t->dstH && (t->srcFormat == FFHIP_PIX_FMT_YUV420P || t->srcFormat == FFHIP_PIX_FMT_YUV422P) &&
           (fmt_rgb(t->dstFormat) || fmt_gbrp(t->dstFormat)) && !(t->flags & FFHIP_SWS_ACCURATE_RND) && !(t->dstH & 1);
}

static int make_k(const FFHipSwsTables &t, FFHipYuv2RgbK *k)
{
    const int64_t yb0 = -(384LL << 16) - 512 * t.yuv2rgb_cy - t.yuv2rgb_oy;
    const int64_t lim = 1LL << 19;
    if (t.yuv2rgb_cy <= 0 || t.yuv2rgb_cy >= lim || llabs(t.yuv2rgb_crv) >= lim || llabs(t.yuv2rgb_cbu) >= lim ||
        llabs(t.yuv2rgb_cgu) >= lim || llabs(t.yuv2rgb_cgv) >= lim || llabs(yb0) >= (1LL << 30)) {
        ffhip_set_error("ffhip_sws: yuv2rgb coefficients outside the int32 closed-form range");
        return FFHIP_EINVAL;
    }
    k->cy = (int)t.yuv2rgb_cy;
    k->kb = (int)(yb0 + 0x8000);
    k->crv = (int)t.yuv2rgb_crv;
    k->cbu = (int)t.yuv2rgb_cbu;
    k->cgu = (int)t.yuv2rgb_cgu;
    k->cgv = (int)t.yuv2rgb_cgv;
    k->off_r = t.yuv2rgb_yoffs - (int)(t.yuv2rgb_crv >> 9);
    k->off_b = t.yuv2rgb_yoffs - (int)(t.yuv2rgb_cbu >> 9);
    k->off_g = t.yuv2rgb_yoffs - (int)(t.yuv2rgb_cgu >> 9) - (int)(t.yuv2rgb_cgv >> 9);
    return 0;
}

/*
 * Fast-path view of the banks: pad 1..3-tap banks to 4 taps.  Zero taps do not change a sum, so bilinear / point /
 * area up-scaling and 1:1 format conversion run on the column walker too.  A 1-tap vertical bank is
 * yuv2plane1_8_c, (h + 64) >> 7 == (64<<12 + h*4096) >> 19: its tap becomes 4096 (planar targets; a packed-RGB
 * target keeps the bank's own coefficient, which yuv2rgb_X multiplies by).  Fills c->nf / c->np / c->dn; false when
 * a bank does not fit.
 */
static bool build_fast_view(FFHipSwsContext *c, const int limits[4], bool packed_rgb)
{
    bool ok = true, padded = false;
    /*
     * Packed RGB: the reference picks yuv2rgb_1 / _2 / _X per output row from the vertical sizes and weights
     * (packed_vscale, vscale.c:126-170).  All three are the same sums with a different rounding seed:
     *   _X   Y = (sum + (1 << 18)) >> 19 — and plain sums do not change when zero taps pad a bank to 4;
     *   _1   Y = (l + 64) >> 7 = (l * 4096 + (1 << 18)) >> 19, chroma (u0 * (4096 - a) + u1 * a + (128 << 11)) >> 19: _X's
     *        formula whenever the single taps are 4096 (output.c:1883-1939);
     *   _2   the two-tap sums with NO rounding term (output.c:1843-1881), taken when both banks have 2 taps and every row's
     *        weights are non-negative and sum to 4096 (bilinear).
     * So the walker serves every shape whose rows agree on the seed; rows that disagree (never seen from initFilter) go to the
     * general kernel.
     */
    if (packed_rgb) {
        const int lf = c->d[2].size, cf = c->d[3].size;
        c->cw_vround = 1 << 18;
        if (lf == 1 || cf == 1) {
            for (int i = 2; i < 4; i++)
                if (c->d[i].size == 1)
                    for (int y = 0; y < c->d[i].n; y++)
                        if (c->f[i][y] != 4096)
                            return false;
        }
        if (lf == 2 && cf == 2) {
            int two = 0;
            const int n = c->d[2].n < c->d[3].n ? c->d[2].n : c->d[3].n; /* chrDstH == dstH for packed targets */
            for (int y = 0; y < n; y++) {
                const int l0 = c->f[2][2 * y], l1 = c->f[2][2 * y + 1], c0 = c->f[3][2 * y], c1 = c->f[3][2 * y + 1];
                two += l0 + l1 == 4096 && (unsigned)l1 <= 4096u && c0 + c1 == 4096 && (unsigned)c1 <= 4096u;
            }
            if (two == n)
                c->cw_vround = 0;
            else if (two)
                return false;
        }
    }
    for (int i = 0; i < 4 && ok; i++) {
        const int fs = c->d[i].size, n = c->d[i].n;
        if (fs > 4 || limits[i] < 4) { ok = false; break; }
        if (fs == 4) { c->nf[i] = c->f[i]; c->np[i] = c->p[i]; continue; }
        padded = true;
        c->nf[i].assign((size_t)n * 4, 0);
        c->np[i].resize(n);
        for (int x = 0; x < n; x++) {
            const int pos = c->p[i][x];
            int npos = pos + 4 > limits[i] ? limits[i] - 4 : pos;
            if (i < 2) {
                /* horizontal: keep the padded window inside the 8-byte span of its 4-column group (the
                 * zero taps may sit in front of the real ones as well as behind them) */
                const int g0 = x & ~3;
                int lo = c->p[i][g0];
                for (int k = 1; k < 4 && g0 + k < n; k++)
                    lo = c->p[i][g0 + k] < lo ? c->p[i][g0 + k] : lo;
                const int base = lo & ~3;
                if (npos > base + 4)
                    npos = base + 4;
            }
            if (npos < 0 || pos < npos || pos - npos + fs > 4) { ok = false; break; }
            c->np[i][x] = npos;
            for (int k = 0; k < fs; k++)
                c->nf[i][(size_t)x * 4 + (pos - npos) + k] = (i >= 2 && fs == 1 && !packed_rgb) ? 4096 : c->f[i][(size_t)x * fs + k];
        }
    }
    if (!ok)
        return false;
    for (int i = 0; i < 4; i++) {
        c->dn[i] = c->d[i];
        c->dn[i].size = 4;
    }
    if (padded) {
        size_t noff[4][2], ntot = 0;
        for (int i = 0; i < 4; i++) {
            noff[i][0] = ntot; ntot += (c->nf[i].size() * 2 + 15) & ~(size_t)15;
            noff[i][1] = ntot; ntot += (c->np[i].size() * 4 + 15) & ~(size_t)15;
        }
        ok = hipMalloc(&c->dev_ntables, ntot) == hipSuccess;
        for (int i = 0; i < 4 && ok; i++) {
            uint8_t *nb = static_cast<uint8_t *>(c->dev_ntables);
            ok = hipMemcpy(nb + noff[i][0], c->nf[i].data(), c->nf[i].size() * 2, hipMemcpyHostToDevice) == hipSuccess &&
                 hipMemcpy(nb + noff[i][1], c->np[i].data(), c->np[i].size() * 4, hipMemcpyHostToDevice) == hipSuccess;
            c->dn[i].filter = reinterpret_cast<const int16_t *>(nb + noff[i][0]);
            c->dn[i].pos = reinterpret_cast<const int32_t *>(nb + noff[i][1]);
        }
    }
    return ok;
}

/*
 * Wide view of the banks for sws_lwalk.hip: all four padded to a common 4*ht horizontal / 2*vt vertical taps, windows
 * shifted back inside the plane where the padding would leave it (the real taps then sit at the end of the window).
 */
static bool build_wide_view(FFHipSwsContext *c, const int limits[4], int min_ht = 0)
{
    int ht = min_ht, vt = 0;
    for (int i = 0; i < 4; i++) {
        const int fs = c->d[i].size;
        if (fs > (i < 2 ? 64 : 32))
            return false;
        const int cls = fs > 32 ? 8 : fs > 16 ? 4 : fs > 8 ? 2 : 1; /* 8, 16, 32 or (across only) 64 taps */
        if (i < 2) ht = 2 * cls > ht ? 2 * cls : ht;
        else       vt = 4 * cls > vt ? 4 * cls : vt;
    }
    if (ht == 16 && vt == 4)
        vt = 8; /* (the instantiated pairs at 64 taps across: 16 and 32 taps down) */
    for (int i = 0; i < 4; i++) {
        const int P = i < 2 ? 4 * ht : 2 * vt, fs = c->d[i].size, n = c->d[i].n;
        if (limits[i] < P)
            return false;
        c->wf[i].assign((size_t)n * P, 0);
        c->wp[i].resize(n);
        for (int x = 0; x < n; x++) {
            const int pos = c->p[i][x];
            const int npos = pos + P > limits[i] ? limits[i] - P : pos;
            if (pos < 0 || pos + fs > limits[i])
                return false;
            c->wp[i][x] = npos;
            for (int k = 0; k < fs; k++)
                c->wf[i][(size_t)x * P + (pos - npos) + k] = (i >= 2 && fs == 1) ? 4096 : c->f[i][(size_t)x * fs + k];
        }
    }
    size_t off[4][2], tot = 0;
    for (int i = 0; i < 4; i++) {
        off[i][0] = tot; tot += (c->wf[i].size() * 2 + 15) & ~(size_t)15;
        off[i][1] = tot; tot += (c->wp[i].size() * 4 + 15) & ~(size_t)15;
    }
    if (c->dev_wtables) { /* a second try with wider padding */
        (void)hipFree(c->dev_wtables);
        c->dev_wtables = nullptr;
    }
    if (hipMalloc(&c->dev_wtables, tot) != hipSuccess)
        return false;
    uint8_t *b = static_cast<uint8_t *>(c->dev_wtables);
    for (int i = 0; i < 4; i++) {
        if (hipMemcpy(b + off[i][0], c->wf[i].data(), c->wf[i].size() * 2, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(b + off[i][1], c->wp[i].data(), c->wp[i].size() * 4, hipMemcpyHostToDevice) != hipSuccess)
            return false;
        c->dw[i].filter = reinterpret_cast<const int16_t *>(b + off[i][0]);
        c->dw[i].pos = reinterpret_cast<const int32_t *>(b + off[i][1]);
        c->dw[i].size = i < 2 ? 4 * ht : 2 * vt;
        c->dw[i].n = c->d[i].n;
    }
    c->lw_ht = ht;
    c->lw_vt = vt;
    return true;
}

/* the wide view + the walker's own checks (ffhip_lw_bank_ok, no int16 wrap of a horizontal sum).  A wave's row buffer grows with the tap
 * class (3, 5, 9 x 256 bytes for 8, 16, 32 taps): a bank of few taps at a steep ratio (bilinear at 1/7) gets the next class's padding
 * when its 256 columns span more source than its own class holds. */
static bool wide_setup(FFHipSwsContext *c, const int limits[4], bool chroma_pair)
{
    for (int min_ht = 0; min_ht <= 16; min_ht = min_ht ? 2 * min_ht : 4) {
        if (!build_wide_view(c, limits, min_ht))
            return false;
        bool ok = true;
        for (int k = 0; k < 2 && ok; k++)
            ok = ffhip_lw_bank_ok(c->wp[k].data(), c->lw_ht, c->d[k].n, limits[k], c->wp[2 + k].data(), c->lw_vt, c->d[2 + k].n,
                                  limits[2 + k], k == 1 && chroma_pair) != 0 &&
                 ffhip_cw_bank_nowrap(c->wf[k].data(), 4 * c->lw_ht, c->d[k].n);
        if (ok)
            return true;
        if (c->lw_ht >= 16)
            break;
        if (min_ht < c->lw_ht)
            min_ht = c->lw_ht; /* (the next class up from the one the taps asked for) */
    }
    return false;
}

/* exact 2x: the 4-tap views (c->nf / c->np) as virtual banks on the regular windows of the edge-replicated rows, on the device;
 * sets c->up2_ok when every bank row is of that shape (sws_up2.hip) */
/* chroma_only: the chroma banks alone (the luma plane is the source's: FFHipSwsContext.mix_up2) */
static void up2_build(FFHipSwsContext *c, const int nsrc[4], bool chroma_only = false)
{
    std::vector<uint32_t> vb[4];
    bool ok = true;
    for (int i = 0; i < 4 && ok; i++)
        if (!(chroma_only && !(i & 1)))
            ok = ffhip_up2_virtual_bank(c->nf[i].data(), c->np[i].data(), c->d[i].n, nsrc[i], &vb[i]) != 0;
    if (!ok)
        return;
    for (int i = 0; i < 2; i++)
        c->up2_hco_ok[i] = !(chroma_only && i == 0) && ffhip_up2_hco(vb[i], c->up2_hco[i]);
    /* vertical banks: one leading row (y = -1) and 17 trailing ones of zeros (the row loop reads ahead) */
    for (int i = chroma_only ? 3 : 2; i < 4; i++) {
        std::vector<uint32_t> pv((size_t)(c->d[i].n + 18) * 2, 0);
        memcpy(pv.data() + 2, vb[i].data(), vb[i].size() * 4);
        vb[i].swap(pv);
    }
    size_t uo[4], ut = 0;
    for (int i = 0; i < 4; i++) {
        uo[i] = ut;
        ut += (vb[i].size() * 4 + 255) & ~(size_t)255;
    }
    if (hipMalloc(&c->up2_dev, ut) != hipSuccess)
        return;
    uint8_t *b = static_cast<uint8_t *>(c->up2_dev);
    for (int i = 0; i < 4 && ok; i++)
        ok = hipMemcpy(b + uo[i], vb[i].data(), vb[i].size() * 4, hipMemcpyHostToDevice) == hipSuccess;
    if (ok) {
        c->up2_h[0] = reinterpret_cast<const uint32_t *>(b + uo[0]);
        c->up2_h[1] = reinterpret_cast<const uint32_t *>(b + uo[1]);
        c->up2_v[0] = reinterpret_cast<const uint32_t *>(b + uo[2]);
        c->up2_v[1] = reinterpret_cast<const uint32_t *>(b + uo[3]);
        if (chroma_only)
            c->mix_up2 = 1;
        else
            c->up2_ok = 1;
    }
}

/* exact 2x of 4:2:0 into packed RGB: luma 2x both ways, chroma 2x horizontally and 4x vertically (a packed target has a
 * chroma line per output line).  The 4-tap views as virtual banks on the regular windows of the edge-replicated rows; the two vertical
 * ones merged into one table of four dwords per output row (the kernel reads it with scalar loads), one zero row in front (y = -1) and
 * zero rows behind (the row loop reads one row pair past the picture).  Sets c->u2r_ok (sws_up2rgb.hip). */
static void up2rgb_build(FFHipSwsContext *c, int srcW, int srcH, int dstW, int dstH)
{
    if (dstW != 2 * srcW || dstH != 2 * srcH || (srcW & 7) || (srcH & 1) || srcW < 16 || srcH < 8)
        return;
    const int chrW = srcW / 2, chrH = srcH / 2;
    if (c->d[0].n != dstW || c->d[1].n != srcW || c->d[2].n != dstH || c->d[3].n != dstH)
        return;
    std::vector<uint32_t> hl, hc, vl, vc;
    if (!ffhip_upn_virtual_bank(c->nf[0].data(), c->np[0].data(), dstW, srcW, 2, &hl) ||
        !ffhip_upn_virtual_bank(c->nf[1].data(), c->np[1].data(), srcW, chrW, 2, &hc) ||
        !ffhip_upn_virtual_bank(c->nf[2].data(), c->np[2].data(), dstH, srcH, 2, &vl) ||
        !ffhip_upn_virtual_bank(c->nf[3].data(), c->np[3].data(), dstH, chrH, 4, &vc))
        return;
    uint32_t hco[32];
    if (!ffhip_up2rgb_hco(hl, hc, hco))
        return;
    std::vector<uint32_t> vt((size_t)(dstH + 6) * 4, 0);
    for (int y = 0; y < dstH; y++) {
        uint32_t *r = vt.data() + (size_t)(y + 1) * 4;
        r[0] = vl[2 * (size_t)y]; r[1] = vl[2 * (size_t)y + 1]; r[2] = vc[2 * (size_t)y]; r[3] = vc[2 * (size_t)y + 1];
    }
    if (hipMalloc(&c->u2r_dev, 256 + vt.size() * 4) != hipSuccess)
        return;
    uint8_t *b = static_cast<uint8_t *>(c->u2r_dev);
    if (hipMemcpy(b, hco, sizeof(hco), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(b + 256, vt.data(), vt.size() * 4, hipMemcpyHostToDevice) != hipSuccess)
        return;
    c->u2r_hco = reinterpret_cast<const uint32_t *>(b);
    c->u2r_vt = reinterpret_cast<const uint32_t *>(b + 256);
    c->u2r_ok = 1;
}

/* 4:2:0 into packed RGB at the source's size: one-tap unit banks for the luma and across, the vertical chroma bank an exact 2x bank
 * (its 4-tap view on the regular windows of the edge-replicated plane).  Sets c->eqr_ok (sws_eqrgb.hip). */
static void eqrgb_build(FFHipSwsContext *c, int srcW, int srcH, int chrW, int chrH)
{
    if ((srcW & 7) || (srcH & 1) || chrH < 4 || 2 * chrH != srcH || 2 * chrW != srcW || c->cw_vround != 1 << 18)
        return;
    const int unit[3] = { 1 << 14, 1 << 14, 1 << 12 }, want_n[3] = { srcW, chrW, srcH };
    for (int i = 0; i < 3; i++) {
        if (c->d[i].size != 1 || c->d[i].n != want_n[i])
            return;
        for (int x = 0; x < c->d[i].n; x++)
            if (c->f[i][x] != unit[i] || c->p[i][x] != x)
                return;
    }
    if (c->d[3].n != srcH)
        return;
    std::vector<uint32_t> vc;
    if (!ffhip_up2_virtual_bank(c->nf[3].data(), c->np[3].data(), srcH, chrH, &vc))
        return;
    std::vector<uint32_t> vt((size_t)(srcH + 6) * 2, 0);
    memcpy(vt.data() + 2, vc.data(), vc.size() * 4);
    if (hipMalloc(&c->eqr_dev, vt.size() * 4) != hipSuccess)
        return;
    if (hipMemcpy(c->eqr_dev, vt.data(), vt.size() * 4, hipMemcpyHostToDevice) != hipSuccess)
        return;
    c->eqr_ok = 1;
}

/* exact 2:1: the banks (up to 8 taps) as virtual banks on the windows 2x - 3 .. 2x + 4 of the edge-replicated rows, on the device;
 * sets c->dn2_ok when every bank row is of that shape (sws_down2.hip) */
static void dn2_build(FFHipSwsContext *c, const int nsrc[4])
{
    std::vector<uint32_t> vb[4];
    bool ok = true;
    for (int i = 0; i < 4 && ok; i++)
        ok = ffhip_down2_virtual_bank(c->f[i].data(), c->p[i].data(), c->d[i].size, c->d[i].n, nsrc[i], &vb[i]) != 0;
    if (!ok)
        return;
    for (int i = 2; i < 4; i++)
        vb[i].resize((size_t)(c->d[i].n + 8) * 4, 0); /* the row loop reads four rows of coefficients at a time */
    size_t uo[4], ut = 0;
    for (int i = 0; i < 4; i++) {
        uo[i] = ut;
        ut += (vb[i].size() * 4 + 255) & ~(size_t)255;
    }
    if (hipMalloc(&c->dn2_dev, ut) != hipSuccess)
        return;
    uint8_t *b = static_cast<uint8_t *>(c->dn2_dev);
    for (int i = 0; i < 4 && ok; i++)
        ok = hipMemcpy(b + uo[i], vb[i].data(), vb[i].size() * 4, hipMemcpyHostToDevice) == hipSuccess;
    if (ok) {
        c->dn2_h[0] = reinterpret_cast<const uint32_t *>(b + uo[0]);
        c->dn2_h[1] = reinterpret_cast<const uint32_t *>(b + uo[1]);
        c->dn2_v[0] = reinterpret_cast<const uint32_t *>(b + uo[2]);
        c->dn2_v[1] = reinterpret_cast<const uint32_t *>(b + uo[3]);
        c->dn2_ok = 1;
    }
}

/* exact 3:2: the banks (up to 6 taps) as virtual banks on the windows 3 (x >> 1) - 2 + (x & 1) .. + 5 of the edge-replicated rows, on the
 * device; sets c->d32_ok when every bank row is of that shape (sws_down32.hip) */
static void d32_build(FFHipSwsContext *c, const int nsrc[4], int pin = 3, int pout = 2)
{
    std::vector<uint32_t> vb[4];
    for (int i = 0; i < 4; i++)
        if (!ffhip_d32_virtual_bank(c->f[i].data(), c->p[i].data(), c->d[i].size, c->d[i].n, nsrc[i], i < 2 ? 3 : 4, &vb[i], pin, pout))
            return;
    size_t uo[4], ut = 0;
    for (int i = 0; i < 4; i++) {
        uo[i] = ut;
        ut += (vb[i].size() * 4 + 255) & ~(size_t)255;
    }
    if (hipMalloc(&c->d32_dev, ut) != hipSuccess)
        return;
    uint8_t *b = static_cast<uint8_t *>(c->d32_dev);
    for (int i = 0; i < 4; i++)
        if (hipMemcpy(b + uo[i], vb[i].data(), vb[i].size() * 4, hipMemcpyHostToDevice) != hipSuccess)
            return;
    c->d32_h[0] = reinterpret_cast<const uint32_t *>(b + uo[0]);
    c->d32_h[1] = reinterpret_cast<const uint32_t *>(b + uo[1]);
    c->d32_v[0] = reinterpret_cast<const uint32_t *>(b + uo[2]);
    c->d32_v[1] = reinterpret_cast<const uint32_t *>(b + uo[3]);
    c->d32_ok = pin == 3 ? 1 : 2;
}

/* exact 3:2 up above 8 bits: the banks (up to 4 taps) as virtual banks on the windows 2 (x / 3) - 2 + x % 3 .. + 3 of the edge-replicated rows, on
 * the device; sets c->u32_ok when every bank row is of that shape (sws_up32.hip) */
static void u32_build(FFHipSwsContext *c, const int nsrc[4], int pin, int pout)
{
    std::vector<uint32_t> vb[4];
    for (int i = 0; i < 4; i++)
        if (!ffhip_u32_virtual_bank(c->f[i].data(), c->p[i].data(), c->d[i].size, c->d[i].n, nsrc[i], pin, pout, &vb[i]))
            return;
    size_t uo[4], ut = 0;
    for (int i = 0; i < 4; i++) {
        uo[i] = ut;
        ut += (vb[i].size() * 4 + 255) & ~(size_t)255;
    }
    if (hipMalloc(&c->u32_dev, ut) != hipSuccess)
        return;
    uint8_t *b = static_cast<uint8_t *>(c->u32_dev);
    for (int i = 0; i < 4; i++)
        if (hipMemcpy(b + uo[i], vb[i].data(), vb[i].size() * 4, hipMemcpyHostToDevice) != hipSuccess)
            return;
    c->u32_h[0] = reinterpret_cast<const uint32_t *>(b + uo[0]);
    c->u32_h[1] = reinterpret_cast<const uint32_t *>(b + uo[1]);
    c->u32_v[0] = reinterpret_cast<const uint32_t *>(b + uo[2]);
    c->u32_v[1] = reinterpret_cast<const uint32_t *>(b + uo[3]);
    c->u32_ok = pin == 2 ? 1 : 2;
}

/* the luma banks alone at exact 2:1, for a packed-RGB target's first stage (its chroma goes 2:1 across but 1:1 or 2:1 down by the
 * source's subsampling, and rides the wide-bank walker): sets c->dn2_luma */
static void dn2_build_luma(FFHipSwsContext *c, int srcW, int srcH)
{
    std::vector<uint32_t> vb[3];
    if (!ffhip_down2_virtual_bank(c->f[0].data(), c->p[0].data(), c->d[0].size, c->d[0].n, srcW, &vb[0]) ||
        !ffhip_down2_virtual_bank(c->f[2].data(), c->p[2].data(), c->d[2].size, c->d[2].n, srcH, &vb[1]))
        return;
    vb[1].resize((size_t)(c->d[2].n + 8) * 4, 0); /* the row loop reads four rows of coefficients at a time */
    /* the chroma planes on the same kernel when they need no vertical filter — a chroma line per output line (4K 4:2:0 -> 1080p RGB),
     * every row of the vertical bank one tap of 4096 on the line itself — and go 2:1 across (FFHipDn2Job.v1) */
    bool chr = c->d[3].n == c->chrSrcH && c->d[1].n >= 6;
    for (int y = 0; y < c->d[3].n && chr; y++)
        for (int i = 0; i < c->d[3].size && chr; i++) {
            const int16_t t = c->f[3][(size_t)y * c->d[3].size + i];
            chr = t == (c->p[3][y] + i == y ? 4096 : 0);
        }
    chr = chr && ffhip_cw_bank_nowrap(c->f[1].data(), c->d[1].size, c->d[1].n) &&
          ffhip_down2_virtual_bank(c->f[1].data(), c->p[1].data(), c->d[1].size, c->d[1].n, c->chrSrcW, &vb[2]) != 0;
    const size_t o1 = (vb[0].size() * 4 + 255) & ~(size_t)255, o2 = o1 + ((vb[1].size() * 4 + 255) & ~(size_t)255);
    if (hipMalloc(&c->dn2_dev, o2 + (chr ? vb[2].size() * 4 : 0)) != hipSuccess)
        return;
    uint8_t *b = static_cast<uint8_t *>(c->dn2_dev);
    if (hipMemcpy(b, vb[0].data(), vb[0].size() * 4, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(b + o1, vb[1].data(), vb[1].size() * 4, hipMemcpyHostToDevice) != hipSuccess ||
        (chr && hipMemcpy(b + o2, vb[2].data(), vb[2].size() * 4, hipMemcpyHostToDevice) != hipSuccess))
        return;
    c->dn2_h[0] = reinterpret_cast<const uint32_t *>(b);
    c->dn2_v[0] = reinterpret_cast<const uint32_t *>(b + o1);
    c->dn2_h[1] = chr ? reinterpret_cast<const uint32_t *>(b + o2) : nullptr;
    c->dn2_luma = 1 + (chr ? 1 : 0);
}

/* no horizontal sum of a 4-tap bank falls below -32768 after >> (depth - 1) on samples of `depth` bits: int16 saturation then
 * equals the reference's min(., 32767) + truncation (ffhip_cw_bank_nowrap is this at 8 bits) */
static bool bank_nowrap_depth(const int16_t *f, int n, int depth, int size = 4)
{
    for (int x = 0; x < n; x++) {
        long long neg = 0;
        for (int j = 0; j < size; j++)
            if (f[(size_t)x * size + j] < 0)
                neg += f[(size_t)x * size + j];
        if (((1LL << depth) - 1) * neg < -32768LL * (1LL << (depth - 1)))
            return false;
    }
    return true;
}

extern "C" FFHipSwsContext *ffhip_sws_from_tables(const FFHipSwsTables *t)
{
    if (!t || !fmt_yuv(t->srcFormat) || !(fmt_yuv(t->dstFormat) || fmt_rgb(t->dstFormat) || (fmt_gbrp(t->dstFormat) && unscaled_rule(t) && !(t->dstW & 1)))) {
        ffhip_set_error("ffhip_sws: unsupported format pair");
        return nullptr;
    }
    /* an odd RGB width and a 4:4:4 source force SWS_FULL_CHR_H_INT in the reference (utils.c:1270-1290): the tables must say so */
    if (fmt_rgb(t->dstFormat) && !t->full_chr_h_int &&
        ((t->dstW & 1) || (fmt_hsub(t->srcFormat) == 0 && fmt_vsub(t->srcFormat) == 0 && !(t->flags & FFHIP_SWS_FAST_BILINEAR)))) {
        ffhip_set_error("ffhip_sws: an odd RGB width / a 4:4:4 source to packed RGB runs with SWS_FULL_CHR_H_INT in the reference: "
                        "FFHipSwsTables.full_chr_h_int and .yuv2rgb_full must be set");
        return nullptr;
    }
    if (unscaled_rule(t) && (t->dstW & 1)) {
        ffhip_set_error("ffhip_sws: equal-size yuv420p / yuv422p to an odd-width RGB target is the table converter's tail case; not on the hip path");
        return nullptr;
    }
    /* alpha: 1 = dst[3] filled with 255, 2 = src[3] scaled into dst[3] by the luma banks (a second pass of the context): a planar 8-bit
     * target, and no range conversion (the reference converts plane 0 only, hscale.c:57-59; the second pass would convert its luma) */
    if (t->dst_alpha_fill < 0 || t->dst_alpha_fill > 2 ||
        (t->dst_alpha_fill == 2 && unscaled_rule(t) && !(rgb_layout(t->dstFormat) >= 2 && rgb_layout(t->dstFormat) <= 5 && t->srcFormat == FFHIP_PIX_FMT_YUV420P)) ||
        (t->dst_alpha_fill == 2 && !unscaled_rule(t) && fmt_rgb(t->dstFormat) && (rgb_layout(t->dstFormat) < 2 || fmt_nv(t->srcFormat) || fmt_hbd(t->srcFormat))) ||
        (t->dst_alpha_fill == 2 && !unscaled_rule(t) && !fmt_rgb(t->dstFormat) && (fmt_nv(t->dstFormat) || ffhip_pixfmt_hbd(t->srcFormat, nullptr, nullptr, nullptr, nullptr) ||
                                    ffhip_pixfmt_hbd(t->dstFormat, nullptr, nullptr, nullptr, nullptr) || t->src_range != t->dst_range))) {
        ffhip_set_error("ffhip_sws: a source alpha plane (dst_alpha_fill 2) goes with a planar 8-bit target and equal ranges, or with a 32-bit "
                        "packed RGB target of an 8-bit planar source");
        return nullptr;
    }
    /* (a packed RGB target takes the source's range through the yuv2rgb coefficients the caller hands over: ff_yuv2rgb_c_init_tables'
     * fullRange branch, libswscale/yuv2rgb.c:760-768; the range fields then play no part) */
    if (!ffhip_have_device()) {
        ffhip_set_error("ffhip_sws: no HIP device (FFHIP_ENOSYS) - keep the C function pointers");
        return nullptr;
    }
    FFHipSwsContext *c = new (std::nothrow) FFHipSwsContext();
    if (!c)
        return nullptr;
    c->device = ffhip_current_device();
    c->t = *t;
    c->flat_dither = g_sws_create_flat_dither;
    c->chrSrcW = -((-t->srcW) >> fmt_hsub(t->srcFormat));
    c->chrSrcH = -((-t->srcH) >> fmt_vsub(t->srcFormat));
    c->unscaled_yuv2rgb = unscaled_rule(t);
    if (make_k(*t, &c->k) < 0) {
        delete c;
        return nullptr;
    }

    /* copy the banks (with the reference's 3 entries of over-read padding when present is not assumed) */
    const FFHipSwsFilter *in[4] = { &t->hLum, &t->hChr, &t->vLum, &t->vChr };
    FFHipSwsFilter *own[4] = { &c->t.hLum, &c->t.hChr, &c->t.vLum, &c->t.vChr };
    size_t off[4][2], total = 0;
    /* the reference builds no banks for a context that got a special converter (ff_sws_init_single_context() returns before
     * initFilter(), libswscale/utils.c:1625-1637): the table converter takes none, the context keeps one-tap identities */
    std::vector<int16_t> idf[4];
    std::vector<int32_t> idp[4];
    FFHipSwsFilter idb[4];
    if (c->unscaled_yuv2rgb && (!t->hLum.filter || !t->hChr.filter || !t->vLum.filter || !t->vChr.filter)) {
        const int n[4] = { t->dstW, (t->dstW + 1) >> 1, t->dstH, (t->dstH + 1) >> 1 };
        for (int i = 0; i < 4; i++) {
            idf[i].assign((size_t)n[i], (int16_t)(i < 2 ? 1 << 14 : 1 << 12));
            idp[i].resize((size_t)n[i]);
            for (int x = 0; x < n[i]; x++)
                idp[i][x] = x;
            idb[i].filter = idf[i].data(); idb[i].pos = idp[i].data(); idb[i].size = 1; idb[i].n = n[i];
            in[i] = &idb[i];
        }
    }
    for (int i = 0; i < 4; i++) {
        if (!in[i]->filter || !in[i]->pos || in[i]->size <= 0 || in[i]->n <= 0) {
            ffhip_set_error("ffhip_sws: filter bank %d missing", i);
            delete c;
            return nullptr;
        }
        c->f[i].assign(in[i]->filter, in[i]->filter + (size_t)in[i]->n * in[i]->size);
        c->p[i].assign(in[i]->pos, in[i]->pos + in[i]->n);
        own[i]->filter = c->f[i].data();
        own[i]->pos = c->p[i].data();
        off[i][0] = total; total += (c->f[i].size() * 2 + 15) & ~(size_t)15;
        off[i][1] = total; total += (c->p[i].size() * 4 + 15) & ~(size_t)15;
    }
    if (hipMalloc(&c->dev_tables, total) != hipSuccess) {
        ffhip_set_error("ffhip_sws: hipMalloc(%zu) for filter banks failed", total);
        delete c;
        return nullptr;
    }
    for (int i = 0; i < 4; i++) {
        uint8_t *base = static_cast<uint8_t *>(c->dev_tables);
        if (hipMemcpy(base + off[i][0], c->f[i].data(), c->f[i].size() * 2, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(base + off[i][1], c->p[i].data(), c->p[i].size() * 4, hipMemcpyHostToDevice) != hipSuccess) {
            ffhip_set_error("ffhip_sws: filter bank upload failed");
            ffhip_sws_freeContext(c);
            return nullptr;
        }
        c->d[i].filter = reinterpret_cast<const int16_t *>(base + off[i][0]);
        c->d[i].pos = reinterpret_cast<const int32_t *>(base + off[i][1]);
        c->d[i].size = in[i]->size;
        c->d[i].n = in[i]->n;
    }

    if (c->unscaled_yuv2rgb)
        return c;

    if (fmt_hbd(t->srcFormat) || fmt_hbd(t->dstFormat)) {
        /* above 8 bits on either side: the 16-bit scaler takes the banks as they are */
        const bool hrgb = fmt_rgb(t->dstFormat);
        if ((!hrgb && t->srcW == t->dstW && t->srcH == t->dstH && t->src_range == t->dst_range && !c->flat_dither /* (an RGB source's converter lines
             * go through the scaler at equal sizes too: the reference has no special converter for them) */) || t->srcFormat == FFHIP_PIX_FMT_NV21 ||
            t->dstFormat == FFHIP_PIX_FMT_NV21 || fmt_gbrp(t->dstFormat)) {
            ffhip_set_error("ffhip_sws: above 8 bits the hip path scales between the YUV formats and into packed RGB (no equal-size YUV conversion, no NV21 mix)");
            ffhip_sws_freeContext(c);
            return nullptr;
        }
        if (hrgb && (t->full_chr_h_int || (t->dstW & 1) || t->dst_alpha_fill == 2 || c->flat_dither)) {
            /* (round 6) a deeper source into packed RGB: the two-stage form below serves the chroma-subsampled writers (yuv2rgb_X / _2 / _1 on
             * pixel pairs); the full-chroma writers (an odd width, a 4:4:4 source, SWS_FULL_CHR_H_INT) and a source alpha plane are not built */
            ffhip_set_error("ffhip_sws: above 8 bits into packed RGB: full-chroma writers (odd width / 4:4:4 source / SWS_FULL_CHR_H_INT) and alpha are not on the hip path");
            ffhip_sws_freeContext(c);
            return nullptr;
        }
        if (c->d[1].n != (hrgb ? (t->dstW + 1) >> 1 : -((-t->dstW) >> fmt_hsub(t->dstFormat))) ||
            c->d[3].n != (hrgb ? t->dstH : -((-t->dstH) >> fmt_vsub(t->dstFormat))) ||
            c->d[0].n != t->dstW || c->d[2].n != t->dstH) {
            ffhip_set_error("ffhip_sws: the banks do not match the target's plane sizes");
            ffhip_sws_freeContext(c);
            return nullptr;
        }
        /* does every 64-column run of both horizontal banks reach at most 320 source columns (the staged kernel's LDS row)? */
        c->hbd = 2;
        c->hbd_sw = 8;
        c->hbd_rows = 1;
        for (int b = 0; b < 2; b++) {
            const std::vector<int32_t> &pos = c->p[b];
            const int fs = c->d[b].size, n = c->d[b].n;
            for (int x0 = 0; x0 < n; x0 += 64) {
                const int x1 = (x0 + 63 < n ? x0 + 63 : n - 1);
                int lo = pos[x0], hi = pos[x0];
                for (int x = x0; x <= x1; x++) { lo = pos[x] < lo ? pos[x] : lo; hi = pos[x] > hi ? pos[x] : hi; }
                if (hi + fs - lo > 320 || lo != pos[x0])
                    c->hbd = 1; /* not staged */
                else if (hi + fs - lo > c->hbd_sw)
                    c->hbd_sw = hi + fs - lo;
            }
            /* source rows a 32-row tile of the vertical bank reaches (capped: taller reaches run in chunks) */
            const std::vector<int32_t> &vp = c->p[2 + b];
            const int vfs = c->d[2 + b].size, vn = c->d[2 + b].n;
            for (int y0 = 0; y0 < vn; y0 += 32) {
                const int y1 = (y0 + 31 < vn ? y0 + 31 : vn - 1);
                if (vp[y1] + vfs - vp[y0] > c->hbd_rows)
                    c->hbd_rows = vp[y1] + vfs - vp[y0];
            }
        }
        c->hbd_sw = (c->hbd_sw + 3) & ~1;
        /* exact 2x between formats of 9..14 bits laid out alike (planar <-> planar, P01x <-> P01x), no range change: the
         * static-schedule kernel's 16-bit twin (k_sws_up2<., ., 1>) */
        {
            int sd = 8, sl = 0, dd = 8, dl = 0;
            (void)ffhip_pixfmt_hbd(t->srcFormat, &sd, &sl, nullptr, nullptr);
            (void)ffhip_pixfmt_hbd(t->dstFormat, &dd, &dl, nullptr, nullptr);
            const int cw = c->chrSrcW, chh = c->chrSrcH;
            const int limits[4] = { t->srcW, cw, t->srcH, chh };
            /* (round 6: from an 8-bit source too — planar into planar, NV12 into P01x — on planes widened to 16-bit samples: `widen8`) */
            const bool w8 = sd == 8 && dd > 8 && dd <= 14 && !c->flat_dither && t->srcFormat != FFHIP_PIX_FMT_NV21 &&
                            (fmt_nv(t->srcFormat) ? dl == 1 : dl == 0);
            if (((sd > 8 && sd <= 14 && sl == dl && sl != 2) || w8) && dd > 8 && dd <= 14 && t->src_range == t->dst_range &&
                t->dstW == 2 * t->srcW && t->dstH == 2 * t->srcH && c->d[1].n == 2 * cw && c->d[3].n == 2 * chh &&
                !(t->srcW & 3) && t->srcW >= 8 && (sl ? !(cw & 1) && cw >= 4 : !(cw & 3) && cw >= 8) &&
                build_fast_view(c, limits, false) && bank_nowrap_depth(c->nf[0].data(), c->d[0].n, sd) &&
                bank_nowrap_depth(c->nf[1].data(), c->d[1].n, sd)) {
                up2_build(c, limits);
                c->widen8 = c->up2_ok && w8;
            }
            /* ... and exact 2:1 (k_sws_down2<1>), banks of up to 8 taps */
            const int cdw = c->d[1].n, cdh = c->d[3].n;
            /* (round 5: also into an 8-bit target laid out alike — P01x -> NV12, planar -> planar — with the ordered dither on the way out) */
            const bool dn8 = dd == 8 && !hrgb && !c->flat_dither /* (k_sws_down2 has the ordered dither only) */ && (sl == 1 ? t->dstFormat == FFHIP_PIX_FMT_NV12 : !fmt_nv(t->dstFormat));
            if (sd > 8 && sd <= 14 && ((dd > 8 && dd <= 14 && sl == dl) || dn8) && sl != 2 && t->src_range == t->dst_range &&
                t->srcW == 2 * t->dstW && t->srcH == 2 * t->dstH && cw == 2 * cdw && chh == 2 * cdh &&
                !(t->dstW & 3) && t->dstW >= 12 && (sl ? !(cdw & 1) && cdw >= 6 : !(cdw & 3) && cdw >= 12) &&
                bank_nowrap_depth(c->f[0].data(), c->d[0].n, sd, c->d[0].size) && bank_nowrap_depth(c->f[1].data(), c->d[1].n, sd, c->d[1].size))
                dn2_build(c, limits);
            /* exact 3:2 up (720p -> 1080p ...) between formats of 9..14 bits laid out alike, or from an 8-bit source widened to words as for the
             * walker below (planar into planar, NV12 into P01x): sws_up32.hip; the walker is set up beside it for planes it cannot take */
            {
                const bool w8u = sd == 8 && dd > 8 && dd <= 14 && !c->flat_dither && t->srcFormat != FFHIP_PIX_FMT_NV21 &&
                                 (fmt_nv(t->srcFormat) ? dl == 1 : dl == 0);
                if (((sd > 8 && sd <= 14 && sl == dl && sl != 2) || w8u) && dd > 8 && dd <= 14 && !hrgb && !c->flat_dither && t->src_range == t->dst_range &&
                    2 * t->dstW == 3 * t->srcW && 2 * t->dstH == 3 * t->srcH && 2 * c->d[1].n == 3 * cw && 2 * c->d[3].n == 3 * chh &&
                    !(t->srcW & 3) && t->srcW >= 12 && !(t->srcH & 1) && !(chh & 1) &&
                    ((sl || fmt_nv(t->srcFormat)) ? !(cw & 1) && cw >= 6 : !(cw & 3) && cw >= 12) &&
                    bank_nowrap_depth(c->f[0].data(), c->d[0].n, sd, c->d[0].size) && bank_nowrap_depth(c->f[1].data(), c->d[1].n, sd, c->d[1].size))
                    u32_build(c, limits, 2, 3);
                /* ... and exact 4:3 (1080p -> 1440p): period (3 in, 4 out) of the same kernel */
                if (((sd > 8 && sd <= 14 && sl == dl && sl != 2) || w8u) && dd > 8 && dd <= 14 && !hrgb && !c->flat_dither && t->src_range == t->dst_range &&
                    3 * t->dstW == 4 * t->srcW && 3 * t->dstH == 4 * t->srcH && 3 * c->d[1].n == 4 * cw && 3 * c->d[3].n == 4 * chh &&
                    !(t->srcW % 6) && t->srcW >= 18 && !(t->srcH % 3) && !(chh % 3) &&
                    ((sl || fmt_nv(t->srcFormat)) ? !(cw % 3) && cw >= 9 : !(cw % 6) && cw >= 18) &&
                    bank_nowrap_depth(c->f[0].data(), c->d[0].n, sd, c->d[0].size) && bank_nowrap_depth(c->f[1].data(), c->d[1].n, sd, c->d[1].size))
                    u32_build(c, limits, 3, 4);
                if (c->u32_ok && w8u)
                    c->widen8 = 1;
            }
            /* ... and exact 3:2 DOWN (1080p -> 720p, 4K -> 1440p) between 9..14-bit formats laid out alike: the 16-bit twin of sws_down32.hip */
            /* (also into an 8-bit target laid out alike — P01x -> NV12, planar -> planar — with the ordered dither on the way out, as `dn8` above) */
            if (sd > 8 && sd <= 14 && ((dd > 8 && dd <= 14 && sl == dl) || dn8) && sl != 2 && !hrgb && !c->flat_dither && t->src_range == t->dst_range &&
                2 * t->srcW == 3 * t->dstW && 2 * t->srcH == 3 * t->dstH && 2 * cw == 3 * cdw && 2 * chh == 3 * cdh &&
                !(t->dstW & 3) && t->dstW >= 12 && !(t->dstH & 1) && !(cdh & 1) && (sl ? !(cdw & 1) && cdw >= 6 : !(cdw & 3) && cdw >= 12) &&
                bank_nowrap_depth(c->f[0].data(), c->d[0].n, sd, c->d[0].size) && bank_nowrap_depth(c->f[1].data(), c->d[1].n, sd, c->d[1].size))
                d32_build(c, limits);
            /* ... and exact 4:3 down (1440p -> 1080p): the twin's second period */
            if (!c->d32_ok && sd > 8 && sd <= 14 && dd > 8 && dd <= 14 && sl == dl && sl != 2 && !hrgb && !c->flat_dither && t->src_range == t->dst_range &&
                3 * t->srcW == 4 * t->dstW && 3 * t->srcH == 4 * t->dstH && 3 * cw == 4 * cdw && 3 * chh == 4 * cdh &&
                !(t->dstW % 6) && t->dstW >= 18 && !(t->dstH % 3) && !(cdh % 3) && (sl ? !(cdw % 3) && cdw >= 9 : !(cdw % 6) && cdw >= 18) &&
                bank_nowrap_depth(c->f[0].data(), c->d[0].n, sd, c->d[0].size) && bank_nowrap_depth(c->f[1].data(), c->d[1].n, sd, c->d[1].size))
                d32_build(c, limits, 4, 3);
            /* every other ratio between 9..14-bit formats whose banks have at most 8 taps: the 16-bit column walker (sws_walk16.hip);
             * no range change (it carries no range stage), no 8-bit side */
            /* (round 5: also an 8-bit planar / NV12 target fed from a 9..14-bit source — a 10-bit decoder's frames for an 8-bit consumer:
             * the 16-bit horizontal pass, yuv2planeX_8_c / yuv2nv12cX_c with the ordered dither on the way out) */
            const bool to8 = dd == 8 && (t->dstFormat == FFHIP_PIX_FMT_NV12 || !fmt_nv(t->dstFormat));
            /* (round 6: an 8-bit planar / NV12 source into a 9..14-bit target too — hScale8To15_c is hScale16To15_c at depth 8: the walker on
             * planes widened to 16-bit samples by a pass of their own; was the tiled k_sws_scale16 at 0.05 of HBM) */
            const bool widen = sd == 8 && dd > 8 && dd <= 14 && t->srcFormat != FFHIP_PIX_FMT_NV21 && !c->flat_dither;
            if (!c->up2_ok && !c->dn2_ok && ((sd > 8 && sd <= 14) || widen) && ((dd > 8 && dd <= 14) || to8) && (sl != 2 || widen) && dl != 2 &&
                (t->src_range == t->dst_range || hrgb /* (the source's range lives in the yuv2rgb tables) */) &&
                c->d[0].size <= 16 && c->d[1].size <= 16 && c->d[2].size <= 16 && c->d[3].size <= 16 &&
                bank_nowrap_depth(c->f[0].data(), c->d[0].n, sd, c->d[0].size) && bank_nowrap_depth(c->f[1].data(), c->d[1].n, sd, c->d[1].size)) {
                const int hmax = c->d[0].size > c->d[1].size ? c->d[0].size : c->d[1].size, vmax = c->d[2].size > c->d[3].size ? c->d[2].size : c->d[3].size;
                int ht = hmax <= 4 ? 4 : hmax <= 8 ? 8 : 16, vt = vmax <= 4 ? 4 : vmax <= 8 ? 8 : 16;
                if (ht == 16 && vt == 4) vt = 8;   /* (the instantiated pairs: 4x4 .. 8x8, 16x8, 8x16, 16x16) */
                if (vt == 16 && ht == 4) ht = 8;
                const int T[4] = { ht, ht, vt, vt };
                std::vector<int16_t> pf[4];
                std::vector<int32_t> pp[4];
                bool ok = true;
                for (int i = 0; i < 4 && ok; i++)
                    ok = ffhip_w16_pad_bank(c->f[i].data(), c->p[i].data(), c->d[i].size, c->d[i].n, limits[i], T[i], &pf[i], &pp[i]);
                size_t off[8], tot = 0;
                for (int i = 0; i < 4; i++) {
                    off[2 * i] = tot;     tot += (pf[i].size() * 2 + 255) & ~(size_t)255;
                    off[2 * i + 1] = tot; tot += (pp[i].size() * 4 + 255) & ~(size_t)255;
                }
                if (ok && hipMalloc(&c->w16_dev, tot) == hipSuccess) {
                    uint8_t *b = static_cast<uint8_t *>(c->w16_dev);
                    for (int i = 0; i < 4 && ok; i++)
                        ok = hipMemcpy(b + off[2 * i], pf[i].data(), pf[i].size() * 2, hipMemcpyHostToDevice) == hipSuccess &&
                             hipMemcpy(b + off[2 * i + 1], pp[i].data(), pp[i].size() * 4, hipMemcpyHostToDevice) == hipSuccess;
                    if (ok) {
                        for (int i = 0; i < 4; i++) {
                            c->w16_f[i] = reinterpret_cast<const int16_t *>(b + off[2 * i]);
                            c->w16_p[i] = reinterpret_cast<const int32_t *>(b + off[2 * i + 1]);
                        }
                        c->w16_ht = ht;
                        c->w16_vt = vt;
                        c->w16_ok = 1;
                        c->widen8 = widen;
                        for (int i = 0; i < 2; i++) {
                            c->w16_span[i][0] = ffhip_w16_span(pp[i].data(), c->d[i].n, 256, 2, ht);
                            c->w16_span[i][1] = ffhip_w16_span(pp[i].data(), c->d[i].n, 128, 2, ht);
                            c->w16_span[i][2] = ffhip_w16_span(pp[i].data(), c->d[i].n, 128, 4, ht);
                        }
                    }
                }
            }
        }
        if (hrgb) {
            /* packed_vscale() picks the writer per target row (vscale.c:126-170): yuv2rgb_1 for one luma tap with one chroma tap or a blending
             * pair, yuv2rgb_2 for blending pairs on both, yuv2rgb_X otherwise.  _X and the one-tap _1 are the walker's sums with the flat 64
             * (Y = (l * 4096 + (64 << 12)) >> 19 = (l + 64) >> 7); _2 has no rounding term (seed 0); _1 with a chroma pair averages or drops
             * a line ((u0 + u1 + 128) >> 8 / (u0 + 64) >> 7, output.c:1889-1939): not the sums — refused */
            const int lfs = c->d[2].size, cfs = c->d[3].size;
            auto blend = [](const int16_t *f) { return (uint16_t)f[1] + (uint16_t)f[0] == 4096 && (uint16_t)f[1] <= 4096U; };
            int n2 = 0, n1 = 0;
            for (int y = 0; y < t->dstH; y++) {
                const bool cb = cfs == 2 && blend(c->f[3].data() + (size_t)2 * y), lb = lfs == 2 && blend(c->f[2].data() + (size_t)2 * y);
                n1 += lfs == 1 && cb;
                n2 += lb && cb;
            }
            if (n1 || (n2 && n2 != t->dstH)) {
                ffhip_set_error("ffhip_sws: above 8 bits into packed RGB: the vertical banks mix yuv2rgb_1 / _2 rows with others (bilinear at this ratio): not on the hip path");
                ffhip_sws_freeContext(c);
                return nullptr;
            }
            c->hrgb_seed0 = n2 == t->dstH;
        }
        if (hrgb && !c->w16_ok) {
            ffhip_set_error("ffhip_sws: above 8 bits into packed RGB runs on the 16-bit walker: banks of at most 16 taps that cannot wrap (and 9..14-bit sources)");
            ffhip_sws_freeContext(c);
            return nullptr;
        }
        return c;
    }
    int r = 0;
    if (fmt_rgb(t->dstFormat)) {
        FFHipScaleRgbArgs &a = c->rgb;
        memset(&a, 0, sizeof(a));
        a.srcW = t->srcW; a.srcH = t->srcH; a.chrSrcW = c->chrSrcW; a.chrSrcH = c->chrSrcH;
        a.dstW = t->dstW; a.dstH = t->dstH;
        a.hl = c->d[0]; a.hc = c->d[1]; a.vl = c->d[2]; a.vc = c->d[3];
        a.bgr = rgb_layout(t->dstFormat);
        a.k = c->k;
        a.full = t->full_chr_h_int != 0; /* SWS_FULL_CHR_H_INT: a chroma sample per pixel, the yuv2rgb_full_* writers */
        for (int i = 0; i < 6; i++)
            a.fk[i] = t->yuv2rgb_full[i];
        a.has_alpha = t->dst_alpha_fill == 2; /* the source's alpha plane into the alpha byte: the general kernel (the walker has no alpha lane) */
        if (a.vc.n != t->dstH || a.hc.n != (a.full ? t->dstW : (t->dstW + 1) / 2)) {
            ffhip_set_error("ffhip_sws: chroma banks do not match a packed-RGB target (need chrDstH == dstH)");
            ffhip_sws_freeContext(c);
            return nullptr;
        }
        r = ffhip_plan_scale_rgb(&a, c->p[0].data(), c->p[1].data(), c->p[2].data(), c->p[3].data());
        /* planar 4:4:4 at the source's size (what sws_scale() runs for yuv444p -> rgb24: full chroma forced, no table converter): every
         * bank one tap on the sample itself -> the streaming kernel of sws_full444.hip */
        if (!r && a.full && !a.has_alpha && t->srcFormat == FFHIP_PIX_FMT_YUV444P && t->srcW == t->dstW && t->srcH == t->dstH && t->dstW >= 8) {
            bool id = true;
            for (int b = 0; b < 4 && id; b++) {
                id = c->d[b].size == 1 && c->d[b].n == (b < 2 ? t->dstW : t->dstH);
                for (int x = 0; x < c->d[b].n && id; x++)
                    id = c->p[b][x] == x && c->f[b][x] == (b < 2 ? 16384 : 4096);
            }
            c->f444_ok = id;
        }
        /* column walker with RGB output: 4-tap vertical banks (yuv2rgb_X), <= 4-tap horizontal banks, no int16 wrap */
        const int limits[4] = { a.srcW, a.chrSrcW, a.srcH, a.chrSrcH };
        if (!r && !a.full && !a.has_alpha && !(t->dstW & 7) && build_fast_view(c, limits, true))
            c->cw_rgb = ffhip_cw_bank_ok(c->np[0].data(), 4, c->d[0].n, a.srcW, c->np[2].data(), 4, c->d[2].n, a.srcH) &&
                        ffhip_cw_bank_ok(c->np[1].data(), 4, c->d[1].n, a.chrSrcW, c->np[3].data(), 4, c->d[3].n, a.chrSrcH) &&
                        c->d[1].n * 2 == t->dstW && ffhip_cw_bank_nowrap(c->nf[0].data(), 4, c->d[0].n) &&
                        ffhip_cw_bank_nowrap(c->nf[1].data(), 4, c->d[1].n);
        /* ... and, for exact 2x of 4:2:0 (planar or NV12 / NV21), the static-schedule kernel with the RGB writer (sws_up2rgb.hip) */
        if (c->cw_rgb && (t->srcFormat == FFHIP_PIX_FMT_YUV420P || fmt_nv(t->srcFormat)))
            up2rgb_build(c, t->srcW, t->srcH, t->dstW, t->dstH);
        /* ... and the same sources at the source's own size (what sws_scale() runs for NV12 -> rgb24: no table converter exists for it) */
        if (c->cw_rgb && (t->srcFormat == FFHIP_PIX_FMT_YUV420P || fmt_nv(t->srcFormat)) && t->srcW == t->dstW && t->srcH == t->dstH)
            eqrgb_build(c, t->srcW, t->srcH, a.chrSrcW, a.chrSrcH);
        /* ... and every ratio the walker does not take (banks above 4 taps: all down-scaling) whose vertical luma bank has 3 taps or more
         * (the reference then runs yuv2rgb_X, seed 1 << 18: vscale.c:126-170) goes in TWO stages: the wide-bank walker on these very banks
         * into the target's own geometry (luma as unclipped int16, a chroma line per output line), then the tables' closed form
         * (sws_y16rgb.hip) — against the LDS-tiled k_scale_rgb at 0.05 of HBM */
        if (!r && !c->cw_rgb && !a.full && !a.has_alpha && !(t->dstW & 1) && c->d[2].size >= 3) {
            bool ok = wide_setup(c, limits, fmt_nv(t->srcFormat));
            if (fmt_nv(t->srcFormat) && (a.chrSrcW & 3))
                ok = false;
            c->lw_ok = ok;
            /* exact 2:1 (4K -> 1080p): the luma on the static-schedule kernel */
            if (ok && t->srcW == 2 * t->dstW && t->srcH == 2 * t->dstH && !(t->dstW & 3) && t->dstW >= 12 &&
                ffhip_cw_bank_nowrap(c->f[0].data(), c->d[0].size, c->d[0].n))
                dn2_build_luma(c, t->srcW, t->srcH);
        }
    } else {
        FFHipScalePlaneArgs &l = c->lum, &ch = c->chr;
        memset(&l, 0, sizeof(l));
        memset(&ch, 0, sizeof(ch));
        l.srcW = t->srcW; l.srcH = t->srcH; l.dstW = t->dstW; l.dstH = t->dstH;
        l.h = c->d[0]; l.v = c->d[2];
        ch.srcW = c->chrSrcW; ch.srcH = c->chrSrcH; ch.dstW = c->d[1].n; ch.dstH = c->d[3].n;
        ch.h = c->d[1]; ch.v = c->d[3];
        const bool rc = t->src_range != t->dst_range;
        if (rc) {
            /* lum / chrRangeToJpeg_c, ...FromJpeg_c (swscale.c:160-207): 16-bit coefficient, 32-bit offset at 15-bit intermediates */
            l.rc_coeff = (int)(uint16_t)t->lumConvertRange_coeff; l.rc_offset = (int)t->lumConvertRange_offset; l.rc_clip = !t->src_range;
            ch.rc_coeff = (int)(uint16_t)t->chrConvertRange_coeff; ch.rc_offset = (int)t->chrConvertRange_offset; ch.rc_clip = !t->src_range;
        }
        r = ffhip_plan_scale_plane(&l, 1, c->p[0].data(), c->p[2].data());
        if (!r)
            r = ffhip_plan_scale_plane(&ch, 2, c->p[1].data(), c->p[3].data());
        if (rc) {
            /* of the fast kernels only the exact-2x one carries the range stage (round 4: k_sws_up2<., ., 0, 1>); everything else
             * about such a context is the general tiled kernel's */
            if (r) {
                ffhip_set_error("ffhip_sws: bank sizes outside the tiled kernel's range");
                ffhip_sws_freeContext(c);
                return nullptr;
            }
            const int lim[4] = { l.srcW, ch.srcW, l.srcH, ch.srcH };
            auto rc_floor_ok = [&](const std::vector<int16_t> &f, int n, int coeff, int offset) {
                /* the lowest horizontal sample a bank row can produce, through the conversion: the int16 pack saturates where the
                 * reference's store wraps, so nothing may fall below -32768 */
                for (int x = 0; x < n; x++) {
                    int neg = 0;
                    for (int j = 0; j < 4; j++)
                        if (f[(size_t)x * 4 + j] < 0)
                            neg += f[(size_t)x * 4 + j];
                    const long long hmin = (255LL * neg) >> 7;
                    if (((hmin * coeff + offset) >> 14) < -32768)
                        return false;
                }
                return true;
            };
            if (l.dstW == 2 * l.srcW && l.dstH == 2 * l.srcH && ch.dstW == 2 * ch.srcW && ch.dstH == 2 * ch.srcH &&
                fmt_nv(t->srcFormat) == fmt_nv(t->dstFormat) && !(l.srcW & 3) && l.srcW >= 8 &&
                (fmt_nv(t->srcFormat) ? !(ch.srcW & 1) && ch.srcW >= 4 : !(ch.srcW & 3) && ch.srcW >= 8) && build_fast_view(c, lim, false) &&
                ffhip_cw_bank_ok(c->np[0].data(), 4, c->d[0].n, l.srcW, c->np[2].data(), 4, c->d[2].n, l.srcH) &&
                ffhip_cw_bank_ok(c->np[1].data(), 4, c->d[1].n, ch.srcW, c->np[3].data(), 4, c->d[3].n, ch.srcH) &&
                ffhip_cw_bank_nowrap(c->nf[0].data(), 4, c->d[0].n) && ffhip_cw_bank_nowrap(c->nf[1].data(), 4, c->d[1].n) &&
                rc_floor_ok(c->nf[0], c->d[0].n, l.rc_coeff, l.rc_offset) && rc_floor_ok(c->nf[1], c->d[1].n, ch.rc_coeff, ch.rc_offset)) {
                const int nsrc[4] = { l.srcW, ch.srcW, l.srcH, ch.srcH };
                up2_build(c, nsrc);
                c->cw_ok = c->up2_ok; /* the dispatcher's gate; c->up2_rc keeps every other fast kernel out */
                c->up2_rc = c->up2_ok;
            }
            return c;
        }
        /* 4:2:0 planar <-> semi-planar at the same size: every bank one tap on the sample itself, the result is the source's bytes laid
         * out differently (sws_copy420.hip) */
        {
            auto is420 = [](int f) { return f == FFHIP_PIX_FMT_YUV420P || f == FFHIP_PIX_FMT_NV12 || f == FFHIP_PIX_FMT_NV21; };
            bool id = !r && is420(t->srcFormat) && is420(t->dstFormat) && t->srcW == t->dstW && t->srcH == t->dstH && t->dst_alpha_fill != 2 &&
                      t->dstW >= 16 && c->d[1].n >= 16;
            for (int b = 0; b < 4 && id; b++) {
                id = c->d[b].size == 1;
                for (int x = 0; x < c->d[b].n && id; x++)
                    id = c->p[b][x] == x && c->f[b][x] == (b < 2 ? 16384 : 4096);
            }
            c->c420_ok = id;
        }
        /* planar 4:4:4 -> 4:2:0 at the same size (a capture's frames subsampled for an encoder): the luma plane is the source's (one-tap
         * banks), the chroma planes go exactly 2:1 both ways — the layout kernel's copy job + the static-schedule kernel on two planes,
         * against the wide walker computing all three with its 8-tap classes */
        if (!r && !c->c420_ok && t->srcFormat == FFHIP_PIX_FMT_YUV444P && t->dstFormat == FFHIP_PIX_FMT_YUV420P && t->srcW == t->dstW &&
            t->srcH == t->dstH && t->dst_alpha_fill != 2 && t->dstW >= 16 && !(c->d[1].n & 3) && c->d[1].n >= 12) {
            bool id = true;
            for (int b = 0; b < 4 && id; b += 2) {
                id = c->d[b].size == 1;
                for (int x = 0; x < c->d[b].n && id; x++)
                    id = c->p[b][x] == x && c->f[b][x] == (b < 2 ? 16384 : 4096);
            }
            std::vector<uint32_t> vb[2];
            if (id && ffhip_cw_bank_nowrap(c->f[1].data(), c->d[1].size, c->d[1].n) &&
                ffhip_down2_virtual_bank(c->f[1].data(), c->p[1].data(), c->d[1].size, c->d[1].n, c->chrSrcW, &vb[0]) &&
                ffhip_down2_virtual_bank(c->f[3].data(), c->p[3].data(), c->d[3].size, c->d[3].n, c->chrSrcH, &vb[1])) {
                vb[1].resize((size_t)(c->d[3].n + 8) * 4, 0); /* the row loop reads four rows of coefficients at a time */
                const size_t o1 = (vb[0].size() * 4 + 255) & ~(size_t)255;
                if (hipMalloc(&c->dn2_dev, o1 + vb[1].size() * 4) == hipSuccess) {
                    uint8_t *bb = static_cast<uint8_t *>(c->dn2_dev);
                    if (hipMemcpy(bb, vb[0].data(), vb[0].size() * 4, hipMemcpyHostToDevice) == hipSuccess &&
                        hipMemcpy(bb + o1, vb[1].data(), vb[1].size() * 4, hipMemcpyHostToDevice) == hipSuccess) {
                        c->dn2_h[1] = reinterpret_cast<const uint32_t *>(bb);
                        c->dn2_v[1] = reinterpret_cast<const uint32_t *>(bb + o1);
                        c->mix_dn2 = 1;
                    }
                }
            }
        }
        const int limits[4] = { l.srcW, ch.srcW, l.srcH, ch.srcH };
        {
            const bool ok = build_fast_view(c, limits, false);
            c->cw_ok = ok && ffhip_cw_bank_ok(c->np[0].data(), 4, c->d[0].n, l.srcW, c->np[2].data(), 4, c->d[2].n, l.srcH) &&
                       ffhip_cw_bank_ok(c->np[1].data(), 4, c->d[1].n, ch.srcW, c->np[3].data(), 4, c->d[3].n, ch.srcH);
        }
        c->cw_opt = c->cw_ok && ffhip_cw_bank_nowrap(c->nf[0].data(), 4, c->d[0].n) &&
                    ffhip_cw_bank_nowrap(c->nf[1].data(), 4, c->d[1].n);
        c->cw_dup = c->cw_opt && ffhip_cw_bank_dup12(c->np[0].data(), c->d[0].n) && ffhip_cw_bank_dup12(c->np[1].data(), c->d[1].n);
        /* exact 2x in both directions, chroma laid out alike on both sides: the static-schedule kernel (sws_up2.hip) */
        if (c->cw_opt && l.dstW == 2 * l.srcW && l.dstH == 2 * l.srcH && ch.dstW == 2 * ch.srcW && ch.dstH == 2 * ch.srcH &&
            fmt_nv(t->srcFormat) == fmt_nv(t->dstFormat) && !(l.srcW & 3) && l.srcW >= 8 &&
            (fmt_nv(t->srcFormat) ? !(ch.srcW & 1) && ch.srcW >= 4 : !(ch.srcW & 3) && ch.srcW >= 8)) {
            const int nsrc[4] = { l.srcW, ch.srcW, l.srcH, ch.srcH };
            up2_build(c, nsrc);
        }
        /* planar 4:2:0 -> 4:4:4 at the same size: the luma plane is the source's, the chroma planes go exactly 2x both ways */
        if (c->cw_opt && !c->up2_ok && t->srcFormat == FFHIP_PIX_FMT_YUV420P && t->dstFormat == FFHIP_PIX_FMT_YUV444P && l.dstW == l.srcW &&
            l.dstH == l.srcH && ch.dstW == 2 * ch.srcW && ch.dstH == 2 * ch.srcH && !(ch.srcW & 3) && ch.srcW >= 8 && l.dstW >= 16 &&
            t->dst_alpha_fill != 2) {
            bool id = true;
            for (int b = 0; b < 4 && id; b += 2) {
                id = c->d[b].size == 1;
                for (int x = 0; x < c->d[b].n && id; x++)
                    id = c->p[b][x] == x && c->f[b][x] == (b < 2 ? 16384 : 4096);
            }
            if (id) {
                const int nsrc[4] = { l.srcW, ch.srcW, l.srcH, ch.srcH };
                up2_build(c, nsrc, true);
            }
        }
        /* wide banks (down-scaling, long kernels): the LDS-backed walker; FFHIP_SWS_WIDE=1 builds it for narrow banks
         * too (parity tests of that kernel on up-scaling cases) */
        {
            const char *ew = FFHIP_KNOB("FFHIP_SWS_WIDE");
            if (!c->cw_ok || (ew && ew[0] == '1')) {
                bool ok = wide_setup(c, limits, fmt_nv(t->srcFormat) || fmt_nv(t->dstFormat));
                if (fmt_nv(t->srcFormat) && (ch.srcW & 3))
                    ok = false;
                c->lw_ok = ok;
            }
        }
        /* exact 2:1 in both directions, chroma laid out alike on both sides, no horizontal sum leaves int16: the
         * static-schedule kernel (sws_down2.hip) */
        if (ffhip_cw_bank_nowrap(c->f[0].data(), c->d[0].size, c->d[0].n) && ffhip_cw_bank_nowrap(c->f[1].data(), c->d[1].size, c->d[1].n) &&
            l.srcW == 2 * l.dstW && l.srcH == 2 * l.dstH && ch.srcW == 2 * ch.dstW && ch.srcH == 2 * ch.dstH &&
            fmt_nv(t->srcFormat) == fmt_nv(t->dstFormat) && !(l.dstW & 3) && l.dstW >= 12 &&
            (fmt_nv(t->srcFormat) ? !(ch.dstW & 1) && ch.dstW >= 6 : !(ch.dstW & 3) && ch.dstW >= 12)) {
            const int nsrc[4] = { l.srcW, ch.srcW, l.srcH, ch.srcH };
            dn2_build(c, nsrc);
        }
        /* exact 3:2 in both directions (1080p -> 720p, 4K -> 1440p), chroma laid out alike on both sides: sws_down32.hip */
        if (ffhip_cw_bank_nowrap(c->f[0].data(), c->d[0].size, c->d[0].n) && ffhip_cw_bank_nowrap(c->f[1].data(), c->d[1].size, c->d[1].n) &&
            2 * l.srcW == 3 * l.dstW && 2 * l.srcH == 3 * l.dstH && 2 * ch.srcW == 3 * ch.dstW && 2 * ch.srcH == 3 * ch.dstH &&
            fmt_nv(t->srcFormat) == fmt_nv(t->dstFormat) && !(l.dstW & 7) && l.dstW >= 24 && !(l.dstH & 1) && !(ch.dstH & 1) &&
            (fmt_nv(t->srcFormat) ? !(ch.dstW & 3) && ch.dstW >= 12 : !(ch.dstW & 7) && ch.dstW >= 24) && t->dst_alpha_fill != 2) {
            const int nsrc[4] = { l.srcW, ch.srcW, l.srcH, ch.srcH };
            d32_build(c, nsrc);
        }
        /* exact 3:2 / 4:3 UP (720p -> 1080p, 1080p -> 1440p), chroma laid out alike and in the same order on both sides: the 8-bit twin of
         * sws_up32.hip (round 6; was the 4-tap column walker at 0.35 - 0.40 of HBM) */
        for (int q = 0; q < 2 && !c->u32_ok; q++) {
            const int pin = q ? 3 : 2, pout = q ? 4 : 3, no = 4 * pout;
            if (ffhip_cw_bank_nowrap(c->f[0].data(), c->d[0].size, c->d[0].n) && ffhip_cw_bank_nowrap(c->f[1].data(), c->d[1].size, c->d[1].n) &&
                pin * l.dstW == pout * l.srcW && pin * l.dstH == pout * l.srcH && pin * ch.dstW == pout * ch.srcW && pin * ch.dstH == pout * ch.srcH &&
                t->srcFormat == t->dstFormat /* (the same layout and channel order) */ && !(l.dstW % no) && l.dstW >= 3 * no && !(l.dstH % pout) &&
                !(ch.dstH % pout) && (fmt_nv(t->srcFormat) ? !(ch.dstW % (no / 2)) && ch.dstW >= 3 * no / 2 : !(ch.dstW % no) && ch.dstW >= 3 * no) &&
                t->dst_alpha_fill != 2) {
                const int nsrc[4] = { l.srcW, ch.srcW, l.srcH, ch.srcH };
                u32_build(c, nsrc, pin, pout);
            }
        }
        /* MFMA variant: same banks; chroma either byte-interleaved on both sides or planar on both sides */
        const bool nv_in = fmt_nv(t->srcFormat), nv_out = fmt_nv(t->dstFormat);
        if (c->cw_opt && nv_in == nv_out) {
            std::vector<uint8_t> tl, tc;
            const int nl = ffhip_mf_build_tiles(&tl, c->nf[0].data(), c->np[0].data(), c->d[0].n, l.srcW, 0, 0);
            const int nc = ffhip_mf_build_tiles(&tc, c->nf[1].data(), c->np[1].data(), c->d[1].n, ch.srcW, nv_in,
                                                t->srcFormat == FFHIP_PIX_FMT_NV21);
            if (nl > 0 && nc > 0) {
                std::vector<int32_t> ys[2];
                const int srcHs[2] = { l.srcH, ch.srcH };
                for (int k = 0; k < 2; k++) {
                    const std::vector<int32_t> &vp = c->np[2 + k];
                    ys[k].assign(srcHs[k] + 1, (int32_t)vp.size());
                    int y = 0;
                    for (int p = 0; p <= srcHs[k]; p++) {
                        while (y < (int)vp.size() && vp[y] < p)
                            y++;
                        ys[k][p] = y;
                    }
                }
                const size_t o1 = (tl.size() + 255) & ~(size_t)255, o2 = o1 + ((tc.size() + 255) & ~(size_t)255);
                const size_t o3 = o2 + ((ys[0].size() * 4 + 255) & ~(size_t)255), tot = o3 + ys[1].size() * 4 + 256;
                if (hipMalloc(&c->mf_dev, tot) == hipSuccess) {
                    uint8_t *b = static_cast<uint8_t *>(c->mf_dev);
                    if (hipMemcpy(b, tl.data(), tl.size(), hipMemcpyHostToDevice) == hipSuccess &&
                        hipMemcpy(b + o1, tc.data(), tc.size(), hipMemcpyHostToDevice) == hipSuccess &&
                        hipMemcpy(b + o2, ys[0].data(), ys[0].size() * 4, hipMemcpyHostToDevice) == hipSuccess &&
                        hipMemcpy(b + o3, ys[1].data(), ys[1].size() * 4, hipMemcpyHostToDevice) == hipSuccess) {
                        c->mf_tiles[0] = b; c->mf_tiles[1] = b + o1;
                        c->mf_ys[0] = reinterpret_cast<const int32_t *>(b + o2);
                        c->mf_ys[1] = reinterpret_cast<const int32_t *>(b + o3);
                        c->mf_ntiles[0] = nl; c->mf_ntiles[1] = nc;
                        c->mf_chr_pair = nv_in;
                        c->mf_ok = 1;
                    }
                }
            }
        }
    }
    if (r < 0) {
        ffhip_sws_freeContext(c);
        return nullptr;
    }
    return c;
}

extern "C" FFHipSwsContext *ffhip_sws_getContext(int srcW, int srcH, int srcFormat, int dstW, int dstH, int dstFormat,
                                                 int flags)
{
    /* a packed RGB source in front of a YUV target: the context of the 14-bit planar lines its converters make (kernels/sws_rgbin.hip) */
    FFHipSwsRgbIn ri;
    const int inner = ffhip_sws_rgb_source_plan(srcW, srcH, srcFormat, dstW, dstH, dstFormat, flags, &ri.half, ri.table, &ri.bpp, ri.ofs);
    if (inner) {
        ri.fmt = srcFormat;
        srcFormat = inner;
    }
    FFHipSwsHostTables *h = ffhip_sws_tables_create(srcW, srcH, srcFormat, dstW, dstH, dstFormat, flags);
    if (!h)
        return nullptr;
    FFHipSwsTables t;
    ffhip_sws_tables_get(h, &t);
    g_sws_create_flat_dither = inner != 0;
    FFHipSwsContext *c = ffhip_sws_from_tables(&t);
    g_sws_create_flat_dither = 0;
    ffhip_sws_tables_free(h);
    if (c && inner) {
        c->rgb_in = ri;
        rgb_in_plan_luma(c);
    }
    return c;
}

extern "C" FFHipSwsContext *ffhip_sws_from_tables_rgb_source(const FFHipSwsTables *t, int rgbFormat, const int32_t rgb2yuv[9])
{
    if (!t || !rgb2yuv || (t->srcFormat != FFHIP_PIX_FMT_YUV422P14LE && t->srcFormat != FFHIP_PIX_FMT_YUV444P14LE) || t->src_range != t->dst_range) {
        ffhip_set_error("ffhip_sws_from_tables_rgb_source: the tables must describe the converter lines (yuv422p14le / yuv444p14le source, equal ranges)");
        return nullptr;
    }
    FFHipSwsRgbIn ri;
    int32_t unused[9];
    int half = 0;
    /* the formats and the component bytes from the plan the stand-alone constructor uses; half and the table are the caller's */
    if (!ffhip_sws_rgb_source_plan(t->srcW, t->srcH, rgbFormat, t->dstW, t->dstH, t->dstFormat, t->flags, &half, unused, &ri.bpp, ri.ofs)) {
        ffhip_set_error("ffhip_sws_from_tables_rgb_source: format pair %d -> %d is not on the hip path", rgbFormat, t->dstFormat);
        return nullptr;
    }
    ri.half = t->srcFormat == FFHIP_PIX_FMT_YUV422P14LE;
    if (ri.half && (t->srcW & 1)) {
        ffhip_set_error("ffhip_sws_from_tables_rgb_source: half-width chroma needs an even source width");
        return nullptr;
    }
    ri.fmt = rgbFormat;
    memcpy(ri.table, rgb2yuv, sizeof(ri.table));
    g_sws_create_flat_dither = 1;
    FFHipSwsContext *c = ffhip_sws_from_tables(t);
    g_sws_create_flat_dither = 0;
    if (c) {
        c->rgb_in = ri;
        rgb_in_plan_luma(c);
    }
    return c;
}

extern "C" int ffhip_sws_set_rgb2yuv(FFHipSwsContext *c, const int32_t rgb2yuv[9])
{
    if (!c || !rgb2yuv || !c->rgb_in.bpp)
        return FFHIP_EINVAL;
    std::lock_guard<std::mutex> lk(c->mu);
    memcpy(c->rgb_in.table, rgb2yuv, sizeof(c->rgb_in.table));
    return 0;
}

/* identity luma banks into an 8-bit luma plane that the 16-bit walker would produce: the converter pass can write it */
static void rgb_in_plan_luma(FFHipSwsContext *c)
{
    const FFHipSwsTables &t = c->t;
    int dd = 8, dl = 0;
    (void)ffhip_pixfmt_hbd(t.dstFormat, &dd, &dl, nullptr, nullptr);
    bool id = c->w16_ok && dd == 8 && !fmt_rgb(t.dstFormat) && t.srcW == t.dstW && t.srcH == t.dstH && c->d[0].size == 1 && c->d[2].size == 1 &&
              t.src_range == t.dst_range;
    for (int x = 0; id && x < t.dstW; x++)
        id = c->f[0][(size_t)x] == 16384 && c->p[0][(size_t)x] == x;
    for (int y = 0; id && y < t.dstH; y++)
        id = c->f[2][(size_t)y] == 4096 && c->p[2][(size_t)y] == y;
    c->rgb_in.y_direct = id;
    /* every bank the identity (round 6): a planar 8-bit target with a chroma sample per converter sample and a chroma row per source row */
    {
        bool dc = id && c->flat_dither == 1 && c->chrSrcH == t.srcH && c->d[3].n == t.srcH && c->d[3].size == 1 && c->d[1].size == 1 &&
                  c->d[1].n == (c->rgb_in.half ? t.srcW / 2 : t.srcW) && !fmt_nv(t.dstFormat) && !t.dst_alpha_fill;
        for (int x = 0; dc && x < c->d[1].n; x++)
            dc = c->f[1][(size_t)x] == 16384 && c->p[1][(size_t)x] == x;
        for (int y = 0; dc && y < c->d[3].n; y++)
            dc = c->f[3][(size_t)y] == 4096 && c->p[3][(size_t)y] == y;
        c->rgb_in.direct_c = dc;
    }
    /* the whole conversion in one kernel (round 6): 4:2:0 target, chroma read at half width through the identity bank, 2:1 down the rows on
     * a bank of the exact-2:1 shape, the flat seed 64 */
    bool f4 = id && c->rgb_in.half && c->flat_dither == 1 && !(t.srcW & 3) && t.srcW >= 4 && !(t.srcH & 1) && c->chrSrcH == t.srcH &&
              c->d[3].n * 2 == t.srcH && c->d[1].n * 2 == t.srcW && c->d[1].size == 1 &&
              (t.dstFormat == FFHIP_PIX_FMT_NV12 || t.dstFormat == FFHIP_PIX_FMT_YUV420P) && !c->rgb_in.vfv;
    for (int x = 0; f4 && x < c->d[1].n; x++)
        f4 = c->f[1][(size_t)x] == 16384 && c->p[1][(size_t)x] == x;
    std::vector<uint32_t> vb;
    if (f4 && ffhip_down2_virtual_bank(c->f[3].data(), c->p[3].data(), c->d[3].size, c->d[3].n, c->chrSrcH, &vb)) {
        vb.resize((size_t)(c->d[3].n + 8) * 4, 0);
        if (hipMalloc(&c->rgb_in.vfv, vb.size() * 4) == hipSuccess &&
            hipMemcpy(c->rgb_in.vfv, vb.data(), vb.size() * 4, hipMemcpyHostToDevice) == hipSuccess)
            c->rgb_in.fused420 = 1;
    }
}

/* the converter pass of an RGB-source context over `rows` source rows of nframes frames: src -> the 14-bit planes at p[] */
static int rgb_in_launch(const FFHipSwsContext *c, int nframes, const uint8_t *src, ptrdiff_t src_stride, size_t src_fp, int rows, uint8_t *const p[3],
                         const int pitch[3], const size_t fp[3], hipStream_t stream, uint8_t *y8 = nullptr, ptrdiff_t y8_stride = 0, size_t y8_fp = 0)
{
    FFHipRgbInArgs a;
    memset(&a, 0, sizeof(a));
    a.y8 = y8; a.y8_stride = y8_stride; a.y8_fp = y8_fp;
    a.src = src; a.src_stride = src_stride; a.src_fp = src_fp;
    for (int i = 0; i < 3; i++) {
        a.dst[i] = p[i]; a.dst_stride[i] = pitch[i]; a.dst_fp[i] = fp[i];
    }
    a.w = c->t.srcW; a.h = rows;
    a.ro = c->rgb_in.ofs[0]; a.go = c->rgb_in.ofs[1]; a.bo = c->rgb_in.ofs[2];
    const int32_t *T = c->rgb_in.table;
    a.ry = T[0]; a.gy = T[1]; a.by = T[2]; a.ru = T[3]; a.gu = T[4]; a.bu = T[5]; a.rv = T[6]; a.gv = T[7]; a.bv = T[8];
    return ffhip_launch_sws_rgb_in(a, c->rgb_in.bpp, c->rgb_in.half, nframes, stream);
}

/* sws_setColorspaceDetails() on a live context (libswscale/utils.c:848-1000 ends in ff_yuv2rgb_c_init_tables() for RGB targets): the
 * coefficient fields of `t` replace the context's, the banks and formats stay */
extern "C" int ffhip_sws_set_yuv2rgb(FFHipSwsContext *c, const FFHipSwsTables *t)
{
    if (!c || !t)
        return FFHIP_EINVAL;
    std::lock_guard<std::mutex> lk(c->mu);
    FFHipSwsTables n = c->t;
    n.yuv2rgb_cy = t->yuv2rgb_cy; n.yuv2rgb_oy = t->yuv2rgb_oy; n.yuv2rgb_crv = t->yuv2rgb_crv; n.yuv2rgb_cbu = t->yuv2rgb_cbu;
    n.yuv2rgb_cgu = t->yuv2rgb_cgu; n.yuv2rgb_cgv = t->yuv2rgb_cgv; n.yuv2rgb_yoffs = t->yuv2rgb_yoffs;
    for (int i = 0; i < 6; i++)
        n.yuv2rgb_full[i] = t->yuv2rgb_full[i];
    FFHipYuv2RgbK k;
    const int r = make_k(n, &k);
    if (r < 0)
        return r;
    c->t = n;
    c->k = k;
    c->rgb.k = k;
    for (int i = 0; i < 6; i++)
        c->rgb.fk[i] = n.yuv2rgb_full[i];
    return 0;
}

extern "C" int ffhip_sws_fast_path(const FFHipSwsContext *c)
{
    return c ? (c->cw_ok || c->cw_rgb) + (c->mf_ok ? 2 : 0) + (c->lw_ok ? 4 : 0) + (c->up2_ok ? 8 : 0) + (c->dn2_ok ? 16 : 0) + (c->w16_ok ? 32 : 0) + (c->u2r_ok ? 64 : 0) + (c->eqr_ok ? 128 : 0) + (c->f444_ok ? 256 : 0) + (c->c420_ok ? 512 : 0) + (c->mix_dn2 ? 1024 : 0) + (c->mix_up2 ? 2048 : 0) + (c->d32_ok ? 4096 : 0) + (c->u32_ok ? 8192 : 0) : 0;
}

extern "C" int ffhip_sws_tuned_numbering(const FFHipSwsContext *c)
{
    return c ? c->tune_choice : -1;
}

extern "C" int ffhip_sws_mfma_tiles_host(const int16_t *filter, const int32_t *pos, int n, int srcW, int pair, int src_swap,
                                         uint8_t *out, size_t out_size)
{
    std::vector<uint8_t> v;
    const int nt = ffhip_mf_build_tiles(&v, filter, pos, n, srcW, pair, src_swap);
    if (nt < 0)
        return FFHIP_EINVAL;
    if (out) {
        if (out_size < v.size())
            return FFHIP_ENOMEM;
        memcpy(out, v.data(), v.size());
    }
    return nt;
}

extern "C" int ffhip_sws_up2_virtual_bank_host(const int16_t *filter, const int32_t *pos, int n_dst, int n_src, uint32_t *out)
{
    std::vector<uint32_t> v;
    if (!filter || !pos || !out || !ffhip_up2_virtual_bank(filter, pos, n_dst, n_src, &v))
        return 0;
    memcpy(out, v.data(), v.size() * 4);
    return 1;
}

extern "C" int ffhip_sws_upn_virtual_bank_host(const int16_t *filter, const int32_t *pos, int n_dst, int n_src, int ratio, uint32_t *out)
{
    std::vector<uint32_t> v;
    if (!filter || !pos || !out || !ffhip_upn_virtual_bank(filter, pos, n_dst, n_src, ratio, &v))
        return 0;
    memcpy(out, v.data(), v.size() * 4);
    return 1;
}

extern "C" int ffhip_sws_up2rgb_hco_host(const uint32_t *hl, int n_hl, const uint32_t *hc, int n_hc, uint32_t out[32])
{
    if (!hl || !hc || !out || n_hl <= 0 || n_hc <= 0)
        return 0;
    const std::vector<uint32_t> a(hl, hl + (size_t)n_hl * 2), b(hc, hc + (size_t)n_hc * 2);
    return ffhip_up2rgb_hco(a, b, out);
}

extern "C" int ffhip_sws_down2_virtual_bank_host(const int16_t *filter, const int32_t *pos, int fsize, int n_dst, int n_src, uint32_t *out)
{
    std::vector<uint32_t> v;
    if (!filter || !pos || !out || !ffhip_down2_virtual_bank(filter, pos, fsize, n_dst, n_src, &v))
        return 0;
    memcpy(out, v.data(), v.size() * 4);
    return 1;
}

extern "C" int ffhip_sws_d32_virtual_bank_host(const int16_t *filter, const int32_t *pos, int fsize, int n_dst, int n_src, uint32_t *out)
{
    std::vector<uint32_t> v;
    if (!filter || !pos || !out || !ffhip_d32_virtual_bank(filter, pos, fsize, n_dst, n_src, 3, &v))
        return 0;
    memcpy(out, v.data(), v.size() * 4);
    return 1;
}

extern "C" void ffhip_sws_freeContext(FFHipSwsContext *c)
{
    if (!c)
        return;
    FFHipDeviceGuard dg(c->device);
    if (c->dev_tables)
        (void)hipFree(c->dev_tables);
    if (c->widen_tmp)
        (void)hipFree(c->widen_tmp);
    if (c->rgb_in.vfv)
        (void)hipFree(c->rgb_in.vfv);
    if (c->rgb_in.planes)
        (void)hipFree(c->rgb_in.planes);
    if (c->rgb_in.stage)
        (void)hipFree(c->rgb_in.stage);
    if (c->mf_dev)
        (void)hipFree(c->mf_dev);
    if (c->up2_dev)
        (void)hipFree(c->up2_dev);
    for (int i = 0; i < 16; i++)
        if (c->tune_ev[i])
            (void)hipEventDestroy(c->tune_ev[i]);
    if (c->u2r_dev)
        (void)hipFree(c->u2r_dev);
    if (c->rgb2_tmp)
        (void)hipFree(c->rgb2_tmp);
    if (c->rgb2_done)
        (void)hipEventDestroy(c->rgb2_done);
    if (c->rgb2_aux) {
        (void)hipStreamSynchronize(c->rgb2_aux);
        (void)hipStreamDestroy(c->rgb2_aux);
    }
    if (c->rgb2_fork)
        (void)hipEventDestroy(c->rgb2_fork);
    if (c->rgb2_join)
        (void)hipEventDestroy(c->rgb2_join);
    if (c->eqr_dev)
        (void)hipFree(c->eqr_dev);
    if (c->w16_dev)
        (void)hipFree(c->w16_dev);
    if (c->dn2_dev)
        (void)hipFree(c->dn2_dev);
    if (c->d32_dev)
        (void)hipFree(c->d32_dev);
    if (c->u32_dev)
        (void)hipFree(c->u32_dev);
    if (c->dev_ntables)
        (void)hipFree(c->dev_ntables);
    if (c->dev_wtables)
        (void)hipFree(c->dev_wtables);
    if (c->stage)
        (void)hipFree(c->stage);
    delete c;
}

/* a context with a side above 8 bits: luma, then the two chroma channels (planes of their own or the halves of an interleaved pair) */
static int scale16(FFHipSwsContext *c, int nframes, const void *const src[4], const int srcStride[4], const size_t srcFramePitch[4],
                   void *const dst[4], const int dstStride[4], const size_t dstFramePitch[4], hipStream_t stream)
{
    const FFHipSwsTables &t = c->t;
    int sd = 8, sl = fmt_nv(t.srcFormat) ? 2 : 0, dd = 8, dl = fmt_nv(t.dstFormat) ? 2 : 0;
    (void)ffhip_pixfmt_hbd(t.srcFormat, &sd, &sl, nullptr, nullptr);
    (void)ffhip_pixfmt_hbd(t.dstFormat, &dd, &dl, nullptr, nullptr);
    const int ssz = sd > 8 ? 2 : 1, dsz = dd > 8 ? 2 : 1;
    const bool hrgb = fmt_rgb(t.dstFormat); /* (round 6) one packed plane: the walker's planes are the context's intermediate */
    for (int pl = 0; pl < (sl ? 2 : 3); pl++)
        if (!src[pl] || (srcStride[pl] % ssz) || (srcFramePitch[pl] % ssz) || ((uintptr_t)src[pl] % ssz))
            return FFHIP_EINVAL;
    for (int pl = 0; pl < (hrgb ? 1 : dl ? 2 : 3); pl++)
        if (!dst[pl] || (dstStride[pl] % dsz) || (dstFramePitch[pl] % dsz) || ((uintptr_t)dst[pl] % dsz))
            return FFHIP_EINVAL;
    /* (round 6) an 8-bit source on the walker: the planes widened to 16-bit samples in the context's own memory; from here on the source is
     * "depth 8 in words", planar or an interleaved pair plane.  When the walker cannot run (alignment) the tiled kernel takes the bytes. */
    const void *wsrc[4] = { src[0], src[1], src[2], nullptr };
    int wss[4] = { srcStride[0], srcStride[1], srcStride[2], 0 };
    size_t wsf[4] = { srcFramePitch[0], srcFramePitch[1], srcFramePitch[2], 0 };
    std::unique_lock<std::mutex> wlk;
    bool widened = false;
    if (c->widen8 && (c->w16_ok || c->up2_ok || c->u32_ok)) {
        bool ok = true;
        for (int pl = 0; ok && pl < (dl ? 2 : 3); pl++)
            ok = dstStride[pl] > 0 && !(((uintptr_t)dst[pl] | (uintptr_t)dstStride[pl] | dstFramePitch[pl]) & 3);
        for (int pl = 0; ok && pl < (sl ? 2 : 3); pl++)
            ok = srcStride[pl] > 0;
        const char *ew = FFHIP_KNOB(c->up2_ok ? "FFHIP_SWS_UP2" : "FFHIP_SWS_WALK16");
        if (ok && !(ew && ew[0] == '0')) {
            const int np = sl ? 2 : 3;
            const int wb[3] = { t.srcW, sl ? 2 * c->chrSrcW : c->chrSrcW, c->chrSrcW }, rows[3] = { t.srcH, c->chrSrcH, c->chrSrcH };
            size_t pitch[3], fp[3], off[3], need = 0;
            for (int pl = 0; pl < np; pl++) {
                pitch[pl] = ((size_t)2 * wb[pl] + 255) & ~(size_t)255;
                fp[pl] = pitch[pl] * (size_t)rows[pl];
                off[pl] = need;
                need += fp[pl] * (size_t)nframes;
            }
            wlk = std::unique_lock<std::mutex>(c->rgb2_mu); /* the planes are the context's: one batch in flight */
            if (need > c->widen_tmp_sz) {
                if (c->widen_tmp) {
                    HIP_TRY(hipDeviceSynchronize());
                    HIP_TRY(hipFree(c->widen_tmp));
                }
                c->widen_tmp = nullptr;
                c->widen_tmp_sz = 0;
                HIP_TRY(hipMalloc(&c->widen_tmp, need));
                c->widen_tmp_sz = need;
            }
            for (int pl = 0; pl < np; pl++) {
                uint8_t *d = static_cast<uint8_t *>(c->widen_tmp) + off[pl];
                const int r = ffhip_launch_sws_widen8(static_cast<const uint8_t *>(src[pl]), srcStride[pl], srcFramePitch[pl], d, (ptrdiff_t)pitch[pl], fp[pl],
                                                      wb[pl], rows[pl], nframes, stream);
                if (r < 0)
                    return r;
                wsrc[pl] = d; wss[pl] = (int)pitch[pl]; wsf[pl] = fp[pl];
            }
            widened = true;
            sl = sl ? 1 : 0; /* an interleaved pair plane of words, as P01x has it (samples in the LOW bits: smsb stays off) */
        }
    }
    if (c->widen8 && !widened) { /* falls to the tiled kernel below with the caller's planes */ }
    {
        uintptr_t al = 0;
        bool neg = false;
        for (int pl = 0; pl < (sl ? 2 : 3); pl++) {
            al |= (uintptr_t)wsrc[pl] | (uintptr_t)wss[pl] | wsf[pl] | (uintptr_t)dst[pl] | (uintptr_t)dstStride[pl] | dstFramePitch[pl];
            neg = neg || wss[pl] < 0 || dstStride[pl] < 0;
        }
        const char *eu = FFHIP_KNOB("FFHIP_SWS_UP2");
        if (c->up2_ok && (widened || !c->widen8) && !(al & 3) && !neg && !(eu && eu[0] == '0')) {
            /* exact 2x above 8 bits: the static-schedule kernel (FFHIP_SWS_UP2=0: the tiled k_sws_scale16) */
            FFHipUp2Args U;
            memset(&U, 0, sizeof(U));
            U.nframes = nframes;
            U.xcd = 1;
            const int cw = c->chrSrcW, chh = c->chrSrcH;
            auto upjob = [&](int which, int plane, int w, int h, int pair) {
                FFHipUp2Job &j = U.job[U.njobs++];
                j.src = static_cast<const uint8_t *>(wsrc[plane]); j.dst = static_cast<uint8_t *>(dst[plane]);
                j.sstride = wss[plane]; j.dstride = dstStride[plane]; j.sfp = wsf[plane]; j.dfp = dstFramePitch[plane];
                j.pair = pair; j.swap = 0;
                j.srcW = w; j.srcH = h;
                j.ngroups = pair ? w / 2 : w / 4;
                j.hfv = c->up2_h[which]; j.vfv = c->up2_v[which];
                j.hb_sdepth = sd; j.hb_ddepth = dd; j.hb_smsb = sl == 1 && !widened; j.hb_dmsb = dl == 1;
            };
            upjob(0, 0, t.srcW, t.srcH, 0);
            if (sl) {
                upjob(1, 1, cw, chh, 1);
            } else {
                upjob(1, 1, cw, chh, 0);
                upjob(1, 2, cw, chh, 0);
            }
            /* frames per wave: as at 8 bits, the split that wastes the fewest lanes at the right edge of the rows */
            int best = 0;
            double bestw = 1e30;
            for (int fsft = 0; fsft <= 2; fsft++) {
                const int lpf = 64 >> fsft;
                bool fits = !(nframes < (1 << fsft) && fsft);
                double w = 0;
                for (int i = 0; i < U.njobs; i++) {
                    const FFHipUp2Job &j = U.job[i];
                    const unsigned long long span_s = (unsigned long long)((1 << fsft) - 1) * j.sfp + (unsigned long long)j.srcH * (size_t)j.sstride;
                    const unsigned long long span_d = (unsigned long long)((1 << fsft) - 1) * j.dfp + 2ull * j.srcH * (size_t)j.dstride;
                    if (span_s >= (1ull << 31) || span_d >= (1ull << 31))
                        fits = false;
                    const int nfull = fsft ? j.ngroups / 64 : 0;
                    w += ((double)nfull + (double)cdiv(j.ngroups - nfull * 64, lpf) / (1 << fsft)) * j.srcH;
                }
                if (fits && w < bestw - 1e-9) { bestw = w; best = fsft; }
            }
            if (bestw < 1e29) {
                U.fshift = best;
                for (int i = 0; i < U.njobs; i++)
                    ffhip_up2_plan_job(&U.job[i], 64 >> U.fshift, 60);
                {
                    const char *ed = FFHIP_KNOB("FFHIP_UP2_DEPTH"), *ev2 = FFHIP_KNOB("FFHIP_UP2_VAR"); /* measure build: rows in flight, 1 = non-temporal stores */
                    /* six rows in flight (round 6, with the straight-line rows: p010 1080p -> 4K 0.615 -> 0.63, planar unchanged) */
                    return ffhip_launch_up2(U, ed && ed[0] == '3' ? 3 : 6, ev2 ? atoi(ev2) : 0, stream);
                }
            }
        }
        const char *e2 = FFHIP_KNOB("FFHIP_SWS_DOWN2");
        if (c->dn2_ok && !(al & 3) && !neg && !(e2 && e2[0] == '0')) {
            /* exact 2:1 above 8 bits: the static-schedule kernel (FFHIP_SWS_DOWN2=0: the tiled k_sws_scale16) */
            FFHipDn2Args D;
            memset(&D, 0, sizeof(D));
            D.nframes = nframes;
            D.xcd = 1;
            auto dnjob = [&](int which, int plane, int dw_, int sh_, int dh_, int pair) {
                FFHipDn2Job &j = D.job[D.njobs++];
                j.src = static_cast<const uint8_t *>(src[plane]); j.dst = static_cast<uint8_t *>(dst[plane]);
                j.sstride = srcStride[plane]; j.dstride = dstStride[plane]; j.sfp = srcFramePitch[plane]; j.dfp = dstFramePitch[plane];
                j.pair = pair; j.swap = 0;
                j.srcH = sh_; j.dstH = dh_;
                j.ngroups = pair ? dw_ / 2 : dw_ / 4;
                j.hfv = c->dn2_h[which]; j.vfv = c->dn2_v[which];
                j.hb_sdepth = sd; j.hb_ddepth = dd; j.hb_smsb = sl == 1; j.hb_dmsb = dd > 8 && dl == 1;
                j.dither_off = plane == 2 ? 3 : 0;
                ffhip_down2_plan_job(&j, 32);
            };
            dnjob(0, 0, t.dstW, t.srcH, t.dstH, 0);
            if (sl) {
                dnjob(1, 1, c->d[1].n, c->chrSrcH, c->d[3].n, 1);
            } else {
                dnjob(1, 1, c->d[1].n, c->chrSrcH, c->d[3].n, 0);
                dnjob(1, 2, c->d[1].n, c->chrSrcH, c->d[3].n, 0);
            }
            return ffhip_launch_down2(D, stream);
        }
    }
    {
        const char *ew = FFHIP_KNOB("FFHIP_SWS_WALK16"); /* measured variant: 0 = the tiled k_sws_scale16 */
        uintptr_t al = 0;
        bool neg = false;
        for (int pl = 0; pl < (sl ? 2 : 3); pl++) {
            al |= (uintptr_t)wsrc[pl] | (uintptr_t)wss[pl] | wsf[pl];
            neg = neg || wss[pl] < 0;
        }
        for (int pl = 0; pl < (hrgb ? 0 : dl ? 2 : 3); pl++) {
            al |= (uintptr_t)dst[pl] | (uintptr_t)dstStride[pl] | dstFramePitch[pl];
            neg = neg || dstStride[pl] < 0;
        }
        if (hrgb)
            neg = neg || dstStride[0] < 0;
        if (hrgb && !(c->w16_ok && !(al & 3) && !neg)) {
            ffhip_set_error("ffhip_sws: above 8 bits into packed RGB needs 4-byte aligned planes and pitches, top-down");
            return FFHIP_EINVAL;
        }
        const char *ed3 = FFHIP_KNOB("FFHIP_SWS_DOWN32");
        if (c->d32_ok && !hrgb && !widened && !c->widen8 && !(al & 3) && !neg && !(ed3 && ed3[0] == '0')) {
            /* exact 3:2 down above 8 bits: the static-schedule kernel's 16-bit twin (FFHIP_SWS_DOWN32=0: the walker) */
            FFHipD32Args D;
            memset(&D, 0, sizeof(D));
            D.nframes = nframes;
            D.hb = 1; D.sdepth = sd; D.ddepth = dd; D.smsb = sl == 1; D.dmsb = dl == 1;
            D.ratio43 = c->d32_ok == 2;
            const int pin = D.ratio43 ? 4 : 3, pout = D.ratio43 ? 3 : 2;
            auto d3job = [&](int which, int plane, int dw_, int sh_, int pair) {
                FFHipD32Job &j = D.job[D.njobs++];
                j.src = static_cast<const uint8_t *>(src[plane]); j.dst = static_cast<uint8_t *>(dst[plane]);
                j.sstride = srcStride[plane]; j.dstride = dstStride[plane]; j.sfp = srcFramePitch[plane]; j.dfp = dstFramePitch[plane];
                j.pair = pair; j.swap = 0;
                j.srcH = sh_; j.dstH = sh_ / pin * pout;
                j.ngroups = pair ? dw_ / pout : dw_ / (2 * pout);
                j.hfv = c->d32_h[which]; j.vfv = c->d32_v[which];
                j.dither_off = plane == 2 ? 3 : 0;
            };
            d3job(0, 0, c->d[0].n, t.srcH, 0);
            if (sl) {
                d3job(1, 1, c->d[1].n, c->chrSrcH, 1);
            } else {
                d3job(1, 1, c->d[1].n, c->chrSrcH, 0);
                d3job(1, 2, c->d[1].n, c->chrSrcH, 0);
            }
            return ffhip_launch_down32(D, stream);
        }
        const char *eu3 = FFHIP_KNOB("FFHIP_SWS_UP32");
        if (c->u32_ok && !hrgb && (widened || !c->widen8) && !(al & 3) && !neg && !(eu3 && eu3[0] == '0')) {
            /* exact 3:2 up above 8 bits: the static-schedule kernel (FFHIP_SWS_UP32=0: the walker) */
            FFHipU32Args U;
            memset(&U, 0, sizeof(U));
            U.nframes = nframes;
            U.sdepth = sd; U.ddepth = dd; U.smsb = sl == 1 && !widened; U.dmsb = dl == 1;
            U.ratio43 = c->u32_ok == 2;
            const int pin = U.ratio43 ? 3 : 2, pout = U.ratio43 ? 4 : 3;
            auto u3job = [&](int which, int plane, int dw_, int sh_, int pair) {
                FFHipU32Job &j = U.job[U.njobs++];
                j.src = static_cast<const uint8_t *>(wsrc[plane]); j.dst = static_cast<uint8_t *>(dst[plane]);
                j.sstride = wss[plane]; j.dstride = dstStride[plane]; j.sfp = wsf[plane]; j.dfp = dstFramePitch[plane];
                j.pair = pair;
                j.srcH = sh_; j.dstH = sh_ / pin * pout;
                j.ngroups = pair ? dw_ / pout : dw_ / (2 * pout);
                j.hfv = c->u32_h[which]; j.vfv = c->u32_v[which];
            };
            u3job(0, 0, c->d[0].n, t.srcH, 0);
            if (sl) {
                u3job(1, 1, c->d[1].n, c->chrSrcH, 1);
            } else {
                u3job(1, 1, c->d[1].n, c->chrSrcH, 0);
                u3job(1, 2, c->d[1].n, c->chrSrcH, 0);
            }
            return ffhip_launch_up32(U, stream);
        }
        if (c->w16_ok && (widened || !c->widen8) && !(al & 3) && !neg && (hrgb || !(ew && ew[0] == '0'))) {
            /* a packed-RGB target (round 6): the walker writes the first stage — an int16 luma plane of unclipped sums, 8-bit chroma planes of
             * half the width with a line per target line, flat dither: what yuv2rgb_X_c_template computes before its tables (output.c:1789-1840) —
             * into the context's intermediate, and k_y16_rgb (sws_y16rgb.hip) turns it into pixels */
            void *xdst[4] = { dst[0], dst[1], dst[2], nullptr };
            int xds[4] = { dstStride[0], dstStride[1], dstStride[2], 0 };
            size_t xdf[4] = { dstFramePitch[0], dstFramePitch[1], dstFramePitch[2], 0 };
            std::unique_lock<std::mutex> rlk;
            size_t ypitch = 0, cpitch = 0, yfp = 0, cfp = 0;
            if (hrgb) {
                ypitch = ((size_t)2 * t.dstW + 255) & ~(size_t)255; cpitch = ((size_t)(t.dstW / 2) + 255) & ~(size_t)255;
                yfp = ypitch * (size_t)t.dstH; cfp = cpitch * (size_t)t.dstH;
                const size_t need = (yfp + 2 * cfp) * (size_t)nframes;
                rlk = std::unique_lock<std::mutex>(c->rgb2_mu);
                if (!c->rgb2_done)
                    HIP_TRY(hipEventCreateWithFlags(&c->rgb2_done, hipEventDisableTiming));
                else
                    HIP_TRY(hipStreamWaitEvent(stream, c->rgb2_done, 0));
                if (need > c->rgb2_tmp_sz) {
                    if (c->rgb2_tmp)
                        HIP_TRY(hipFree(c->rgb2_tmp));
                    c->rgb2_tmp = nullptr;
                    c->rgb2_tmp_sz = 0;
                    HIP_TRY(hipMalloc(&c->rgb2_tmp, need));
                    c->rgb2_tmp_sz = need;
                }
                uint8_t *ty = static_cast<uint8_t *>(c->rgb2_tmp);
                xdst[0] = ty; xdst[1] = ty + yfp * (size_t)nframes; xdst[2] = ty + (yfp + cfp) * (size_t)nframes;
                xds[0] = (int)ypitch; xds[1] = xds[2] = (int)cpitch;
                xdf[0] = yfp; xdf[1] = xdf[2] = cfp;
            }
            FFHipW16Args W;
            memset(&W, 0, sizeof(W));
            W.nframes = nframes;
            W.ht = c->w16_ht; W.vt = c->w16_vt;
            W.sdepth = sd; W.ddepth = dd; W.smsb = sl == 1 && !widened; W.dmsb = dl == 1;
            W.flat_dither = hrgb ? (c->hrgb_seed0 ? 2 : 1) : c->flat_dither;
            /* rows per strip: 64 when the batch fills the chip several times over; a strip re-filters VT - 1 source rows, but a
             * wave is one dependent chain of rows, and a launch of fewer waves than the chip holds (32 frames of 720p -> 1080p: 6,656
             * against 7,168 slots) runs at the speed of one chain: halve until there are 1.5 slots' worth (measured, 720p -> 1080p:
             * 64 rows 0.263, 32 rows 0.308, 16 rows 0.298 of HBM; the larger pictures are best at 64) */
            int strip = 64;
            {
                const char *es = FFHIP_KNOB("FFHIP_W16_STRIP"); /* measured variant */
                auto waves = [&](int st) {
                    const long long lum = (long long)cdiv(c->d[0].n, 256) * cdiv(c->d[2].n, st);
                    const long long chr = (sl || dl) ? (long long)cdiv(c->d[1].n, 128) * cdiv(c->d[3].n, st)
                                                     : 2LL * cdiv(c->d[1].n, 256) * cdiv(c->d[3].n, st);
                    return (lum + chr) * nframes;
                };
                if (es && atoi(es) > 0)
                    strip = atoi(es) > 64 ? 64 : atoi(es);
                else
                    while (strip > 16 && waves(strip) < 3 * 7168 / 2)
                        strip >>= 1;
            }
            auto job = [&](int which, int nch, int splane0, int dplane0) {
                FFHipW16Job &j = W.job[W.njobs++];
                j.nch = nch;
                j.sstep = which && sl ? 2 : 1; j.dstep = which && dl ? 2 : 1;
                j.dither_off = dplane0 == 2 ? 3 : 0;
                for (int k = 0; k < nch; k++) {
                    /* an interleaved side: both channels live in plane 1, the second one sample on; a planar side: planes 1 and 2 */
                    const int sp = which && sl ? 1 : splane0 + k, dp = which && dl ? 1 : dplane0 + k;
                    j.src[k] = static_cast<const uint8_t *>(wsrc[sp]) + (which && sl ? 2 * k : 0);
                    j.dst[k] = static_cast<uint8_t *>(xdst[dp]) + (which && dl ? (dd == 8 ? 1 : 2) * k : 0);
                    j.sstride[k] = wss[sp]; j.dstride[k] = xds[dp];
                    j.sfp[k] = wsf[sp]; j.dfp[k] = xdf[dp];
                }
                j.srcH = which ? c->chrSrcH : t.srcH;
                j.y16 = hrgb && !which;
                j.dstW = c->d[which].n; j.dstH = c->d[2 + which].n;
                j.hf = c->w16_f[which]; j.hp = c->w16_p[which]; j.vf = c->w16_f[2 + which]; j.vp = c->w16_p[2 + which];
                j.srcW = which ? c->chrSrcW : t.srcW;
                {
                    const char *eg = FFHIP_KNOB("FFHIP_W16_STAGE"); /* measure build: 0 keeps the per-lane global loads */
                    const int sp = c->w16_span[which][nch == 1 ? 0 : j.sstep == 2 ? 2 : 1];
                    /* staged when a wave's windows cover at most 512 bytes of a source row, i.e. when the picture grows: adjacent lanes'
                     * windows then overlap several times over and the per-lane loads fetched every sample four to eight times (measured,
                     * profiles/r06_walk16_lds.txt: p010 720p -> 1080p 0.270 -> 0.353, yuv420p10 1080p -> 1440p 0.329 -> 0.401 of HBM); wider
                     * spans (down-scaling: 4K -> 1440p flat, 1080p -> 720p -4 %) keep the direct loads; FFHIP_W16_STAGE=1 stages up to 1 KiB */
                    j.stage = sp > 0 && sp <= (eg && eg[0] == '1' ? 1024 : 512) && !(eg && eg[0] == '0');
                }
                ffhip_w16_plan_job(&j, strip);
            };
            if (!c->rgb_in_luma_done)
                job(0, 1, 0, 0);
            if (sl || dl) {
                job(1, 2, 1, 1);
            } else {
                job(1, 1, 1, 1);
                job(1, 1, 2, 2);
            }
            const int rw = ffhip_launch_walk16(W, stream);
            if (rw < 0 || !hrgb)
                return rw;
            FFHipY16RgbArgs Y;
            memset(&Y, 0, sizeof(Y));
            Y.y = static_cast<const uint8_t *>(xdst[0]); Y.u = static_cast<const uint8_t *>(xdst[1]); Y.v = static_cast<const uint8_t *>(xdst[2]);
            Y.dst = static_cast<uint8_t *>(dst[0]);
            Y.ystride = (ptrdiff_t)ypitch; Y.cstride = (ptrdiff_t)cpitch; Y.dstride = dstStride[0];
            Y.yfp = yfp; Y.cfp = cfp; Y.dfp = dstFramePitch[0];
            Y.w = t.dstW; Y.h = t.dstH; Y.nframes = nframes; Y.lay = rgb_layout(t.dstFormat); Y.k = c->k;
            const int ry = ffhip_launch_y16_rgb(Y, stream);
            if (ry >= 0)
                HIP_TRY(hipEventRecord(c->rgb2_done, stream));
            return ry;
        }
    }
    if (c->rgb_in_luma_done) {
        ffhip_set_error("ffhip_sws: internal: the converter pass wrote the luma plane but the walker does not run");
        return FFHIP_EINVAL;
    }
    FFHipScale16Args a;
    memset(&a, 0, sizeof(a));
    a.nplanes = 3;
    for (int pl = 0; pl < 3; pl++) {
        FFHipScale16Plane &p = a.pl[pl];
        const int sp = pl == 0 ? 0 : sl ? 1 : pl, dp = pl == 0 ? 0 : dl ? 1 : pl;   /* plane index on either side */
        p.src = static_cast<const uint8_t *>(src[sp]);
        p.dst = static_cast<uint8_t *>(dst[dp]);
        p.src_stride = srcStride[sp]; p.dst_stride = dstStride[dp];
        p.src_fp = srcFramePitch[sp]; p.dst_fp = dstFramePitch[dp];
        p.sdepth = sd; p.smsb = sl == 1; p.sstep = pl && sl ? 2 : 1; p.schan = pl && sl ? pl - 1 : 0;
        p.ddepth = dd; p.dmsb = dl == 1; p.dstep = pl && dl ? 2 : 1; p.dchan = pl && dl ? pl - 1 : 0;
        p.h = c->d[pl ? 1 : 0]; p.v = c->d[pl ? 3 : 2];
        p.dstW = p.h.n; p.dstH = p.v.n;
        p.dither = dd == 8 && sd > 8 && !c->flat_dither; /* swscale.c:291: should_dither = isNBPS(src) || is16BPS(src) — of the caller's format */
        p.dither_off = pl == 2 ? 3 : 0;     /* vscale.c: the V plane reads the dither row three entries on; yuv2nv12cX_c: (i + 3) & 7 */
        if (t.src_range != t.dst_range) {
            p.rc_coeff = pl ? t.chrConvertRange_coeff : t.lumConvertRange_coeff;
            p.rc_offset = pl ? t.chrConvertRange_offset : t.lumConvertRange_offset;
            p.rc_clip = !t.src_range;
        }
    }
    a.staged = c->hbd == 2;
    a.sw_pitch = c->hbd_sw;
    a.max_rows = c->hbd_rows;
    return ffhip_launch_scale16(a, nframes, stream);
}

static int scale_batch_dev(FFHipSwsContext *c, int nframes, const void *const src[4], const int srcStride[4], const size_t srcFramePitch[4],
                           void *const dst[4], const int dstStride[4], const size_t dstFramePitch[4], void *stream_);

struct PlaneDesc { int wbytes, rows; };

static int plane_list(int fmt, int w, int h, PlaneDesc out[3])
{
    if (fmt_yuv(fmt)) {
        const int cw = -((-w) >> fmt_hsub(fmt)), chh = -((-h) >> fmt_vsub(fmt));
        int depth = 8, layout = 0;
        const int bs = ffhip_pixfmt_hbd(fmt, &depth, &layout, nullptr, nullptr) ? 2 : 1; /* bytes per sample */
        if (fmt_nv(fmt) || layout == 1) { out[0] = { bs * w, h }; out[1] = { bs * 2 * cw, chh }; return 2; }
        out[0] = { bs * w, h }; out[1] = { bs * cw, chh }; out[2] = { bs * cw, chh };
        return 3;
    }
    if (fmt_gbrp(fmt)) {
        out[0] = out[1] = out[2] = { w, h };
        return 3;
    }
    out[0] = { (rgb_layout(fmt) < 2 ? 3 : 4) * w, h };
    return 1;
}


extern "C" int ffhip_sws_scale_batch_dev(FFHipSwsContext *c, int nframes, const void *const src[4],
                                         const int srcStride[4], const size_t srcFramePitch[4], void *const dst[4],
                                         const int dstStride[4], const size_t dstFramePitch[4], void *stream_)
{
    if (c && c->rgb_in.bpp) {
        /* a packed RGB source: the converter pass into the context's own planes (kept for the next call; one batch in flight per context:
         * the host face's lock), then the 14-bit planar context on those */
        if (!src || !src[0] || !srcStride || !srcFramePitch || nframes <= 0)
            return FFHIP_EINVAL;
        FFHipDeviceGuard dg(c->device);
        std::lock_guard<std::mutex> lk(c->mu);
        if (c->rgb_in.direct_c && dst && dstStride && dstFramePitch && dst[0] && dst[1] && dst[2] && !FFHIP_KNOB("FFHIP_SWS_RGB_DIRECT_OFF")) {
            /* every bank the identity: the converter pass writes the three 8-bit planes of the target (FFHipRgbInArgs.c8) */
            FFHipRgbInArgs a;
            memset(&a, 0, sizeof(a));
            a.src = static_cast<const uint8_t *>(src[0]); a.src_stride = srcStride[0]; a.src_fp = srcFramePitch[0];
            a.y8 = static_cast<uint8_t *>(dst[0]); a.y8_stride = dstStride[0]; a.y8_fp = dstFramePitch[0];
            a.c8 = 1;
            for (int i = 1; i < 3; i++) {
                a.dst[i] = static_cast<uint8_t *>(dst[i]); a.dst_stride[i] = dstStride[i]; a.dst_fp[i] = dstFramePitch[i];
            }
            a.dst[0] = a.y8; a.dst_stride[0] = dstStride[0]; a.dst_fp[0] = dstFramePitch[0];
            a.w = c->t.srcW; a.h = c->t.srcH;
            a.ro = c->rgb_in.ofs[0]; a.go = c->rgb_in.ofs[1]; a.bo = c->rgb_in.ofs[2];
            const int32_t *T = c->rgb_in.table;
            a.ry = T[0]; a.gy = T[1]; a.by = T[2]; a.ru = T[3]; a.gu = T[4]; a.bu = T[5]; a.rv = T[6]; a.gv = T[7]; a.bv = T[8];
            return ffhip_launch_sws_rgb_in(a, c->rgb_in.bpp, c->rgb_in.half, nframes, (hipStream_t)stream_);
        }
        {
            /* one kernel for RGB -> yuv420p / NV12 at the source's size (FFHIP_SWS_RGB420=0 in the measure build: the two-stage form) */
            const char *e4 = FFHIP_KNOB("FFHIP_SWS_RGB420");
            const bool nv = fmt_nv(c->t.dstFormat);
            bool ok = c->rgb_in.fused420 && dst && dstStride && dstFramePitch && !(e4 && e4[0] == '0') && srcStride[0] > 0 &&
                      !(((uintptr_t)src[0] | (uintptr_t)srcStride[0] | srcFramePitch[0]) & 3);
            for (int pl = 0; ok && pl < (nv ? 2 : 3); pl++)
                ok = dst[pl] && dstStride[pl] > 0 && !(((uintptr_t)dst[pl] | (uintptr_t)dstStride[pl] | dstFramePitch[pl]) & (pl == 0 || nv ? 3 : 1));
            ok = ok && (nv || (dstStride[1] == dstStride[2] && dstFramePitch[1] == dstFramePitch[2]));
            if (ok) {
                FFHipRgb420Args A;
                memset(&A, 0, sizeof(A));
                A.in.src = static_cast<const uint8_t *>(src[0]); A.in.src_stride = srcStride[0]; A.in.src_fp = srcFramePitch[0];
                A.in.y8 = static_cast<uint8_t *>(dst[0]); A.in.y8_stride = dstStride[0]; A.in.y8_fp = dstFramePitch[0];
                A.in.w = c->t.srcW; A.in.h = c->t.srcH;
                A.in.ro = c->rgb_in.ofs[0]; A.in.go = c->rgb_in.ofs[1]; A.in.bo = c->rgb_in.ofs[2];
                const int32_t *T = c->rgb_in.table;
                A.in.ry = T[0]; A.in.gy = T[1]; A.in.by = T[2]; A.in.ru = T[3]; A.in.gu = T[4]; A.in.bu = T[5]; A.in.rv = T[6]; A.in.gv = T[7]; A.in.bv = T[8];
                A.cdst[0] = static_cast<uint8_t *>(dst[1]); A.cdst[1] = nv ? nullptr : static_cast<uint8_t *>(dst[2]);
                A.cstride = dstStride[1]; A.cfp = dstFramePitch[1];
                A.chrH = c->t.srcH / 2;
                A.vfv = c->rgb_in.vfv;
                A.nframes = nframes;
                return ffhip_launch_sws_rgb420(A, c->rgb_in.bpp, nv, (hipStream_t)stream_);
            }
        }
        const int cw = c->rgb_in.half ? c->t.srcW / 2 : c->t.srcW;
        const int pitch[3] = { ((c->t.srcW * 2) + 255) & ~255, ((cw * 2) + 255) & ~255, ((cw * 2) + 255) & ~255 };
        const size_t fp[3] = { (size_t)pitch[0] * c->t.srcH, (size_t)pitch[1] * c->t.srcH, (size_t)pitch[2] * c->t.srcH };
        const size_t need = (fp[0] + fp[1] + fp[2]) * (size_t)nframes + 256;
        if (need > c->rgb_in.planes_sz) {
            if (c->rgb_in.planes) {
                (void)hipDeviceSynchronize(); /* an earlier batch may still read them */
                (void)hipFree(c->rgb_in.planes);
            }
            c->rgb_in.planes = nullptr;
            c->rgb_in.planes_sz = 0;
            if (hipMalloc(&c->rgb_in.planes, need) != hipSuccess) {
                ffhip_set_error("ffhip_sws_scale_batch_dev: hipMalloc(%zu) for an RGB source's converter lines failed", need);
                return FFHIP_ENOMEM;
            }
            c->rgb_in.planes_sz = need;
        }
        uint8_t *b = static_cast<uint8_t *>(c->rgb_in.planes);
        uint8_t *const p[3] = { b, b + fp[0] * nframes, b + (fp[0] + fp[1]) * nframes };
        /* (the walker — the kernel that can leave the luma out — wants 4-byte aligned planes and pitches, top-down: scale16()) */
        bool yd = c->rgb_in.y_direct && dst && dstStride && dstFramePitch;
        for (int pl = 0; yd && pl < (fmt_nv(c->t.dstFormat) ? 2 : 3); pl++)
            yd = dst[pl] && dstStride[pl] > 0 && !(((uintptr_t)dst[pl] | (uintptr_t)dstStride[pl] | dstFramePitch[pl]) & 3);
        const int r = rgb_in_launch(c, nframes, static_cast<const uint8_t *>(src[0]), srcStride[0], srcFramePitch[0], c->t.srcH, p, pitch, fp, (hipStream_t)stream_,
                                    yd ? static_cast<uint8_t *>(dst[0]) : nullptr, yd ? dstStride[0] : 0, yd ? dstFramePitch[0] : 0);
        c->rgb_in_luma_done = yd;
        if (r < 0)
            return r;
        const void *s2[4] = { p[0], p[1], p[2], nullptr };
        const int ss2[4] = { pitch[0], pitch[1], pitch[2], 0 };
        const size_t sf2[4] = { fp[0], fp[1], fp[2], 0 };
        const int r2 = scale_batch_dev(c, nframes, s2, ss2, sf2, dst, dstStride, dstFramePitch, stream_);
        c->rgb_in_luma_done = false;
        if (r2 < 0 || !c->t.dst_alpha_fill)
            return r2;
        if (!dst[3]) {
            ffhip_set_error("ffhip_sws_scale_batch_dev: the format has an alpha plane: plane 3 is NULL");
            return FFHIP_EINVAL;
        }
        for (int f = 0; f < nframes; f++) /* opaque: the source carries no alpha into a planar target here (swscale.c:536-553) */
            if (hipMemset2DAsync(static_cast<uint8_t *>(dst[3]) + (size_t)f * dstFramePitch[3], (size_t)dstStride[3], 255, (size_t)c->t.dstW, (size_t)c->t.dstH,
                                 (hipStream_t)stream_) != hipSuccess)
                return FFHIP_EINVAL;
        return r2;
    }
    if (c && (c->unscaled_yuv2rgb || (c->t.dst_alpha_fill == 2 && fmt_rgb(c->t.dstFormat)))) /* one launch; a source alpha plane (src[3]) is read by it */
        return scale_batch_dev(c, nframes, src, srcStride, srcFramePitch, dst, dstStride, dstFramePitch, stream_);
    if (c && c->t.dst_alpha_fill && (!dst || !dst[3] || (c->t.dst_alpha_fill == 2 && (!src || !src[3])))) {
        ffhip_set_error("ffhip_sws_scale_batch_dev: the format has an alpha plane: plane 3 is NULL");
        return FFHIP_EINVAL;
    }
    /* the alpha pass flips the context's planner state (luma_pass): a context with a scaled alpha plane serves one call at a time,
     * both passes under the lock the host face holds too */
    std::unique_lock<std::mutex> lk;
    if (c && c->t.dst_alpha_fill == 2)
        lk = std::unique_lock<std::mutex>(c->mu);
    const int r = scale_batch_dev(c, nframes, src, srcStride, srcFramePitch, dst, dstStride, dstFramePitch, stream_);
    if (r < 0 || !c->t.dst_alpha_fill)
        return r;
    FFHipDeviceGuard dg(c->device);
    if (c->t.dst_alpha_fill == 2) {
        /* alpha on both sides: the alpha plane is the luma of a second pass (lum_h_scale / lum_planar_vscale run the luma banks on plane
         * 3, hscale.c:63-79, vscale.c:57-70) in which the planners enumerate the luma job only; the chroma slots keep the picture's own
         * planes (were a planner to scale them again it would write the bytes they already hold) */
        if (fmt_rgb(c->t.dstFormat) || fmt_nv(c->t.dstFormat)) {
            ffhip_set_error("ffhip_sws_scale_batch_dev: a scaled alpha plane belongs to a planar target");
            return FFHIP_EINVAL;
        }
        const void *s2[4] = { src[3], src[1], src[2], nullptr };
        const int ss2[4] = { srcStride[3], srcStride[1], srcStride[2], 0 };
        const size_t sf2[4] = { srcFramePitch[3], srcFramePitch[1], srcFramePitch[2], 0 };
        void *d2[4] = { dst[3], dst[1], dst[2], nullptr };
        const int ds2[4] = { dstStride[3], dstStride[1], dstStride[2], 0 };
        const size_t df2[4] = { dstFramePitch[3], dstFramePitch[1], dstFramePitch[2], 0 };
        c->luma_pass = true; /* (a context serves one call at a time: the planners leave the chroma jobs out) */
        const int r2 = scale_batch_dev(c, nframes, s2, ss2, sf2, d2, ds2, df2, stream_);
        c->luma_pass = false;
        return r2;
    }
    /* a target with an alpha plane the source does not drive: opaque (ff_swscale's fillPlane, libswscale/swscale.c:536-553) */
    for (int f = 0; f < nframes; f++)
        HIP_TRY(hipMemset2DAsync((uint8_t *)dst[3] + (size_t)f * dstFramePitch[3], (size_t)dstStride[3], 255, (size_t)c->t.dstW, (size_t)c->t.dstH,
                                 (hipStream_t)stream_));
    return r;
}

/* The numbering of a large launch of the table converter: 0 plain, 1 an eighth per XCD.  While undecided, launches 0..7 of the context run
 * plain, eighth, eighth, plain, plain, eighth, eighth, plain (a drift of the clocks cancels) and *slot names the event pair to record
 * around this one; once all eight have finished the faster numbering is kept — the eighth only when it wins by 2 %.  No decision is forced: a query that finds an
 * event pending leaves the default (plain) in place for this call.  Not while the stream is being captured into a graph. */
static int tune_pick(FFHipSwsContext *c, hipStream_t stream, int *slot)
{
    *slot = -1;
    if (c->tune_choice >= 0)
        return c->tune_choice;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) {
        (void)hipGetLastError();
        return 0;
    }
    if (c->tune_n < 8) {
        if (!c->tune_ev[0])
            for (int i = 0; i < 16; i++)
                if (hipEventCreate(&c->tune_ev[i]) != hipSuccess) {
                    (void)hipGetLastError();
                    c->tune_choice = 0;
                    return 0;
                }
        *slot = c->tune_n;
        return (c->tune_n + 1) >> 1 & 1;       /* plain, eighth, eighth, plain, plain, eighth, eighth, plain */
    }
    float t[2] = { 0, 0 };
    for (int i = 0; i < 8; i++) {
        float e;
        if (hipEventQuery(c->tune_ev[2 * i + 1]) != hipSuccess || hipEventElapsedTime(&e, c->tune_ev[2 * i], c->tune_ev[2 * i + 1]) != hipSuccess) {
            (void)hipGetLastError();
            return 0;
        }
        t[(i + 1) >> 1 & 1] += e;
    }
    c->tune_choice = t[1] < 0.98f * t[0];
    for (int i = 0; i < 16; i++) {
        (void)hipEventDestroy(c->tune_ev[i]);
        c->tune_ev[i] = nullptr;
    }
    return c->tune_choice;
}

/* the table converter on `rows` lines (a whole frame, or a 2-line aligned slice) of device planes: one launch */
static int unscaled_launch(FFHipSwsContext *c, int nframes, int rows, const void *const src[4], const int srcStride[4], const size_t srcFramePitch[4],
                           void *const dst[4], const int dstStride[4], const size_t dstFramePitch[4], hipStream_t stream)
{
    const FFHipSwsTables &t = c->t;
    FFHipYuv2RgbArgs a;
    a.y = (const uint8_t *)src[0]; a.u = (const uint8_t *)src[1]; a.v = (const uint8_t *)src[2]; a.dst = (uint8_t *)dst[0];
    a.y_stride = srcStride[0]; a.u_stride = srcStride[1]; a.v_stride = srcStride[2]; a.dst_stride = dstStride[0];
    a.y_fp = srcFramePitch[0]; a.u_fp = srcFramePitch[1]; a.v_fp = srcFramePitch[2]; a.dst_fp = dstFramePitch[0];
    a.wvalid = t.dstW & ~1; a.h = rows; a.dst_y0 = 0; a.nframes = nframes; a.flat = 0; a.xcd = 0; a.k = c->k;
    a.c422 = t.srcFormat == FFHIP_PIX_FMT_YUV422P;
    if (!a.y || !a.u || !a.v || !a.dst)
        return FFHIP_EINVAL;
    if (t.dst_alpha_fill == 2) {
        if (!src[3]) {
            ffhip_set_error("ffhip_sws: the source's alpha plane (plane 3) is NULL");
            return FFHIP_EINVAL;
        }
        a.alpha = (const uint8_t *)src[3]; a.alpha_stride = srcStride[3]; a.alpha_fp = srcFramePitch[3];
    }
    if (fmt_gbrp(t.dstFormat)) {
        if (!dst[1] || !dst[2])
            return FFHIP_EINVAL;
        a.dst1 = (uint8_t *)dst[1]; a.dst1_stride = dstStride[1]; a.dst1_fp = dstFramePitch[1];
        a.dst2 = (uint8_t *)dst[2]; a.dst2_stride = dstStride[2]; a.dst2_fp = dstFramePitch[2];
        return ffhip_launch_yuv420p_rgb24(a, 6, stream);
    }
    /* 24-bit targets from planar 4:2:0, launches of 64 MiB and more: the numbering this box prefers */
    int slot = -1;
    if (rgb_layout(t.dstFormat) < 2 && !a.c422 && !a.alpha && (long long)nframes * rows * t.dstW * 3 >= (64LL << 20))
        a.xcd = tune_pick(c, stream, &slot);
    if (slot >= 0 && hipEventRecord(c->tune_ev[2 * slot], stream) != hipSuccess) {
        (void)hipGetLastError();
        slot = -1;
    }
    const int r = ffhip_launch_yuv420p_rgb24(a, rgb_layout(t.dstFormat), stream);
    if (slot >= 0) {
        if (r >= 0 && hipEventRecord(c->tune_ev[2 * slot + 1], stream) == hipSuccess)
            c->tune_n++;
        else
            (void)hipGetLastError();
    }
    return r;
}

static int scale_batch_dev(FFHipSwsContext *c, int nframes, const void *const src[4], const int srcStride[4], const size_t srcFramePitch[4],
                           void *const dst[4], const int dstStride[4], const size_t dstFramePitch[4], void *stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (!c || nframes < 0 || !src || !dst)
        return FFHIP_EINVAL;
    FFHipDeviceGuard dg(c->device);
    const FFHipSwsTables &t = c->t;
    const uint8_t *s0 = (const uint8_t *)src[0], *s1 = (const uint8_t *)src[1], *s2 = (const uint8_t *)src[2];
    if (c->hbd)
        return scale16(c, nframes, src, srcStride, srcFramePitch, dst, dstStride, dstFramePitch, stream);

    if (c->unscaled_yuv2rgb)
        return unscaled_launch(c, nframes, t.srcH, src, srcStride, srcFramePitch, dst, dstStride, dstFramePitch, stream);

    /* chroma source description */
    const uint8_t *cu, *cv;
    ptrdiff_t cus, cvs;
    size_t cuf, cvf;
    int cstep;
    if (fmt_nv(t.srcFormat)) {
        const int sw = t.srcFormat == FFHIP_PIX_FMT_NV21;
        cu = s1 + sw; cv = s1 + !sw;
        cus = cvs = srcStride[1]; cuf = cvf = srcFramePitch[1]; cstep = 2;
    } else {
        cu = s1; cv = s2; cus = srcStride[1]; cvs = srcStride[2]; cuf = srcFramePitch[1]; cvf = srcFramePitch[2];
        cstep = 1;
    }

    if (fmt_rgb(t.dstFormat)) {
        FFHipScaleRgbArgs a = c->rgb;
        a.src[0] = s0; a.src[1] = cu; a.src[2] = cv;
        a.src_stride[0] = srcStride[0]; a.src_stride[1] = cus; a.src_stride[2] = cvs;
        a.src_fp[0] = srcFramePitch[0]; a.src_fp[1] = cuf; a.src_fp[2] = cvf;
        a.chr_step = cstep;
        a.dst = (uint8_t *)dst[0]; a.dst_stride = dstStride[0]; a.dst_fp = dstFramePitch[0];
        a.nframes = nframes;
        if (a.has_alpha) {
            if (!src[3]) {
                ffhip_set_error("ffhip_sws_scale_batch_dev: the source's alpha plane (plane 3) is NULL");
                return FFHIP_EINVAL;
            }
            a.alpha = (const uint8_t *)src[3]; a.alpha_stride = srcStride[3]; a.alpha_fp = srcFramePitch[3];
        }
        const char *ev = FFHIP_KNOB("FFHIP_SWS_FAST");
        uintptr_t al = (uintptr_t)s0 | (size_t)srcStride[0] | srcFramePitch[0] | (uintptr_t)a.dst | (size_t)a.dst_stride | a.dst_fp |
                       (size_t)cus | (size_t)cvs | cuf | cvf | (uintptr_t)(cstep == 2 ? s1 : cu) | (uintptr_t)(cstep == 2 ? s1 : cv);
        /* (the round-5 kernels walk their rows with running pointers and were written for top-down pictures: a negative stride keeps the
         * older kernels) */
        const bool topdown = srcStride[0] > 0 && cus > 0 && cvs > 0 && a.dst_stride > 0;
        if (c->f444_ok && topdown && !(ev && ev[0] == '0') && !(al & 3)) {
            FFHipFull444Args F;
            memset(&F, 0, sizeof(F));
            F.src[0] = s0; F.src[1] = cu; F.src[2] = cv;
            F.sstride[0] = srcStride[0]; F.sstride[1] = cus; F.sstride[2] = cvs;
            F.sfp[0] = srcFramePitch[0]; F.sfp[1] = cuf; F.sfp[2] = cvf;
            F.dst = a.dst; F.dstride = a.dst_stride; F.dfp = a.dst_fp;
            F.w = a.dstW; F.h = a.dstH; F.nframes = nframes; F.lay = a.bgr;
            for (int i = 0; i < 6; i++)
                F.fk[i] = a.fk[i];
            return ffhip_launch_full444(F, stream);
        }
        const char *eq = FFHIP_KNOB("FFHIP_SWS_EQRGB"); /* measure build: 0 keeps the column walker */
        if (c->eqr_ok && topdown && !(ev && ev[0] == '0') && !(eq && eq[0] == '0') && !(al & 3)) {
            /* the source's size: chroma lines interpolated by the exact-2x vertical bank, nothing else scaled (sws_eqrgb.hip) */
            FFHipEqRgbArgs E;
            memset(&E, 0, sizeof(E));
            E.src[0] = s0; E.src[1] = cstep == 2 ? s1 : cu; E.src[2] = cstep == 2 ? s1 : cv;
            E.sil = cstep == 2; E.swap = t.srcFormat == FFHIP_PIX_FMT_NV21;
            E.sstride[0] = srcStride[0]; E.sstride[1] = cus; E.sstride[2] = cvs;
            E.sfp[0] = srcFramePitch[0]; E.sfp[1] = cuf; E.sfp[2] = cvf;
            E.dst = a.dst; E.dstride = a.dst_stride; E.dfp = a.dst_fp;
            E.chrH = a.chrSrcH; E.ngroups = a.dstW / 8; E.nframes = nframes;
            E.vt = static_cast<const uint32_t *>(c->eqr_dev);
            E.vround = c->cw_vround; E.lay = a.bgr; E.k = c->k;
            const char *es = FFHIP_KNOB("FFHIP_EQRGB_STEPS"), *ef = FFHIP_KNOB("FFHIP_EQRGB_FPP");
            /* (frames that share a wave address their planes by 32-bit lane offsets from the pack's first frame: up to three frame pitches) */
            const bool far = (E.sfp[0] | E.sfp[1] | E.sfp[2] | E.dfp) >= ((size_t)1 << 30);
            ffhip_eqrgb_plan(&E, es && atoi(es) > 0 ? atoi(es) : 30, ef ? atoi(ef) : E.lay < 2 && !far ? 0 : 1);
            return ffhip_launch_eqrgb(E, stream);
        }
        const char *eu2 = FFHIP_KNOB("FFHIP_SWS_UP2RGB"); /* measure build: 0 keeps the column walker, v<n> a measured variant */
        if (c->u2r_ok && topdown && !(ev && ev[0] == '0') && !(eu2 && eu2[0] == '0') && !(al & 3)) {
            /* exact 2x of 4:2:0: static schedule, regular windows, the RGB writer fused (sws_up2rgb.hip) */
            FFHipUp2RgbArgs U;
            memset(&U, 0, sizeof(U));
            U.src[0] = s0; U.src[1] = cstep == 2 ? s1 : cu; U.src[2] = cstep == 2 ? s1 : cv;
            U.sil = cstep == 2; U.swap = t.srcFormat == FFHIP_PIX_FMT_NV21;
            U.sstride[0] = srcStride[0]; U.sstride[1] = cus; U.sstride[2] = cvs;
            U.sfp[0] = srcFramePitch[0]; U.sfp[1] = cuf; U.sfp[2] = cvf;
            U.dst = a.dst; U.dstride = a.dst_stride; U.dfp = a.dst_fp;
            U.srcW = a.srcW; U.srcH = a.srcH; U.ngroups = a.srcW / 4; U.nframes = nframes;
            U.hco = c->u2r_hco; U.vt = c->u2r_vt;
            U.vround = c->cw_vround; U.lay = a.bgr; U.k = c->k;
            const char *es = FFHIP_KNOB("FFHIP_UP2RGB_STEPS");
            const char *ef = FFHIP_KNOB("FFHIP_UP2RGB_FPP"); /* measure build: frames per pack */
            /* measured (profiles/r05_up2rgb_f.txt): 24-bit pixels — the kernel is bound by its instruction stream: frames share the ragged
             * last lane block (no idle lanes), strips of 24 luma rows; 32-bit pixels — closer to the write path: one frame per pack
             * (two frames' rows written side by side cost 8 %, more when the frame pitch aliases), strips of 36 */
            const bool far = (U.sfp[0] | U.sfp[1] | U.sfp[2] | U.dfp) >= ((size_t)1 << 30); /* (32-bit lane offsets span up to three frame pitches) */
            ffhip_up2rgb_plan(&U, es && atoi(es) > 0 ? atoi(es) : U.lay < 2 ? 24 : 36, ef ? atoi(ef) : U.lay < 2 && !far ? 0 : 1);
            return ffhip_launch_up2rgb(U, eu2 && eu2[0] == 'v' ? atoi(eu2 + 1) : 0, stream);
        }
        if (c->cw_rgb && !(ev && ev[0] == '0') && !(al & 3)) {
            FFHipCwRgbArgs R;
            memset(&R, 0, sizeof(R));
            R.src[0] = s0; R.src[1] = cstep == 2 ? s1 : cu; R.src[2] = cstep == 2 ? s1 : cv;
            R.sstride[0] = srcStride[0]; R.sstride[1] = cus; R.sstride[2] = cvs;
            R.sfp[0] = srcFramePitch[0]; R.sfp[1] = cuf; R.sfp[2] = cvf;
            R.dst = a.dst; R.dstride = a.dst_stride; R.dfp = a.dst_fp;
            R.sil = cstep == 2; R.src_swap = t.srcFormat == FFHIP_PIX_FMT_NV21; R.bgr = a.bgr;
            R.srcW = a.srcW; R.srcH = a.srcH; R.chrSrcW = a.chrSrcW; R.chrSrcH = a.chrSrcH; R.dstW = a.dstW; R.dstH = a.dstH;
            R.hlf = c->dn[0].filter; R.hlp = c->dn[0].pos; R.hcf = c->dn[1].filter; R.hcp = c->dn[1].pos;
            R.vlf = c->dn[2].filter; R.vlp = c->dn[2].pos; R.vcf = c->dn[3].filter; R.vcp = c->dn[3].pos;
            R.nframes = nframes; R.k = c->k; R.vround = c->cw_vround;
            { const char *en = FFHIP_KNOB("FFHIP_CWRGB_NTS"); R.nts = !(en && en[0] == '0'); }
            return ffhip_launch_colwalk_rgb(R, stream);
        }
        const char *e2 = FFHIP_KNOB("FFHIP_SWS_RGB2"); /* measure build: 0 keeps the LDS-tiled kernel */
        if (c->lw_ok && topdown && !(ev && ev[0] == '0') && !(e2 && e2[0] == '0') && !(al & 3)) {
            const char *ed = FFHIP_KNOB("FFHIP_SWS_DOWN2");
            const int chrW = a.dstW / 2;
            /* exact 2:1 from a 4:2:0 source: fused, no intermediate — before anything of the two-stage form (its lock, its buffer) is touched */
            if (c->dn2_luma == 2 && (cstep == 2 || (cus == cvs && cuf == cvf)) && srcStride[0] > 0 && cus > 0 &&
                !(ed && (ed[0] == '0' || ed[0] == 'l' || ed[0] == 't')) && !(a.dstW & 3) && a.dstW >= 12) {
                /* ... and the whole conversion in ONE kernel, no intermediate (k_sws_down2_rgb; planar chroma: planes laid out alike;
                 * FFHIP_SWS_DOWN2=t: the two-stage form below) */
                FFHipDn2RgbArgs F;
                memset(&F, 0, sizeof(F));
                F.ysrc = s0; F.ysstride = srcStride[0]; F.ysfp = srcFramePitch[0];
                F.csrc = cstep == 2 ? s1 : cu; F.csrc2 = cstep == 2 ? nullptr : cv; F.csstride = cus; F.csfp = cuf; F.swap = cstep == 2 && cv < cu;
                F.dst = a.dst; F.dstride = a.dst_stride; F.dfp = a.dst_fp;
                F.srcH = a.srcH; F.chrH = a.chrSrcH; F.dstH = a.dstH; F.ngroups = a.dstW / 4;
                F.hfv_l = c->dn2_h[0]; F.hfv_c = c->dn2_h[1]; F.vfv = c->dn2_v[0];
                F.nframes = nframes; F.xcd = 1; F.lay = a.bgr; F.k = c->k;
                return ffhip_launch_down2_rgb(F, 32, stream);
            }
            /* two stages (see the context's creation): planes of the target's geometry, pitches and frames 256-byte aligned */
            const size_t ypitch = ((size_t)2 * a.dstW + 255) & ~(size_t)255, cpitch = ((size_t)(a.dstW / 2) + 255) & ~(size_t)255;
            const size_t yfp = ypitch * (size_t)a.dstH, cfp = cpitch * (size_t)a.dstH, need = (yfp + 2 * cfp) * (size_t)nframes;
            /* the intermediate is the context's: calls are serialised here, and a call on another stream waits for the last reader */
            std::lock_guard<std::mutex> lk(c->rgb2_mu);
            if (!c->rgb2_done)
                HIP_TRY(hipEventCreateWithFlags(&c->rgb2_done, hipEventDisableTiming));
            else
                HIP_TRY(hipStreamWaitEvent(stream, c->rgb2_done, 0));
            if (need > c->rgb2_tmp_sz) {
                if (c->rgb2_tmp)
                    HIP_TRY(hipFree(c->rgb2_tmp)); /* (waits for the launches that still use it) */
                c->rgb2_tmp = nullptr;
                c->rgb2_tmp_sz = 0;
                HIP_TRY(hipMalloc(&c->rgb2_tmp, need));
                c->rgb2_tmp_sz = need;
            }
            uint8_t *ty = static_cast<uint8_t *>(c->rgb2_tmp), *tu = ty + yfp * (size_t)nframes, *tv = tu + cfp * (size_t)nframes;
            if (c->dn2_luma == 2 && srcStride[0] > 0 && !(ed && (ed[0] == '0' || ed[0] == 'l')) && chrW % (cstep == 2 ? 2 : 4) == 0 &&
                chrW / (cstep == 2 ? 2 : 4) >= 3 && cus > 0 && cvs > 0) {
                /* exact 2:1 with a chroma line per output line: luma and chroma in ONE launch of the static-schedule kernel (the chroma
                 * jobs without a vertical filter), an interleaved pair leaves as a plane of (u, v) bytes (FFHIP_SWS_DOWN2=l: luma only,
                 * the chroma on the wide walker as before round 5's last step) */
                FFHipDn2Args D;
                memset(&D, 0, sizeof(D));
                D.nframes = nframes;
                D.xcd = 1;
                {
                    FFHipDn2Job &j = D.job[D.njobs++];
                    j.src = s0; j.dst = ty; j.sstride = srcStride[0]; j.dstride = (ptrdiff_t)ypitch; j.sfp = srcFramePitch[0]; j.dfp = yfp;
                    j.srcH = a.srcH; j.dstH = a.dstH; j.ngroups = a.dstW / 4;
                    j.hfv = c->dn2_h[0]; j.vfv = c->dn2_v[0];
                    j.y16 = 1;
                    ffhip_down2_plan_job(&j, 32);
                }
                for (int k = 0; k < (cstep == 2 ? 1 : 2); k++) {
                    FFHipDn2Job &j = D.job[D.njobs++];
                    j.pair = cstep == 2; j.swap = cstep == 2 && cv < cu; j.v1 = 1;
                    j.src = cstep == 2 ? s1 : k ? cv : cu; j.sstride = k ? cvs : cus; j.sfp = k ? cvf : cuf;
                    j.dst = k ? tv : tu; j.dstride = (ptrdiff_t)(cstep == 2 ? 2 * cpitch : cpitch); j.dfp = cstep == 2 ? 2 * cfp : cfp;
                    j.srcH = a.chrSrcH; j.dstH = a.dstH; j.ngroups = chrW / (cstep == 2 ? 2 : 4);
                    j.hfv = c->dn2_h[1]; j.vfv = c->dn2_v[0];
                    ffhip_down2_plan_job(&j, 32);
                }
                int r1 = ffhip_launch_down2(D, stream);
                if (r1 < 0)
                    return r1;
                FFHipY16RgbArgs Y;
                memset(&Y, 0, sizeof(Y));
                Y.y = ty; Y.u = tu; Y.v = tv; Y.dst = a.dst;
                Y.uvi = cstep == 2;
                Y.ystride = (ptrdiff_t)ypitch; Y.cstride = (ptrdiff_t)(cstep == 2 ? 2 * cpitch : cpitch); Y.dstride = a.dst_stride;
                Y.yfp = yfp; Y.cfp = cstep == 2 ? 2 * cfp : cfp; Y.dfp = a.dst_fp;
                Y.w = a.dstW; Y.h = a.dstH; Y.nframes = nframes; Y.lay = a.bgr; Y.k = c->k;
                r1 = ffhip_launch_y16_rgb(Y, stream);
                if (r1 >= 0)
                    HIP_TRY(hipEventRecord(c->rgb2_done, stream));
                return r1;
            }
            FFHipLwArgs W;
            memset(&W, 0, sizeof(W));
            W.nframes = nframes; W.ht = c->lw_ht; W.vt = c->lw_vt;
            auto wbank = [&](FFHipLwJob &j, int srcW, int srcH, int dstW, int which) {
                j.srcW = srcW; j.srcH = srcH; j.dstW = dstW; j.dstH = a.dstH;
                j.hf = c->dw[which].filter; j.hp = c->dw[which].pos; j.vf = c->dw[2 + which].filter; j.vp = c->dw[2 + which].pos;
                ffhip_lw_plan_job(&j);
            };
            if (cstep == 1) {
                for (int k = 0; k < 2; k++) {
                    FFHipLwJob &j = W.job[W.njobs++];
                    j.src[0] = k ? cv : cu; j.sstride[0] = k ? cvs : cus; j.sfp[0] = k ? cvf : cuf;
                    j.dst[0] = k ? tv : tu; j.dstride[0] = (ptrdiff_t)cpitch; j.dfp[0] = cfp;
                    wbank(j, a.chrSrcW, a.chrSrcH, a.dstW / 2, 1);
                }
            } else {
                FFHipLwJob &j = W.job[W.njobs++];
                j.pair = 1; j.sil = 1; j.dil = 0;
                j.src_swap = cv < cu;
                j.src[0] = j.src[1] = s1; j.sstride[0] = j.sstride[1] = cus; j.sfp[0] = j.sfp[1] = cuf;
                j.dst[0] = tu; j.dst[1] = tv; j.dstride[0] = j.dstride[1] = (ptrdiff_t)cpitch; j.dfp[0] = j.dfp[1] = cfp;
                wbank(j, a.chrSrcW, a.chrSrcH, a.dstW / 2, 1);
            }
            int r2 = 0;
            bool joined = false;
            if (c->dn2_luma && srcStride[0] > 0 && !(ed && ed[0] == '0')) {
                /* the two first-stage kernels do not depend on each other and neither fills the chip for long (the chroma walker waits on
                 * LDS round trips, VALU 29 % busy): side by side on two streams (FFHIP_SWS_RGB2=s: one after the other) */
                hipStream_t ls = stream;
                if (!(e2 && e2[0] == 's')) {
                    if (!c->rgb2_aux) {
                        HIP_TRY(hipStreamCreateWithFlags(&c->rgb2_aux, hipStreamNonBlocking));
                        HIP_TRY(hipEventCreateWithFlags(&c->rgb2_fork, hipEventDisableTiming));
                        HIP_TRY(hipEventCreateWithFlags(&c->rgb2_join, hipEventDisableTiming));
                    }
                    HIP_TRY(hipEventRecord(c->rgb2_fork, stream));
                    HIP_TRY(hipStreamWaitEvent(c->rgb2_aux, c->rgb2_fork, 0));
                    ls = c->rgb2_aux;
                    joined = true;
                }
                FFHipDn2Args D;
                memset(&D, 0, sizeof(D));
                D.nframes = nframes;
                D.xcd = 1;
                FFHipDn2Job &j = D.job[D.njobs++];
                j.src = s0; j.dst = ty; j.sstride = srcStride[0]; j.dstride = (ptrdiff_t)ypitch; j.sfp = srcFramePitch[0]; j.dfp = yfp;
                j.srcH = a.srcH; j.dstH = a.dstH; j.ngroups = a.dstW / 4;
                j.hfv = c->dn2_h[0]; j.vfv = c->dn2_v[0];
                j.y16 = 1;
                ffhip_down2_plan_job(&j, 32);
                r2 = ffhip_launch_down2(D, ls);
                if (joined)
                    HIP_TRY(hipEventRecord(c->rgb2_join, ls));
            } else {
                FFHipLwJob &jl = W.job[W.njobs++];
                jl.src[0] = s0; jl.sstride[0] = srcStride[0]; jl.sfp[0] = srcFramePitch[0];
                jl.dst[0] = ty; jl.dstride[0] = (ptrdiff_t)ypitch; jl.dfp[0] = yfp;
                jl.y16 = 1;
                wbank(jl, a.srcW, a.srcH, a.dstW, 0);
            }
            if (r2 >= 0)
                r2 = ffhip_launch_lwalk(W, stream);
            if (joined)
                HIP_TRY(hipStreamWaitEvent(stream, c->rgb2_join, 0));
            if (r2 < 0)
                return r2;
            FFHipY16RgbArgs Y;
            memset(&Y, 0, sizeof(Y));
            Y.y = ty; Y.u = tu; Y.v = tv; Y.dst = a.dst;
            Y.ystride = (ptrdiff_t)ypitch; Y.cstride = (ptrdiff_t)cpitch; Y.dstride = a.dst_stride;
            Y.yfp = yfp; Y.cfp = cfp; Y.dfp = a.dst_fp;
            Y.w = a.dstW; Y.h = a.dstH; Y.nframes = nframes; Y.lay = a.bgr; Y.k = c->k;
            r2 = ffhip_launch_y16_rgb(Y, stream);
            if (r2 >= 0)
                HIP_TRY(hipEventRecord(c->rgb2_done, stream));
            return r2;
        }
        return ffhip_launch_scale_rgb(a, stream);
    }

    FFHipScalePlaneArgs l = c->lum, ch = c->chr;
    l.src[0] = l.src[1] = s0; l.src_stride[0] = l.src_stride[1] = srcStride[0];
    l.src_fp[0] = l.src_fp[1] = srcFramePitch[0]; l.src_step = 1;
    l.dst[0] = l.dst[1] = (uint8_t *)dst[0]; l.dst_stride[0] = l.dst_stride[1] = dstStride[0];
    l.dst_fp[0] = l.dst_fp[1] = dstFramePitch[0]; l.dst_step = 1;
    l.nframes = nframes;
    ch.src[0] = cu; ch.src[1] = cv; ch.src_stride[0] = cus; ch.src_stride[1] = cvs;
    ch.src_fp[0] = cuf; ch.src_fp[1] = cvf; ch.src_step = cstep;
    if (fmt_nv(t.dstFormat)) {
        const int sw = t.dstFormat == FFHIP_PIX_FMT_NV21;
        ch.dst[0] = (uint8_t *)dst[1] + sw; ch.dst[1] = (uint8_t *)dst[1] + !sw;
        ch.dst_stride[0] = ch.dst_stride[1] = dstStride[1];
        ch.dst_fp[0] = ch.dst_fp[1] = dstFramePitch[1];
        ch.dst_step = 2;
    } else {
        ch.dst[0] = (uint8_t *)dst[1]; ch.dst[1] = (uint8_t *)dst[2];
        ch.dst_stride[0] = dstStride[1]; ch.dst_stride[1] = dstStride[2];
        ch.dst_fp[0] = dstFramePitch[1]; ch.dst_fp[1] = dstFramePitch[2];
        ch.dst_step = 1;
    }
    ch.nframes = nframes;

    /* fast path: 4x4-tap banks, dword-aligned planes.  FFHIP_SWS_FAST=0 forces the LDS-tiled kernel;
     * FFHIP_CW_LUMA_GROUPS / FFHIP_CW_PLAIN select measured variants. */
    const char *ev = FFHIP_KNOB("FFHIP_SWS_FAST");
    if (c->c420_ok && !(ev && ev[0] == '0') && srcStride[0] > 0 && cus > 0 && cvs > 0 && dstStride[0] > 0 && dstStride[1] > 0 &&
        (fmt_nv(t.dstFormat) || dstStride[2] > 0)) {
        FFHipCopy420Args K;
        memset(&K, 0, sizeof(K));
        K.nframes = nframes;
        const int cw = c->d[1].n, chh = c->d[3].n;
        {
            FFHipCopy420Job &j = K.job[K.njobs++];
            j.src[0] = s0; j.sstride[0] = srcStride[0]; j.sfp[0] = srcFramePitch[0];
            j.dst = (uint8_t *)dst[0]; j.dstride = dstStride[0]; j.dfp = dstFramePitch[0];
            j.kind = 0; j.wbytes = t.dstW; j.rows = t.dstH;
        }
        if (fmt_nv(t.dstFormat)) {
            FFHipCopy420Job &j = K.job[K.njobs++];
            j.dst = (uint8_t *)dst[1]; j.dstride = dstStride[1]; j.dfp = dstFramePitch[1];
            j.wbytes = 2 * cw; j.rows = chh;
            const bool dsw = t.dstFormat == FFHIP_PIX_FMT_NV21;
            if (cstep == 2) { /* pairs in, pairs out: as they are, or each pair turned round */
                j.src[0] = s1; j.sstride[0] = cus; j.sfp[0] = cuf;
                j.kind = dsw == (t.srcFormat == FFHIP_PIX_FMT_NV21) ? 0 : 3;
            } else {
                j.kind = 2;
                j.src[0] = dsw ? cv : cu; j.sstride[0] = dsw ? cvs : cus; j.sfp[0] = dsw ? cvf : cuf;
                j.src[1] = dsw ? cu : cv; j.sstride[1] = dsw ? cus : cvs; j.sfp[1] = dsw ? cuf : cvf;
            }
        } else {
            for (int k = 0; k < 2; k++) {
                FFHipCopy420Job &j = K.job[K.njobs++];
                j.dst = (uint8_t *)dst[1 + k]; j.dstride = dstStride[1 + k]; j.dfp = dstFramePitch[1 + k];
                j.wbytes = cw; j.rows = chh;
                if (cstep == 2) {
                    j.src[0] = s1; j.sstride[0] = cus; j.sfp[0] = cuf;
                    j.kind = 1; j.k = (t.srcFormat == FFHIP_PIX_FMT_NV21) ? !k : k;
                } else {
                    j.src[0] = k ? cv : cu; j.sstride[0] = k ? cvs : cus; j.sfp[0] = k ? cvf : cuf;
                    j.kind = 0;
                }
            }
        }
        return ffhip_launch_copy420(K, stream);
    }
    if (c->mix_dn2 && !(ev && ev[0] == '0') && srcStride[0] > 0 && cus > 0 && cvs > 0 && dstStride[0] > 0 && dstStride[1] > 0 && dstStride[2] > 0 &&
        !(((uintptr_t)cu | (uintptr_t)cv | (size_t)cus | (size_t)cvs | cuf | cvf | (uintptr_t)dst[1] | (uintptr_t)dst[2] | (size_t)dstStride[1] |
           (size_t)dstStride[2] | dstFramePitch[1] | dstFramePitch[2]) & 3)) {
        FFHipCopy420Args K;
        memset(&K, 0, sizeof(K));
        K.nframes = nframes;
        FFHipCopy420Job &kj = K.job[K.njobs++];
        kj.src[0] = s0; kj.sstride[0] = srcStride[0]; kj.sfp[0] = srcFramePitch[0];
        kj.dst = (uint8_t *)dst[0]; kj.dstride = dstStride[0]; kj.dfp = dstFramePitch[0];
        kj.kind = 0; kj.wbytes = t.dstW; kj.rows = t.dstH;
        int r1 = ffhip_launch_copy420(K, stream);
        if (r1 < 0)
            return r1;
        FFHipDn2Args D;
        memset(&D, 0, sizeof(D));
        D.nframes = nframes;
        D.xcd = 1;
        for (int k = 0; k < 2; k++) {
            FFHipDn2Job &j = D.job[D.njobs++];
            j.src = k ? cv : cu; j.sstride = k ? cvs : cus; j.sfp = k ? cvf : cuf;
            j.dst = (uint8_t *)dst[1 + k]; j.dstride = dstStride[1 + k]; j.dfp = dstFramePitch[1 + k];
            j.srcH = c->chrSrcH; j.dstH = c->d[3].n; j.ngroups = c->d[1].n / 4;
            j.hfv = c->dn2_h[1]; j.vfv = c->dn2_v[1];
            ffhip_down2_plan_job(&j, 32);
        }
        return ffhip_launch_down2(D, stream);
    }
    const char *eu3b = FFHIP_KNOB("FFHIP_SWS_UP32"); /* measure build: 0 keeps the column walker */
    uintptr_t al3 = (uintptr_t)l.src[0] | (size_t)l.src_stride[0] | l.src_fp[0] | (uintptr_t)l.dst[0] | (size_t)l.dst_stride[0] | l.dst_fp[0];
    bool neg3 = l.src_stride[0] < 0 || l.dst_stride[0] < 0;
    for (int i = 0; i < 2; i++) {
        al3 |= (size_t)ch.src_stride[i] | ch.src_fp[i] | (size_t)ch.dst_stride[i] | ch.dst_fp[i];
        al3 |= ch.src_step == 2 ? (uintptr_t)(ch.src[0] < ch.src[1] ? ch.src[0] : ch.src[1]) : (uintptr_t)ch.src[i];
        al3 |= ch.dst_step == 2 ? (uintptr_t)(ch.dst[0] < ch.dst[1] ? ch.dst[0] : ch.dst[1]) : (uintptr_t)ch.dst[i];
        neg3 = neg3 || ch.src_stride[i] < 0 || ch.dst_stride[i] < 0;
    }
    if (!(al3 & 3) && c->u32_ok && !neg3 && !c->luma_pass && !(eu3b && eu3b[0] == '0')) {
        /* exact 3:2 / 4:3 up: static schedule, no LDS (the 8-bit twin in sws_up32.hip) */
        FFHipU32Args U;
        memset(&U, 0, sizeof(U));
        U.nframes = nframes;
        U.bytes = 1;
        U.ratio43 = c->u32_ok == 2;
        const int no = U.ratio43 ? 16 : 12;
        auto ujob = [&](const FFHipScalePlaneArgs &p, int which, const uint8_t *src, ptrdiff_t ss, size_t sf, uint8_t *dst, ptrdiff_t dsr, size_t df, int pair) {
            FFHipU32Job &j = U.job[U.njobs++];
            j.src = src; j.dst = dst; j.sstride = ss; j.dstride = dsr; j.sfp = sf; j.dfp = df;
            j.pair = pair;
            j.srcH = p.srcH; j.dstH = p.dstH;
            j.ngroups = pair ? p.dstW / (no / 2) : p.dstW / no;
            j.hfv = c->u32_h[which]; j.vfv = c->u32_v[which];
        };
        ujob(l, 0, l.src[0], l.src_stride[0], l.src_fp[0], l.dst[0], l.dst_stride[0], l.dst_fp[0], 0);
        if (ch.src_step == 2) {
            ujob(ch, 1, ch.src[1] < ch.src[0] ? ch.src[1] : ch.src[0], ch.src_stride[0], ch.src_fp[0], ch.dst[1] < ch.dst[0] ? ch.dst[1] : ch.dst[0],
                 ch.dst_stride[0], ch.dst_fp[0], 1);
        } else {
            for (int k = 0; k < 2; k++)
                ujob(ch, 1, ch.src[k], ch.src_stride[k], ch.src_fp[k], ch.dst[k], ch.dst_stride[k], ch.dst_fp[k], 0);
        }
        return ffhip_launch_up32(U, stream);
    }
    if (c->cw_ok && !(ev && ev[0] == '0')) {
        uintptr_t al = 0;
        for (int i = 0; i < 2; i++) {
            al |= (uintptr_t)l.src[i] | (size_t)l.src_stride[i] | l.src_fp[i] | (uintptr_t)l.dst[i] |
                  (size_t)l.dst_stride[i] | l.dst_fp[i];
            al |= (size_t)ch.src_stride[i] | ch.src_fp[i] | (size_t)ch.dst_stride[i] | ch.dst_fp[i];
            /* an interleaved pair is addressed through its lower pointer */
            al |= ch.src_step == 2 ? (uintptr_t)(ch.src[0] < ch.src[1] ? ch.src[0] : ch.src[1]) : (uintptr_t)ch.src[i];
            al |= ch.dst_step == 2 ? (uintptr_t)(ch.dst[0] < ch.dst[1] ? ch.dst[0] : ch.dst[1]) : (uintptr_t)ch.dst[i];
        }
        const char *eu = FFHIP_KNOB("FFHIP_SWS_UP2");
        if (!(al & 3) && (c->up2_ok || (c->mix_up2 && !c->luma_pass && l.src_stride[0] > 0 && l.dst_stride[0] > 0 && ch.src_stride[0] > 0 && ch.src_stride[1] > 0 &&
                                            ch.dst_stride[0] > 0 && ch.dst_stride[1] > 0)) && !(eu && eu[0] == '0') && !(em_forced())) {
            /* exact 2x: static schedule, regular windows (sws_up2.hip).  FFHIP_SWS_UP2=0 takes the general column walker. */
            FFHipUp2Args U;
            memset(&U, 0, sizeof(U));
            U.nframes = nframes;
            auto upjob = [&](const FFHipScalePlaneArgs &p, int which, const uint8_t *src, ptrdiff_t ss, size_t sf, uint8_t *dst,
                             ptrdiff_t dsr, size_t df, int pair, int swap) {
                FFHipUp2Job &j = U.job[U.njobs++];
                j.src = src; j.dst = dst; j.sstride = ss; j.dstride = dsr; j.sfp = sf; j.dfp = df;
                j.pair = pair; j.swap = swap;
                j.srcW = p.srcW; j.srcH = p.srcH;
                j.ngroups = pair ? p.srcW / 2 : p.srcW / 4;
                j.hfv = c->up2_h[which]; j.vfv = c->up2_v[which];
                j.rc_coeff = p.rc_coeff; j.rc_offset = p.rc_offset;
                j.hco_ok = c->up2_hco_ok[which];
                memcpy(j.hco, c->up2_hco[which], sizeof(j.hco));
            };
            if (c->mix_up2) {
                FFHipCopy420Args K;
                memset(&K, 0, sizeof(K));
                K.nframes = nframes;
                FFHipCopy420Job &kj = K.job[K.njobs++];
                kj.src[0] = l.src[0]; kj.sstride[0] = l.src_stride[0]; kj.sfp[0] = l.src_fp[0];
                kj.dst = l.dst[0]; kj.dstride = l.dst_stride[0]; kj.dfp = l.dst_fp[0];
                kj.kind = 0; kj.wbytes = l.dstW; kj.rows = l.dstH;
                const int r1 = ffhip_launch_copy420(K, stream);
                if (r1 < 0)
                    return r1;
            } else {
                upjob(l, 0, l.src[0], l.src_stride[0], l.src_fp[0], l.dst[0], l.dst_stride[0], l.dst_fp[0], 0, 0);
            }
            if (c->luma_pass) {
            } else if (ch.src_step == 2) {
                const bool ssw = ch.src[1] < ch.src[0], dsw = ch.dst[1] < ch.dst[0];
                upjob(ch, 1, ssw ? ch.src[1] : ch.src[0], ch.src_stride[0], ch.src_fp[0], dsw ? ch.dst[1] : ch.dst[0],
                      ch.dst_stride[0], ch.dst_fp[0], 1, ssw != dsw);
            } else {
                for (int k = 0; k < 2; k++)
                    upjob(ch, 1, ch.src[k], ch.src_stride[k], ch.src_fp[k], ch.dst[k], ch.dst_stride[k], ch.dst_fp[k], 0, 0);
            }
            /* frames per wave: the split that wastes the fewest lanes at the right edge of the widest job's rows;
             * lane offsets (frame pitch included) must stay below 2^32 */
            const char *ef = FFHIP_KNOB("FFHIP_UP2_FSHIFT"), *es = FFHIP_KNOB("FFHIP_UP2_STRIP"), *ed = FFHIP_KNOB("FFHIP_UP2_DEPTH");
            const char *ev2 = FFHIP_KNOB("FFHIP_UP2_VAR"), *ex = FFHIP_KNOB("FFHIP_UP2_XCD");
            U.xcd = ex ? atoi(ex) : 1;     /* measure build: 0 plain numbering, 1 an eighth per XCD (the product), 1 + k chunks of 2^k workgroups */
            int best = 0;
            double bestw = 1e30;
            for (int fsft = 0; fsft <= 2; fsft++) {
                const int lpf = 64 >> fsft;
                bool fits = true;
                double w = 0;
                for (int i = 0; i < U.njobs; i++) {
                    const FFHipUp2Job &j = U.job[i];
                    const unsigned long long span_s = (unsigned long long)((1 << fsft) - 1) * j.sfp + (unsigned long long)j.srcH * (size_t)(j.sstride < 0 ? -j.sstride : j.sstride);
                    const unsigned long long span_d = (unsigned long long)((1 << fsft) - 1) * j.dfp + 2ull * j.srcH * (size_t)(j.dstride < 0 ? -j.dstride : j.dstride);
                    if (span_s >= (1ull << 31) || span_d >= (1ull << 31))
                        fits = false;
                    /* waves per frame and strip: the full blocks, plus this frame's share of the shared ragged-end blocks */
                    const int nfull = fsft ? j.ngroups / 64 : 0;
                    w += ((double)nfull + (double)cdiv(j.ngroups - nfull * 64, lpf) / (1 << fsft)) * j.srcH;
                }
                if (nframes < (1 << fsft) && fsft)
                    fits = false;
                if (fits && w < bestw - 1e-9) { bestw = w; best = fsft; }
            }
            U.fshift = ef && ef[0] >= '0' && ef[0] <= '2' ? ef[0] - '0' : best;
            bool neg = false;
            for (int i = 0; i < U.njobs; i++)
                neg = neg || U.job[i].sstride < 0 || U.job[i].dstride < 0;
            if (!neg) {
                /* source rows per strip: 60 when that is eight waves per SIMD or more; shorter strips for a small batch — a strip
                 * re-reads the rows above it and pays a prologue, but ONE 1080p -> 4K frame in 60-row strips is 232 waves on 1,024
                 * SIMDs (measured, nv12: 1 frame 27.7 -> 20.6 us, 4 frames 30.5 -> 23.8 us, 16 frames 63.5 -> 56.6 us, 64 frames
                 * unchanged; what sws_scale_frame() on a filter graph's frames sees.  FFHIP_UP2_STRIP fixes it in the measure build) */
                static const int wants[] = { 60, 36, 24, 12 };
                for (int t = 0; t < 4; t++) {
                    long long u = 0;
                    for (int i = 0; i < U.njobs; i++) {
                        ffhip_up2_plan_job(&U.job[i], 64 >> U.fshift, es && atoi(es) > 0 ? atoi(es) : wants[t]);
                        u += (long long)U.job[i].upj * U.job[i].nstrips;
                    }
                    if ((es && atoi(es) > 0) || u * ((nframes + (1 << U.fshift) - 1) >> U.fshift) >= 8192)
                        break;
                }
                /* FFHIP_UP2_VAR: 0 the product; 16 / 48 / 64 = measurement-only builds (no stores / arithmetic only / bytes only) */
                /* round 6, the product: the horizontal bank in SGPRs, six source rows in flight, non-temporal stores (variant 3 at depth 6:
                 * +0.4 .. +0.7 % over rounds 2-5's kernel on two boxes, profiles/r06_up2_variants.txt; each ingredient alone is within
                 * +-0.5 %); FFHIP_UP2_VAR=0 FFHIP_UP2_DEPTH=3 (measure build) is that kernel.  Banks whose interior columns are not the two
                 * phase rows (ffhip_up2_hco) fall back to it inside the launcher. */
                /* (the range-converting twin: the SGPR bank and non-temporal stores at three rows in flight — yuvj420p -> yuv420p 1080p -> 4K
                 * 0.550 -> 0.592 of HBM, six rows 0.585; profiles/r06_up2_twins.txt) */
                bool scal = true;
                for (int i = 0; i < U.njobs; i++)
                    scal = scal && U.job[i].hco_ok;
                return ffhip_launch_up2(U, ed ? (ed[0] == '6' ? 6 : 3) : scal && !U.job[0].rc_coeff ? 6 : 3, ev2 ? atoi(ev2) : scal ? 3 : 0, stream);
            }
        }
        if (!(al & 3) && !c->up2_rc) {
            const char *em = FFHIP_KNOB("FFHIP_SWS_MFMA");
            if (c->mf_ok && em && em[0] == '1') {
                /* horizontal pass on the matrix cores (k_sws_mfma) */
                const char *est = FFHIP_KNOB("FFHIP_MF_STRIP");
                FFHipMfArgs M;
                memset(&M, 0, sizeof(M));
                M.nframes = nframes;
                auto mfjob = [&](const FFHipScalePlaneArgs &p, int which, const uint8_t *src, ptrdiff_t ss, size_t sf, uint8_t *dst,
                                 ptrdiff_t dsr, size_t df, int pair, int dswap) {
                    FFHipMfJob &j = M.job[M.njobs++];
                    j.src = src; j.dst = dst; j.sstride = ss; j.dstride = dsr; j.sfp = sf; j.dfp = df;
                    j.pair = pair; j.dst_swap = dswap;
                    j.srcH = p.srcH; j.dstW = p.dstW; j.dstH = p.dstH;
                    j.tiles = c->mf_tiles[which]; j.vf = c->dn[2 + which].filter; j.vp = c->dn[2 + which].pos; j.ys = c->mf_ys[which];
                    j.ntiles = c->mf_ntiles[which];
                    j.ncb = cdiv(j.ntiles, 16);
                    const int want = est && atoi(est) > 0 ? atoi(est) : 540;
                    const int ns = cdiv(p.dstH, want);
                    j.strip_rows = cdiv(p.dstH, ns);
                    j.nstrips = cdiv(p.dstH, j.strip_rows);
                };
                mfjob(l, 0, l.src[0], l.src_stride[0], l.src_fp[0], l.dst[0], l.dst_stride[0], l.dst_fp[0], 0, 0);
                if (c->luma_pass) {
                } else if (c->mf_chr_pair) {
                    const uint8_t *sp = ch.src[0] < ch.src[1] ? ch.src[0] : ch.src[1];
                    uint8_t *dp = ch.dst[0] < ch.dst[1] ? ch.dst[0] : ch.dst[1];
                    mfjob(ch, 1, sp, ch.src_stride[0], ch.src_fp[0], dp, ch.dst_stride[0], ch.dst_fp[0], 1, ch.dst[1] < ch.dst[0]);
                } else {
                    for (int k = 0; k < 2; k++)
                        mfjob(ch, 1, ch.src[k], ch.src_stride[k], ch.src_fp[k], ch.dst[k], ch.dst_stride[k], ch.dst_fp[k], 0, 0);
                }
                return ffhip_launch_mfma(M, stream);
            }
            const char *eg = FFHIP_KNOB("FFHIP_CW_LUMA_GROUPS"), *ep = FFHIP_KNOB("FFHIP_CW_PLAIN");
            const char *ed = FFHIP_KNOB("FFHIP_CW_DEPTH"), *es = FFHIP_KNOB("FFHIP_CW_STRIP");
            const int lg = (eg && eg[0] == '1') || (ep && ep[0] == '1') ? 1 : 2; /* measured: 2 groups/lane is 12 % faster */
            const int depth = ed && ed[0] == '3' ? 3 : 6; /* measured: 6 rows in flight is 4 % faster with OPT */
            /* output rows per strip: 120 when that is four waves per SIMD or more, shorter strips for a smaller batch (measured, yuv420p
             * 720p -> 1080p: 1 frame 49 -> 19 us, 8 frames 50 -> 29 us at 24 rows, 32 frames 64 -> 57 us at 40: a lone frame in
             * 120-row strips is 56 waves) */
            static const int strips[] = { 120, 60, 40, 24 };
            int gpl[3], strip = 120;
            FFHipCwArgs A;
            memset(&A, 0, sizeof(A));
            A.nframes = nframes;
            const char *eo = FFHIP_KNOB("FFHIP_CW_OPT");
            A.flags = ep && ep[0] == '1' ? 1 : 0;
            if (c->cw_opt && !A.flags && !(eo && eo[0] == '0'))
                A.flags |= 2;
            const char *edup = FFHIP_KNOB("FFHIP_CW_DUP");
            if ((A.flags & 2) && c->cw_dup && !(edup && edup[0] == '0'))
                A.flags |= 4;
            auto bank = [&](FFHipCwJob &j, const FFHipScalePlaneArgs &p) {
                const int which = &p == &ch ? 1 : 0; /* the padded 4-tap view of the banks */
                j.srcW = p.srcW; j.srcH = p.srcH; j.dstW = p.dstW; j.dstH = p.dstH;
                j.hf = c->dn[which].filter; j.hp = c->dn[which].pos; j.vf = c->dn[2 + which].filter; j.vp = c->dn[2 + which].pos;
            };
            FFHipCwJob &jl = A.job[0];
            bank(jl, l);
            jl.kind = lg == 2 ? 1 : 0;
            jl.src[0] = l.src[0]; jl.sstride[0] = l.src_stride[0]; jl.sfp[0] = l.src_fp[0];
            jl.dst[0] = l.dst[0]; jl.dstride[0] = l.dst_stride[0]; jl.dfp[0] = l.dst_fp[0];
            gpl[0] = lg;
            A.njobs = 1;
            if (c->luma_pass) {
            } else if (ch.src_step == 1 && ch.dst_step == 1) {
                for (int k = 0; k < 2; k++) {
                    FFHipCwJob &j = A.job[A.njobs++];
                    bank(j, ch);
                    j.kind = jl.kind;
                    j.src[0] = ch.src[k]; j.sstride[0] = ch.src_stride[k]; j.sfp[0] = ch.src_fp[k];
                    j.dst[0] = ch.dst[k]; j.dstride[0] = ch.dst_stride[k]; j.dfp[0] = ch.dst_fp[k];
                    gpl[A.njobs - 1] = lg;
                }
            } else {
                FFHipCwJob &j = A.job[A.njobs++];
                bank(j, ch);
                const bool src_il = ch.src_step == 2, dst_il = ch.dst_step == 2;
                j.kind = src_il && dst_il ? 2 : src_il ? 3 : 4;
                for (int k = 0; k < 2; k++) {
                    j.src[k] = ch.src[k]; j.sstride[k] = ch.src_stride[k]; j.sfp[k] = ch.src_fp[k];
                    j.dst[k] = ch.dst[k]; j.dstride[k] = ch.dst_stride[k]; j.dfp[k] = ch.dst_fp[k];
                }
                if (src_il) {
                    j.src_swap = ch.src[1] < ch.src[0];
                    j.src[0] = j.src_swap ? ch.src[1] : ch.src[0];
                }
                if (dst_il) {
                    j.dst_swap = ch.dst[1] < ch.dst[0];
                    j.dst[0] = j.dst_swap ? ch.dst[1] : ch.dst[0];
                }
                gpl[A.njobs - 1] = 1;
            }
            for (int t = 0; t < 4; t++) {
                strip = es && atoi(es) > 0 ? atoi(es) : strips[t];
                long long u = 0;
                for (int i = 0; i < A.njobs; i++) {
                    ffhip_cw_plan_job(&A.job[i], gpl[i], strip);
                    u += (long long)A.job[i].ncb * A.job[i].nstrips;
                }
                if ((es && atoi(es) > 0) || u * nframes >= 4096)
                    break;
            }
            return ffhip_launch_colwalk(A, lg, depth, stream);
        }
    }
    uintptr_t al2 = (uintptr_t)l.src[0] | (size_t)l.src_stride[0] | l.src_fp[0] | (uintptr_t)l.dst[0] | (size_t)l.dst_stride[0] | l.dst_fp[0];
    for (int i = 0; i < 2; i++) {
        al2 |= (size_t)ch.src_stride[i] | ch.src_fp[i] | (size_t)ch.dst_stride[i] | ch.dst_fp[i];
        al2 |= ch.src_step == 2 ? (uintptr_t)(ch.src[0] < ch.src[1] ? ch.src[0] : ch.src[1]) : (uintptr_t)ch.src[i];
        al2 |= ch.dst_step == 2 ? (uintptr_t)(ch.dst[0] < ch.dst[1] ? ch.dst[0] : ch.dst[1]) : (uintptr_t)ch.dst[i];
    }
    const char *e2 = FFHIP_KNOB("FFHIP_SWS_DOWN2");
    bool neg = false;
    for (int i = 0; i < 2; i++)
        neg = neg || l.src_stride[i] < 0 || l.dst_stride[i] < 0 || ch.src_stride[i] < 0 || ch.dst_stride[i] < 0;
    if (!(al2 & 3) && c->dn2_ok && !neg && !(e2 && e2[0] == '0') && !(ev && ev[0] == '0')) {
        /* exact 2:1: static schedule, regular windows, no LDS (sws_down2.hip).  FFHIP_SWS_DOWN2=0 takes the wide walker. */
        FFHipDn2Args D;
        memset(&D, 0, sizeof(D));
        D.nframes = nframes;
        const char *ex = FFHIP_KNOB("FFHIP_DN2_XCD"), *es = FFHIP_KNOB("FFHIP_DN2_STRIP");
        D.xcd = !(ex && ex[0] == '0');
        auto dnjob = [&](const FFHipScalePlaneArgs &p, int which, const uint8_t *src, ptrdiff_t ss, size_t sf, uint8_t *dst,
                         ptrdiff_t dsr, size_t df, int pair, int swap) {
            FFHipDn2Job &j = D.job[D.njobs++];
            j.src = src; j.dst = dst; j.sstride = ss; j.dstride = dsr; j.sfp = sf; j.dfp = df;
            j.pair = pair; j.swap = swap;
            j.srcH = p.srcH; j.dstH = p.dstH;
            j.ngroups = pair ? p.dstW / 2 : p.dstW / 4;
            j.hfv = c->dn2_h[which]; j.vfv = c->dn2_v[which];
            ffhip_down2_plan_job(&j, es && atoi(es) > 0 ? atoi(es) : 32); /* measured: 28..36 rows per strip */
        };
        dnjob(l, 0, l.src[0], l.src_stride[0], l.src_fp[0], l.dst[0], l.dst_stride[0], l.dst_fp[0], 0, 0);
        if (c->luma_pass) {
        } else if (ch.src_step == 2) {
            const bool ssw = ch.src[1] < ch.src[0], dsw = ch.dst[1] < ch.dst[0];
            dnjob(ch, 1, ssw ? ch.src[1] : ch.src[0], ch.src_stride[0], ch.src_fp[0], dsw ? ch.dst[1] : ch.dst[0],
                  ch.dst_stride[0], ch.dst_fp[0], 1, ssw != dsw);
        } else {
            for (int k = 0; k < 2; k++)
                dnjob(ch, 1, ch.src[k], ch.src_stride[k], ch.src_fp[k], ch.dst[k], ch.dst_stride[k], ch.dst_fp[k], 0, 0);
        }
        return ffhip_launch_down2(D, stream);
    }
    const char *e3 = FFHIP_KNOB("FFHIP_SWS_DOWN32"); /* measure build: 0 keeps the wide walker */
    if (!(al2 & 3) && c->d32_ok && !neg && !c->luma_pass && !(e3 && e3[0] == '0') && !(ev && ev[0] == '0')) {
        /* exact 3:2: static schedule with period (3 in, 2 out), no LDS (sws_down32.hip) */
        FFHipD32Args D;
        memset(&D, 0, sizeof(D));
        D.nframes = nframes;
        auto djob = [&](const FFHipScalePlaneArgs &p, int which, const uint8_t *src, ptrdiff_t ss, size_t sf, uint8_t *dst, ptrdiff_t dsr, size_t df,
                        int pair, int swap) {
            FFHipD32Job &j = D.job[D.njobs++];
            j.src = src; j.dst = dst; j.sstride = ss; j.dstride = dsr; j.sfp = sf; j.dfp = df;
            j.pair = pair; j.swap = swap;
            j.srcH = p.srcH; j.dstH = p.dstH;
            j.ngroups = pair ? p.dstW / 4 : p.dstW / 8;
            j.hfv = c->d32_h[which]; j.vfv = c->d32_v[which];
        };
        djob(l, 0, l.src[0], l.src_stride[0], l.src_fp[0], l.dst[0], l.dst_stride[0], l.dst_fp[0], 0, 0);
        if (ch.src_step == 2) {
            const bool ssw = ch.src[1] < ch.src[0], dsw = ch.dst[1] < ch.dst[0];
            djob(ch, 1, ssw ? ch.src[1] : ch.src[0], ch.src_stride[0], ch.src_fp[0], dsw ? ch.dst[1] : ch.dst[0], ch.dst_stride[0], ch.dst_fp[0], 1,
                 ssw != dsw);
        } else {
            for (int k = 0; k < 2; k++)
                djob(ch, 1, ch.src[k], ch.src_stride[k], ch.src_fp[k], ch.dst[k], ch.dst_stride[k], ch.dst_fp[k], 0, 0);
        }
        return ffhip_launch_down32(D, stream);
    }
    /* wide banks: the LDS-backed walker (FFHIP_SWS_WIDE=0 forces the LDS-tiled kernel) */
    const char *ew = FFHIP_KNOB("FFHIP_SWS_WIDE");
    const bool cw_taken_off = ev && ev[0] == '0';
    if (c->lw_ok && !(ew && ew[0] == '0') && !(cw_taken_off && !(ew && ew[0] == '1'))) {
        uintptr_t al = (uintptr_t)l.src[0] | (size_t)l.src_stride[0] | l.src_fp[0] | (uintptr_t)l.dst[0] | (size_t)l.dst_stride[0] |
                       l.dst_fp[0];
        for (int i = 0; i < 2; i++) {
            al |= (size_t)ch.src_stride[i] | ch.src_fp[i] | (size_t)ch.dst_stride[i] | ch.dst_fp[i];
            al |= ch.src_step == 2 ? (uintptr_t)(ch.src[0] < ch.src[1] ? ch.src[0] : ch.src[1]) : (uintptr_t)ch.src[i];
            al |= ch.dst_step == 2 ? (uintptr_t)(ch.dst[0] < ch.dst[1] ? ch.dst[0] : ch.dst[1]) : (uintptr_t)ch.dst[i];
        }
        if (!(al & 3)) {
            FFHipLwArgs W;
            memset(&W, 0, sizeof(W));
            W.nframes = nframes; W.ht = c->lw_ht; W.vt = c->lw_vt;
            auto wbank = [&](FFHipLwJob &j, const FFHipScalePlaneArgs &p, int which) {
                j.srcW = p.srcW; j.srcH = p.srcH; j.dstW = p.dstW; j.dstH = p.dstH;
                j.hf = c->dw[which].filter; j.hp = c->dw[which].pos; j.vf = c->dw[2 + which].filter; j.vp = c->dw[2 + which].pos;
                ffhip_lw_plan_job(&j);
            };
            /* the heavier units (a U/V pair is twice a plane) are enumerated first: they start first */
            if (c->luma_pass) {
            } else if (ch.src_step == 1 && ch.dst_step == 1) {
                for (int k = 0; k < 2; k++) {
                    FFHipLwJob &j = W.job[W.njobs++];
                    j.src[0] = ch.src[k]; j.sstride[0] = ch.src_stride[k]; j.sfp[0] = ch.src_fp[k];
                    j.dst[0] = ch.dst[k]; j.dstride[0] = ch.dst_stride[k]; j.dfp[0] = ch.dst_fp[k];
                    wbank(j, ch, 1);
                }
            } else {
                FFHipLwJob &j = W.job[W.njobs++];
                j.pair = 1; j.sil = ch.src_step == 2; j.dil = ch.dst_step == 2;
                for (int k = 0; k < 2; k++) {
                    j.src[k] = ch.src[k]; j.sstride[k] = ch.src_stride[k]; j.sfp[k] = ch.src_fp[k];
                    j.dst[k] = ch.dst[k]; j.dstride[k] = ch.dst_stride[k]; j.dfp[k] = ch.dst_fp[k];
                }
                if (j.sil) {
                    j.src_swap = ch.src[1] < ch.src[0];
                    j.src[0] = j.src_swap ? ch.src[1] : ch.src[0];
                }
                if (j.dil) {
                    j.dst_swap = ch.dst[1] < ch.dst[0];
                    j.dst[0] = j.dst_swap ? ch.dst[1] : ch.dst[0];
                }
                wbank(j, ch, 1);
            }
            FFHipLwJob &jl = W.job[W.njobs++];
            jl.src[0] = l.src[0]; jl.sstride[0] = l.src_stride[0]; jl.sfp[0] = l.src_fp[0];
            jl.dst[0] = l.dst[0]; jl.dstride[0] = l.dst_stride[0]; jl.dfp[0] = l.dst_fp[0];
            wbank(jl, l, 0);
            return ffhip_launch_lwalk(W, stream);
        }
    }
    if (c->luma_pass)
        ch.tiles_y = 0; /* the tiled kernel's grid is the luma tiles followed by the chroma tiles: none of the latter */
    return ffhip_launch_scale_yuv(l, ch, stream);
}

/* ---- host-pointer face ---------------------------------------------------------------------- */
static hipError_t copy2d(void *dst, ptrdiff_t dpitch, const void *src, ptrdiff_t spitch, size_t wbytes, int rows,
                         hipMemcpyKind kind)
{
    /* rows that follow each other on both sides: ONE linear copy.  Round 6: the runtime moves pageable memory through a 2-D copy at about
     * half the rate of a linear one (nv12 1080p -> 4K through ffhip_sws_scale: 0.75 -> 0.38 ms per frame, 41 GB/s over PCIe both ways,
     * profiles/r06_host_face.txt), so the staging buffer keeps rows of a multiple of 64 bytes tight and a tight host plane is copied
     * linearly.  (Registering the caller's buffers after a few sightings was measured too: no faster than this, and not kept.) */
    if (dpitch == (ptrdiff_t)wbytes && spitch == (ptrdiff_t)wbytes)
        return hipMemcpy(dst, src, wbytes * (size_t)rows, kind);
    if (dpitch >= (ptrdiff_t)wbytes && spitch >= (ptrdiff_t)wbytes)
        return hipMemcpy2D(dst, dpitch, src, spitch, wbytes, rows, kind);
    for (int r = 0; r < rows; r++) { /* negative (bottom-up) strides: swscale.c:1141-1158 */
        hipError_t e = hipMemcpy((uint8_t *)dst + r * dpitch, (const uint8_t *)src + r * spitch, wbytes, kind);
        if (e != hipSuccess)
            return e;
    }
    return hipSuccess;
}

static int sws_scale_locked(FFHipSwsContext *c, const uint8_t *const src[], const int srcStride[], int srcSliceY, int srcSliceH,
                            uint8_t *const dst[], const int dstStride[]);

extern "C" int ffhip_sws_scale(FFHipSwsContext *c, const uint8_t *const src[], const int srcStride[], int srcSliceY,
                               int srcSliceH, uint8_t *const dst[], const int dstStride[])
{
    if (!c || !src || !dst || srcSliceH <= 0)
        return FFHIP_EINVAL;
    FFHipDeviceGuard dg(c->device);
    std::lock_guard<std::mutex> lk(c->mu);
    if (c->t.dst_alpha_fill != 2 || c->unscaled_yuv2rgb || fmt_rgb(c->t.dstFormat))
        return sws_scale_locked(c, src, srcStride, srcSliceY, srcSliceH, dst, dstStride);
    /* alpha on both sides: a second pass whose luma is the alpha plane (see ffhip_sws_scale_batch_dev); whole frames only — the slice
     * collection holds one frame's source */
    if (srcSliceY != 0 || srcSliceH != c->t.srcH) {
        ffhip_set_error("ffhip_sws_scale: source slices together with a scaled alpha plane are not on the hip path");
        return FFHIP_ENOSYS;
    }
    if (!src[3] || !dst[3]) {
        ffhip_set_error("ffhip_sws_scale: the formats have an alpha plane: plane 3 is NULL");
        return FFHIP_EINVAL;
    }
    const int r = sws_scale_locked(c, src, srcStride, srcSliceY, srcSliceH, dst, dstStride);
    if (r < 0)
        return r;
    PlaneDesc dp[3];
    if (plane_list(c->t.dstFormat, c->t.dstW, c->t.dstH, dp) != 3)
        return FFHIP_EINVAL;
    std::vector<uint8_t> scratch((size_t)dp[1].wbytes * (size_t)dp[1].rows * 2);
    const uint8_t *const s2[4] = { src[3], src[1], src[2], nullptr };
    const int ss2[4] = { srcStride[3], srcStride[1], srcStride[2], 0 };
    uint8_t *const d2[4] = { dst[3], scratch.data(), scratch.data() + (size_t)dp[1].wbytes * (size_t)dp[1].rows, nullptr };
    const int ds2[4] = { dstStride[3], dp[1].wbytes, dp[1].wbytes, 0 };
    c->luma_pass = true;
    const int r2 = sws_scale_locked(c, s2, ss2, 0, srcSliceH, d2, ds2);
    c->luma_pass = false;
    return r2 < 0 ? r2 : r;
}

static int sws_scale_locked(FFHipSwsContext *c, const uint8_t *const src[], const int srcStride[], int srcSliceY, int srcSliceH,
                            uint8_t *const dst[], const int dstStride[])
{
    const FFHipSwsTables &t = c->t;
    const bool unscaled = c->unscaled_yuv2rgb;
    /*
     * Scaled contexts and source slices (sws_scale()'s legacy contract: slices arrive in order, libswscale/swscale.c:1085-1090):
     * the result must not depend on the slicing (libswscale/tests, tools/scale_slice_test.c).  The fused kernels work on whole
     * frames, so slices are collected in the context's device staging and the frame is produced when the last one arrives:
     * the calls before that return 0 output lines, the last one dstH.  Top-to-bottom order only.
     */
    const bool sliced = !unscaled && (srcSliceY != 0 || srcSliceH != t.srcH);
    if (sliced) {
        if (srcSliceY < 0 || srcSliceY + srcSliceH > t.srcH || (srcSliceY != 0 && srcSliceY != c->slice_next)) {
            ffhip_set_error("ffhip_sws_scale: slice [%d, +%d) is out of order (next expected line %d of %d)", srcSliceY, srcSliceH,
                            srcSliceY ? c->slice_next : 0, t.srcH);
            return FFHIP_EINVAL;
        }
        if (srcSliceY & 1) {
            ffhip_set_error("ffhip_sws_scale: slices of a vertically subsampled source must start on an even line");
            return FFHIP_EINVAL;
        }
    }
    if (unscaled && ((srcSliceY | srcSliceH) & 1)) { /* dst_slice_align = 2, swscale_unscaled.c:2430 */
        ffhip_set_error("ffhip_sws_scale: slices of the unscaled yuv2rgb converter must be 2-line aligned");
        return FFHIP_EINVAL;
    }
    const int srcRows = unscaled ? srcSliceH : t.srcH;
    PlaneDesc sp[4], dp[3];
    int ns = plane_list(t.srcFormat, t.srcW, srcRows, sp);
    const int nd = plane_list(t.dstFormat, t.dstW, unscaled ? srcSliceH : t.dstH, dp);
    if ((unscaled || fmt_rgb(t.dstFormat)) && t.dst_alpha_fill == 2) { /* the source's alpha plane rides along as a fourth plane of the luma's size */
        if (!src[3]) {
            ffhip_set_error("ffhip_sws_scale: the source's alpha plane (plane 3) is NULL");
            return FFHIP_EINVAL;
        }
        sp[ns++] = sp[0];
    }
    size_t off_s[4], off_d[3], total = 0;
    int pitch_s[4] = { 0, 0, 0, 0 }, pitch_d[4] = { 0, 0, 0, 0 };
    for (int i = 0; i < ns; i++) {
        pitch_s[i] = sp[i].wbytes % 64 ? (sp[i].wbytes + 255) & ~255 : sp[i].wbytes; /* (tight rows copy linearly from a tight host plane) */
        off_s[i] = total;
        total += (size_t)pitch_s[i] * sp[i].rows + 256;
    }
    for (int i = 0; i < nd; i++) {
        pitch_d[i] = dp[i].wbytes % 64 ? (dp[i].wbytes + 255) & ~255 : dp[i].wbytes;
        off_d[i] = total;
        total += (size_t)pitch_d[i] * dp[i].rows + 256;
    }
    if (total > c->stage_sz) {
        if (c->stage)
            (void)hipFree(c->stage);
        c->stage = nullptr;
        c->stage_sz = 0;
        if (hipMalloc(&c->stage, total) != hipSuccess) {
            ffhip_set_error("ffhip_sws_scale: staging hipMalloc(%zu) failed", total);
            return FFHIP_ENOMEM;
        }
        c->stage_sz = total;
    }
    uint8_t *base = (uint8_t *)c->stage;
    const void *dsrc[4] = { 0, 0, 0, 0 };
    void *ddst[4] = { 0, 0, 0, 0 };
    size_t fp[4] = { 0, 0, 0, 0 };
    if (c->rgb_in.bpp) {
        /* a packed RGB source: the slice's rows go up as they are and the converter pass writes rows [y, y + h) of the three 14-bit planes
         * (the converters work line by line, and the 4:2:2 / 4:4:4 lines have no vertical subsampling: a slice is a slice of every plane) */
        const int rows = sliced ? srcSliceH : srcRows, row0 = sliced ? srcSliceY : 0;
        const int wb = t.srcW * c->rgb_in.bpp, rp = (wb + 255) & ~255;
        const size_t need = (size_t)rp * rows + 256;
        if (need > c->rgb_in.stage_sz) {
            if (c->rgb_in.stage)
                (void)hipFree(c->rgb_in.stage);
            c->rgb_in.stage = nullptr;
            c->rgb_in.stage_sz = 0;
            if (hipMalloc(&c->rgb_in.stage, need) != hipSuccess) {
                ffhip_set_error("ffhip_sws_scale: staging hipMalloc(%zu) failed", need);
                return FFHIP_ENOMEM;
            }
            c->rgb_in.stage_sz = need;
        }
        HIP_TRY(copy2d(c->rgb_in.stage, rp, src[0], srcStride[0], wb, rows, hipMemcpyHostToDevice));
        uint8_t *const pp[3] = { base + off_s[0] + (size_t)row0 * pitch_s[0], base + off_s[1] + (size_t)row0 * pitch_s[1],
                                 base + off_s[2] + (size_t)row0 * pitch_s[2] };
        /* (the target's staging planes lie behind the source's: base + off_d[]) */
        const bool yd = c->rgb_in.y_direct;
        const int r = rgb_in_launch(c, 1, static_cast<const uint8_t *>(c->rgb_in.stage), rp, 0, rows, pp, pitch_s, fp, 0,
                                    yd ? base + off_d[0] + (size_t)row0 * pitch_d[0] : nullptr, yd ? pitch_d[0] : 0, 0);
        if (r < 0)
            return r;
        c->rgb_in_luma_done = yd;
        for (int i = 0; i < 3; i++)
            dsrc[i] = base + off_s[i];
    } else
    for (int i = 0; i < ns; i++) {
        /* a slice brings luma rows [y, y + h) and chroma rows [y >> 1, (y + h + 1) >> 1) (swscale.c:280-283) */
        const int vs = fmt_vsub(c->t.srcFormat);
        const bool cpl = i == 1 || i == 2; /* (plane 3, when there is one, is the alpha plane: the luma's geometry) */
        const int row0 = !sliced ? 0 : cpl ? srcSliceY >> vs : srcSliceY;
        const int rows = !sliced ? sp[i].rows : cpl ? (-((-(srcSliceY + srcSliceH)) >> vs)) - (srcSliceY >> vs) : srcSliceH;
        HIP_TRY(copy2d(base + off_s[i] + (size_t)row0 * pitch_s[i], pitch_s[i], src[i], srcStride[i], sp[i].wbytes, rows,
                       hipMemcpyHostToDevice));
        dsrc[i] = base + off_s[i];
    }
    if (sliced) {
        c->slice_next = srcSliceY + srcSliceH;
        if (c->slice_next < t.srcH)
            return 0; /* nothing can be written before the frame is complete */
        c->slice_next = 0;
    }
    for (int i = 0; i < nd; i++)
        ddst[i] = base + off_d[i];
    /* an odd RGB trailing column / rows a converter leaves untouched must survive the round trip */
    if (unscaled && (t.dstW & 1))
        HIP_TRY(copy2d(base + off_d[0], pitch_d[0], dst[0] + (ptrdiff_t)srcSliceY * dstStride[0], dstStride[0],
                       dp[0].wbytes, dp[0].rows, hipMemcpyHostToDevice));
    int r;
    if (unscaled) {
        r = unscaled_launch(c, 1, srcSliceH, dsrc, pitch_s, fp, ddst, pitch_d, fp, 0);
    } else {
        r = scale_batch_dev(c, 1, dsrc, pitch_s, fp, ddst, pitch_d, fp, 0);
    }
    c->rgb_in_luma_done = false;
    if (r < 0)
        return r;
    HIP_TRY(hipStreamSynchronize(0));
    for (int i = 0; i < nd; i++) {
        uint8_t *hd = dst[i] + (unscaled ? (ptrdiff_t)srcSliceY * dstStride[i] : 0);
        HIP_TRY(copy2d(hd, dstStride[i], base + off_d[i], pitch_d[i], dp[i].wbytes, dp[i].rows, hipMemcpyDeviceToHost));
    }
    if (t.dst_alpha_fill == 1 && dst[3]) /* the alpha plane of a target whose source has none: opaque (swscale.c:536-553) */
        for (int y = 0; y < t.dstH; y++)
            memset(dst[3] + (ptrdiff_t)y * dstStride[3], 255, (size_t)t.dstW);
    return unscaled ? srcSliceH : t.dstH;
}

/* ---- per-line parity faces ------------------------------------------------------------------- */
extern "C" int ffhip_sws_hscale8to15_dev(int16_t *dst, int dstW, ptrdiff_t dstPitch, const uint8_t *src,
                                         ptrdiff_t srcPitch, int nlines, const int16_t *filter, const int32_t *filterPos,
                                         int filterSize, void *stream)
{
    if (!dst || !src || !filter || !filterPos || filterSize <= 0)
        return FFHIP_EINVAL;
    return ffhip_launch_hscale8to15(dst, dstW, dstPitch, src, srcPitch, nlines, filter, filterPos, filterSize,
                                    (hipStream_t)stream);
}

extern "C" int ffhip_sws_yuv2planeX8_dev(const int16_t *filter, int filterSize, const int16_t *src, ptrdiff_t srcPitch,
                                         uint8_t *dest, int dstW, const uint8_t *dither8, int offset, void *stream)
{
    if (!src || !dest || !dither8 || filterSize <= 0 || (filterSize > 1 && !filter))
        return FFHIP_EINVAL;
    return ffhip_launch_yuv2planeX8(filter, filterSize, src, srcPitch, dest, dstW, dither8, offset, (hipStream_t)stream);
}

/* ---- signature-exact per-line host faces (swscale_internal.h:128-266, 648-653; installed by ff_sws_init_swscale_<arch>(),
 * swscale.c:697-714): what tests/checkasm/sw_scale.c exercises.  One call = one launch through the scratch arena; a call that
 * cannot run on the device is answered by the displaced C function (kernels/shim_arena.h). -------------------------------- */
static FFHipSwsLineContext g_fb_line;

static bool hscale_line_gpu(int16_t *dst, int dstW, const uint8_t *src, const int16_t *filter, const int32_t *filterPos, int filterSize)
{
    if (dstW <= 0 || filterSize <= 0 || filterSize > 4096)
        return false;
    int span = 0; /* source bytes the line reads: the windows are in bounds of the (padded) line by construction */
    for (int i = 0; i < dstW; i++)
        if (filterPos[i] < 0)
            return false;
        else if (filterPos[i] + filterSize > span)
            span = filterPos[i] + filterSize;
    const size_t bs = ((size_t)span + 63) & ~(size_t)63, bf = ((size_t)dstW * filterSize * 2 + 63) & ~(size_t)63;
    const size_t bp = ((size_t)dstW * 4 + 63) & ~(size_t)63, bd = ((size_t)dstW * 2 + 63) & ~(size_t)63;
    Arena A(bs + bf + bp + bd);
    if (!A.ok)
        return false;
    uint8_t *dsrc = A.buf, *df = dsrc + bs, *dp = df + bf, *dd = dp + bp;
    if (hipMemcpy(dsrc, src, span, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(df, filter, (size_t)dstW * filterSize * 2, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(dp, filterPos, (size_t)dstW * 4, hipMemcpyHostToDevice) != hipSuccess)
        return false;
    if (ffhip_launch_hscale8to15((int16_t *)dd, dstW, 0, dsrc, 0, 1, (const int16_t *)df, (const int32_t *)dp, filterSize, 0) < 0 || !A.down())
        return false;
    memcpy(dst, A.host(dd), (size_t)dstW * 2);
    return true;
}
static void s_hyscale(void *c, int16_t *dst, int dstW, const uint8_t *src, const int16_t *filter, const int32_t *filterPos, int filterSize)
{ if (!hscale_line_gpu(dst, dstW, src, filter, filterPos, filterSize)) SHIM_FB(g_fb_line, hyScale, c, dst, dstW, src, filter, filterPos, filterSize); }
static void s_hcscale(void *c, int16_t *dst, int dstW, const uint8_t *src, const int16_t *filter, const int32_t *filterPos, int filterSize)
{ if (!hscale_line_gpu(dst, dstW, src, filter, filterPos, filterSize)) SHIM_FB(g_fb_line, hcScale, c, dst, dstW, src, filter, filterPos, filterSize); }

/* nsrc int16 lines of n samples each, packed at a pitch of `pitch` bytes from `at` */
static bool lines_up(uint8_t *at, size_t pitch, const int16_t *const *src, int nsrc, int n)
{
    for (int j = 0; j < nsrc; j++)
        if (!src[j] || hipMemcpy(at + j * pitch, src[j], (size_t)n * 2, hipMemcpyHostToDevice) != hipSuccess)
            return false;
    return true;
}

static bool planex_line_gpu(const int16_t *filter, int filterSize, const int16_t *const *src, uint8_t *dest, int dstW, const uint8_t *dither,
                            int offset)
{
    if (dstW <= 0 || filterSize <= 0 || filterSize > 256 || !dither)
        return false;
    const size_t pitch = ((size_t)dstW * 2 + 63) & ~(size_t)63, bd = ((size_t)dstW + 63) & ~(size_t)63;
    Arena A(64 + 512 + pitch * filterSize + bd);
    if (!A.ok)
        return false;
    uint8_t *ddi = A.buf, *df = ddi + 64, *dl = df + 512, *dd = dl + pitch * filterSize;
    if (hipMemcpy(ddi, dither, 8, hipMemcpyHostToDevice) != hipSuccess ||
        (filter && hipMemcpy(df, filter, (size_t)filterSize * 2, hipMemcpyHostToDevice) != hipSuccess) || !lines_up(dl, pitch, src, filterSize, dstW))
        return false;
    if (ffhip_launch_yuv2planeX8((const int16_t *)df, filterSize, (const int16_t *)dl, (ptrdiff_t)pitch, dd, dstW, ddi, offset, 0) < 0 || !A.down())
        return false;
    memcpy(dest, A.host(dd), dstW);
    return true;
}
static void s_yuv2plane1(const int16_t *src, uint8_t *dest, int dstW, const uint8_t *dither, int offset)
{ if (!planex_line_gpu(nullptr, 1, &src, dest, dstW, dither, offset)) SHIM_FB(g_fb_line, yuv2plane1, src, dest, dstW, dither, offset); }
static void s_yuv2planex(const int16_t *filter, int filterSize, const int16_t **src, uint8_t *dest, int dstW, const uint8_t *dither, int offset)
{
    /* a 1-tap call is NOT yuv2plane1 (its arithmetic is (src * filter + dither << 12) >> 19): the kernel's fs == 1 form is plane1's,
     * so single taps take the general loop through a second, zero tap */
    bool ok;
    if (filterSize == 1) {
        const int16_t f2[2] = { filter[0], 0 };
        const int16_t *s2[2] = { src[0], src[0] };
        ok = planex_line_gpu(f2, 2, s2, dest, dstW, dither, offset);
    } else {
        ok = filter && planex_line_gpu(filter, filterSize, src, dest, dstW, dither, offset);
    }
    if (!ok)
        SHIM_FB(g_fb_line, yuv2planeX, filter, filterSize, src, dest, dstW, dither, offset);
}

static bool nv12cx_line_gpu(int dstFormat, const uint8_t *chrDither, const int16_t *chrFilter, int chrFilterSize, const int16_t *const *chrUSrc,
                            const int16_t *const *chrVSrc, uint8_t *dest, int chrDstW)
{
    if (chrDstW <= 0 || chrFilterSize <= 0 || chrFilterSize > 256 || !fmt_nv(dstFormat) || !chrDither || !chrFilter)
        return false;
    const size_t pitch = ((size_t)chrDstW * 2 + 63) & ~(size_t)63, bd = ((size_t)chrDstW * 2 + 63) & ~(size_t)63;
    Arena A(64 + 512 + 2 * pitch * chrFilterSize + bd);
    if (!A.ok)
        return false;
    uint8_t *ddi = A.buf, *df = ddi + 64, *du = df + 512, *dv = du + pitch * chrFilterSize, *dd = dv + pitch * chrFilterSize;
    if (hipMemcpy(ddi, chrDither, 8, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(df, chrFilter, (size_t)chrFilterSize * 2, hipMemcpyHostToDevice) != hipSuccess ||
        !lines_up(du, pitch, chrUSrc, chrFilterSize, chrDstW) || !lines_up(dv, pitch, chrVSrc, chrFilterSize, chrDstW))
        return false;
    if (ffhip_launch_yuv2nv12cX(dstFormat == FFHIP_PIX_FMT_NV21, ddi, (const int16_t *)df, chrFilterSize, (const int16_t *)du, (const int16_t *)dv,
                                (ptrdiff_t)pitch, dd, chrDstW, 0) < 0 || !A.down())
        return false;
    memcpy(dest, A.host(dd), (size_t)chrDstW * 2);
    return true;
}
static void s_yuv2nv12cx(int dstFormat, const uint8_t *chrDither, const int16_t *chrFilter, int chrFilterSize, const int16_t **chrUSrc,
                         const int16_t **chrVSrc, uint8_t *dest, int dstW)
{
    if (!nv12cx_line_gpu(dstFormat, chrDither, chrFilter, chrFilterSize, chrUSrc, chrVSrc, dest, dstW))
        SHIM_FB(g_fb_line, yuv2nv12cX, dstFormat, chrDither, chrFilter, chrFilterSize, chrUSrc, chrVSrc, dest, dstW);
}

extern "C" int ff_sws_init_swscale_hip(FFHipSwsLineContext *lc, int srcFormat, int dstFormat)
{
    if (!lc || !fmt_yuv(srcFormat) || !(fmt_yuv(dstFormat) || fmt_rgb(dstFormat)))
        return FFHIP_EINVAL;
    if (!ffhip_have_device())
        return FFHIP_ENOSYS;
    FFHipSwsLineContext o = *lc;
    o.hyScale = s_hyscale; /* 8-bit sources: hScale8To15_c (swscale.c:608-611) */
    o.hcScale = s_hcscale;
    if (fmt_yuv(dstFormat)) { /* 8-bit planar / NV targets: yuv2plane1_8_c, yuv2planeX_8_c, yuv2nv12cX_c (output.c:3261-3275) */
        o.yuv2plane1 = s_yuv2plane1;
        o.yuv2planeX = s_yuv2planex;
        if (fmt_nv(dstFormat))
            o.yuv2nv12cX = s_yuv2nv12cx;
    }
    fb_snapshot(g_fb_line, *lc, o);
    *lc = o;
    return 0;
}

/* yuv2packed1 / yuv2packed2 / yuv2packedX (swscale_internal.h:201-266) for the packed RGB targets: these read the context's yuv2rgb
 * tables, so their first argument is OUR context where the reference passes its SwsInternal, and they RETURN their status —
 * the FFmpeg-side wrapper calls the C function it displaced when that is negative. */
static int packed_line(FFHipSwsContext *c, int mode, const int16_t *lf, const int16_t *const *lum, int lfs, const int16_t *cf,
                       const int16_t *const *cu, const int16_t *const *cv, int cfs, uint8_t *dest, int dstW, int yalpha, int uvalpha)
{
    if (!c || !dest || !lum || !cu || !cv || dstW < 2 || (dstW & 1) || lfs < 1 || cfs < 1 || lfs > 256 || cfs > 256 || !fmt_rgb(c->t.dstFormat))
        return FFHIP_EINVAL;
    FFHipDeviceGuard dg(c->device);
    const int lay = rgb_layout(c->t.dstFormat), bpp = lay < 2 ? 3 : 4, cw = dstW >> 1;
    const size_t pitch = ((size_t)dstW * 2 + 63) & ~(size_t)63, bd = ((size_t)dstW * bpp + 63) & ~(size_t)63;
    Arena A(1024 + pitch * (lfs + 2 * cfs) + bd);
    if (!A.ok)
        return FFHIP_EIO;
    uint8_t *dlf = A.buf, *dcf = dlf + 512, *dl = dcf + 512, *du = dl + pitch * lfs, *dv = du + pitch * cfs, *dd = dv + pitch * cfs;
    if ((lf && hipMemcpy(dlf, lf, (size_t)lfs * 2, hipMemcpyHostToDevice) != hipSuccess) ||
        (cf && hipMemcpy(dcf, cf, (size_t)cfs * 2, hipMemcpyHostToDevice) != hipSuccess) || !lines_up(dl, pitch, lum, lfs, dstW) ||
        !lines_up(du, pitch, cu, cfs, cw) || !lines_up(dv, pitch, cv, cfs, cw))
        return FFHIP_EIO;
    if (ffhip_launch_yuv2packed_line(mode, (const int16_t *)dlf, (const int16_t *)dl, lfs, (const int16_t *)dcf, (const int16_t *)du,
                                     (const int16_t *)dv, cfs, (ptrdiff_t)pitch, yalpha, uvalpha, dd, dstW, lay, c->k, 0) < 0 || !A.down())
        return FFHIP_EIO;
    memcpy(dest, A.host(dd), (size_t)dstW * bpp);
    return 0;
}

extern "C" int ffhip_sws_yuv2packedX(FFHipSwsContext *c, const int16_t *lumFilter, const int16_t **lumSrc, int lumFilterSize,
                                     const int16_t *chrFilter, const int16_t **chrUSrc, const int16_t **chrVSrc, int chrFilterSize,
                                     const int16_t **alpSrc, uint8_t *dest, int dstW, int y)
{
    (void)alpSrc; (void)y; /* no alpha plane among the supported sources; y only matters to the dithered 8-bit-and-below targets */
    if (!lumFilter || !chrFilter)
        return FFHIP_EINVAL;
    return packed_line(c, 0, lumFilter, lumSrc, lumFilterSize, chrFilter, chrUSrc, chrVSrc, chrFilterSize, dest, dstW, 0, 0);
}

extern "C" int ffhip_sws_yuv2packed2(FFHipSwsContext *c, const int16_t *lumSrc[2], const int16_t *chrUSrc[2], const int16_t *chrVSrc[2],
                                     const int16_t *alpSrc[2], uint8_t *dest, int dstW, int yalpha, int uvalpha, int y)
{
    (void)alpSrc; (void)y;
    if ((unsigned)yalpha > 4096u || (unsigned)uvalpha > 4096u)
        return FFHIP_EINVAL;
    return packed_line(c, 1, nullptr, lumSrc, 2, nullptr, chrUSrc, chrVSrc, 2, dest, dstW, yalpha, uvalpha);
}

extern "C" int ffhip_sws_yuv2packed1(FFHipSwsContext *c, const int16_t *lumSrc, const int16_t *chrUSrc[2], const int16_t *chrVSrc[2],
                                     const int16_t *alpSrc, uint8_t *dest, int dstW, int uvalpha, int y)
{
    (void)alpSrc; (void)y;
    if ((unsigned)uvalpha > 4096u || !lumSrc)
        return FFHIP_EINVAL;
    const int16_t *const l1[1] = { lumSrc };
    return packed_line(c, 2, nullptr, l1, 1, nullptr, chrUSrc, chrVSrc, uvalpha ? 2 : 1, dest, dstW, 0, uvalpha);
}
