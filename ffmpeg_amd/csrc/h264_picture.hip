/*
 * h264_picture.hip — caller-side batching for the H.264 macroblock loop.
 *
 * The reference reconstructs a picture by calling the dsp pointers once per block from hl_decode_mb()
 * (libavcodec/h264_mb_template.c:41-270: mc_part_std / mc_part_weighted -> qpel + chroma + weight tables, h264_mb.c:206-420;
 * hl_decode_mb_predict_luma / idct_add, h264_mb.c:612-800) and, a row behind, ff_h264_filter_mb() (h264_loopfilter.c:716).
 * On a GPU one call per block is ~10^4 times slower than C, so the decoder RECORDS those calls here while it parses the
 * picture — the same operands it would have passed — and flush() runs them as a handful of launches in the one order that
 * preserves the reference's data flow:
 *
 *     per plane:  MC put -> dst | MC put -> bi-pred scratch | MC avg -> dst | weight / biweight | IDCT + add |
 *                 intra macroblocks (all planes, reconstruction wavefront) | deblock (frame order)
 *
 * Blocks of one stage are disjoint, stages depend on each other only in that order.  Intra macroblocks predict from their
 * neighbours' reconstructed, unfiltered samples — inter neighbours are complete after the residual stage, intra ones chain
 * through k_h264_intra_frame's wavefront (kernels/h264_intra.hip); deblocking is the decoder-order wavefront behind it.
 * Records travel in ONE host-to-device copy per picture from a pinned buffer.  Entropy decoding stays on the CPU.
 */
#include <algorithm>
#include <new>
#include <string.h>
#include <string>
#include <thread>
#include <vector>

#include "kernels/common.h"
#include "kernels/h264_kernels.h"

namespace {
enum { ST_PUT = 0, ST_TMP = 1, ST_AVG = 2 };

struct Section { size_t off = 0; int n = 0; };

template <typename T>
size_t place(size_t &total, const std::vector<T> &v, Section &s)
{
    total = (total + 15) & ~(size_t)15;
    s.off = total;
    s.n = (int)v.size();
    total += v.size() * sizeof(T);
    return s.off;
}
} // namespace

struct FFHipH264Picture {
    int last_status = 0; /* what the picture's last flush came to: 0, or the FFHIP_E* that left its planes incomplete (ffhip_h264_picture_status) */
    int device = 0; /* staging and scratch planes live on this device; flush() makes it current for its duration */
    int mb_w = 0, mb_h = 0;
    int bd = 8;     /* sample depth: above 8 the planes hold uint16_t, coefficient blocks int32_t (dctcoef, bit_depth_template.c:39-50) */
    int cfmt = 1;   /* sps->chroma_format_idc: 1 (4:2:0), 2 (4:2:2: 8 x 16 chroma) or 3 (4:4:4: Cb and Cr through the LUMA members, hl_decode_mb_444) */
    std::vector<FFHipQpelBlock> qpel[3][3];       /* luma-table MC: plane (4:2:0: plane 0 only) x stage */
    std::vector<FFHipChromaBlock> cmc[2][3];      /* chroma MC: plane (Cb, Cr) x stage     */
    std::vector<FFHipWeightBlock> wt[3];          /* weight / biweight per plane           */
    std::vector<int32_t> idct_off[3][6];          /* per plane x FFHIP_H264_IDCT* kind (4, 5: add_pixels4 / 8_clear, the lossless bypass) */
    std::vector<int16_t> idct_coef[3][6];
    /* intra macroblocks in recording order and their packed coefficient runs.  4:2:0: [0] holds whole macroblocks; 4:4:4: [pl] holds
     * plane pl's share of every intra macroblock as a luma-only record (the wavefront runs once per plane, side by side) */
    std::vector<FFHipH264IntraMB> intra[3];
    std::vector<int16_t> intra_coef[3];
    std::vector<FFHipH264IntraMB> intra_sorted[3]; /* flush(): by (mb_y, mb_x)               */
    std::vector<int32_t> intra_rows[3];            /* flush(): mb_h + 1 row starts           */
    /* 4:2:2: [0] above holds the macroblocks' luma as luma-only records, the chroma planes have records and a wavefront of their own */
    std::vector<FFHipH264IntraC422> intra_c422, intra_c422_sorted;
    std::vector<int16_t> intra_c422_coef;
    std::vector<int32_t> intra_c422_rows;
    std::vector<FFHipH264Edge> edges[3];          /* whole-picture edge arrays, zero = skip */
    bool any_edge[3] = { false, false, false };
    void *pinned = nullptr, *dev = nullptr;
    size_t pinned_sz = 0, dev_sz = 0;
    uint8_t *tmp[3] = { nullptr, nullptr, nullptr }; /* bi-prediction scratch planes (sl->bipred_scratchpad) */
    size_t tmp_sz[3] = { 0, 0, 0 };
    hipEvent_t copied = nullptr;
    bool copy_pending = false;
    /* the chroma planes' deblocking wavefront runs beside the luma one on a second stream, forked and joined with events */
    hipStream_t aux = nullptr;
    hipEvent_t fork = nullptr, join = nullptr;
    /* geometry of plane pl in samples / rows, and the edge records a macroblock has in it */
    int plane_w(int pl) const { return (pl && cfmt != 3 ? 8 : 16) * mb_w; }
    int plane_h(int pl) const { return (pl && cfmt == 1 ? 8 : 16) * mb_h; }
    int edges_per_mb(int pl) const { return !pl || cfmt == 3 ? 8 : cfmt == 2 ? 6 : 4; } /* 4:2:2 chroma: x = 0, 4, then y = 0, 4, 8, 12 */
    int intra_sets() const { return cfmt == 3 ? 3 : 1; }
};

extern "C" void ffhip_h264_picture_free(FFHipH264Picture **pp)
{
    FFHipDeviceGuard dg(pp && *pp ? (*pp)->device : -1);
    if (!pp || !*pp)
        return;
    FFHipH264Picture *p = *pp;
    if (p->copy_pending)
        (void)hipEventSynchronize(p->copied);
    if (p->copied)
        (void)hipEventDestroy(p->copied);
    if (p->aux) {
        (void)hipStreamSynchronize(p->aux);
        (void)hipStreamDestroy(p->aux);
    }
    if (p->fork)
        (void)hipEventDestroy(p->fork);
    if (p->join)
        (void)hipEventDestroy(p->join);
    if (p->pinned)
        (void)hipHostFree(p->pinned);
    if (p->dev)
        (void)hipFree(p->dev);
    for (int i = 0; i < 3; i++)
        if (p->tmp[i])
            (void)hipFree(p->tmp[i]);
    delete p;
    *pp = nullptr;
}

extern "C" void ffhip_h264_picture_begin(FFHipH264Picture *p)
{
    if (!p)
        return;
    for (int s = 0; s < 3; s++) {
        for (int pl = 0; pl < 3; pl++)
            p->qpel[pl][s].clear();
        p->cmc[0][s].clear();
        p->cmc[1][s].clear();
    }
    p->intra_c422.clear();
    p->intra_c422_coef.clear();
    for (int pl = 0; pl < 3; pl++) {
        p->intra[pl].clear();
        p->intra_coef[pl].clear();
        p->wt[pl].clear();
        for (int k = 0; k < 6; k++) {
            p->idct_off[pl][k].clear();
            p->idct_coef[pl][k].clear();
        }
        if (p->any_edge[pl])
            memset(p->edges[pl].data(), 0, p->edges[pl].size() * sizeof(FFHipH264Edge));
        p->any_edge[pl] = false;
    }
}

extern "C" int ffhip_h264_picture_create(FFHipH264Picture **pp, int mb_w, int mb_h)
{
    return ffhip_h264_picture_create_hbd(pp, mb_w, mb_h, 8);
}

extern "C" int ffhip_h264_picture_create_hbd(FFHipH264Picture **pp, int mb_w, int mb_h, int bit_depth)
{
    return ffhip_h264_picture_create_fmt(pp, mb_w, mb_h, bit_depth, 1);
}

extern "C" int ffhip_h264_picture_create_fmt(FFHipH264Picture **pp, int mb_w, int mb_h, int bit_depth, int chroma_format_idc)
{
    if (!pp || mb_w <= 0 || mb_h <= 0)
        return FFHIP_EINVAL;
    *pp = nullptr;
    if (chroma_format_idc < 0 || chroma_format_idc > 3) {
        ffhip_set_error("ffhip_h264_picture_create_fmt: chroma_format_idc %d (0 .. 3)", chroma_format_idc);
        return FFHIP_EINVAL;
    }
    /* monochrome: the decoder reconstructs it as a 4:2:0 picture whose chroma planes come out mid-grey through the ordinary members
     * (DC_128 chroma prediction, chroma MC from mid-grey references, no chroma residual, no chroma edges: h264_mb_template.c:112-148,
     * h264_cavlc.c decode_chroma, h264_loopfilter.c:726) — the same object */
    if (chroma_format_idc == 0)
        chroma_format_idc = 1;
    if (bit_depth != 8 && bit_depth != 9 && bit_depth != 10 && bit_depth != 12 && bit_depth != 14) {
        ffhip_set_error("ffhip_h264_picture_create_hbd: bit depth %d (8, 9, 10, 12 and 14 are the depths H.264 defines)", bit_depth);
        return FFHIP_EINVAL;
    }
    /* Recording is host work and needs no device (the decoder's macroblock loop can be exercised, and the lists inspected through
     * ffhip_h264_picture_lists(), on any machine); an object made without one refuses flush() with FFHIP_ENOSYS. */
    const bool have_dev = ffhip_have_device() != 0;
    FFHipH264Picture *p = new (std::nothrow) FFHipH264Picture();
    if (p)
        p->device = have_dev ? ffhip_current_device() : -1;
    if (!p)
        return FFHIP_ENOMEM;
    p->mb_w = mb_w;
    p->mb_h = mb_h;
    p->bd = bit_depth;
    p->cfmt = chroma_format_idc;
    const size_t nmb = (size_t)mb_w * mb_h;
    for (int pl = 0; pl < 3; pl++)
        p->edges[pl].assign(nmb * (size_t)p->edges_per_mb(pl), FFHipH264Edge());
    if (!have_dev) {
        *pp = p;
        return 0;
    }
    if (hipEventCreateWithFlags(&p->copied, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&p->fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&p->join, hipEventDisableTiming) != hipSuccess ||
        hipStreamCreateWithFlags(&p->aux, hipStreamNonBlocking) != hipSuccess) {
        ffhip_h264_picture_free(&p);
        return FFHIP_ENOMEM;
    }
    *pp = p;
    return 0;
}

/* ---- recording: cheap appends on the host ----------------------------------------------------------- */
extern "C" int ffhip_h264_picture_mc_luma(FFHipH264Picture *p, int stage, const FFHipQpelBlock *blk)
{
    return ffhip_h264_picture_mc_luma_plane(p, 0, stage, blk);
}

extern "C" int ffhip_h264_picture_mc_luma_plane(FFHipH264Picture *p, int plane, int stage, const FFHipQpelBlock *blk)
{
    if (!p || !blk || plane < 0 || plane > 2 || stage < 0 || stage > 2)
        return FFHIP_EINVAL;
    if (plane && p->cfmt != 3) {
        ffhip_set_error("ffhip_h264_picture_mc_luma_plane: plane %d of a 4:2:0 picture takes chroma MC records", plane);
        return FFHIP_EINVAL;
    }
    FFHipQpelBlock b = *blk;
    b.avg = stage == ST_AVG;
    p->qpel[plane][stage].push_back(b);
    return 0;
}

extern "C" int ffhip_h264_picture_mc_chroma(FFHipH264Picture *p, int plane, int stage, const FFHipChromaBlock *blk)
{
    if (!p || !blk || plane < 1 || plane > 2 || stage < 0 || stage > 2)
        return FFHIP_EINVAL;
    if (p->cfmt == 3) {
        ffhip_set_error("ffhip_h264_picture_mc_chroma: the chroma planes of a 4:4:4 picture are predicted by the luma tables (mc_luma_plane)");
        return FFHIP_EINVAL;
    }
    FFHipChromaBlock b = *blk;
    b.avg = stage == ST_AVG;
    p->cmc[plane - 1][stage].push_back(b);
    return 0;
}

extern "C" int ffhip_h264_picture_weight(FFHipH264Picture *p, int plane, const FFHipWeightBlock *blk)
{
    if (!p || !blk || plane < 0 || plane > 2)
        return FFHIP_EINVAL;
    p->wt[plane].push_back(*blk);
    return 0;
}

extern "C" int ffhip_h264_picture_idct_add(FFHipH264Picture *p, int plane, int kind, int32_t dst_offset, int16_t *block)
{
    if (!p || !block || plane < 0 || plane > 2 || kind < FFHIP_H264_IDCT4 || kind > FFHIP_H264_ADD_PIXELS8_CLEAR)
        return FFHIP_EINVAL;
    /* int16 units: above 8 bits a coefficient is an int32 (dctcoef), the caller's `block` is the decoder's sl->mb as it stands */
    const int wide = p->bd > 8 ? 2 : 1;
    const int ncoef = ((kind == FFHIP_H264_IDCT8 || kind == FFHIP_H264_IDCT8_DC || kind == FFHIP_H264_ADD_PIXELS8_CLEAR) ? 64 : 16) * wide;
    p->idct_off[plane][kind].push_back(dst_offset);
    std::vector<int16_t> &c = p->idct_coef[plane][kind];
    c.insert(c.end(), block, block + ncoef);
    /* the side effect of the dsp function the decoder relies on: coefficients are consumed (h264idct_template.c:66,142,
     * and block[0] = 0 for the dc forms) */
    if (kind == FFHIP_H264_IDCT4_DC || kind == FFHIP_H264_IDCT8_DC)
        block[0] = block[wide - 1] = 0;
    else
        memset(block, 0, sizeof(int16_t) * ncoef);
    return 0;
}

/* scan8[] (libavcodec/h264_parse.h:40-57): luma block i, chroma block k of plane pl (1 Cb, 2 Cr), and the three DC entries */
static int scan8_luma(int i) { return 4 + (i & 1) + ((i >> 2) & 1) * 2 + (1 + ((i >> 1) & 1) + ((i >> 3) & 1) * 2) * 8; }
static int scan8_chroma(int pl, int k) { return 4 + (k & 1) + (5 * pl + 1 + (k >> 1)) * 8; }

/* The macroblock-level members hl_decode_mb() calls for an inter macroblock's residual (h264_mb.c:780-797, h264_mb_template.c:254-257),
 * expanded into the per-block records exactly as the dsp functions dispatch them (h264idct_template.c:176-228): which 0 idct_add16,
 * 1 idct8_add4 (plane 0..2, one destination), 3 idct_add8 (dst_offset[0] Cb, [1] Cr; block = sl->mb, blocks 16..19 and 32..35).
 * block_offset is the decoder's h->block_offset (bytes), nnzc the pointer the member gets.  Consumes `block` as the functions do. */
extern "C" int ffhip_h264_picture_idct_mb(FFHipH264Picture *p, int which, int plane, const int32_t dst_offset[2], const int *block_offset,
                                          int16_t *block, const uint8_t *nnzc)
{
    if (!p || !dst_offset || !block_offset || !block || !nnzc || plane < 0 || plane > 2)
        return FFHIP_EINVAL;
    const int wide = p->bd > 8 ? 2 : 1; /* int16 per coefficient: block + i*16*sizeof(pixel) in the reference */
    auto coef0 = [&](int i) { return wide == 2 ? reinterpret_cast<const int32_t *>(block)[i * 16] : (int32_t)block[i * 16]; };
    int r = 0;
    if (which == 0) {
        for (int i = 0; i < 16 && r >= 0; i++) {
            const int nnz = nnzc[scan8_luma(i)];
            if (nnz)
                r = ffhip_h264_picture_idct_add(p, plane, nnz == 1 && coef0(i) ? FFHIP_H264_IDCT4_DC : FFHIP_H264_IDCT4, dst_offset[0] + block_offset[i],
                                                block + i * 16 * wide);
        }
    } else if (which == 1) {
        for (int i = 0; i < 16 && r >= 0; i += 4) {
            const int nnz = nnzc[scan8_luma(i)];
            if (nnz)
                r = ffhip_h264_picture_idct_add(p, plane, nnz == 1 && coef0(i) ? FFHIP_H264_IDCT8_DC : FFHIP_H264_IDCT8, dst_offset[0] + block_offset[i],
                                                block + i * 16 * wide);
        }
    } else if (which == 3 && p->cfmt == 2) {
        /* ff_h264_idct_add8_422 (h264idct_template.c:230-252): eight blocks per plane — coefficients 16 j + k, cache entries and offsets
         * 16 j + k for the upper four, 16 j + k + 4 for the lower four (scan8_chroma(j, k) runs on down the cache for k = 4..7) */
        for (int j = 1; j < 3 && r >= 0; j++)
            for (int k = 0; k < 8 && r >= 0; k++) {
                const int i = j * 16 + k, at = k < 4 ? i : i + 4;
                if (nnzc[scan8_chroma(j, k)])
                    r = ffhip_h264_picture_idct_add(p, j, FFHIP_H264_IDCT4, dst_offset[j - 1] + block_offset[at], block + i * 16 * wide);
                else if (coef0(i))
                    r = ffhip_h264_picture_idct_add(p, j, FFHIP_H264_IDCT4_DC, dst_offset[j - 1] + block_offset[at], block + i * 16 * wide);
            }
    } else if (which == 3 && p->cfmt == 1) {
        for (int j = 1; j < 3 && r >= 0; j++)
            for (int i = j * 16; i < j * 16 + 4 && r >= 0; i++) {
                if (nnzc[scan8_chroma(j, i - j * 16)])
                    r = ffhip_h264_picture_idct_add(p, j, FFHIP_H264_IDCT4, dst_offset[j - 1] + block_offset[i], block + i * 16 * wide);
                else if (coef0(i))
                    r = ffhip_h264_picture_idct_add(p, j, FFHIP_H264_IDCT4_DC, dst_offset[j - 1] + block_offset[i], block + i * 16 * wide);
            }
    } else {
        ffhip_set_error("ffhip_h264_picture_idct_mb: which = %d (0 idct_add16, 1 idct8_add4, 3 idct_add8 of a 4:2:0 picture)", which);
        return FFHIP_EINVAL;
    }
    return r;
}

extern "C" int ffhip_h264_picture_deblock_mb(FFHipH264Picture *p, int plane, int mb_x, int mb_y, const FFHipH264Edge *e)
{
    if (!p || !e || plane < 0 || plane > 2 || mb_x < 0 || mb_x >= p->mb_w || mb_y < 0 || mb_y >= p->mb_h)
        return FFHIP_EINVAL;
    const int per = p->edges_per_mb(plane);
    memcpy(&p->edges[plane][((size_t)mb_y * p->mb_w + mb_x) * per], e, sizeof(FFHipH264Edge) * per);
    p->any_edge[plane] = true;
    return 0;
}

/* CF = dctcoef of the depth (int16_t at 8 bits, int32_t above); runs, R.coef, *ncoefs and cap count int16 entries at every depth */
/* luma_only (a plane of a 4:4:4 macroblock): the record carries sixteen luma-type blocks and nothing else; an I_PCM run holds the plane's
 * 256 samples (the run keeps its 4:2:0 length, the rest zero: the kernel's run fetch is sized by the macroblock type) */
template <typename CF>
static int intra_pack(int bd, FFHipH264IntraMB *rec, const uint8_t *nnzc, int16_t *mb_, const int16_t *mb_luma_dc_, const uint8_t *pcm, int16_t *coefs,
                      int32_t *ncoefs, int32_t cap, bool luma_only = false)
{
    constexpr int W = (int)(sizeof(CF) / sizeof(int16_t));
    if (!rec || !coefs || !ncoefs || rec->type > FFHIP_H264_INTRA_PCM || *ncoefs < 0)
        return FFHIP_EINVAL;
    FFHipH264IntraMB &R = *rec;
    CF *mb = reinterpret_cast<CF *>(mb_);
    const CF *mb_luma_dc = reinterpret_cast<const CF *>(mb_luma_dc_);
    const bool bypass = R.flags & FFHIP_H264_INTRA_BYPASS, dpcm = bypass && (R.flags & FFHIP_H264_INTRA_DPCM);
    if (bypass && W != 1) {
        ffhip_set_error("ffhip_h264_intra_pack: the transform bypass is taken at 8 bits only");
        return FFHIP_ENOSYS;
    }
    R.flags = bypass ? FFHIP_H264_INTRA_BYPASS : 0;
    memset(R.pad, 0, sizeof(R.pad));
    memset(R.nnz, 0, sizeof(R.nnz));
    memset(R.luma_dc, 0, sizeof(R.luma_dc));
    R.blocks = 0;
    int32_t n = (*ncoefs + 7) & ~7; /* runs start on 16 bytes */
    if ((int64_t)n + (384 + 16) * W > cap)
        return FFHIP_ENOMEM;
    for (int32_t i = *ncoefs; i < n; i++)
        coefs[i] = 0;
    R.coef = n;
    if (R.type == FFHIP_H264_INTRA_PCM) {
        if (!pcm)
            return FFHIP_EINVAL;
        const int nsamp = luma_only ? 256 : 384;
        if (W == 1) {
            memcpy(coefs + n, pcm, (size_t)nsamp);
            memset(reinterpret_cast<uint8_t *>(coefs + n) + nsamp, 0, (size_t)(384 - nsamp));
        } else {
            /* get_bits(&gb, bit_depth) 384 times over sl->intra_pcm_ptr (h264_mb_template.c:100-131): MSB-first fields */
            uint16_t *out = reinterpret_cast<uint16_t *>(coefs + n);
            uint32_t acc = 0;
            int have = 0;
            memset(out, 0, 384 * sizeof(uint16_t));
            for (int k = 0; k < nsamp; k++) {
                while (have < bd) {
                    acc = (acc << 8) | *pcm++;
                    have += 8;
                }
                have -= bd;
                out[k] = (uint16_t)((acc >> have) & ((1u << bd) - 1u));
            }
        }
        *ncoefs = n + 192 * W;
        return 0;
    }
    if (!nnzc || !mb)
        return FFHIP_EINVAL;
    if (bypass) {
        /* hl_decode_mb() with transform_bypass (h264_mb.c:614-770, h264_mb_template.c:190-213): the blocks hold residual SAMPLES (row-major:
         * the decoder reads them with the untransposed scans, h264_slice.c:770-777), added by add_pixels4 / 8_clear to the prediction, or —
         * _DPCM, vertical / horizontal prediction — by the pred*_add forms, which add each residual to the sample before it along the
         * direction: here the residuals of such a block become their running sums along the direction (modulo 2^16; the sample keeps its
         * low 8 bits), after which they are residuals of the plain prediction.  A block that travels is marked nnz = 16: "add it whole". */
        auto sum = [](CF a, CF b) { return (CF)(uint16_t)((uint32_t)(uint16_t)a + (uint32_t)(uint16_t)b); };
        /* running sums over an n x n region made of 4x4 blocks (base[blk(x4, y4) * 16 + x + 4 y]) or one 8x8 block; vertical: down the columns */
        auto chain = [&](CF *base, int n, bool vertical, auto blk_at) {
            for (int a = 0; a < n; a++) {       /* the line across the direction */
                CF run = 0;
                for (int t = 0; t < n; t++) {   /* along the direction */
                    const int x = vertical ? a : t, y = vertical ? t : a;
                    CF *c = blk_at(base, x, y);
                    run = sum(run, *c);
                    *c = run;
                }
            }
        };
        auto any = [](const CF *b, int cnt) {
            for (int i = 0; i < cnt; i++)
                if (b[i])
                    return true;
            return false;
        };
        auto take_all = [&](CF *b, int cnt) {
            memcpy(coefs + n, b, sizeof(CF) * cnt);
            n += cnt * W;
            memset(b, 0, sizeof(CF) * cnt);
        };
        auto in4 = [](CF *base, int x, int y) { /* luma 4x4 blocks in decoding order: block i at (imb_bx(i), imb_by(i)) */
            const int i = ((x >> 2) & 1) | ((y >> 2) & 1) << 1 | (x >> 3) << 2 | (y >> 3) << 3;
            return base + 16 * i + (x & 3) + 4 * (y & 3);
        };
        if (R.type == FFHIP_H264_INTRA_16x16) {
            if (nnzc[0]) { /* the DC block's samples go to their blocks' first positions (dc_mapping, h264_mb.c:713-723) */
                static const uint8_t dc_mapping[16] = { 0, 1, 4, 5, 2, 3, 6, 7, 8, 9, 12, 13, 10, 11, 14, 15 };
                if (!mb_luma_dc)
                    return FFHIP_EINVAL;
                for (int i = 0; i < 16; i++)
                    mb[16 * dc_mapping[i]] = mb_luma_dc[i];
            }
            const bool ch = dpcm && (R.pred16 == 2 /* VERT_PRED8x8 */ || R.pred16 == 1 /* HOR_PRED8x8 */);
            if (ch) { /* pred16x16_{vertical,horizontal}_add: the sixteen blocks' pred4x4_*_add run into one another (h264pred_template.c:1305-1330) */
                chain(mb, 16, R.pred16 == 2, in4);
                R.pad[0]++;
            }
            for (int i = 0; i < 16; i++)
                if (ch ? any(mb + 16 * i, 16) : (nnzc[scan8_luma(i)] || mb[16 * i])) {
                    R.nnz[i] = 16;
                    R.blocks |= 1u << i;
                    take_all(mb + 16 * i, 16);
                }
        } else if (R.type == FFHIP_H264_INTRA_4x4) {
            for (int i = 0; i < 16; i++) {
                const bool ch = dpcm && R.pred4[i] <= 1; /* VERT_PRED 0, HOR_PRED 1: pred4x4_*_add, whatever the count says */
                if (ch) {
                    chain(mb + 16 * i, 4, R.pred4[i] == 0, [](CF *b, int x, int y) { return b + x + 4 * y; });
                    R.pad[0]++;
                }
                if (ch ? any(mb + 16 * i, 16) : nnzc[scan8_luma(i)] != 0) {
                    R.nnz[i] = 16;
                    R.blocks |= 1u << i;
                    take_all(mb + 16 * i, 16);
                }
            }
        } else {
            for (int i = 0; i < 16; i += 4) {
                const bool ch = dpcm && R.pred4[i] <= 1; /* pred8x8l_*_filter_add: the filtered edge sample plus the running sum */
                if (ch) {
                    chain(mb + 16 * i, 8, R.pred4[i] == 0, [](CF *b, int x, int y) { return b + x + 8 * y; });
                    R.pad[0]++;
                }
                if (ch ? any(mb + 16 * i, 64) : nnzc[scan8_luma(i)] != 0) {
                    R.nnz[i] = 16;
                    R.blocks |= 1u << i;
                    take_all(mb + 16 * i, 64);
                }
            }
        }
        if (luma_only)
            R.cbp &= 0x0f;
        if (R.cbp & 0x30) { /* (h264_mb_template.c:193-213: no DC transform; pred8x8_*_add chains the plane's four blocks) */
            const bool ch = dpcm && (R.chroma_pred == 2 || R.chroma_pred == 1);
            for (int pl = 1; pl < 3; pl++) {
                CF *base = mb + 256 * pl;
                if (ch) {
                    chain(base, 8, R.chroma_pred == 2, [](CF *b, int x, int y) { return b + 16 * ((x >> 2) + 2 * (y >> 2)) + (x & 3) + 4 * (y & 3); });
                    R.pad[0]++;
                }
                for (int k = 0; k < 4; k++)
                    if (ch ? any(base + 16 * k, 16) : (nnzc[scan8_chroma(pl, k)] || base[16 * k])) {
                        R.nnz[16 + 4 * (pl - 1) + k] = 16;
                        R.blocks |= 1u << (16 + 4 * (pl - 1) + k);
                        take_all(base + 16 * k, 16);
                    }
            }
        }
        *ncoefs = n;
        return 0;
    }
    /* a block travels when the dsp function hl_decode_mb() would call on it reads it; the caller's copy is consumed the way that
     * function consumes it: zeroed by idct_add / idct8_add (h264idct_template.c:66,142), [0] = 0 by the dc forms (:150,166) */
    auto take = [&](CF *b, int cnt, bool full) {
        memcpy(coefs + n, b, sizeof(CF) * cnt);
        n += cnt * W;
        if (full)
            memset(b, 0, sizeof(CF) * cnt);
        else
            b[0] = 0;
    };
    if (R.type == FFHIP_H264_INTRA_16x16 && nnzc[0]) {
        /* scan8[LUMA_DC_BLOCK_INDEX]: luma_dc_dequant_idct writes the 16 DC positions of sl->mb (h264_mb.c:707-711) */
        if (!mb_luma_dc)
            return FFHIP_EINVAL;
        R.flags |= FFHIP_H264_INTRA_LUMA_DC;
        if (W == 1) {
            memcpy(R.luma_dc, mb_luma_dc, sizeof(R.luma_dc));
        } else { /* dctcoef does not fit the record's int16 field: the sixteen DCs lead the run (h264_intra_mb.h imb_block) */
            memcpy(coefs + n, mb_luma_dc, sizeof(CF) * 16);
            n += 16 * W;
        }
    }
    if (R.type == FFHIP_H264_INTRA_4x4) {
        for (int i = 0; i < 16; i++) {
            const int nnz = nnzc[scan8_luma(i)];
            R.nnz[i] = (uint8_t)nnz;
            if (nnz) {
                R.blocks |= 1u << i;
                take(mb + i * 16, 16, !(nnz == 1 && mb[i * 16]));
            }
        }
    } else if (R.type == FFHIP_H264_INTRA_8x8) {
        for (int i = 0; i < 16; i += 4) {
            const int nnz = nnzc[scan8_luma(i)];
            R.nnz[i] = (uint8_t)nnz;
            if (nnz) {
                R.blocks |= 1u << i;
                take(mb + i * 16, 64, !(nnz == 1 && mb[i * 16]));
            }
        }
    } else {
        for (int i = 0; i < 16; i++) { /* idct_add16intra (h264idct_template.c:191-200) */
            const int nnz = nnzc[scan8_luma(i)];
            R.nnz[i] = (uint8_t)nnz;
            if (nnz || mb[i * 16]) {
                R.blocks |= 1u << i;
                take(mb + i * 16, 16, nnz != 0);
            }
        }
    }
    if (luma_only)
        R.cbp &= 0x0f;
    if (R.cbp & 0x30) { /* chroma_dc_dequant_idct + idct_add8 (h264_mb_template.c:246-258, h264idct_template.c:216-228) */
        for (int pl = 1; pl < 3; pl++) {
            if (nnzc[40 * pl]) /* scan8[CHROMA_DC_BLOCK_INDEX + pl - 1] */
                R.flags |= (uint8_t)(FFHIP_H264_INTRA_CB_DC << (pl - 1));
            for (int k = 0; k < 4; k++) {
                CF *b = mb + 256 * pl + 16 * k;
                const int nnz = nnzc[scan8_chroma(pl, k)];
                R.nnz[16 + 4 * (pl - 1) + k] = (uint8_t)nnz;
                if (nnz || b[0]) {
                    R.blocks |= 1u << (16 + 4 * (pl - 1) + k);
                    take(b, 16, nnz != 0);
                }
            }
        }
    }
    *ncoefs = n;
    return 0;
}

extern "C" int ffhip_h264_intra_pack(FFHipH264IntraMB *rec, const uint8_t *nnzc, int16_t *mb, const int16_t *mb_luma_dc, const uint8_t *pcm,
                                     int16_t *coefs, int32_t *ncoefs, int32_t cap)
{
    return intra_pack<int16_t>(8, rec, nnzc, mb, mb_luma_dc, pcm, coefs, ncoefs, cap);
}

extern "C" int ffhip_h264_intra_pack_hbd(int bit_depth, FFHipH264IntraMB *rec, const uint8_t *nnzc, int16_t *mb, const int16_t *mb_luma_dc,
                                         const uint8_t *pcm, int16_t *coefs, int32_t *ncoefs, int32_t cap)
{
    if (bit_depth == 8)
        return intra_pack<int16_t>(8, rec, nnzc, mb, mb_luma_dc, pcm, coefs, ncoefs, cap);
    if (bit_depth != 9 && bit_depth != 10 && bit_depth != 12 && bit_depth != 14)
        return FFHIP_EINVAL;
    return intra_pack<int32_t>(bit_depth, rec, nnzc, mb, mb_luma_dc, pcm, coefs, ncoefs, cap);
}

/* One plane of a 4:4:4 macroblock (hl_decode_mb_444: hl_decode_mb_predict_luma / hl_decode_mb_idct_luma with p = plane,
 * h264_mb.c:614-800): the arguments are the plane's slices of the decoder's arrays — nnzc + 5 * 8 * p (scan8[i + 16 p] = scan8[i] + 40 p,
 * the DC entry scan8[LUMA_DC_BLOCK_INDEX + p] = 40 p), sl->mb + 256 p, sl->mb_luma_dc[p], the plane's 256 I_PCM samples — and
 * rec->qmul[0] = pps->dequant4_coeff[p][p ? chroma_qp[p - 1] : qscale][0]. */
extern "C" int ffhip_h264_intra_pack_plane(int bit_depth, FFHipH264IntraMB *rec, const uint8_t *nnzc, int16_t *mb, const int16_t *mb_luma_dc,
                                           const uint8_t *pcm, int16_t *coefs, int32_t *ncoefs, int32_t cap)
{
    if (bit_depth == 8)
        return intra_pack<int16_t>(8, rec, nnzc, mb, mb_luma_dc, pcm, coefs, ncoefs, cap, true);
    if (bit_depth != 9 && bit_depth != 10 && bit_depth != 12 && bit_depth != 14)
        return FFHIP_EINVAL;
    return intra_pack<int32_t>(bit_depth, rec, nnzc, mb, mb_luma_dc, pcm, coefs, ncoefs, cap, true);
}

/* The chroma planes of an intra macroblock of a 4:2:2 picture: which of the 2 x 8 blocks travel, consumed as chroma422_dc_dequant_idct +
 * idct_add8_422 leave them (h264idct_template.c:230-252: idct_add zeroes a block, idct_dc_add its DC; the DC positions the dequantiser
 * wrote are among them).  qmul: pps->dequant4_coeff[1 + p][chroma_qp[p] + 3][0].  pcm: the 2 x 128 chroma fields (after the 256 luma ones). */
template <typename CF>
static int intra_pack_c422(int bd, FFHipH264IntraC422 *rec, const FFHipH264IntraMB *d, const uint8_t *nnzc, int16_t *mb_, const uint8_t *pcm,
                           std::vector<int16_t> &coefs)
{
    constexpr int W = (int)(sizeof(CF) / sizeof(int16_t));
    FFHipH264IntraC422 &R = *rec;
    memset(&R, 0, sizeof(R));
    R.mb_x = d->mb_x;
    R.mb_y = d->mb_y;
    R.type = d->type;
    R.chroma_pred = d->chroma_pred;
    R.cbp = d->cbp & 0x30;
    R.qmul[0] = d->qmul[1];
    R.qmul[1] = d->qmul[2];
    while (coefs.size() & 7)
        coefs.push_back(0); /* runs start on 16 bytes */
    if (coefs.size() > (size_t)INT32_MAX - 4096)
        return FFHIP_EINVAL;
    R.coef = (int32_t)coefs.size();
    if (R.type == FFHIP_H264_INTRA_PCM) {
        if (!pcm)
            return FFHIP_EINVAL;
        const size_t at = coefs.size();
        coefs.resize(at + 128 * W);
        if (W == 1) {
            memcpy(coefs.data() + at, pcm, 256);
        } else {
            uint16_t *out = reinterpret_cast<uint16_t *>(coefs.data() + at);
            uint32_t acc = 0;
            int have = 0;
            for (int k = 0; k < 256; k++) { /* MSB-first bit_depth-bit fields (h264_mb_template.c:100-131) */
                while (have < bd) {
                    acc = (acc << 8) | *pcm++;
                    have += 8;
                }
                have -= bd;
                out[k] = (uint16_t)((acc >> have) & ((1u << bd) - 1u));
            }
        }
        return 0;
    }
    if (!nnzc || !mb_)
        return FFHIP_EINVAL;
    CF *mb = reinterpret_cast<CF *>(mb_);
    if (d->flags & FFHIP_H264_INTRA_BYPASS) {
        /* the transform bypass at 4:2:2 (h264_mb_template.c:193-221 with chroma422): residual samples, no DC transform; _DPCM with vertical /
         * horizontal prediction: pred8x16_{vertical,horizontal}_add chains the plane's eight blocks (h264pred_template.c:1280-1303) — the
         * residuals become running sums along the direction, as in intra_pack() */
        if (W != 1)
            return FFHIP_ENOSYS;
        R.flags = FFHIP_H264_INTRA_BYPASS;
        if (R.cbp & 0x30) {
            const bool ch = (d->flags & FFHIP_H264_INTRA_DPCM) && (R.chroma_pred == 2 || R.chroma_pred == 1), vertical = R.chroma_pred == 2;
            for (int pl = 1; pl < 3; pl++) {
                CF *base = mb + 256 * pl;
                auto at = [&](int x, int y) { return base + 16 * ((x >> 2) + 2 * (y >> 2)) + (x & 3) + 4 * (y & 3); };
                if (ch)
                    for (int a = 0; a < (vertical ? 8 : 16); a++) {
                        CF run = 0;
                        for (int t = 0; t < (vertical ? 16 : 8); t++) {
                            CF *c = vertical ? at(a, t) : at(t, a);
                            run = (CF)(uint16_t)((uint32_t)(uint16_t)run + (uint32_t)(uint16_t)*c);
                            *c = run;
                        }
                    }
                for (int k = 0; k < 8; k++) {
                    CF *b = base + 16 * k;
                    bool present = nnzc[scan8_chroma(pl, k)] || b[0];
                    if (ch) {
                        present = false;
                        for (int i = 0; i < 16; i++)
                            present = present || b[i];
                    }
                    if (present) {
                        const int bit = 8 * (pl - 1) + k;
                        R.full |= (uint16_t)(1u << bit);
                        R.blocks |= (uint16_t)(1u << bit);
                        const size_t o = coefs.size();
                        coefs.resize(o + 16 * W);
                        memcpy(coefs.data() + o, b, sizeof(CF) * 16);
                        memset(b, 0, sizeof(CF) * 16);
                    }
                }
            }
        }
        return 0;
    }
    if (R.cbp & 0x30)
        for (int pl = 1; pl < 3; pl++) {
            if (nnzc[40 * pl]) /* scan8[CHROMA_DC_BLOCK_INDEX + pl - 1] */
                R.flags |= (uint8_t)(1 << (pl - 1));
            for (int k = 0; k < 8; k++) {
                CF *b = mb + 256 * pl + 16 * k;
                const int nnz = nnzc[scan8_chroma(pl, k)], bit = 8 * (pl - 1) + k;
                if (nnz)
                    R.full |= (uint16_t)(1u << bit);
                if (nnz || b[0]) {
                    R.blocks |= (uint16_t)(1u << bit);
                    const size_t at = coefs.size();
                    coefs.resize(at + 16 * W);
                    memcpy(coefs.data() + at, b, sizeof(CF) * 16);
                    if (nnz)
                        memset(b, 0, sizeof(CF) * 16);
                    else
                        b[0] = 0;
                }
            }
        }
    return 0;
}

extern "C" int ffhip_h264_picture_intra_mb(FFHipH264Picture *p, const FFHipH264IntraMB *d, const uint8_t *nnzc, int16_t *mb,
                                           const int16_t *mb_luma_dc, const uint8_t *pcm)
{
    if (!p || !d || d->mb_x < 0 || d->mb_x >= p->mb_w || d->mb_y < 0 || d->mb_y >= p->mb_h)
        return FFHIP_EINVAL;
    const int wide = p->bd > 8 ? 2 : 1; /* int16 entries per dctcoef */
    if ((d->flags & FFHIP_H264_INTRA_BYPASS) && p->bd != 8) {
        ffhip_set_error("ffhip_h264_picture_intra_mb: the transform bypass is taken at 8 bits");
        return FFHIP_ENOSYS;
    }
    if (p->cfmt == 2) {
        /* 4:2:2: the luma as a luma-only record of the wavefront, the two 8 x 16 chroma planes as a record of their own */
        FFHipH264IntraMB R = *d;
        std::vector<int16_t> &c = p->intra_coef[0];
        if (c.size() > (size_t)INT32_MAX - 2048)
            return FFHIP_EINVAL;
        int32_t n = (int32_t)c.size();
        c.resize((size_t)n + 816);
        int r = ffhip_h264_intra_pack_plane(p->bd, &R, nnzc, mb, mb_luma_dc, pcm, c.data(), &n, (int32_t)c.size());
        c.resize((size_t)n);
        if (r < 0)
            return r;
        FFHipH264IntraC422 C;
        const uint8_t *pcm_c = pcm ? pcm + (p->bd > 8 ? 32 * p->bd : 256) : nullptr; /* behind the 256 luma fields */
        r = p->bd > 8 ? intra_pack_c422<int32_t>(p->bd, &C, d, nnzc, mb, pcm_c, p->intra_c422_coef)
                      : intra_pack_c422<int16_t>(8, &C, d, nnzc, mb, pcm_c, p->intra_c422_coef);
        if (r < 0)
            return r;
        p->intra[0].push_back(R);
        p->intra_c422.push_back(C);
        return 0;
    }
    for (int pl = 0; pl < p->intra_sets(); pl++) {
        FFHipH264IntraMB R = *d;
        std::vector<int16_t> &c = p->intra_coef[pl];
        if (c.size() > (size_t)INT32_MAX - 2048)
            return FFHIP_EINVAL;
        int32_t n = (int32_t)c.size();
        c.resize((size_t)n + 816);
        int r;
        if (p->cfmt == 3) {
            /* the plane's share of the macroblock as a luma-only record: same prediction modes and availability in all three planes */
            R.qmul[0] = d->qmul[pl];
            const uint8_t *pcm_pl = pcm ? pcm + (p->bd > 8 ? 32 * p->bd * pl : 256 * pl) : nullptr; /* 256 bit_depth-bit fields per plane */
            r = ffhip_h264_intra_pack_plane(p->bd, &R, nnzc ? nnzc + 40 * pl : nullptr, mb ? mb + 256 * pl * wide : nullptr,
                                            mb_luma_dc ? mb_luma_dc + 32 * pl : nullptr /* int16_t mb_luma_dc[3][16 * 2], h264dec.h */, pcm_pl,
                                            c.data(), &n, (int32_t)c.size());
        } else {
            r = ffhip_h264_intra_pack_hbd(p->bd, &R, nnzc, mb, mb_luma_dc, pcm, c.data(), &n, (int32_t)c.size());
        }
        c.resize((size_t)n);
        if (r < 0)
            return r;
        p->intra[pl].push_back(R);
    }
    return 0;
}

extern "C" int ffhip_h264_intra_frame_dev(uint8_t *y, uint8_t *cb, uint8_t *cr, ptrdiff_t stride_y, ptrdiff_t stride_c, int mb_w, int mb_h,
                                          const FFHipH264IntraMB *recs, const int32_t *row_start, const int16_t *coefs, void *stream)
{
    if (!ffhip_have_device())
        return FFHIP_ENOSYS;
    return ffhip_launch_h264_intra_frame(y, cb, cr, stride_y, stride_c, mb_w, mb_h, recs, row_start, coefs, (hipStream_t)stream);
}

extern "C" int ffhip_h264_intra_frame_dev_hbd(int bit_depth, uint8_t *y, uint8_t *cb, uint8_t *cr, ptrdiff_t stride_y, ptrdiff_t stride_c, int mb_w,
                                              int mb_h, const FFHipH264IntraMB *recs, const int32_t *row_start, const int16_t *coefs, void *stream)
{
    if (!ffhip_have_device())
        return FFHIP_ENOSYS;
    return ffhip_launch_h264_intra_frame_bd(bit_depth, y, cb, cr, stride_y, stride_c, mb_w, mb_h, recs, row_start, coefs, (hipStream_t)stream);
}

extern "C" int ffhip_h264_intra_frames_dev(int bit_depth, int npics, const FFHipH264IntraPic *pics, ptrdiff_t stride_y, ptrdiff_t stride_c,
                                           int mb_w, int mb_h, void *stream)
{
    if (npics < 0 || (npics && !pics))
        return FFHIP_EINVAL;
    if (!ffhip_have_device())
        return FFHIP_ENOSYS;
    return ffhip_launch_h264_intra_frames_bd(bit_depth, npics, pics, stride_y, stride_c, mb_w, mb_h, (hipStream_t)stream);
}

/* what has been recorded since begin(), as it stands: pointers into the object, valid until the next record call */
extern "C" int ffhip_h264_picture_lists(const FFHipH264Picture *p, FFHipH264PictureLists *out)
{
    if (!p || !out)
        return FFHIP_EINVAL;
    memset(out, 0, sizeof(*out));
    out->mb_w = p->mb_w;
    out->mb_h = p->mb_h;
    out->bit_depth = p->bd;
    out->chroma_format_idc = p->cfmt;
    for (int pl = 0; pl < 3; pl++) {
        for (int s = 0; s < 3; s++) {
            out->qpel[pl][s] = p->qpel[pl][s].data();
            out->nqpel[pl][s] = (int)p->qpel[pl][s].size();
            if (pl) {
                out->cmc[pl - 1][s] = p->cmc[pl - 1][s].data();
                out->ncmc[pl - 1][s] = (int)p->cmc[pl - 1][s].size();
            }
        }
        out->wt[pl] = p->wt[pl].data();
        out->nwt[pl] = (int)p->wt[pl].size();
        for (int k = 0; k < 4; k++) {
            out->idct_off[pl][k] = p->idct_off[pl][k].data();
            out->idct_coef[pl][k] = p->idct_coef[pl][k].data();
            out->nidct[pl][k] = (int)p->idct_off[pl][k].size();
        }
        for (int k = 0; k < 2; k++) {
            out->addpx_off[pl][k] = p->idct_off[pl][4 + k].data();
            out->addpx_coef[pl][k] = p->idct_coef[pl][4 + k].data();
            out->naddpx[pl][k] = (int)p->idct_off[pl][4 + k].size();
        }
        out->intra[pl] = p->intra[pl].data();
        out->nintra[pl] = (int)p->intra[pl].size();
        out->intra_coef[pl] = p->intra_coef[pl].data();
        out->nintra_coef[pl] = (int)p->intra_coef[pl].size();
        out->edges[pl] = p->any_edge[pl] ? p->edges[pl].data() : nullptr;
    }
    out->intra_c422 = p->intra_c422.data();
    out->nintra_c422 = (int)p->intra_c422.size();
    out->intra_c422_coef = p->intra_c422_coef.data();
    out->nintra_c422_coef = (int)p->intra_c422_coef.size();
    return 0;
}

extern "C" int ffhip_h264_intra_planes_dev(int bit_depth, int nplanes, const FFHipH264IntraPic *planes, ptrdiff_t stride, int mb_w, int mb_h,
                                           void *stream)
{
    if (nplanes < 0 || (nplanes && !planes))
        return FFHIP_EINVAL;
    if (!ffhip_have_device())
        return FFHIP_ENOSYS;
    return ffhip_launch_h264_intra_frames_bd(bit_depth, nplanes, planes, stride, stride, mb_w, mb_h, (hipStream_t)stream, 1);
}

/* ---- flush ------------------------------------------------------------------------------------------ */
/* what a batched flush (ffhip_h264_pictures_flush) takes over from a picture after its prediction and residual stages */
struct FlushBack {
    int nintra = 0;                 /* wavefronts of this picture: 1 (4:2:0, all three planes), or one per plane that has records (4:4:4) */
    FFHipH264IntraPic ip[3] = {};
    const FFHipH264Edge *edges[3] = { nullptr, nullptr, nullptr };
    /* 4:2:2: the chroma planes' wavefront */
    const FFHipH264IntraC422 *c422 = nullptr;
    const int32_t *c422_rows = nullptr;
    const int16_t *c422_coef = nullptr;
};

/* the picture's intra wavefront(s) and in-loop filter, from what the front half left in `B` */
static int flush_tail(FFHipH264Picture *p, uint8_t *const dst[3], const int stride[3], const FlushBack &B, hipStream_t stream)
{
    const int bd = p->bd;
    int r = 0;
    /* ---- intra macroblocks: every inter macroblock is complete now; one wavefront over the three planes (4:4:4: one per plane, side by
     * side in one launch) ---- */
    if (B.c422) {
        /* 4:2:2: the luma wavefront below is luma-only; the 8 x 16 chroma planes are a wavefront of their own, beside it on the picture's
         * second stream (prediction never crosses planes) */
        HIP_TRY(hipEventRecord(p->fork, stream));
        HIP_TRY(hipStreamWaitEvent(p->aux, p->fork, 0));
        ffhip_progress_report_to(stream, true); /* a hand-off lost on the second stream is the caller's stream's to hear about */
        r = ffhip_launch_h264_intra_c422(bd, dst[1], dst[2], stride[1], p->mb_w, p->mb_h, B.c422, B.c422_rows, B.c422_coef, p->aux);
        ffhip_progress_report_to(nullptr, false);
        HIP_TRY(hipEventRecord(p->join, p->aux));
    }
    if (r >= 0 && B.nintra)
        r = ffhip_launch_h264_intra_frames_bd(bd, B.nintra, B.ip, stride[0], stride[1], p->mb_w, p->mb_h, stream, p->cfmt != 1);
    if (B.c422)
        HIP_TRY(hipStreamWaitEvent(stream, p->join, 0));
    if (r < 0)
        return r;
    /* ---- in-loop filter, decoder order: the planes are independent, and a lone wavefront is a chain of dependent hand-offs that
     * leaves the GPU mostly idle — the chroma planes run beside the luma plane on the second stream ---- */
    const bool chroma = B.edges[1] || B.edges[2];
    if (p->cfmt == 2) {
        /* 4:2:2: 8 x 16 chroma macroblocks with six edges each: a frame-order kernel of their own (both planes in one launch when they
         * share a stride), beside the luma plane's on the second stream */
        if (chroma) {
            HIP_TRY(hipEventRecord(p->fork, stream));
            HIP_TRY(hipStreamWaitEvent(p->aux, p->fork, 0));
            ffhip_progress_report_to(stream, true);
            if (B.edges[1] && B.edges[2] && stride[1] == stride[2]) {
                uint8_t *const pl_[2] = { dst[1], dst[2] };
                const FFHipH264Edge *const ed_[2] = { B.edges[1], B.edges[2] };
                r = ffhip_launch_h264_deblock_c422_planes(bd, 2, pl_, ed_, stride[1], p->mb_w, p->mb_h, p->aux);
            } else {
                for (int pl = 1; pl < 3 && r >= 0; pl++)
                    if (B.edges[pl])
                        r = ffhip_launch_h264_deblock_c422(bd, dst[pl], stride[pl], p->mb_w, p->mb_h, B.edges[pl], p->aux);
            }
            ffhip_progress_report_to(nullptr, false);
            HIP_TRY(hipEventRecord(p->join, p->aux));
        }
        if (r >= 0 && B.edges[0])
            r = ffhip_launch_h264_deblock_frames_bd(bd, 0, dst[0], 0, 1, stride[0], p->mb_w, p->mb_h, B.edges[0], stream);
        if (chroma)
            HIP_TRY(hipStreamWaitEvent(stream, p->join, 0));
        return r < 0 ? r : 0;
    }
    if (p->cfmt == 3) {
        /* all three planes by the luma filter (filter_mb_edgev / edgeh on img_cb / img_cr, h264_loopfilter.c:601-703): one launch of
         * three "pictures" when the skewed-rows kernel can address them by table, else plane by plane */
        uint8_t *pl_[3];
        const FFHipH264Edge *ed_[3];
        int n = 0;
        bool tab = stride[0] == stride[1] && stride[0] == stride[2] && !(stride[0] & 15);
        for (int pl = 0; pl < 3; pl++)
            if (B.edges[pl]) {
                pl_[n] = dst[pl];
                ed_[n++] = B.edges[pl];
                tab = tab && !((uintptr_t)dst[pl] & 15);
            }
        if (n > 1 && tab)
            return ffhip_launch_h264_deblock_pictures_bd(bd, 0, pl_, ed_, n, stride[0], p->mb_w, p->mb_h, stream);
        n = 0;
        for (int pl = 0; pl < 3 && r >= 0; pl++)
            if (B.edges[pl])
                r = ffhip_launch_h264_deblock_frames_bd(bd, 0, dst[pl], 0, 1, stride[pl], p->mb_w, p->mb_h, B.edges[pl], stream);
        return r < 0 ? r : 0;
    }
    if (chroma) {
        HIP_TRY(hipEventRecord(p->fork, stream));
        HIP_TRY(hipStreamWaitEvent(p->aux, p->fork, 0));
        const ptrdiff_t gap = dst[2] - dst[1];
        ffhip_progress_report_to(stream, true); /* a hand-off lost on the second stream is the caller's stream's to hear about */
        /* (8 bits: Cb and Cr as ONE launch of two "pictures" when Cr follows Cb at a 4-byte aligned distance and both are filtered) */
        if (bd == 8 && B.edges[1] && B.edges[2] && stride[1] == stride[2] && gap > 0 && !(gap & 3) &&
            B.edges[2] == B.edges[1] + p->edges[1].size()) {
            r = ffhip_launch_h264_deblock_frames_chroma(dst[1], (size_t)gap, 2, stride[1], p->mb_w, p->mb_h, B.edges[1], p->aux);
        } else {
            for (int pl = 1; pl < 3 && r >= 0; pl++)
                if (B.edges[pl])
                    r = ffhip_launch_h264_deblock_frames_bd(bd, 1, dst[pl], 0, 1, stride[pl], p->mb_w, p->mb_h, B.edges[pl], p->aux);
        }
        ffhip_progress_report_to(nullptr, false);
        HIP_TRY(hipEventRecord(p->join, p->aux));
    }
    if (r >= 0 && B.edges[0])
        r = ffhip_launch_h264_deblock_frames_bd(bd, 0, dst[0], 0, 1, stride[0], p->mb_w, p->mb_h, B.edges[0], stream);
    if (chroma)
        HIP_TRY(hipStreamWaitEvent(stream, p->join, 0));
    return r < 0 ? r : 0;
}

static int flush_impl(FFHipH264Picture *p, uint8_t *const dst[3], const int stride[3], const uint8_t *const ref[3], void *stream_, FlushBack *defer)
{
    FFHipDeviceGuard dg(p ? p->device : -1);
    if (!p || !dst || !stride || !ref)
        return FFHIP_EINVAL;
    hipStream_t stream = (hipStream_t)stream_;
    for (int pl = 0; pl < 3; pl++)
        if (!dst[pl] || !ref[pl] || stride[pl] <= 0)
            return FFHIP_EINVAL;
    if (p->device < 0) {
        ffhip_set_error("ffhip_h264_picture_flush: the picture object was made without a HIP device (records only)");
        return FFHIP_ENOSYS;
    }
    /* a finished deblocking wavefront (an earlier picture's) that lost a hand-off is reported now rather than never */
    {
        /* (the chroma planes' wavefront runs on the picture's second stream but files its failures under the caller's) */
        const int r = ffhip_progress_check(stream);
        if (r < 0)
            return r;
    }
    /* everything a later stage would refuse is refused before anything is queued: the stages modify dst in place */
    if (p->bd > 8)
        for (int pl = 0; pl < 3; pl++)
            if (p->any_edge[pl] && ((stride[pl] & 15) || ((uintptr_t)dst[pl] & 15))) {
                ffhip_set_error("ffhip_h264_picture_flush: plane %d of a %d-bit picture with deblocking records must be 16-byte aligned "
                                "(plane and stride)", pl, p->bd);
                return FFHIP_EINVAL;
            }
    if (p->bd > 8)
        for (int pl = 0; pl < 3; pl++)
            if ((stride[pl] & 1) || ((uintptr_t)dst[pl] & 1) || ((uintptr_t)ref[pl] & 1)) {
                ffhip_set_error("ffhip_h264_picture_flush: plane %d of a %d-bit picture is not 2-byte aligned", pl, p->bd);
                return FFHIP_EINVAL;
            }
    const int nsets = p->intra_sets();
    if (p->cfmt == 3 && (!p->intra[0].empty() || !p->intra[1].empty() || !p->intra[2].empty())) {
        /* the wavefront launch takes one luma stride for every "picture" of a launch: the three planes of a 4:4:4 picture share it (the
         * decoder's linesize == uvlinesize there); four samples per access */
        const unsigned amask = p->bd > 8 ? 7u : 3u;
        if (stride[1] != stride[0] || stride[2] != stride[0] || (((uintptr_t)dst[0] | (uintptr_t)dst[1] | (uintptr_t)dst[2] | (unsigned)stride[0]) & amask)) {
            ffhip_set_error("ffhip_h264_picture_flush: the planes of a 4:4:4 picture with intra macroblocks share one stride and are %u-byte aligned",
                            amask + 1);
            return FFHIP_EINVAL;
        }
    }

    if (!p->intra_c422.empty() && (stride[1] != stride[2] || (((uintptr_t)dst[1] | (uintptr_t)dst[2] | (unsigned)stride[1]) & (p->bd > 8 ? 7u : 3u)))) {
        ffhip_set_error("ffhip_h264_picture_flush: the chroma planes of a 4:2:2 picture with intra macroblocks share one stride and are %d-byte "
                        "aligned", p->bd > 8 ? 8 : 4);
        return FFHIP_EINVAL;
    }
    /* layout of the one staging buffer */
    size_t total = 0;
    Section s_qpel[3][3], s_cmc[2][3], s_wt[3], s_ioff[3][6], s_icoef[3][6], s_edge[3], s_intra[3], s_irows[3], s_intracoef[3];
    Section s_c422, s_c422rows, s_c422coef;
    if (!p->intra_c422.empty()) {
        p->intra_c422_sorted = p->intra_c422;
        std::stable_sort(p->intra_c422_sorted.begin(), p->intra_c422_sorted.end(), [](const FFHipH264IntraC422 &a, const FFHipH264IntraC422 &b) {
            return a.mb_y != b.mb_y ? a.mb_y < b.mb_y : a.mb_x < b.mb_x;
        });
        p->intra_c422_rows.assign((size_t)p->mb_h + 1, 0);
        for (const FFHipH264IntraC422 &a : p->intra_c422_sorted)
            p->intra_c422_rows[(size_t)a.mb_y + 1]++; /* (a macroblock recorded twice is caught on its luma record below) */
        for (int r = 0; r < p->mb_h; r++)
            p->intra_c422_rows[(size_t)r + 1] += p->intra_c422_rows[r];
        if (p->intra_c422_coef.empty())
            p->intra_c422_coef.assign(8, 0);
        place(total, p->intra_c422_sorted, s_c422);
        place(total, p->intra_c422_rows, s_c422rows);
        place(total, p->intra_c422_coef, s_c422coef);
    }
    for (int q = 0; q < nsets; q++) {
        if (p->intra[q].empty())
            continue;
        /* the wavefront walks a row's intra macroblocks left to right: by (mb_y, mb_x), one record per macroblock */
        p->intra_sorted[q] = p->intra[q];
        std::stable_sort(p->intra_sorted[q].begin(), p->intra_sorted[q].end(), [](const FFHipH264IntraMB &a, const FFHipH264IntraMB &b) {
            return a.mb_y != b.mb_y ? a.mb_y < b.mb_y : a.mb_x < b.mb_x;
        });
        p->intra_rows[q].assign((size_t)p->mb_h + 1, 0);
        for (size_t i = 0; i < p->intra_sorted[q].size(); i++) {
            const FFHipH264IntraMB &a = p->intra_sorted[q][i];
            if (i && a.mb_y == p->intra_sorted[q][i - 1].mb_y && a.mb_x == p->intra_sorted[q][i - 1].mb_x) {
                ffhip_set_error("ffhip_h264_picture_flush: macroblock (%d, %d) recorded twice as intra", a.mb_x, a.mb_y);
                return FFHIP_EINVAL;
            }
            p->intra_rows[q][(size_t)a.mb_y + 1]++;
        }
        for (int r = 0; r < p->mb_h; r++)
            p->intra_rows[q][(size_t)r + 1] += p->intra_rows[q][r];
        if (p->intra_coef[q].empty())
            p->intra_coef[q].assign(8, 0); /* a picture of coefficient-free intra macroblocks still hands the kernel a base */
        place(total, p->intra_sorted[q], s_intra[q]);
        place(total, p->intra_rows[q], s_irows[q]);
        place(total, p->intra_coef[q], s_intracoef[q]);
    }
    for (int s = 0; s < 3; s++) {
        for (int pl = 0; pl < 3; pl++)
            place(total, p->qpel[pl][s], s_qpel[pl][s]);
        place(total, p->cmc[0][s], s_cmc[0][s]);
        place(total, p->cmc[1][s], s_cmc[1][s]);
    }
    for (int pl = 0; pl < 3; pl++) {
        place(total, p->wt[pl], s_wt[pl]);
        for (int k = 0; k < 6; k++) {
            place(total, p->idct_off[pl][k], s_ioff[pl][k]);
            place(total, p->idct_coef[pl][k], s_icoef[pl][k]);
        }
        if (p->any_edge[pl])
            place(total, p->edges[pl], s_edge[pl]);
    }
    total = (total + 255) & ~(size_t)255;
    if (p->copy_pending) { /* the previous picture's records are still leaving the pinned buffer */
        HIP_TRY(hipEventSynchronize(p->copied));
        p->copy_pending = false;
    }
    if (total > p->pinned_sz) {
        if (p->pinned)
            (void)hipHostFree(p->pinned);
        p->pinned = nullptr;
        p->pinned_sz = 0;
        const size_t want = total + total / 2;
        if (hipHostMalloc(&p->pinned, want, hipHostMallocDefault) != hipSuccess) {
            ffhip_set_error("ffhip_h264_picture_flush: hipHostMalloc(%zu) failed", want);
            return FFHIP_ENOMEM;
        }
        p->pinned_sz = want;
    }
    if (total > p->dev_sz) {
        /* the old buffer may still be read by launches of the previous picture on this stream */
        HIP_TRY(hipStreamSynchronize(stream));
        if (p->dev)
            (void)hipFree(p->dev);
        p->dev = nullptr;
        p->dev_sz = 0;
        const size_t want = total + total / 2;
        if (hipMalloc(&p->dev, want) != hipSuccess) {
            ffhip_set_error("ffhip_h264_picture_flush: hipMalloc(%zu) failed", want);
            return FFHIP_ENOMEM;
        }
        p->dev_sz = want;
    }
    uint8_t *hb = (uint8_t *)p->pinned, *db = (uint8_t *)p->dev;
    auto put = [&](const Section &s, const void *src, size_t bytes) {
        if (bytes)
            memcpy(hb + s.off, src, bytes);
    };
    for (int q = 0; q < nsets; q++)
        if (!p->intra[q].empty()) {
            put(s_intra[q], p->intra_sorted[q].data(), p->intra_sorted[q].size() * sizeof(FFHipH264IntraMB));
            put(s_irows[q], p->intra_rows[q].data(), p->intra_rows[q].size() * sizeof(int32_t));
            put(s_intracoef[q], p->intra_coef[q].data(), p->intra_coef[q].size() * sizeof(int16_t));
        }
    if (!p->intra_c422.empty()) {
        put(s_c422, p->intra_c422_sorted.data(), p->intra_c422_sorted.size() * sizeof(FFHipH264IntraC422));
        put(s_c422rows, p->intra_c422_rows.data(), p->intra_c422_rows.size() * sizeof(int32_t));
        put(s_c422coef, p->intra_c422_coef.data(), p->intra_c422_coef.size() * sizeof(int16_t));
    }
    bool need_tmp[3] = { false, false, false };
    for (int s = 0; s < 3; s++) {
        for (int pl = 0; pl < 3; pl++)
            put(s_qpel[pl][s], p->qpel[pl][s].data(), p->qpel[pl][s].size() * sizeof(FFHipQpelBlock));
        for (int c = 0; c < 2; c++)
            put(s_cmc[c][s], p->cmc[c][s].data(), p->cmc[c][s].size() * sizeof(FFHipChromaBlock));
    }
    for (int pl = 0; pl < 3; pl++)
        need_tmp[pl] = !p->qpel[pl][ST_TMP].empty() || (pl && !p->cmc[pl - 1][ST_TMP].empty());
    for (int pl = 0; pl < 3; pl++) {
        put(s_wt[pl], p->wt[pl].data(), p->wt[pl].size() * sizeof(FFHipWeightBlock));
        for (const FFHipWeightBlock &w : p->wt[pl])
            need_tmp[pl] = need_tmp[pl] || w.bi;
        for (int k = 0; k < 6; k++) {
            put(s_ioff[pl][k], p->idct_off[pl][k].data(), p->idct_off[pl][k].size() * sizeof(int32_t));
            put(s_icoef[pl][k], p->idct_coef[pl][k].data(), p->idct_coef[pl][k].size() * sizeof(int16_t));
        }
        if (p->any_edge[pl])
            put(s_edge[pl], p->edges[pl].data(), p->edges[pl].size() * sizeof(FFHipH264Edge));
    }
    /* bi-prediction scratch planes: same stride as the picture (the MC kernels take one stride for both operands) */
    for (int pl = 0; pl < 3; pl++) {
        if (!need_tmp[pl])
            continue;
        const size_t rows = (size_t)p->plane_h(pl), need = rows * (size_t)stride[pl] + 64;
        if (need > p->tmp_sz[pl]) {
            HIP_TRY(hipStreamSynchronize(stream));
            if (p->tmp[pl])
                (void)hipFree(p->tmp[pl]);
            p->tmp[pl] = nullptr;
            p->tmp_sz[pl] = 0;
            if (hipMalloc((void **)&p->tmp[pl], need) != hipSuccess) {
                ffhip_set_error("ffhip_h264_picture_flush: scratch plane hipMalloc(%zu) failed", need);
                return FFHIP_ENOMEM;
            }
            p->tmp_sz[pl] = need;
        }
    }
    if (total) {
        HIP_TRY(hipMemcpyAsync(db, hb, total, hipMemcpyHostToDevice, stream));
        HIP_TRY(hipEventRecord(p->copied, stream));
        p->copy_pending = true;
    }

    FlushBack B;
    for (int q = 0; q < nsets; q++)
        if (!p->intra[q].empty()) {
            /* 4:4:4: the plane as the "luma" of a luma-only wavefront (cb / cr are never touched there) */
            uint8_t *const y = p->cfmt == 3 ? dst[q] : dst[0];
            B.ip[B.nintra++] = FFHipH264IntraPic{ y, p->cfmt == 3 ? y : dst[1], p->cfmt == 3 ? y : dst[2], (const FFHipH264IntraMB *)(db + s_intra[q].off),
                                                  (const int32_t *)(db + s_irows[q].off), (const int16_t *)(db + s_intracoef[q].off) };
        }
    for (int pl = 0; pl < 3; pl++)
        if (p->any_edge[pl])
            B.edges[pl] = (const FFHipH264Edge *)(db + s_edge[pl].off);
    if (!p->intra_c422.empty()) {
        B.c422 = (const FFHipH264IntraC422 *)(db + s_c422.off);
        B.c422_rows = (const int32_t *)(db + s_c422rows.off);
        B.c422_coef = (const int16_t *)(db + s_c422coef.off);
    }

    int r = 0;
    if (p->bd > 8) {
        /* the same stages on the kernels templated on the sample type (kernels/h264_hbd.hip): one launch per list */
        const int bd = p->bd;
        for (int s = 0; s < 3 && r >= 0; s++) {
            for (int pl = 0; pl < 3 && r >= 0; pl++)
                if (s_qpel[pl][s].n)
                    r = ffhip_launch_h264_qpel_bd(bd, s == ST_TMP ? p->tmp[pl] : dst[pl], ref[pl], stride[pl], (const FFHipQpelBlock *)(db + s_qpel[pl][s].off),
                                                  s_qpel[pl][s].n, stream, p->plane_w(pl), p->plane_h(pl));
            for (int c = 0; c < 2 && r >= 0; c++)
                if (s_cmc[c][s].n)
                    r = ffhip_launch_h264_chroma_mc_bd(bd, s == ST_TMP ? p->tmp[1 + c] : dst[1 + c], ref[1 + c], stride[1 + c],
                                                       (const FFHipChromaBlock *)(db + s_cmc[c][s].off), s_cmc[c][s].n, stream, p->plane_w(1),
                                                       p->plane_h(1));
        }
        for (int pl = 0; pl < 3 && r >= 0; pl++)
            if (s_wt[pl].n)
                r = ffhip_launch_h264_weight_bd(bd, dst[pl], p->tmp[pl] ? p->tmp[pl] : dst[pl], stride[pl], (const FFHipWeightBlock *)(db + s_wt[pl].off),
                                                s_wt[pl].n, stream);
        for (int pl = 0; pl < 3 && r >= 0; pl++)
            for (int k = 0; k < 6 && r >= 0; k++) /* (4, 5: add_pixels4 / 8_clear of the lossless bypass) */
                if (s_ioff[pl][k].n)
                    r = ffhip_launch_h264_idct_add_bd(bd, k, dst[pl], stride[pl], (const int32_t *)(db + s_ioff[pl][k].off),
                                                      (int16_t *)(db + s_icoef[pl][k].off), s_ioff[pl][k].n, stream);
    } else {
        /* ---- prediction ---- */
        for (int s = 0; s < 3 && r >= 0; s++) {
            for (int pl = 0; pl < 3 && r >= 0; pl++)
                if (s_qpel[pl][s].n)
                    r = ffhip_launch_h264_qpel(s == ST_TMP ? p->tmp[pl] : dst[pl], ref[pl], stride[pl], (const FFHipQpelBlock *)(db + s_qpel[pl][s].off),
                                               s_qpel[pl][s].n, stream, p->plane_w(pl), p->plane_h(pl));
            if (r >= 0) { /* Cb and Cr of the stage: one launch */
                FFHipPlaneMulti M;
                M.nseg = 0;
                M.pic_w = p->plane_w(1); /* records flagged FFHIP_MC_EMU clamp to the reference pictures' chroma planes */
                M.pic_h = p->plane_h(1);
                for (int c = 0; c < 2; c++)
                    if (s_cmc[c][s].n) {
                        FFHipPlaneSeg &g = M.seg[M.nseg++];
                        g.dst = s == ST_TMP ? p->tmp[1 + c] : dst[1 + c]; g.src = ref[1 + c]; g.blocks = db + s_cmc[c][s].off;
                        g.stride = stride[1 + c]; g.n = (int)s_cmc[c][s].n; g.first = 0;
                    }
                r = ffhip_launch_h264_chroma_mc_multi(M, stream);
            }
        }
        if (r >= 0) { /* explicit weights of the three planes: one launch */
            FFHipPlaneMulti M;
            M.nseg = 0;
            for (int pl = 0; pl < 3; pl++)
                if (s_wt[pl].n) {
                    FFHipPlaneSeg &g = M.seg[M.nseg++];
                    g.dst = dst[pl]; g.src = p->tmp[pl] ? p->tmp[pl] : dst[pl]; g.blocks = db + s_wt[pl].off;
                    g.stride = stride[pl]; g.n = (int)s_wt[pl].n; g.first = 0;
                }
            r = ffhip_launch_h264_weight_multi(M, stream);
        }
        /* ---- residual ---- */
        if (r >= 0) {
            /* one launch for the three planes' four kinds: no block is named twice, so the lists are independent */
            FFHipIdctMulti M;
            M.nseg = 0;
            for (int pl = 0; pl < 3; pl++)
                for (int k = 0; k < 4; k++)
                    if (s_ioff[pl][k].n) {
                        FFHipIdctSeg &g = M.seg[M.nseg++];
                        g.dst = dst[pl]; g.offs = (const int32_t *)(db + s_ioff[pl][k].off); g.coef = (int16_t *)(db + s_icoef[pl][k].off);
                        g.stride = stride[pl]; g.n = (int)s_ioff[pl][k].n; g.kind = k; g.first = 0;
                    }
            r = ffhip_launch_h264_idct_multi(M, stream);
        }
        /* the lossless bypass of inter macroblocks: add_pixels4 / 8_clear (blocks of other macroblocks than the lists above) */
        for (int pl = 0; pl < 3 && r >= 0; pl++)
            for (int k = 4; k < 6 && r >= 0; k++)
                if (s_ioff[pl][k].n)
                    r = ffhip_launch_h264_idct_add(k, dst[pl], stride[pl], (const int32_t *)(db + s_ioff[pl][k].off), (int16_t *)(db + s_icoef[pl][k].off),
                                                   (int)s_ioff[pl][k].n, stream);
    }
    if (r < 0)
        return r;
    if (defer) {
        *defer = B;
        return 0;
    }
    return flush_tail(p, dst, stride, B, stream);
}

extern "C" int ffhip_h264_picture_flush(FFHipH264Picture *p, uint8_t *const dst[3], const int stride[3], const uint8_t *const ref[3],
                                        void *stream)
{
    int r;
    try { /* no C++ exception crosses the C boundary */
        r = flush_impl(p, dst, stride, ref, stream, nullptr);
    } catch (...) {
        ffhip_set_error("ffhip_h264_picture_flush: out of host memory");
        r = FFHIP_ENOMEM;
    }
    if (p)
        p->last_status = r < 0 ? r : 0;
    return r;
}

/* Several pictures together: every picture's own staging copy, prediction and residual launches (throughput kernels), then what is a
 * latency chain per picture — the intra reconstruction wavefront and the in-loop filter — ONCE for all of them, side by side. */
static int pictures_flush(FFHipH264Picture *const *pics, int n, uint8_t *const *dst, const int stride[3], const uint8_t *const *ref, void *stream_);

extern "C" int ffhip_h264_picture_status(const FFHipH264Picture *p) { return p ? p->last_status : FFHIP_EINVAL; }

extern "C" int ffhip_h264_pictures_flush(FFHipH264Picture *const *pics, int n, uint8_t *const *dst, const int stride[3], const uint8_t *const *ref,
                                         void *stream_)
{
    /* no C++ exception crosses the C boundary: the host side of a flush allocates (vectors, strings) */
    try {
        return pictures_flush(pics, n, dst, stride, ref, stream_);
    } catch (const std::bad_alloc &) {
        ffhip_set_error("ffhip_h264_pictures_flush: out of host memory");
        return FFHIP_ENOMEM;
    } catch (...) {
        ffhip_set_error("ffhip_h264_pictures_flush: unexpected failure on the host side");
        return FFHIP_EIO;
    }
}

static int pictures_flush(FFHipH264Picture *const *pics, int n, uint8_t *const *dst, const int stride[3], const uint8_t *const *ref, void *stream_)
{
    if (n < 0 || (n && (!pics || !dst || !stride || !ref)))
        return FFHIP_EINVAL;
    if (n == 0)
        return 0;
    for (int i = 0; i < n; i++)
        if (!pics[i] || pics[i]->mb_w != pics[0]->mb_w || pics[i]->mb_h != pics[0]->mb_h || pics[i]->bd != pics[0]->bd ||
            pics[i]->cfmt != pics[0]->cfmt || pics[i]->device != pics[0]->device) {
            ffhip_set_error("ffhip_h264_pictures_flush: the pictures of a batch share geometry, chroma format, depth and device");
            return FFHIP_EINVAL;
        }
    for (int i = 0; i < n; i++)
        for (int j = 0; j < i; j++)
            if (pics[i] == pics[j]) {
                ffhip_set_error("ffhip_h264_pictures_flush: picture object %d appears twice", i);
                return FFHIP_EINVAL;
            }
    for (int i = 0; i < n; i++)
        pics[i]->last_status = 0;
    if (n == 1)
        return pics[0]->last_status = flush_impl(pics[0], dst, stride, ref, stream_, nullptr);
    if (pics[0]->cfmt == 3 && (stride[1] != stride[0] || stride[2] != stride[0])) {
        ffhip_set_error("ffhip_h264_pictures_flush: the planes of 4:4:4 pictures share one stride");
        return FFHIP_EINVAL;
    }
    /* what the shared in-loop filter launches would refuse is refused before anything is queued (the stages modify dst in place): they
     * address the pictures by table, which only the skewed-rows kernel takes — 16-byte aligned planes and strides (8 for 8-bit chroma) */
    for (int i = 0; i < n; i++)
        for (int pl = 0; pl < 3; pl++) {
            if (!pics[i]->any_edge[pl])
                continue;
            const uintptr_t amask = pl && pics[0]->bd == 8 && pics[0]->cfmt != 3 ? 7 : 15;
            if (!dst[3 * i + pl] || (((uintptr_t)dst[3 * i + pl] | (uintptr_t)stride[pl]) & amask) || (pl == 2 && stride[1] != stride[2])) {
                ffhip_set_error("ffhip_h264_pictures_flush: plane %d of picture %d carries deblocking records: plane and stride must be %d-byte "
                                "aligned (Cb and Cr share a stride)", pl, i, (int)amask + 1);
                return FFHIP_EINVAL;
            }
        }
    FFHipH264Picture *const p0 = pics[0];
    FFHipDeviceGuard dg(p0->device);
    hipStream_t stream = (hipStream_t)stream_;
    const int bd = p0->bd, mb_w = p0->mb_w, mb_h = p0->mb_h;
    std::vector<FlushBack> B((size_t)n);
    int first_err = 0;
    {
        /* the front half of a flush is host work — sorting a picture's intra records, copying megabytes of records and coefficients
         * into its pinned buffer — before a handful of launches: up to eight pictures are prepared at a time by threads of this call
         * (launch order between the pictures is free; the stages that follow wait for all of them) */
        /* (two pictures or fewer: on the caller's thread — starting a thread costs more than it hides there) */
        const int nthr = n <= 2 ? 1 : n < 8 ? n : 8;
        std::vector<std::string> errs((size_t)n);
        std::vector<std::thread> th;
        auto work = [&](int t) {
            for (int i = t; i < n; i += nthr) {
                int r;
                try {
                    r = flush_impl(pics[i], dst + 3 * i, stride, ref + 3 * i, stream_, &B[(size_t)i]);
                } catch (...) {
                    r = FFHIP_ENOMEM;
                }
                if (r < 0) { /* this picture is out; the others of the share go on (their tails run below) */
                    pics[i]->last_status = r;
                    errs[(size_t)i] = r == FFHIP_ENOMEM && !ffhip_last_error()[0] ? "out of host memory" : ffhip_last_error(); /* (the error text is per thread) */
                    B[(size_t)i] = FlushBack();
                }
            }
        };
        int started = 0;
        try {
            for (; started < nthr - 1; started++)
                th.emplace_back(work, started);
        } catch (...) { /* no more threads to be had: this one takes the shares that found none */
        }
        for (int t = started; t < nthr; t++)
            work(t);
        for (std::thread &t : th)
            t.join();
        /* A picture whose front half failed is left out of what follows — its planes are incomplete and its status says why
         * (ffhip_h264_picture_status) — while every other picture of the batch is finished: they have had their prediction and residual
         * stages queued against their planes, and stopping here would leave those half reconstructed.  The call returns the first failure. */
        for (int i = 0; i < n && !first_err; i++)
            if (pics[i]->last_status < 0) {
                first_err = pics[i]->last_status;
                ffhip_set_error("ffhip_h264_pictures_flush: picture %d: %s", i, errs[(size_t)i].c_str());
            }
    }
    /* from here on the stages are shared by the batch: whichever way the function leaves before its last line, no picture is known to be
     * complete and every picture that was still good says so (ffhip_h264_picture_status) */
    struct SharedStageGuard {
        FFHipH264Picture *const *pics; int n; bool armed = true;
        ~SharedStageGuard() {
            if (armed)
                for (int i = 0; i < n; i++)
                    if (!pics[i]->last_status)
                        pics[i]->last_status = FFHIP_EIO;
        }
    } shared_guard{ pics, n };
    const bool c444 = p0->cfmt == 3; /* every plane is a luma plane: wavefronts and filters alike */
    const bool c422 = p0->cfmt == 2; /* 8 x 16 chroma: the luma plane through the luma kernels, the chroma planes through kernels/h264_c422.hip */
    std::vector<FFHipH264IntraPic> ip;
    std::vector<FFHipH264C422Pic> ic;
    for (int i = 0; i < n; i++) {
        for (int q = 0; q < B[i].nintra; q++)
            ip.push_back(B[i].ip[q]);
        if (B[i].c422)
            ic.push_back(FFHipH264C422Pic{ dst[3 * i + 1], dst[3 * i + 2], B[i].c422, B[i].c422_rows, B[i].c422_coef });
    }
    int r = 0;
    if (!ic.empty()) { /* the chroma planes' wavefronts of all pictures beside the luma ones, on the first object's second stream */
        if (stride[1] != stride[2]) {
            ffhip_set_error("ffhip_h264_pictures_flush: Cb and Cr share a stride");
            return FFHIP_EINVAL;
        }
        HIP_TRY(hipEventRecord(p0->fork, stream));
        HIP_TRY(hipStreamWaitEvent(p0->aux, p0->fork, 0));
        ffhip_progress_report_to(stream, true);
        r = ffhip_launch_h264_intra_c422_pics(bd, (int)ic.size(), ic.data(), stride[1], mb_w, mb_h, p0->aux);
        ffhip_progress_report_to(nullptr, false);
        HIP_TRY(hipEventRecord(p0->join, p0->aux));
    }
    if (r >= 0 && !ip.empty())
        r = ffhip_launch_h264_intra_frames_bd(bd, (int)ip.size(), ip.data(), stride[0], stride[1], mb_w, mb_h, stream, c444 || c422);
    if (!ic.empty())
        HIP_TRY(hipStreamWaitEvent(stream, p0->join, 0));
    if (r < 0)
        return r;
    /* the in-loop filter: all chroma planes (Cb and Cr of every picture: up to 2 n "pictures") on the first object's second stream
     * beside all luma planes on the caller's */
    std::vector<uint8_t *> pl_c, pl_y;
    std::vector<const FFHipH264Edge *> ed_c, ed_y;
    for (int i = 0; i < n; i++) {
        for (int pl = 1; pl < 3; pl++)
            if (B[i].edges[pl]) {
                (c444 ? pl_y : pl_c).push_back(dst[3 * i + pl]);
                (c444 ? ed_y : ed_c).push_back(B[i].edges[pl]);
            }
        if (B[i].edges[0]) {
            pl_y.push_back(dst[3 * i]);
            ed_y.push_back(B[i].edges[0]);
        }
    }
    if (!pl_c.empty() && stride[1] != stride[2]) {
        ffhip_set_error("ffhip_h264_pictures_flush: Cb and Cr share a stride");
        return FFHIP_EINVAL;
    }

    if (!pl_c.empty()) {
        HIP_TRY(hipEventRecord(p0->fork, stream));
        HIP_TRY(hipStreamWaitEvent(p0->aux, p0->fork, 0));
        ffhip_progress_report_to(stream, true);
        r = c422 ? ffhip_launch_h264_deblock_c422_planes(bd, (int)pl_c.size(), pl_c.data(), ed_c.data(), stride[1], mb_w, mb_h, p0->aux)
                 : ffhip_launch_h264_deblock_pictures_bd(bd, 1, pl_c.data(), ed_c.data(), (int)pl_c.size(), stride[1], mb_w, mb_h, p0->aux);
        ffhip_progress_report_to(nullptr, false);
        HIP_TRY(hipEventRecord(p0->join, p0->aux));
    }
    if (r >= 0 && !pl_y.empty())
        r = ffhip_launch_h264_deblock_pictures_bd(bd, 0, pl_y.data(), ed_y.data(), (int)pl_y.size(), stride[0], mb_w, mb_h, stream);
    if (!pl_c.empty())
        HIP_TRY(hipStreamWaitEvent(stream, p0->join, 0));
    if (r < 0) { /* a shared stage failed: no picture of the batch is known to be complete */
        for (int i = 0; i < n; i++)
            if (!pics[i]->last_status)
                pics[i]->last_status = r;
        return r;
    }
    shared_guard.armed = false;
    return first_err;
}
