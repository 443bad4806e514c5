/*
 * hevc_mc.hip — HEVC motion compensation, 8-bit, batched: put_hevc_{qpel,epel}[..][!!my][!!mx], _uni, _uni_w,
 * _bi and _bi_w (libavcodec/h26x/h2656_inter_template.c:29-88,97-340,342-578; libavcodec/hevc/dsp_template.c:368-815).
 *
 * One wave per prediction block, four blocks per workgroup.  A lane owns 4 adjacent samples of a row:
 *   horizontal pass   the 11 (chroma: 7) source bytes of the four windows arrive as 3 (2) dword loads at the block's own byte
 *                     alignment; bytes are biased by -128 so that v_dot4_i32_i8 applies (signed taps x signed bytes; every filter
 *                     sums to 64, so the bias is a constant 8192 seeded into the accumulator), windows are cut with v_alignbyte:
 *                     2 dot4 per sample instead of 8 loads + 8 multiply-adds;
 *   vertical pass     the rows it needs — horizontally filtered (hv) or raw (v only) — sit in wave-private LDS as int16 PAIRS of
 *                     vertically adjacent rows (one dword = rows 2q, 2q+1 of a column), so an output is 4 (2) v_dot2_i32_i16 when
 *                     its first row is even and 5 (3) with a one-row-shifted coefficient set when it is odd: no repacking;
 *   output stage      the mode's rounding / weighting on the four 14-bit values, one dword (put: 8-byte) store per lane when the
 *                     destination allows.
 * Rows are read in whole dwords: a block's last group may read up to 3 bytes beyond the right margin the reference needs (and up to
 * 3 int16 beyond `width` in a src2 row) — inside any frame allocation with the usual line padding.
 */
#include <type_traits>

#include "common.h"
#include "h264_kernels.h"

typedef short mc_s2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void hevc_wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

static_assert(sizeof(FFHipHevcMcBlock) == 12, "FFHipHevcMcBlock is a 12-byte record");
/* 16-byte aligned: the tuned kernel reads a row of taps as one or two dwords (scalar loads) */
__constant__ __attribute__((aligned(16))) int8_t hevc_lf8[4][8] = { { 0 }, { -1, 4, -10, 58, 17, -5, 1, 0 }, { -1, 4, -11, 40, 40, -11, 4, -1 }, { 0, 1, -5, 17, 58, -10, 4, -1 } };
__constant__ __attribute__((aligned(16))) int8_t hevc_cf4[8][4] = { { 0 }, { -2, 58, 10, -2 }, { -4, 54, 16, -2 }, { -6, 46, 28, -4 }, { -4, 36, 36, -4 },
                                       { -4, 28, 46, -6 }, { -2, 16, 54, -4 }, { -2, 10, 58, -2 } };


/* the vertical taps as v_dot2_i32_i16 operands over row pairs (generated: pk(a, b) = a | b << 16):
 * [my][0..NP-1] first row even: (c0,c1)(c2,c3)..(0,0);  [my][NP..2NP-1] first row odd: (0,c0)(c1,c2)..(c_last,0) */
__constant__ __attribute__((aligned(16))) uint32_t hevc_lv2[4][12] = {
    { 0x00000000u, 0x00000000u, 0x00000000u, 0x00000000u, 0x00000000u, 0x00000000u, 0x00000000u, 0x00000000u, 0x00000000u, 0x00000000u, 0x00000000u, 0x00000000u },
    { 0x0004ffffu, 0x003afff6u, 0xfffb0011u, 0x00000001u, 0x00000000u, 0xffff0000u, 0xfff60004u, 0x0011003au, 0x0001fffbu, 0x00000000u, 0x00000000u, 0x00000000u },
    { 0x0004ffffu, 0x0028fff5u, 0xfff50028u, 0xffff0004u, 0x00000000u, 0xffff0000u, 0xfff50004u, 0x00280028u, 0x0004fff5u, 0x0000ffffu, 0x00000000u, 0x00000000u },
    { 0x00010000u, 0x0011fffbu, 0xfff6003au, 0xffff0004u, 0x00000000u, 0x00000000u, 0xfffb0001u, 0x003a0011u, 0x0004fff6u, 0x0000ffffu, 0x00000000u, 0x00000000u } };
__constant__ __attribute__((aligned(16))) uint32_t hevc_cv2[8][8] = {
    { 0x00000000u, 0x00000000u, 0x00000000u, 0x00000000u, 0x00000000u, 0x00000000u, 0x00000000u, 0x00000000u },
    { 0x003afffeu, 0xfffe000au, 0x00000000u, 0xfffe0000u, 0x000a003au, 0x0000fffeu, 0x00000000u, 0x00000000u },
    { 0x0036fffcu, 0xfffe0010u, 0x00000000u, 0xfffc0000u, 0x00100036u, 0x0000fffeu, 0x00000000u, 0x00000000u },
    { 0x002efffau, 0xfffc001cu, 0x00000000u, 0xfffa0000u, 0x001c002eu, 0x0000fffcu, 0x00000000u, 0x00000000u },
    { 0x0024fffcu, 0xfffc0024u, 0x00000000u, 0xfffc0000u, 0x00240024u, 0x0000fffcu, 0x00000000u, 0x00000000u },
    { 0x001cfffcu, 0xfffa002eu, 0x00000000u, 0xfffc0000u, 0x002e001cu, 0x0000fffau, 0x00000000u, 0x00000000u },
    { 0x0010fffeu, 0xfffc0036u, 0x00000000u, 0xfffe0000u, 0x00360010u, 0x0000fffcu, 0x00000000u, 0x00000000u },
    { 0x000afffeu, 0xfffe003au, 0x00000000u, 0xfffe0000u, 0x003a000au, 0x0000fffeu, 0x00000000u, 0x00000000u } };
/* ceil(65536 / ng), ng = 1..16 groups of 4 samples per row: i / ng == (i * inv) >> 16 for i < 4096 */
__constant__ uint32_t hevc_mc_inv[17] = { 0, 65536, 32768, 21846, 16384, 13108, 10923, 9363, 8192, 7282, 6554, 5958, 5462, 5042, 4682, 4370, 4096 };

constexpr int MC_TW = 16;    /* the vertical pass works on column tiles of 16: 2960 bytes of LDS per wave keep 8 waves per SIMD resident
                              * (measured: 1.5x over a full 64-column plane, which capped the CU at 16 waves) */
constexpr int MC_PITCH = 20; /* dwords per row pair: MC_TW columns + 4 (16-byte rows, lanes 16 bytes apart) */
constexpr int MC_PAIRS = 37; /* (64 + 7 + 1) / 2 row pairs + the spare one an even first row reads with a zero coefficient */

/* four horizontally filtered samples: p = the first window's first byte (x0 - BEFORE), any alignment */
template <bool CHROMA>
__device__ __forceinline__ void mc_hrow4(const uint8_t *p, int clo, int chi, int (&o)[4])
{
    const uint32_t *q = reinterpret_cast<const uint32_t *>(p);
    const uint32_t d0 = q[0] ^ 0x80808080u, d1 = q[1] ^ 0x80808080u;
    if (CHROMA) {
        o[0] = __builtin_amdgcn_sdot4((int)d0, clo, 8192, false);
        o[1] = __builtin_amdgcn_sdot4((int)__builtin_amdgcn_alignbyte(d1, d0, 1), clo, 8192, false);
        o[2] = __builtin_amdgcn_sdot4((int)__builtin_amdgcn_alignbyte(d1, d0, 2), clo, 8192, false);
        o[3] = __builtin_amdgcn_sdot4((int)__builtin_amdgcn_alignbyte(d1, d0, 3), clo, 8192, false);
    } else {
        const uint32_t d2 = q[2] ^ 0x80808080u;
        o[0] = __builtin_amdgcn_sdot4((int)d1, chi, __builtin_amdgcn_sdot4((int)d0, clo, 8192, false), false);
        o[1] = __builtin_amdgcn_sdot4((int)__builtin_amdgcn_alignbyte(d2, d1, 1), chi,
                                      __builtin_amdgcn_sdot4((int)__builtin_amdgcn_alignbyte(d1, d0, 1), clo, 8192, false), false);
        o[2] = __builtin_amdgcn_sdot4((int)__builtin_amdgcn_alignbyte(d2, d1, 2), chi,
                                      __builtin_amdgcn_sdot4((int)__builtin_amdgcn_alignbyte(d1, d0, 2), clo, 8192, false), false);
        o[3] = __builtin_amdgcn_sdot4((int)__builtin_amdgcn_alignbyte(d2, d1, 3), chi,
                                      __builtin_amdgcn_sdot4((int)__builtin_amdgcn_alignbyte(d1, d0, 3), clo, 8192, false), false);
    }
}

__device__ __forceinline__ uint32_t mc_pk16(int lo, int hi) { return ((uint32_t)lo & 0xffffu) | ((uint32_t)hi << 16); }

/* MODE 0 put (int16), 1 uni, 2 uni_w, 3 bi, 4 bi_w; modes 2..4 read the 24-byte weighted record */
/* SKIP16: the 16 x 16 blocks of the batch are k_hevc_qpel_m's (hevc_qpel_m.hip) — this launch leaves them alone */
template <bool CHROMA, int MODE, bool SKIP16 = false>
__global__ __launch_bounds__(256) void k_hevc_mc(void *dst_, ptrdiff_t dststride, const uint8_t *src, ptrdiff_t srcstride,
                                                 const int16_t *src2, const void *blocks_, int n)
{
    constexpr int TAPS = CHROMA ? 4 : 8, BEFORE = CHROMA ? 1 : 3, NP = TAPS / 2 + 1, FMASK = CHROMA ? 7 : 3;
    using Rec = typename std::conditional<(MODE >= 2), FFHipHevcMcWBlock, FFHipHevcMcBlock>::type;
    __shared__ uint32_t tmp_all[4][MC_PAIRS * MC_PITCH];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    /* SKIP16: a launch of a few thousand workgroups walks the whole batch for the blocks k_hevc_qpel_m left (a wave reads a record and
     * moves on when it is a 16 x 16 one); otherwise one block per wave */
    /* SKIP16: the wave looks at 64 records at a time — a lane each — and works through the ones that are left to this kernel (a record
     * per dependent load was 45 us for a quarter of a million blocks; now 5) */
    for (int c0 = SKIP16 ? blockIdx.x * 64 : blockIdx.x * 4 + wave; c0 < n; c0 += SKIP16 ? (int)gridDim.x * 64 : n) {
    unsigned long todo = 1;
    int turn = 0;
    if (SKIP16) {
        const int bl = c0 + lane;
        const Rec &rr = static_cast<const Rec *>(blocks_)[min(bl, n - 1)];
        todo = __ballot(bl < n && !(rr.width == 16 && rr.height == 16));
    }
    while (todo) {
    const int b = SKIP16 ? c0 + (int)__builtin_ctzl(todo) : c0;
    todo &= todo - 1;
    if (SKIP16 && (turn++ & 3) != wave)
        continue; /* the four waves of the workgroup look at the same 64 records and take turns */
    const Rec k = static_cast<const Rec *>(blocks_)[b];
    const int w = __builtin_amdgcn_readfirstlane((int)k.width), h = __builtin_amdgcn_readfirstlane((int)k.height);
    const int mx = __builtin_amdgcn_readfirstlane((int)k.mx) & FMASK, my = __builtin_amdgcn_readfirstlane((int)k.my) & FMASK;
    const uint8_t *s = src + __builtin_amdgcn_readfirstlane(k.src_offset);
    const int dofs = __builtin_amdgcn_readfirstlane(k.dst_offset);
    const int ng = (w + 3) >> 2, inv = (int)hevc_mc_inv[ng];
    uint32_t *tmp = tmp_all[wave];

    /* the taps as the dot instructions want them */
    const int hlo = CHROMA ? reinterpret_cast<const int *>(hevc_cf4)[mx] : reinterpret_cast<const int *>(hevc_lf8)[2 * mx];
    const int hhi = CHROMA ? 0 : reinterpret_cast<const int *>(hevc_lf8)[2 * mx + 1];
    uint32_t ce[NP], co[NP];
#pragma unroll
    for (int q = 0; q < NP; q++) {
        ce[q] = CHROMA ? hevc_cv2[my][q] : hevc_lv2[my][q];
        co[q] = CHROMA ? hevc_cv2[my][NP + q] : hevc_lv2[my][NP + q];
    }

    int wx0 = 0, wx1 = 0, wofs = 0, wsh = 0, ox = 0;
    const int16_t *s2 = nullptr;
    bool s2_al = false;
    if constexpr (MODE >= 2) {
        wx0 = __builtin_amdgcn_readfirstlane((int)k.wx0); wx1 = __builtin_amdgcn_readfirstlane((int)k.wx1);
        ox = __builtin_amdgcn_readfirstlane((int)k.ox);
        wsh = __builtin_amdgcn_readfirstlane((int)k.denom) + 6; /* uni_w: shift = denom + 14 - 8;  bi_w: log2Wd = denom + 6 */
        wofs = MODE == 2 ? 1 << (wsh - 1) : (ox + 1) << wsh;
        s2 = src2 + __builtin_amdgcn_readfirstlane(k.src2_offset);
        s2_al = (reinterpret_cast<uintptr_t>(s2) & 7) == 0;
    }
    int16_t *d16 = static_cast<int16_t *>(dst_) + dofs;
    uint8_t *d8 = static_cast<uint8_t *>(dst_) + dofs;
    const bool d_al = MODE == 0 ? (reinterpret_cast<uintptr_t>(d16) & 7) == 0
                                : ((reinterpret_cast<uintptr_t>(d8) | (uintptr_t)dststride) & 3) == 0;

    /* the output stage for samples x0..x0+3 of row y (those below w) */
    auto emit = [&](int y, int x0, const int (&val)[4]) {
        const int cnt = w - x0;
        if constexpr (MODE == 0) {
            int16_t *d = d16 + y * 64 + x0;
            if (cnt >= 4 && d_al) {
                *reinterpret_cast<uint2 *>(d) = make_uint2(mc_pk16(val[0], val[1]), mc_pk16(val[2], val[3]));
            } else {
#pragma unroll
                for (int j = 0; j < 4; j++)
                    if (j < cnt)
                        d[j] = (int16_t)val[j];
            }
        } else {
            int o2[4] = { 0, 0, 0, 0 };
            if constexpr (MODE >= 3) {
                const int16_t *q = s2 + y * 64 + x0;
                if (s2_al) {
                    const uint2 v = *reinterpret_cast<const uint2 *>(q);
                    o2[0] = (int16_t)(v.x & 0xffff); o2[1] = (int)v.x >> 16; o2[2] = (int16_t)(v.y & 0xffff); o2[3] = (int)v.y >> 16;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; j++)
                        o2[j] = q[j];
                }
            }
            int out[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                int v;
                if constexpr (MODE == 1)
                    v = (val[j] + 32) >> 6;
                else if constexpr (MODE == 2)
                    v = ((val[j] * wx0 + wofs) >> wsh) + ox;
                else if constexpr (MODE == 3)
                    v = (val[j] + o2[j] + 64) >> 7;
                else
                    v = (val[j] * wx1 + o2[j] * wx0 + wofs) >> (wsh + 1);
                out[j] = clip_u8(v);
            }
            uint8_t *d = d8 + (ptrdiff_t)y * dststride + x0;
            if (cnt >= 4 && d_al) {
                *reinterpret_cast<uint32_t *>(d) = (uint32_t)out[0] | (uint32_t)out[1] << 8 | (uint32_t)out[2] << 16 | (uint32_t)out[3] << 24;
            } else {
#pragma unroll
                for (int j = 0; j < 4; j++)
                    if (j < cnt)
                        d[j] = (uint8_t)out[j];
            }
        }
    };

    if (my) {
      for (int tx = 0; tx < w; tx += MC_TW) { /* column tiles; the horizontal pass costs the same per sample however the block is cut */
        const int ngt = min(ng - (tx >> 2), MC_TW / 4), invt = (int)hevc_mc_inv[ngt];
        /* rows -BEFORE .. h+TAPS-2-BEFORE of the tile, horizontally filtered (or raw), into the row-pair plane */
        const int items = (h + TAPS - 1) * ngt;
        for (int i = lane; i < items; i += 64) {
            const int r = (i * invt) >> 16, xg = i - r * ngt;
            const uint8_t *p = s + (ptrdiff_t)(r - BEFORE) * srcstride + tx + 4 * xg;
            int o[4];
            if (mx) {
                mc_hrow4<CHROMA>(p - BEFORE, hlo, hhi, o);
            } else {
                const uint32_t d = *reinterpret_cast<const uint32_t *>(p);
                o[0] = d & 255; o[1] = (d >> 8) & 255; o[2] = (d >> 16) & 255; o[3] = d >> 24;
            }
            int16_t *t16 = reinterpret_cast<int16_t *>(tmp + (r >> 1) * MC_PITCH + 4 * xg) + (r & 1);
            t16[0] = (int16_t)o[0]; t16[2] = (int16_t)o[1]; t16[4] = (int16_t)o[2]; t16[6] = (int16_t)o[3];
        }
        hevc_wave_sync();
        for (int i = lane; i < h * ngt; i += 64) {
            const int y = (i * invt) >> 16, xg = i - y * ngt;
            const uint4 *t = reinterpret_cast<const uint4 *>(tmp + (y >> 1) * MC_PITCH + 4 * xg);
            const bool odd = y & 1;
            int acc[4] = { 0, 0, 0, 0 };
#pragma unroll
            for (int q = 0; q < NP; q++) {
                const uint4 v = t[q * (MC_PITCH / 4)];
                const mc_s2 c = __builtin_bit_cast(mc_s2, odd ? co[q] : ce[q]);
                acc[0] = __builtin_amdgcn_sdot2(__builtin_bit_cast(mc_s2, v.x), c, acc[0], false);
                acc[1] = __builtin_amdgcn_sdot2(__builtin_bit_cast(mc_s2, v.y), c, acc[1], false);
                acc[2] = __builtin_amdgcn_sdot2(__builtin_bit_cast(mc_s2, v.z), c, acc[2], false);
                acc[3] = __builtin_amdgcn_sdot2(__builtin_bit_cast(mc_s2, v.w), c, acc[3], false);
            }
            if (mx) {
#pragma unroll
                for (int j = 0; j < 4; j++)
                    acc[j] >>= 6;
            }
            emit(y, tx + 4 * xg, acc);
        }
        hevc_wave_sync(); /* the next tile overwrites the plane */
      }
    } else {
        for (int i = lane; i < h * ng; i += 64) {
            const int y = (i * inv) >> 16, xg = i - y * ng;
            const uint8_t *p = s + (ptrdiff_t)y * srcstride + 4 * xg;
            int o[4];
            if (mx) {
                mc_hrow4<CHROMA>(p - BEFORE, hlo, hhi, o);
            } else {
                const uint32_t d = *reinterpret_cast<const uint32_t *>(p);
                o[0] = (d & 255) << 6; o[1] = ((d >> 8) & 255) << 6; o[2] = ((d >> 16) & 255) << 6; o[3] = (d >> 24) << 6;
            }
            emit(y, 4 * xg, o);
        }
    }
    if (SKIP16)
        hevc_wave_sync(); /* the next block reuses the plane */
    }
    }
}

/*
 * The first kernel of this row, kept as the A/B reference (FFHIP_HEVC_MC_OLD=1): lanes sweep the block's samples one by one.
 * MODE 0 put (int16), 1 uni, 2 uni_w, 3 bi, 4 bi_w (put_hevc_{qpel,epel}_{uni_w,bi,bi_w}: h26x/h2656_inter_template.c:60-88,247-340,
 * 487-578; hevc/dsp_template.c:368-420,432-625,630-815 — the same interpolation, a different output stage).  Modes 2..4 read the
 * 24-byte weighted record.
 */
static_assert(sizeof(FFHipHevcMcWBlock) == 24, "FFHipHevcMcWBlock is a 24-byte record");
template <typename PIX, bool CHROMA, int MODE>
__global__ __launch_bounds__(256) void k_hevc_mc_s(void *dst_, ptrdiff_t dststride, const uint8_t *src, ptrdiff_t srcstride,
                                                 const int16_t *src2, const void *blocks_, int n, int bd)
{
    /* PIX = uint8_t (bd 8) / uint16_t (bd 10, 12; strides and offsets stay in bytes).  Above 8 bits the one-dimensional sums drop
     * bd - 8 bits, the unfiltered copy gains 14 - bd, and every output stage shifts by its 8-bit amount minus (bd - 8), offsets
     * scaled by << (bd - 8) (h2656_inter_template.c:29-88,113-245; hevc/dsp_template.c:368-440) */
    constexpr bool UNI = MODE == 1;
    using Rec = typename std::conditional<(MODE >= 2), FFHipHevcMcWBlock, FFHipHevcMcBlock>::type;
    const Rec *blocks = static_cast<const Rec *>(blocks_);
    constexpr int TAPS = CHROMA ? 4 : 8, BEFORE = CHROMA ? 1 : 3;
    __shared__ int16_t tmp_all[4][(64 + TAPS - 1) * 64];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + wave;
    if (b >= n)
        return;
    const Rec k = blocks[b];
    const int w = k.width, h = k.height, mx = k.mx & (CHROMA ? 7 : 3), my = k.my & (CHROMA ? 7 : 3);
    const int sh1 = bd - 8, shu = 14 - bd, maxv = (1 << bd) - 1;
    const ptrdiff_t sst = srcstride / (ptrdiff_t)sizeof(PIX);
    int wx0 = 0, wx1 = 0, wofs = 0, wsh = 0, ox = 0;
    const int16_t *s2 = nullptr;
    if constexpr (MODE >= 2) {
        wx0 = k.wx0; wx1 = k.wx1; ox = k.ox * (1 << sh1);
        wsh = k.denom + shu;                                 /* uni_w: shift = denom + 14 - bd;  bi_w: log2Wd = denom + 14 - bd */
        wofs = MODE == 2 ? 1 << (wsh - 1) : (ox + 1) << wsh;
        s2 = src2 + k.src2_offset;
    }
    const PIX *s = reinterpret_cast<const PIX *>(src + k.src_offset);
    int16_t *tmp = tmp_all[wave];
    int hf[TAPS], vf[TAPS];
#pragma unroll
    for (int t = 0; t < TAPS; t++) {
        hf[t] = CHROMA ? hevc_cf4[mx][t] : hevc_lf8[mx][t];
        vf[t] = CHROMA ? hevc_cf4[my][t] : hevc_lf8[my][t];
    }
    if (mx && my) {
        for (int i = lane; i < w * (h + TAPS - 1); i += 64) {
            const int r = i / w, x = i - r * w;
            const PIX *p = s + (ptrdiff_t)(r - BEFORE) * sst + x - BEFORE;
            int acc = 0;
#pragma unroll
            for (int t = 0; t < TAPS; t++)
                acc += hf[t] * p[t];
            tmp[r * 64 + x] = (int16_t)(acc >> sh1);
        }
        hevc_wave_sync();
    }
    for (int i = lane; i < w * h; i += 64) {
        const int y = i / w, x = i - y * w;
        const PIX *p = s + (ptrdiff_t)y * sst + x;
        int val;
        if (!mx && !my) {
            val = p[0] << shu;
        } else if (!my) {
            val = 0;
#pragma unroll
            for (int t = 0; t < TAPS; t++)
                val += hf[t] * p[t - BEFORE];
            val >>= sh1;
        } else if (!mx) {
            val = 0;
#pragma unroll
            for (int t = 0; t < TAPS; t++)
                val += vf[t] * p[(ptrdiff_t)(t - BEFORE) * sst];
            val >>= sh1;
        } else {
            int acc = 0;
#pragma unroll
            for (int t = 0; t < TAPS; t++)
                acc += vf[t] * tmp[(y + t) * 64 + x];
            val = acc >> 6;
        }
        if constexpr (MODE >= 1) {
            PIX *d = reinterpret_cast<PIX *>(static_cast<uint8_t *>(dst_) + k.dst_offset + (ptrdiff_t)y * dststride) + x;
            int out;
            if constexpr (UNI)
                out = (!mx && !my) ? p[0] : (val + (1 << (shu - 1))) >> shu;
            else if constexpr (MODE == 2)
                out = ((val * wx0 + wofs) >> wsh) + ox;
            else if constexpr (MODE == 3)
                out = (val + s2[y * 64 + x] + (1 << shu)) >> (shu + 1);
            else
                out = (val * wx1 + s2[y * 64 + x] * wx0 + wofs) >> (wsh + 1);
            *d = (PIX)min(max(out, 0), maxv);
        } else {
            static_cast<int16_t *>(dst_)[(ptrdiff_t)k.dst_offset + y * 64 + x] = (int16_t)val;
        }
    }
}

template <bool CHROMA>
static void hevc_mc_launch(int mode, bool old, void *dst, ptrdiff_t dststride, const uint8_t *src, ptrdiff_t srcstride, const int16_t *src2,
                           const void *blocks, int n, hipStream_t stream)
{
    const dim3 grid(cdiv(n, 4)), block(256);
#define MC_CASE(M)                                                                                                                  \
    case M:                                                                                                                         \
        if (old) hipLaunchKernelGGL((k_hevc_mc_s<uint8_t, CHROMA, M>), grid, block, 0, stream, dst, dststride, src, srcstride, src2, blocks, n, 8); \
        else     hipLaunchKernelGGL((k_hevc_mc<CHROMA, M>), grid, block, 0, stream, dst, dststride, src, srcstride, src2, blocks, n);   \
        break;
    switch (mode) {
    MC_CASE(0) MC_CASE(1) MC_CASE(2) MC_CASE(3)
    default:
    MC_CASE(4)
    }
#undef MC_CASE
}

/* mode 0/1: blocks are FFHipHevcMcBlock, src2 unused; mode 2..4: FFHipHevcMcWBlock */
int ffhip_launch_hevc_mc(int chroma, int mode, void *dst, ptrdiff_t dststride, const uint8_t *src, ptrdiff_t srcstride, const int16_t *src2,
                         const void *blocks, int n, hipStream_t stream)
{
    if (n <= 0)
        return 0;
    const char *e = FFHIP_KNOB("FFHIP_HEVC_MC_OLD");
    const bool old = e && e[0] == '1';
    const char *em = FFHIP_KNOB("FFHIP_HEVC_MC_M"); /* measured variant: 0 = without the matrix-core kernel */
    if (!chroma && !old && !(em && em[0] == '0') && ffhip_hevc_qpel_m_ok(mode, dststride, srcstride)) {
        /* luma: the 16 x 16 blocks on the matrix cores, everything else in a second launch that skips those */
        ffhip_launch_hevc_qpel_m(mode, dst, dststride, src, srcstride, src2, blocks, n, stream);
        const dim3 grid(min(cdiv(n, 64), 32768)), block(256); /* 64 records per wave and step; enough waves for a batch that is all other sizes */
#define MC_SKIP(M) case M: hipLaunchKernelGGL((k_hevc_mc<false, M, true>), grid, block, 0, stream, dst, dststride, src, srcstride, src2, blocks, n); break;
        switch (mode) {
        MC_SKIP(0) MC_SKIP(1) MC_SKIP(2) MC_SKIP(3)
        default:
        MC_SKIP(4)
        }
#undef MC_SKIP
        LAUNCH_CHECK();
        return 0;
    }
    if (chroma)
        hevc_mc_launch<true>(mode, old, dst, dststride, src, srcstride, src2, blocks, n, stream);
    else
        hevc_mc_launch<false>(mode, old, dst, dststride, src, srcstride, src2, blocks, n, stream);
    LAUNCH_CHECK();
    return 0;
}

/* 16-bit samples (bd 10 / 12): the sample-per-lane kernel on uint16_t */
template <bool CHROMA>
static void hevc_mc_launch16(int mode, int bd, void *dst, ptrdiff_t dststride, const uint8_t *src, ptrdiff_t srcstride, const int16_t *src2,
                             const void *blocks, int n, hipStream_t stream)
{
    const dim3 grid(cdiv(n, 4)), block(256);
#define MC_CASE(M) case M: hipLaunchKernelGGL((k_hevc_mc_s<uint16_t, CHROMA, M>), grid, block, 0, stream, dst, dststride, src, srcstride, src2, blocks, n, bd); break;
    switch (mode) {
    MC_CASE(0) MC_CASE(1) MC_CASE(2) MC_CASE(3)
    default:
    MC_CASE(4)
    }
#undef MC_CASE
}

int ffhip_launch_hevc_mc_bd(int bd, int chroma, int mode, void *dst, ptrdiff_t dststride, const uint8_t *src, ptrdiff_t srcstride,
                            const int16_t *src2, const void *blocks, int n, hipStream_t stream)
{
    if (bd == 8)
        return ffhip_launch_hevc_mc(chroma, mode, dst, dststride, src, srcstride, src2, blocks, n, stream);
    if (n <= 0)
        return 0;
    if ((bd != 10 && bd != 12) || (((uintptr_t)dst | (uintptr_t)src | (size_t)dststride | (size_t)srcstride) & 1)) {
        ffhip_set_error("ffhip_hevc_mc: bit depth %d (8, 10, 12) / 16-bit planes must be 2-byte aligned", bd);
        return FFHIP_EINVAL;
    }
    if (chroma)
        hevc_mc_launch16<true>(mode, bd, dst, dststride, src, srcstride, src2, blocks, n, stream);
    else
        hevc_mc_launch16<false>(mode, bd, dst, dststride, src, srcstride, src2, blocks, n, stream);
    LAUNCH_CHECK();
    return 0;
}
