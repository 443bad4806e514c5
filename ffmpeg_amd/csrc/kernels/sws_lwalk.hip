/*
 * sws_lwalk.hip — fused H+V scaler for WIDE banks (5..32 taps on either axis: every down-scaling ratio to about 1/8,
 * and up-scaling with long kernels), planar and NV12/NV21 in and out.
 *
 * Same arithmetic as sws_scale.hip / sws_colwalk.hip (hScale8To15_c, libswscale/swscale.c:128-142; yuv2planeX_8_c /
 * yuv2nv12cX_c, libswscale/output.c:468-529; nv12ToUV_c, input.c:936).  The column walker (sws_colwalk.hip) keeps
 * everything in registers, which needs every register index static: a column's window must sit in one 8-byte span
 * and the vertical history is a 3-deep ring.  Wide windows break both, so here the two dynamically indexed
 * things live in LDS, everything else stays as it was:
 *
 *   one WAVE (= one 64-thread workgroup) owns 64 lanes x 4 output columns of a strip of <= 64 output rows and
 *   walks DOWN the source rows.  Per source row
 *     1. the wave copies the row segment its 256 columns read (coalesced dword loads, prefetched two rows ahead;
 *        NV12 is de-interleaved on the way) into its LDS row buffer;
 *     2. a lane reads each of its columns' windows from there at a dword-aligned DYNAMIC address (HT+1 dwords) and
 *        turns them into int16 pairs with two v_perm_b32 per 4 taps whose selectors carry the byte phase, then
 *        v_dot2_i32_i16 against register-resident coefficient pairs: the 15-bit sample h[r];
 *     3. it appends the vertical pairs (h[r-1], h[r]) of its 4 columns as one 16-byte record to a ring of 2*VT
 *        source rows in LDS;
 *     4. every output row whose window ends at r reads VT such records (rows p+1, p+3, ...) and is VT more
 *        v_dot2 per sample against the row's wave-uniform coefficient pairs, then v_ashr_pk_u8_i32 and one store.
 *   A wave executes its LDS operations in order, so no barrier is needed anywhere.  HBM traffic = source in +
 *   destination out (+ 2*VT-1 halo rows per strip).
 *
 * Banks are padded on the host to 4*HT horizontal and 2*VT vertical taps (zero taps change no sum).
 * Integer semantics are the reference's, as in sws_colwalk.hip; the int16 saturation of v_cvt_pk_i16_i32 equals
 * min(.,32767) + truncation because the host admits only banks whose horizontal sums cannot wrap.
 */
#include <stdlib.h>

#include "common.h"
#include "sws_kernels.h"

typedef short lw_short2 __attribute__((ext_vector_type(2)));
typedef uint32_t lw_u2 __attribute__((ext_vector_type(2)));
typedef const uint8_t __attribute__((address_space(1))) *lw_gcptr;
typedef uint8_t __attribute__((address_space(1))) *lw_gptr;
typedef const uint32_t __attribute__((address_space(1))) *lw_gc1;
typedef const lw_u2 __attribute__((address_space(1))) *lw_gc2;
typedef uint32_t __attribute__((address_space(1))) *lw_g1;
typedef lw_u2 __attribute__((address_space(1))) *lw_g2;

__device__ __forceinline__ int lw_dot2(uint32_t a, uint32_t b, int c)
{
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(lw_short2, a), __builtin_bit_cast(lw_short2, b), c, false);
}

__device__ __forceinline__ uint32_t lw_pk_u8(int a, int b)
{
    uint32_t r;
    asm("v_ashr_pk_u8_i32 %0, %1, %2, 19" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

/*
 * NG = 1: one plane, 4 columns per lane.  NG = 2: a U/V pair (byte-interleaved or planar on either side, same banks
 * for both), 2 columns of each per lane — every unit is 4 samples per lane per row, one 16-byte ring record.
 */
template <int HT, int VT, int NL, int NG, bool AHEAD>
__device__ __forceinline__ void lw_unit(const FFHipLwJob &J, int f, int strip, int cb, int lane, uint32_t *lds)
{
    constexpr int CPL = 4 / NG;                      /* columns per lane and channel                      */
    constexpr int NLG = NG == 2 ? (NL + 1) / 2 : NL; /* segment loads per lane and row (a pair's block is half as wide) */
    constexpr int RAWD = NL * 64 + 8;  /* dwords of one row buffer (+ slack: a window read may touch one dword past the segment) */
    constexpr int RING = 2 * VT;       /* source rows of vertical history */
    uint32_t *raw = lds;                                        /* [NG][RAWD]        */
    uint4 *ring = reinterpret_cast<uint4 *>(lds + 2 * RAWD);    /* [RING][64]        */
    const bool sil = NG == 2 && J.sil, dil = NG == 2 && J.dil;
    const bool y16 = NG == 1 && J.y16;
    const int y0 = strip * J.strip_rows;
    const int y1 = min(y0 + J.strip_rows, J.dstH);
    const int ny = y1 - y0;

    /* ---- horizontal descriptors ---- */
    const int X0 = (cb * 64 + lane) * CPL;
    const int segb = __builtin_amdgcn_readfirstlane(J.hp[min(cb * 64 * CPL, J.dstW - 1)]) & ~3;
    uint32_t cf[CPL][2 * HT], sel_a[CPL], sel_b[CPL], woff[CPL];
#pragma unroll
    for (int i = 0; i < CPL; i++) {
        const int xi = min(X0 + i, J.dstW - 1);
        const int o = J.hp[xi] - segb;
        woff[i] = (uint32_t)(o >> 2);
        const uint32_t s = (uint32_t)(o & 3);
        sel_a[i] = 0x0c000c00u | s | ((s + 1) << 16);
        sel_b[i] = 0x0c000c00u | (s + 2) | ((s + 3) << 16);
        const uint2 *c = reinterpret_cast<const uint2 *>(J.hf + (size_t)xi * (4 * HT));
#pragma unroll
        for (int k = 0; k < HT; k++) {
            const uint2 v = c[k];
            cf[i][2 * k] = v.x;
            cf[i][2 * k + 1] = v.y;
        }
    }
    /* the 256-byte pieces of a source row this wave's windows reach (the row buffer is sized for the tap class's steepest ratio; at
     * 1.5 : 1 a third of it would be fetched and staged for nothing: measured traffic 1.9x the algorithmic bytes) */
    const int xlast = min(cb * 64 * CPL + 64 * CPL - 1, J.dstW - 1);
    const int nlu = min(NLG, (__builtin_amdgcn_readfirstlane(J.hp[xlast]) + 4 * HT - segb + 255) >> 8);
    const bool act = X0 < J.dstW;
    const int nval = J.dstW - X0;          /* this lane's valid columns (per channel) when it holds the row's ragged end */
    const bool whole = nval >= CPL;

    /* ---- vertical descriptors of the strip's <= 64 rows ---- */
    int vpl;
    uint32_t vcf[VT];
    {
        const int y = min(y0 + lane, J.dstH - 1);
        vpl = J.vp[y];
        const uint32_t *c = reinterpret_cast<const uint32_t *>(J.vf + (size_t)y * (2 * VT));
#pragma unroll
        for (int k = 0; k < VT; k++)
            vcf[k] = c[k];
    }

    /* ---- source rows: this lane's dwords of the wave's segment ---- */
    const uint8_t *s0 = J.src[0] + (size_t)f * J.sfp[0];
    const uint8_t *s1 = NG == 2 ? J.src[1] + (size_t)f * J.sfp[1] : s0;
    const ptrdiff_t sstride0 = J.sstride[0], sstride1 = J.sstride[1];
    uint32_t goff[NLG]; /* byte offset in the row of load j (clamped to the last whole unit of the row) */
#pragma unroll
    for (int j = 0; j < NLG; j++) {
        if (sil)
            goff[j] = (uint32_t)min(2 * segb + 8 * (lane + 64 * j), 2 * J.srcW - 8);
        else
            goff[j] = (uint32_t)min(segb + 4 * (lane + 64 * j), (J.srcW - 1) & ~3);
    }
    const uint32_t sel_u = J.src_swap ? 0x07050301u : 0x06040200u, sel_v = J.src_swap ? 0x06040200u : 0x07050301u;
    const uint32_t sel_uv = J.dst_swap ? 0x04050001u : 0x05040100u;

    struct Row { uint32_t q[NG][NLG]; };
    const int rfirst = __builtin_amdgcn_readlane(vpl, 0);
    const int rlast = __builtin_amdgcn_readfirstlane(J.vp[y1 - 1]) + RING - 1;
    int pfrow = rfirst;
    const uint8_t *pf0 = s0 + (ptrdiff_t)rfirst * sstride0, *pf1 = s1 + (ptrdiff_t)rfirst * sstride1;
    asm("" : "+s"(pf0), "+s"(pf1));
    auto load_next = [&](Row &o) {
#pragma unroll
        for (int j = 0; j < NLG; j++) {
            if (j >= nlu) /* uniform */
                break;
            uint32_t off = goff[j];
            asm volatile("" : "+v"(off));
            if (sil) {
                const lw_u2 w = *(lw_gc2)((lw_gcptr)pf0 + off);
                o.q[0][j] = w.x;
                o.q[NG - 1][j] = w.y;
            } else {
                o.q[0][j] = *(lw_gc1)((lw_gcptr)pf0 + off);
                if (NG == 2)
                    o.q[NG - 1][j] = *(lw_gc1)((lw_gcptr)pf1 + off);
            }
        }
        const bool adv = pfrow < rlast;
        pfrow = min(pfrow + 1, rlast);
        pf0 += adv ? sstride0 : 0;
        pf1 += adv ? sstride1 : 0;
        asm("" : "+s"(pf0), "+s"(pf1));
    };

    int hprev[4];
#pragma unroll
    for (int i = 0; i < 4; i++)
        hprev[i] = 0;

    uint8_t *d0 = J.dst[0] + (size_t)f * J.dfp[0] + (ptrdiff_t)y0 * J.dstride[0];
    uint8_t *d1 = NG == 2 ? J.dst[1] + (size_t)f * J.dfp[1] + (ptrdiff_t)y0 * J.dstride[1] : d0;
    const ptrdiff_t dstride0 = J.dstride[0], dstride1 = J.dstride[1];
    asm("" : "+s"(d0), "+s"(d1));

    int yy = 0;
    int need = rfirst + RING - 1;

    /* 1. row segment -> LDS (NV12: bytes u0 v0 u1 v1 ... become one dword of U and one of V) */
    auto stage = [&](const Row &cur) {
#pragma unroll
        for (int j = 0; j < NLG; j++) {
            if (j >= nlu) /* uniform */
                break;
            if (sil) {
                raw[lane + 64 * j] = __builtin_amdgcn_perm(cur.q[1][j], cur.q[0][j], sel_u);
                raw[RAWD + lane + 64 * j] = __builtin_amdgcn_perm(cur.q[1][j], cur.q[0][j], sel_v);
            } else {
#pragma unroll
                for (int g = 0; g < NG; g++)
                    raw[g * RAWD + lane + 64 * j] = cur.q[g][j];
            }
        }
        __builtin_amdgcn_wave_barrier();
    };
    /* 2a. the windows of my columns (AHEAD: read one row before their use — a wave's LDS operations execute in order, so
     * the next row may overwrite the buffer as soon as these reads are issued) */
    struct Win { uint32_t d[NG][CPL][HT + 1]; };
    auto read_windows = [&](Win &w) {
#pragma unroll
        for (int g = 0; g < NG; g++)
#pragma unroll
            for (int i = 0; i < CPL; i++)
#pragma unroll
                for (int k = 0; k <= HT; k++)
                    w.d[g][i][k] = raw[g * RAWD + woff[i] + k];
        asm volatile("" ::: "memory");
    };
    auto compute = [&](const Win &w, int rr) {
        /* 2b. + 3. horizontal pass of my columns, vertical pairs into the ring */
        uint32_t pr[4];
#pragma unroll
        for (int g = 0; g < NG; g++) {
#pragma unroll
            for (int i = 0; i < CPL; i++) {
                int acc = 0;
#pragma unroll
                for (int k = 0; k < HT; k++) {
                    acc = lw_dot2(__builtin_amdgcn_perm(w.d[g][i][k + 1], w.d[g][i][k], sel_a[i]), cf[i][2 * k], acc);
                    acc = lw_dot2(__builtin_amdgcn_perm(w.d[g][i][k + 1], w.d[g][i][k], sel_b[i]), cf[i][2 * k + 1], acc);
                }
                const int h = acc >> 7;
                pr[g * CPL + i] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pk_i16(hprev[g * CPL + i], h));
                hprev[g * CPL + i] = h;
            }
        }
        ring[(rr & (RING - 1)) * 64 + lane] = make_uint4(pr[0], pr[1], pr[2], pr[3]);
        __builtin_amdgcn_wave_barrier();
        /* 4. output rows whose window ends here */
        while (yy < ny && need <= rr) {
            const int p = need - (RING - 1);
            uint32_t fk[VT];
#pragma unroll
            for (int k = 0; k < VT; k++)
                fk[k] = __builtin_amdgcn_readlane(vcf[k], yy);
            int v[4];
#pragma unroll
            for (int i = 0; i < 4; i++)
                v[i] = 64 << 12;
#pragma unroll
            for (int k = 0; k < VT; k++) {
                const uint4 P = ring[((p + 1 + 2 * k) & (RING - 1)) * 64 + lane];
                v[0] = lw_dot2(P.x, fk[k], v[0]);
                v[1] = lw_dot2(P.y, fk[k], v[1]);
                v[2] = lw_dot2(P.z, fk[k], v[2]);
                v[3] = lw_dot2(P.w, fk[k], v[3]);
            }
            typedef uint16_t __attribute__((address_space(1))) *lw_gh;
            typedef uint8_t __attribute__((address_space(1))) *lw_gb;
            if (NG == 1 && y16) {
                /* the luma of a packed-RGB target's first stage: the sums >> 19 as they are, int16 — yuv2rgb_X does not clip Y before
                 * the tables (libswscale/output.c:1814-1835); no 4-tap-or-wider bank of weights summing to 4096 leaves int16 */
                lw_u2 w2;
                w2.x = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pk_i16(v[0] >> 19, v[1] >> 19));
                w2.y = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pk_i16(v[2] >> 19, v[3] >> 19));
                if (whole) {
                    *(lw_g2)((lw_gptr)d0 + (uint32_t)(2 * X0)) = w2;
                } else if (act) { /* the ragged last group of a row (round 5: widths that are not multiples of 4) */
                    lw_gh d = (lw_gh)((lw_gptr)d0 + (uint32_t)(2 * X0));
                    d[0] = (uint16_t)w2.x;
                    if (nval > 1) d[1] = (uint16_t)(w2.x >> 16);
                    if (nval > 2) d[2] = (uint16_t)w2.y;
                }
            } else if (NG == 1) {
                const uint32_t w1 = __builtin_amdgcn_perm(lw_pk_u8(v[2], v[3]), lw_pk_u8(v[0], v[1]), 0x05040100);
                if (whole) {
                    *(lw_g1)((lw_gptr)d0 + (uint32_t)X0) = w1;
                } else if (act) {
                    lw_gb d = (lw_gb)((lw_gptr)d0 + (uint32_t)X0);
                    d[0] = (uint8_t)w1;
                    if (nval > 1) d[1] = (uint8_t)(w1 >> 8);
                    if (nval > 2) d[2] = (uint8_t)(w1 >> 16);
                }
            } else if (dil) {
                /* v = U0 U1 V0 V1 -> bytes U0 V0 U1 V1 (V first for NV21) */
                const uint32_t w1 = __builtin_amdgcn_perm(lw_pk_u8(v[1], v[3]), lw_pk_u8(v[0], v[2]), sel_uv);
                if (whole)
                    *(lw_g1)((lw_gptr)d0 + (uint32_t)(2 * X0)) = w1;
                else if (act)
                    *(lw_gh)((lw_gptr)d0 + (uint32_t)(2 * X0)) = (uint16_t)w1; /* one (u, v) pair */
            } else if (whole) {
                *(lw_gh)((lw_gptr)d0 + (uint32_t)X0) = (uint16_t)lw_pk_u8(v[0], v[1]);
                *(lw_gh)((lw_gptr)d1 + (uint32_t)X0) = (uint16_t)lw_pk_u8(v[2], v[3]);
            } else if (act) {
                *(lw_gb)((lw_gptr)d0 + (uint32_t)X0) = (uint8_t)lw_pk_u8(v[0], v[1]);
                *(lw_gb)((lw_gptr)d1 + (uint32_t)X0) = (uint8_t)lw_pk_u8(v[2], v[3]);
            }
            d0 += dstride0;
            d1 += dstride1;
            asm("" : "+s"(d0), "+s"(d1));
            yy++;
            if (yy < ny)
                need = __builtin_amdgcn_readlane(vpl, yy) + RING - 1;
        }
    };

    /* software pipeline: global rows two ahead in registers, LDS windows one ahead in registers */
    Row b0, b1;
    Win w0, w1;
    load_next(b0);
    load_next(b1);
    if (!AHEAD) {
        /* windows read in the row they are used (the default) */
        for (int r = rfirst; r <= rlast; r += 2) {
            stage(b0);
            load_next(b0);
            read_windows(w0);
            compute(w0, r);
            stage(b1);
            load_next(b1);
            if (r + 1 <= rlast) {
                read_windows(w0);
                compute(w0, r + 1);
            }
        }
        return;
    }
    stage(b0);
    load_next(b0);
    read_windows(w0);
    for (int r = rfirst; r <= rlast; r += 2) {
        stage(b1);
        load_next(b1);
        read_windows(w1);
        compute(w0, r);
        stage(b0);
        load_next(b0);
        read_windows(w0);
        if (r + 1 <= rlast)
            compute(w1, r + 1);
    }
}

template <int HT, int VT, int NL, bool AHEAD>
__global__ __launch_bounds__(64) void k_sws_lwalk(FFHipLwArgs A)
{
    extern __shared__ __align__(16) uint32_t lw_lds[];
    const int lane = threadIdx.x;
    const uint32_t gw = blockIdx.x;
    const int f = (int)(gw / (uint32_t)A.units_per_frame);
    const int u = (int)(gw - (uint32_t)f * (uint32_t)A.units_per_frame);
    int j = 0;
    if (A.njobs > 1 && u >= A.job[1].unit_begin) j = 1;
    if (A.njobs > 2 && u >= A.job[2].unit_begin) j = 2;
    const FFHipLwJob &J = A.job[j];
    const int local = u - J.unit_begin;
    const int strip = local / J.ncb, cb = local - strip * J.ncb;
    if (J.pair)
        lw_unit<HT, VT, NL, 2, AHEAD>(J, f, strip, cb, lane, lw_lds);
    else
        lw_unit<HT, VT, NL, 1, AHEAD>(J, f, strip, cb, lane, lw_lds);
}

/* ---- host side ---------------------------------------------------------------------------------- */
/*
 * Can a (padded) bank pair run here?  hpos/vpos are host copies of the padded positions; ht/vt the padded sizes in
 * units of 4 / 2 taps; pair: the bank serves a U/V pair unit (128 columns per wave).  Returns the NL class (3 or 5)
 * or 0.
 */
int ffhip_lw_bank_ok(const int32_t *hpos, int ht, int hn, int srcW, const int32_t *vpos, int vt, int vn, int srcH, int pair)
{
    if ((ht != 2 && ht != 4 && ht != 8 && ht != 16) || (vt != 4 && vt != 8 && vt != 16) || hn <= 0 || vn <= 0 || srcW < 4 * ht || srcW < 8 ||
        srcH < 2 * vt)
        return 0;
    const int block = pair ? 128 : 256; /* output columns of one wave */
    int span = 0;
    for (int x = 0; x < hn; x++)
        if (hpos[x] < 0 || hpos[x] + 4 * ht > srcW || (x && hpos[x] < hpos[x - 1]))
            return 0;
    for (int x0 = 0; x0 < hn; x0 += block) {
        const int xe = x0 + block - 1 < hn ? x0 + block - 1 : hn - 1;
        const int s = hpos[xe] + 4 * ht - (hpos[x0] & ~3);
        if (s > span)
            span = s;
    }
    for (int y = 0; y < vn; y++)
        if (vpos[y] < 0 || vpos[y] + 2 * vt > srcH || (y && vpos[y] < vpos[y - 1]))
            return 0;
    const int nl = ht == 2 ? 3 : ht == 4 ? 5 : ht == 8 ? 9 : 17;
    return span <= (pair ? (nl + 1) / 2 : nl) * 256 ? nl : 0;
}

static void lw_plan(FFHipLwJob *j, int want)
{
    j->ncb = cdiv(j->dstW, j->pair ? 128 : 256);
    const int n = cdiv(j->dstH, want);
    j->strip_rows = cdiv(j->dstH, n);
    j->nstrips = cdiv(j->dstH, j->strip_rows);
}

void ffhip_lw_plan_job(FFHipLwJob *j)
{
    const char *es = FFHIP_KNOB("FFHIP_LW_STRIP"); /* measured variant: shorter strips (more, lighter waves; more halo rows) */
    lw_plan(j, es && atoi(es) > 0 && atoi(es) < 64 ? atoi(es) : 64);
}

int ffhip_launch_lwalk(FFHipLwArgs &A, hipStream_t stream)
{
    if (A.nframes <= 0)
        return 0;
    /* a wave is one dependent chain down its strip: a launch of fewer waves than the chip holds (a thumbnail, a network input, a
     * lone frame, 16 frames of 720p) runs at the speed of its chains — strips of 32, 16 rows then, until the launch has the waves the
     * chip can keep resident (four per SIMD at this kernel's LDS; measured, 1080p -> 720p nv12, 16 frames = 1,440 waves at 64 rows:
     * 0.062 ms, at 32 rows 0.047; a strip re-filters 2 VT - 1 source rows, which is why large launches stay at 64) */
    {
        const char *es = FFHIP_KNOB("FFHIP_LW_STRIP");
        for (int want = 64; !(es && atoi(es) > 0); want >>= 1) {
            long long w = 0;
            for (int i = 0; i < A.njobs; i++) {
                lw_plan(&A.job[i], want);
                w += (long long)A.job[i].ncb * A.job[i].nstrips;
            }
            if (w * A.nframes >= 4096 || want <= 16)
                break;
        }
    }
    int u = 0;
    for (int i = 0; i < A.njobs; i++) {
        A.job[i].unit_begin = u;
        u += A.job[i].ncb * A.job[i].nstrips;
    }
    A.units_per_frame = u;
    const long long waves = (long long)u * A.nframes;
    if (waves >= (1LL << 31)) {
        ffhip_set_error("ffhip_sws: batch too large for one launch (%lld waves)", waves);
        return FFHIP_EINVAL;
    }
    const int nl = A.ht == 2 ? 3 : A.ht == 4 ? 5 : A.ht == 8 ? 9 : 17;
    const size_t lds = (size_t)4 * 2 * (nl * 64 + 8) + (size_t)2 * A.vt * 64 * 16;
    const dim3 grid((unsigned)waves), block(64);
    /* measured (nv12 4K -> 1080p / 720p / 540p): reading the windows one row ahead is 1-2 % SLOWER than reading them in
     * the row they are used (0.78 vs 0.77 ms per 128 frames) — the extra registers cost more than the latency they hide once
     * the units are light; FFHIP_LW_AHEAD=1 selects the pipelined form */
    const char *ea = FFHIP_KNOB("FFHIP_LW_AHEAD");
    const bool ahead = ea && ea[0] == '1';
#define LW_LAUNCH(H, V, N)                                                                                   \
    do {                                                                                                     \
        static FFHipPerDeviceOnce attr_done;                                                                 \
        if (attr_done.enter()) {                                                                             \
            (void)hipFuncSetAttribute((const void *)k_sws_lwalk<H, V, N, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024); \
            (void)hipFuncSetAttribute((const void *)k_sws_lwalk<H, V, N, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024); \
            attr_done.leave(true);                                                                           \
        }                                                                                                    \
        if (ahead) hipLaunchKernelGGL((k_sws_lwalk<H, V, N, true>), grid, block, lds, stream, A);            \
        else       hipLaunchKernelGGL((k_sws_lwalk<H, V, N, false>), grid, block, lds, stream, A);           \
    } while (0)
    /* round 5: banks of up to 32 taps on either axis (ratios down to about 1/8: a 1080p frame into a 224 x 224 network input has 36 and
     * 20 taps... 32 x 32 covers 1080p -> 240p) — the default form only */
#define LW_LAUNCH1(H, V, N)                                                                                  \
    do {                                                                                                     \
        static FFHipPerDeviceOnce attr_done;                                                                 \
        if (attr_done.enter()) {                                                                             \
            (void)hipFuncSetAttribute((const void *)k_sws_lwalk<H, V, N, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024); \
            attr_done.leave(true);                                                                           \
        }                                                                                                    \
        hipLaunchKernelGGL((k_sws_lwalk<H, V, N, false>), grid, block, lds, stream, A);                      \
    } while (0)
    if (A.ht == 2 && A.vt == 4) LW_LAUNCH(2, 4, 3);
    else if (A.ht == 2 && A.vt == 8) LW_LAUNCH(2, 8, 3);
    else if (A.ht == 4 && A.vt == 4) LW_LAUNCH(4, 4, 5);
    else if (A.ht == 4 && A.vt == 8) LW_LAUNCH(4, 8, 5);
    else if (A.ht == 2 && A.vt == 16) LW_LAUNCH1(2, 16, 3);
    else if (A.ht == 4 && A.vt == 16) LW_LAUNCH1(4, 16, 5);
    else if (A.ht == 8 && A.vt == 4) LW_LAUNCH1(8, 4, 9);
    else if (A.ht == 8 && A.vt == 8) LW_LAUNCH1(8, 8, 9);
    else if (A.ht == 8 && A.vt == 16) LW_LAUNCH1(8, 16, 9);
    /* 64 taps across (a 1080p frame into a 224-wide network input: 36), up to 32 down */
    else if (A.ht == 16 && A.vt == 8) LW_LAUNCH1(16, 8, 17);
    else if (A.ht == 16 && A.vt == 16) LW_LAUNCH1(16, 16, 17);
    else {
        ffhip_set_error("ffhip_sws: no wide-bank kernel for %d x %d taps", 4 * A.ht, 2 * A.vt);
        return FFHIP_EINVAL;
    }
#undef LW_LAUNCH
#undef LW_LAUNCH1
    LAUNCH_CHECK();
    return 0;
}
