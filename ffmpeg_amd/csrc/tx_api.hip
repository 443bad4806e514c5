/*
 * tx_api.hip — libavutil av_tx float MDCT (AV_TX_FLOAT_MDCT, power-of-two lengths) for libffhip.
 *
 * Reference path:
 *   ff_tx_mdct_init / ff_tx_mdct_fwd / ff_tx_mdct_inv      libavutil/tx_template.c:1223-1342
 *   ff_tx_mdct_gen_exp                                      libavutil/tx_template.c:2107-2134
 *   split-radix codelets + ff_tx_fft_sr_combine             libavutil/tx_template.c:540-722
 *   cosine tables ff_tx_tab_N_float                         libavutil/tx_template.c:65-77
 *   ff_tx_gen_ptwo_revtab / split_radix_permutation         libavutil/tx.c:125-155
 *
 * One transform = fold/pre-twiddle into a split-radix-permuted N/2-point complex array, an in-place
 * N/2-point split-radix FFT, post-twiddle.  The reference recursion  FFT(m) = FFT(m/2) + 2 x FFT(m/4)
 * + combine(m)  is flattened on the host into one butterfly list per level (all size-2 blocks, then all
 * size-4 combines, ... up to size N/2: a level only consumes lower levels, blocks of a level are
 * disjoint).  Every butterfly performs the reference's float operations in the reference's order and
 * the library is built with -ffp-contract=off, so the results are the reference's, bit for bit.
 *
 * GPU design (HBM-bound, 12 KiB moved per forward N=1024 transform for ~25 kFLOP): one WAVE per
 * transform, several waves per workgroup.  Input is staged HBM -> LDS with coalesced 16-byte loads,
 * folded + pre-twiddled into the LDS complex array, transformed there level by level (64 butterflies per
 * wave instruction, wave-local synchronisation only), post-twiddled back into the staging area and
 * written out with coalesced 16-byte stores.  Twiddles, the permutation and the butterfly lists are
 * small read-only tables that stay resident in L2.
 */
#include <math.h>
#include <mutex>
#include <new>
#include <stdlib.h>
#include <string.h>
#include <vector>

#include "kernels/common.h"
#include "kernels/tx_kernels.h"
#include "kernels/tx_radix_core.h"

struct TxDev {
    int n, lg;                 /* complex FFT size (= len/2) and its log2                         */
    const int *map;            /* n: where input-order element j goes in the (padded) work array   */
    const float2 *exp;         /* n entries, natural order                                          */
    const float *cos_tab;      /* concatenated per level, cos_off[l] = first entry of level l       */
    const uint32_t *sched;     /* butterflies: a0 | k << 16, concatenated per level                 */
    const uint16_t *blocks2;   /* offsets of the size-2 blocks                                      */
    int nblocks2;
    int max_cnt, ahead;        /* largest butterfly list of a level; 1: fetch lists/twiddles one level ahead (FFHIP_TX_AHEAD) */
    int cos_off[16], sched_off[16], sched_cnt[16];
    int half;                  /* forward RDFT: 1 real-to-real, 2 real-to-imaginary output */
};

/* 2 * F * 2^k MDCT lengths, F = 3, 5, 7, 9, 15 (ff_tx_mdct_pfa_<F>xM): per-transform geometry and the extra tables of the
 * prime-factor kernel */
struct TxPfa {
    int n1, m, G;              /* complex points per transform (F m), sub-transform size, transforms per wave (G * m = 64; 1 for m >= 64) */
    int F;                     /* the small factor */
    int fft;                   /* 1: ff_tx_fft_pfa (n1 complex in, n1 complex out, no twiddles) rather than the MDCT around it */
    int magic_q;               /* ceil(2^24 / (n1 / 2)): e / (n1 / 2) == (e * magic) >> 24 for e * n1 / 2 < 2^24 */
    const int *in_map;         /* n1: ((i * F + j) -> k >> 1, the point sub-transform i takes as its j-th input */
    const int *out_map;        /* n1: CRT output map                                                     */
    const int *sub_map;        /* m: where sub-transform i's F-point outputs start                       */
};

struct FFHipTXContext {
    int device = 0;          /* the tables live on this device; every call of the context makes it current for its duration */
    int type, inv, len;
    int half = 0;            /* AV_TX_REAL_TO_REAL / _IMAGINARY (1 / 2) */
    int full = 0;            /* AV_TX_FULL_IMDCT: the inverse writes 2 * len outputs (half transform in the middle, mirrored) */
    float scale;
    TxDev d;
    TxPfa pfa = {};
    void *dev = nullptr;
    size_t blob_bytes = 0;   /* size of the table blob at `dev` (multiple of 16) */
    float2 *wtab = nullptr;  /* exp(-2 pi i k / n), k < n: the register-resident kernels' twiddles (kernels/tx_radix.hip), or null */
    FFHipTxWide *wide = nullptr; /* AV_TX_DOUBLE_* / AV_TX_INT32_* contexts: everything lives in kernels/tx_wide.hip */
    FFHipTxDcst1 *dcst1 = nullptr; /* AV_TX_FLOAT_DCT_I / _DST_I contexts: kernels/tx_dcst1.hip */
    /* host-pointer shim staging */
    void *stage = nullptr;
    size_t stage_sz = 0;
    std::mutex mu;
};

/*
 * LDS layout of the complex work array: element i lives at i + (i >> 5), one pad element per 32 (= per 256-byte row
 * of the 64 LDS banks).  Every power-of-two operand stride of the split-radix levels then falls on distinct banks for
 * the 32 lanes an 8-byte access serves per cycle; with the plain layout the low levels (operands of neighbouring
 * lanes 32..512 bytes apart) serialised 4-16 ways.  The butterfly lists and the forward scatter map carry padded
 * indices from the host; a level's operand offsets k*q pad independently (no carry across bit 5: blocks are 4q
 * aligned).
 */
#define TX_PAD(i) ((i) + ((i) >> 5))
__host__ __device__ static inline size_t tx_z_bytes(int n) { return ((size_t)TX_PAD(n) * 8 + 15) & ~(size_t)15; }

__device__ __forceinline__ void tx_wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

/* ff_tx_fft_sr_combine's TRANSFORM (libavutil/tx_template.c:540-586) on one butterfly: the reference's float
 * operations in the reference's order */
__device__ __forceinline__ void tx_butterfly(float2 &v0, float2 &v1, float2 &v2, float2 &v3, float wre, float wim)
{
    const float nwim = -wim;
    const float t1 = v2.x * wre - v2.y * nwim;
    const float t2 = v2.x * nwim + v2.y * wre;
    float t5 = v3.x * wre - v3.y * wim;
    float t6 = v3.x * wim + v3.y * wre;
    const float t3 = t5 - t1;
    t5 = t5 + t1;
    const float t4 = t2 - t6;
    t6 = t2 + t6;
    const float2 a = v0, b = v1;
    v0 = make_float2(a.x + t5, a.y + t6);
    v1 = make_float2(b.x + t4, b.y + t3);
    v2 = make_float2(a.x - t5, a.y - t6);
    v3 = make_float2(b.x - t4, b.y - t3);
}

/*
 * The levels of the in-place split-radix FFT when no level has more than 64*MI butterflies (N <= 1024: MI = 2).
 * A level's critical path would be  list entry -> twiddles + operands -> arithmetic -> write-back, i.e. two dependent
 * LDS round trips before any arithmetic; the list entries and twiddles do not depend on the data, so they are
 * fetched one level AHEAD, in the shadow of the previous level's arithmetic and barrier.
 */
template <int MI>
__device__ __forceinline__ void tx_fft_levels_ahead(float2 *z, const TxDev &d, const float *cos_tab, const uint32_t *sched, int lane)
{
    uint32_t e[MI];
    float wre[MI], wim[MI];
    auto load_entries = [&](int l, uint32_t (&E)[MI]) {
        const uint32_t *sc = sched + d.sched_off[l];
        const int cnt = d.sched_cnt[l];
#pragma unroll
        for (int j = 0; j < MI; j++)
            E[j] = lane + 64 * j < cnt ? sc[lane + 64 * j] : 0xFFFFFFFFu;
    };
    auto load_twiddles = [&](int l, const uint32_t (&E)[MI], float (&R)[MI], float (&I)[MI]) {
        const int q = 1 << (l - 2);
        const float *tab = cos_tab + d.cos_off[l];
#pragma unroll
        for (int j = 0; j < MI; j++) {
            const int k = E[j] == 0xFFFFFFFFu ? 0 : (int)(E[j] >> 16);
            R[j] = tab[k];
            I[j] = tab[q - k];
        }
    };
    load_entries(2, e);
    load_twiddles(2, e, wre, wim);
    for (int l = 2; l <= d.lg; l++) {
        tx_wave_sync();
        const int q = 1 << (l - 2);
        const int o1 = TX_PAD(q), o2 = TX_PAD(2 * q), o3 = TX_PAD(3 * q);
        float2 v[MI][4];
#pragma unroll
        for (int j = 0; j < MI; j++) {
            const int a0 = e[j] == 0xFFFFFFFFu ? 0 : (int)(e[j] & 0xFFFF);
            v[j][0] = z[a0]; v[j][1] = z[a0 + o1]; v[j][2] = z[a0 + o2]; v[j][3] = z[a0 + o3];
        }
        uint32_t en[MI];
        float nr[MI], ni[MI];
        if (l < d.lg)
            load_entries(l + 1, en);
#pragma unroll
        for (int j = 0; j < MI; j++) {
            tx_butterfly(v[j][0], v[j][1], v[j][2], v[j][3], wre[j], wim[j]);
            if (e[j] != 0xFFFFFFFFu) {
                const int a0 = (int)(e[j] & 0xFFFF);
                z[a0] = v[j][0]; z[a0 + o1] = v[j][1]; z[a0 + o2] = v[j][2]; z[a0 + o3] = v[j][3];
            }
        }
        if (l < d.lg) {
            load_twiddles(l + 1, en, nr, ni);
#pragma unroll
            for (int j = 0; j < MI; j++) {
                e[j] = en[j]; wre[j] = nr[j]; wim[j] = ni[j];
            }
        }
    }
}

/* a transform's team: one wave (n <= 2048: the work array is a wave's own, wave-local synchronisation) or, WG, the whole
 * workgroup on one transform (n = 4096..16384: the work array is the workgroup's LDS, barriers between the levels) */
template <bool WG>
__device__ __forceinline__ void tx_team_sync()
{
    if (WG)
        __syncthreads();
    else
        tx_wave_sync();
}

/* in-place split-radix FFT of z[0..n) held in LDS, one team; `lane` is the thread's index in the team */
template <bool WG = false>
__device__ __forceinline__ void tx_fft_lds(float2 *z, const TxDev &d, const float *cos_tab, const uint32_t *sched,
                                           const uint16_t *blocks2, int lane)
{
    const int TS = WG ? (int)blockDim.x : 64;
    for (int b = lane; b < d.nblocks2; b += TS) {
        const int o = blocks2[b];
        const float2 x = z[o], y = z[o + 1];
        z[o] = make_float2(x.x + y.x, x.y + y.y);
        z[o + 1] = make_float2(x.x - y.x, x.y - y.y);
    }
    if (!WG && d.max_cnt <= 128 && d.ahead) {
        tx_fft_levels_ahead<2>(z, d, cos_tab, sched, lane);
        tx_wave_sync();
        return;
    }
    for (int l = 2; l <= d.lg; l++) {
        tx_team_sync<WG>();
        const int q = 1 << (l - 2);
        const int o1 = TX_PAD(q), o2 = TX_PAD(2 * q), o3 = TX_PAD(3 * q);
        const float *tab = cos_tab + d.cos_off[l];
        const uint32_t *sc = sched + d.sched_off[l];
        for (int b = lane; b < d.sched_cnt[l]; b += TS) {
            const uint32_t e = sc[b];
            const int a0 = e & 0xFFFF, k = e >> 16;
            float2 v0 = z[a0], v1 = z[a0 + o1], v2 = z[a0 + o2], v3 = z[a0 + o3];
            tx_butterfly(v0, v1, v2, v3, tab[k], tab[q - k]);
            z[a0] = v0; z[a0 + o1] = v1; z[a0 + o2] = v2; z[a0 + o3] = v3;
        }
    }
    tx_team_sync<WG>();
}

/*
 * INV == 0: in = 4n floats (contiguous), out = 2n floats `ostride` elements apart.
 * INV == 1: in = 2n floats `istride` elements apart, out = 2n floats (contiguous).
 */
template <int INV>
__global__ __launch_bounds__(256) void k_mdct(TxDev d, const float *in, size_t in_pitch, float *out, size_t out_pitch,
                                              ptrdiff_t stride, int nt, int waves_per_block, int vec_in, int vec_out,
                                              int ftab_bytes)
{
    extern __shared__ __align__(16) uint8_t lds_raw[];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    const int t = blockIdx.x * waves_per_block + wave;
    const int n = d.n, q = n >> 1;
    const size_t zb = tx_z_bytes(n), per_wave = zb + (size_t)n * 16; /* z (padded) + staging (4n floats) */
    /* (the kernel argument itself must stay untouched: a conditionally modified TxDev is spilled to scratch) */
    const float *f_cos = d.cos_tab;
    const uint32_t *f_sched = d.sched;
    const uint16_t *f_b2 = d.blocks2;
    if (ftab_bytes) {
        /* the tables the FFT levels chain through (twiddles, butterfly lists, size-2 list: ~5 KiB at N=1024,
         * contiguous in the context blob from cos_tab on) go to LDS: a level's list entry -> twiddle -> operands
         * chain then costs LDS latencies instead of three dependent L2 round trips */
        const uint4 *s4 = reinterpret_cast<const uint4 *>(d.cos_tab);
        uint8_t *ft = lds_raw + waves_per_block * per_wave;
        uint4 *l4 = reinterpret_cast<uint4 *>(ft);
        for (int i = threadIdx.x; i < ftab_bytes / 16; i += blockDim.x)
            l4[i] = s4[i];
        __syncthreads();
        const uint8_t *g0 = reinterpret_cast<const uint8_t *>(d.cos_tab);
        f_sched = reinterpret_cast<const uint32_t *>(ft + (reinterpret_cast<const uint8_t *>(d.sched) - g0));
        /* (the size-2 block list stays in L2: its 342 bytes would push the workgroup over 42 LDS allocation units of
         * 1280 bytes, i.e. from 3 workgroups per CU to 2) */
        f_cos = reinterpret_cast<const float *>(ft);
    }
    if (wave >= waves_per_block || t >= nt)
        return;
    float2 *z = reinterpret_cast<float2 *>(lds_raw + wave * per_wave);
    float *st = reinterpret_cast<float *>(lds_raw + wave * per_wave + zb);
    const float *src = reinterpret_cast<const float *>(reinterpret_cast<const uint8_t *>(in) + (size_t)t * in_pitch);
    float *dst = reinterpret_cast<float *>(reinterpret_cast<uint8_t *>(out) + (size_t)t * out_pitch);

    if (!INV) {
        /* ---- stage the 4n input samples ---- */
        if (vec_in) {
            const float4 *s4 = reinterpret_cast<const float4 *>(src);
            float4 *d4 = reinterpret_cast<float4 *>(st);
            for (int j = lane; j < n; j += 64)
                d4[j] = s4[j];
        } else {
            for (int j = lane; j < 4 * n; j += 64)
                st[j] = src[j];
        }
        tx_wave_sync();
        /* ---- fold + pre-twiddle, scattered through map (ff_tx_mdct_fwd, tx_template.c:1285-1296) ---- */
        const int len3 = 3 * n;
        for (int i = lane; i < n; i += 64) {
            const int k = 2 * i;
            float re, im;
            if (k < n) {
                re = -st[n + k] + st[n - 1 - k];
                im = -st[len3 + k] + -st[len3 - 1 - k];
            } else {
                re = -st[n + k] + -st[5 * n - 1 - k];
                im = st[k - n] + -st[len3 - 1 - k];
            }
            const float2 e = d.exp[i];
            z[d.map[i]] = make_float2(re * e.y + im * e.x, re * e.x - im * e.y);
        }
        tx_wave_sync();
        tx_fft_lds(z, d, f_cos, f_sched, f_b2, lane);
        /* ---- post-twiddle (tx_template.c:1300-1309) ---- */
        for (int i = lane; i < q; i += 64) {
            const int i0 = q + i, i1 = q - i - 1;
            const float2 s1 = z[TX_PAD(i1)], s0 = z[TX_PAD(i0)], e0 = d.exp[i0], e1 = d.exp[i1];
            const float a = s0.x * e0.y - s0.y * e0.x; /* out[2*i1+1] */
            const float b = s0.x * e0.x + s0.y * e0.y; /* out[2*i0]   */
            const float c = s1.x * e1.y - s1.y * e1.x; /* out[2*i0+1] */
            const float f = s1.x * e1.x + s1.y * e1.y; /* out[2*i1]   */
            if (vec_out) {
                /* (2*i1, 2*i1+1) and (2*i0, 2*i0+1) as 8-byte writes: 4-byte writes two floats apart conflict 2 ways */
                reinterpret_cast<float2 *>(st)[i1] = make_float2(f, a);
                reinterpret_cast<float2 *>(st)[i0] = make_float2(b, c);
            } else {
                dst[(2 * i1 + 1) * stride] = a; dst[2 * i0 * stride] = b;
                dst[(2 * i0 + 1) * stride] = c; dst[2 * i1 * stride] = f;
            }
        }
    } else {
        /* ---- stage the 2n coefficients ---- */
        if (vec_in) {
            const float4 *s4 = reinterpret_cast<const float4 *>(src);
            float4 *d4 = reinterpret_cast<float4 *>(st);
            for (int j = lane; j < (n >> 1); j += 64)
                d4[j] = s4[j];
        } else {
            for (int j = lane; j < 2 * n; j += 64)
                st[j] = src[j * stride];
        }
        tx_wave_sync();
        /* ---- pre-twiddle (ff_tx_mdct_inv, tx_template.c:1321-1328: z[i] from in[map[i]]) walked in INPUT order j =
         * map[i] and scattered through the inverse permutation: the staging area is then read in order instead of
         * through a bit-reversal-like gather (up to 32 lanes on one LDS bank); same operands, same operations ---- */
        for (int j = lane; j < n; j += 64) {
            const int k = j << 1;
            const float tre = st[2 * n - 1 - k], tim = st[k];
            const float2 e = d.exp[j];
            z[d.map[j]] = make_float2(tre * e.x - tim * e.y, tre * e.y + tim * e.x);
        }
        tx_wave_sync();
        tx_fft_lds(z, d, f_cos, f_sched, f_b2, lane);
        /* ---- post-twiddle (tx_template.c:1332-1341) ---- */
        const float2 *ex = d.exp;
        for (int i = lane; i < q; i += 64) {
            const int i0 = q + i, i1 = q - i - 1;
            const float2 z1 = z[TX_PAD(i1)], z0 = z[TX_PAD(i0)], s1 = make_float2(z1.y, z1.x), s0 = make_float2(z0.y, z0.x);
            const float2 e0 = ex[i0], e1 = ex[i1];
            const float a = s1.x * e1.y - s1.y * e1.x; /* o[i1].re */
            const float b = s1.x * e1.x + s1.y * e1.y; /* o[i0].im */
            const float c = s0.x * e0.y - s0.y * e0.x; /* o[i0].re */
            const float f = s0.x * e0.x + s0.y * e0.y; /* o[i1].im */
            if (vec_out) {
                reinterpret_cast<float2 *>(st)[i1] = make_float2(a, f);
                reinterpret_cast<float2 *>(st)[i0] = make_float2(c, b);
            } else {
                dst[2 * i1] = a; dst[2 * i0 + 1] = b; dst[2 * i0] = c; dst[2 * i1 + 1] = f;
            }
        }
    }
    if (vec_out) {
        tx_wave_sync();
        const float4 *s4 = reinterpret_cast<const float4 *>(st);
        float4 *d4 = reinterpret_cast<float4 *>(dst);
        for (int j = lane; j < (n >> 1); j += 64)
            d4[j] = s4[j];
    }
}

/*
 * Persistent variant for contiguous, 16-byte aligned batches (what a codec hands over): the workgroup first
 * copies the context's whole table blob (permutation, exp[], twiddles, butterfly lists: 11-15 KiB for N=1024)
 * into LDS, then each of its waves loops over transforms.  A transform then touches HBM only for its own
 * samples; every table access is an LDS read (~100 cycles) instead of a dependent L2 round trip (~500+) per
 * butterfly level, which is what bounded the one-shot kernel above.
 */
template <int INV>
__global__ __launch_bounds__(256) void k_mdct_l(TxDev d, const uint8_t *blob, int blob_bytes, const float *in, size_t in_pitch,
                                                float *out, size_t out_pitch, int nt, int waves_total)
{
    extern __shared__ __align__(16) uint8_t lds_raw[];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    {
        const uint4 *s4 = reinterpret_cast<const uint4 *>(blob);
        uint4 *l4 = reinterpret_cast<uint4 *>(lds_raw);
        for (int i = threadIdx.x; i < blob_bytes / 16; i += 256)
            l4[i] = s4[i];
    }
    __syncthreads();
    /* rebase the table pointers into LDS (as locals: a modified copy of the argument struct would be spilled) */
    const int *l_map = reinterpret_cast<const int *>(lds_raw + (reinterpret_cast<const uint8_t *>(d.map) - blob));
    const float2 *l_exp = reinterpret_cast<const float2 *>(lds_raw + (reinterpret_cast<const uint8_t *>(d.exp) - blob));
    const float *l_cos = reinterpret_cast<const float *>(lds_raw + (reinterpret_cast<const uint8_t *>(d.cos_tab) - blob));
    const uint32_t *l_sched = reinterpret_cast<const uint32_t *>(lds_raw + (reinterpret_cast<const uint8_t *>(d.sched) - blob));
    const uint16_t *l_b2 = reinterpret_cast<const uint16_t *>(lds_raw + (reinterpret_cast<const uint8_t *>(d.blocks2) - blob));
    const int n = d.n, q = n >> 1;
    const size_t zb = tx_z_bytes(n), per_wave = zb + (size_t)n * 16;
    uint8_t *mine = lds_raw + ((blob_bytes + 15) & ~15) + wave * per_wave;
    float2 *z = reinterpret_cast<float2 *>(mine);
    float *st = reinterpret_cast<float *>(mine + zb);
    float4 *l4 = reinterpret_cast<float4 *>(st);
    const int nin4 = INV ? n / 2 : n;

    for (int t = blockIdx.x * 4 + wave; t < nt; t += waves_total) {
        const float4 *s4 = reinterpret_cast<const float4 *>(reinterpret_cast<const uint8_t *>(in) + (size_t)t * in_pitch);
        float4 *d4 = reinterpret_cast<float4 *>(reinterpret_cast<uint8_t *>(out) + (size_t)t * out_pitch);
        for (int j = lane; j < nin4; j += 64)
            l4[j] = s4[j];
        tx_wave_sync();
        if (!INV) {
            const int len3 = 3 * n;
            for (int i = lane; i < n; i += 64) {
                const int k = 2 * i;
                float re, im;
                if (k < n) {
                    re = -st[n + k] + st[n - 1 - k];
                    im = -st[len3 + k] + -st[len3 - 1 - k];
                } else {
                    re = -st[n + k] + -st[5 * n - 1 - k];
                    im = st[k - n] + -st[len3 - 1 - k];
                }
                const float2 e = l_exp[i];
                z[l_map[i]] = make_float2(re * e.y + im * e.x, re * e.x - im * e.y);
            }
        } else {
            for (int j = lane; j < n; j += 64) {
                const int k = j << 1;
                const float tre = st[2 * n - 1 - k], tim = st[k];
                const float2 e = l_exp[j];
                z[l_map[j]] = make_float2(tre * e.x - tim * e.y, tre * e.y + tim * e.x);
            }
        }
        tx_wave_sync();
        tx_fft_lds(z, d, l_cos, l_sched, l_b2, lane);
        const float2 *ex = l_exp;
        for (int i = lane; i < q; i += 64) {
            const int i0 = q + i, i1 = q - i - 1;
            const float2 e0 = ex[i0], e1 = ex[i1];
            if (!INV) {
                const float2 s1 = z[TX_PAD(i1)], s0 = z[TX_PAD(i0)];
                const float a = s0.x * e0.y - s0.y * e0.x, b = s0.x * e0.x + s0.y * e0.y;
                const float c = s1.x * e1.y - s1.y * e1.x, f = s1.x * e1.x + s1.y * e1.y;
                reinterpret_cast<float2 *>(st)[i1] = make_float2(f, a);
                reinterpret_cast<float2 *>(st)[i0] = make_float2(b, c);
            } else {
                const float2 z1 = z[TX_PAD(i1)], z0 = z[TX_PAD(i0)], s1 = make_float2(z1.y, z1.x), s0 = make_float2(z0.y, z0.x);
                const float a = s1.x * e1.y - s1.y * e1.x, b = s1.x * e1.x + s1.y * e1.y;
                const float c = s0.x * e0.y - s0.y * e0.x, f = s0.x * e0.x + s0.y * e0.y;
                reinterpret_cast<float2 *>(st)[i1] = make_float2(a, f);
                reinterpret_cast<float2 *>(st)[i0] = make_float2(c, b);
            }
        }
        tx_wave_sync();
        for (int j = lane; j < n / 2; j += 64)
            d4[j] = l4[j];
        tx_wave_sync();
    }
}

/*
 * k_mdct_z — the kernel for contiguous, 8-byte aligned batches: no staging area at all, only the padded work array
 * (4.1 KiB at N = 1024) per wave, so 20 waves fit a CU instead of 12 (the transform is a chain of dependent LDS round
 * trips; PMC had LDS 40 %, VALU 30 %, HBM 27 % busy - more waves in flight is what it lacked).
 *   - fold / pre-twiddle straight from global memory: work-array elements i and n-1-i read the two halves of the
 *     SAME four (forward) or two (inverse) float2 pairs of the input, so one lane produces both from coalesced 8-byte
 *     loads and every input byte is loaded exactly once;
 *   - post-twiddle straight to global memory: outputs (2*i1, 2*i1+1) and (2*i0, 2*i0+1) leave as two coalesced 8-byte
 *     stores.
 * Tables (map, exp, twiddles, butterfly lists) are copied to LDS once per workgroup; waves loop over transforms.
 * Same float operations in the same order as k_mdct.
 */
template <int INV, bool TL, bool WG = false>
__global__ __launch_bounds__(1024) void k_mdct_z(TxDev d, const uint8_t *blob, int blob_bytes, const float *in, size_t in_pitch,
                                                 float *out, size_t out_pitch, int nt, int waves_total)
{
    extern __shared__ __align__(16) uint8_t lds_raw[];
    /* WG: the workgroup is one team on one transform at a time (`wave` 0, `lane` the thread index, stride the workgroup) */
    const int wave = WG ? 0 : __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = WG ? (int)threadIdx.x : (int)(threadIdx.x & 63);
    const int TS = WG ? (int)blockDim.x : 64;
    /* TL: the context's tables are copied to LDS once per workgroup (blob_bytes > 0); otherwise they stay in global memory
     * (L2-resident) and blob_bytes is 0: large transforms, where the copy would cost the workgroup most of its waves */
    if (TL) {
        const uint4 *s4 = reinterpret_cast<const uint4 *>(blob);
        uint4 *l4 = reinterpret_cast<uint4 *>(lds_raw);
        for (int i = threadIdx.x; i < blob_bytes / 16; i += blockDim.x)
            l4[i] = s4[i];
        __syncthreads();
    }
    const uint8_t *const tbase = TL ? lds_raw : blob;
    const int *l_map = reinterpret_cast<const int *>(tbase + (reinterpret_cast<const uint8_t *>(d.map) - blob));
    const float2 *l_exp = reinterpret_cast<const float2 *>(tbase + (reinterpret_cast<const uint8_t *>(d.exp) - blob));
    const float *l_cos = reinterpret_cast<const float *>(tbase + (reinterpret_cast<const uint8_t *>(d.cos_tab) - blob));
    const uint32_t *l_sched = reinterpret_cast<const uint32_t *>(tbase + (reinterpret_cast<const uint8_t *>(d.sched) - blob));
    const uint16_t *l_b2 = reinterpret_cast<const uint16_t *>(tbase + (reinterpret_cast<const uint8_t *>(d.blocks2) - blob));
    const int n = d.n, q = n >> 1;
    float2 *z = reinterpret_cast<float2 *>(lds_raw + ((blob_bytes + 15) & ~15) + wave * tx_z_bytes(n));

    for (int t = WG ? (int)blockIdx.x : (int)(blockIdx.x * (blockDim.x >> 6)) + wave; t < nt; t += WG ? (int)gridDim.x : waves_total) {
        const float2 *in2 = reinterpret_cast<const float2 *>(reinterpret_cast<const uint8_t *>(in) + (size_t)t * in_pitch);
        float2 *out2 = reinterpret_cast<float2 *>(reinterpret_cast<uint8_t *>(out) + (size_t)t * out_pitch);
        if (!INV) {
            /* ff_tx_mdct_fwd's fold (tx_template.c:1285-1296), elements i (k = 2i < n) and n-1-i (k >= n) together */
            for (int i = lane; i < q; i += TS) {
                const float2 p1 = in2[q + i];          /* x[n+2i],   x[n+2i+1]   */
                const float2 p2 = in2[q - 1 - i];      /* x[n-2-2i], x[n-1-2i]   */
                const float2 p3 = in2[3 * q + i];      /* x[3n+2i],  x[3n+2i+1]  */
                const float2 p4 = in2[3 * q - 1 - i];  /* x[3n-2-2i], x[3n-1-2i] */
                const int j = n - 1 - i;
                const float re0 = -p1.x + p2.y, im0 = -p3.x + -p4.y;
                const float re1 = -p4.x + -p3.y, im1 = p2.x + -p1.y;
                const float2 e0 = l_exp[i], e1 = l_exp[j];
                z[l_map[i]] = make_float2(re0 * e0.y + im0 * e0.x, re0 * e0.x - im0 * e0.y);
                z[l_map[j]] = make_float2(re1 * e1.y + im1 * e1.x, re1 * e1.x - im1 * e1.y);
            }
        } else {
            /* ff_tx_mdct_inv's pre-twiddle (tx_template.c:1321-1328) in input order, elements j and n-1-j together */
            for (int j = lane; j < q; j += TS) {
                const float2 f = in2[j];               /* x[2j],      x[2j+1]     */
                const float2 g = in2[n - 1 - j];       /* x[2n-2-2j], x[2n-1-2j]  */
                const int j1 = n - 1 - j;
                const float2 e0 = l_exp[j], e1 = l_exp[j1];
                z[l_map[j]] = make_float2(g.y * e0.x - f.x * e0.y, g.y * e0.y + f.x * e0.x);
                z[l_map[j1]] = make_float2(f.y * e1.x - g.x * e1.y, f.y * e1.y + g.x * e1.x);
            }
        }
        tx_team_sync<WG>();
        tx_fft_lds<WG>(z, d, l_cos, l_sched, l_b2, lane);
        for (int i = lane; i < q; i += TS) {
            const int i0 = q + i, i1 = q - i - 1;
            const float2 e0 = l_exp[i0], e1 = l_exp[i1];
            if (!INV) {
                const float2 s1 = z[TX_PAD(i1)], s0 = z[TX_PAD(i0)];
                const float a = s0.x * e0.y - s0.y * e0.x, b = s0.x * e0.x + s0.y * e0.y;
                const float c = s1.x * e1.y - s1.y * e1.x, f = s1.x * e1.x + s1.y * e1.y;
                out2[i1] = make_float2(f, a);
                out2[i0] = make_float2(b, c);
            } else {
                const float2 z1 = z[TX_PAD(i1)], z0 = z[TX_PAD(i0)], s1 = make_float2(z1.y, z1.x), s0 = make_float2(z0.y, z0.x);
                const float a = s1.x * e1.y - s1.y * e1.x, b = s1.x * e1.x + s1.y * e1.y;
                const float c = s0.x * e0.y - s0.y * e0.x, f = s0.x * e0.x + s0.y * e0.y;
                out2[i1] = make_float2(a, f);
                out2[i0] = make_float2(c, b);
            }
        }
        tx_team_sync<WG>();
    }
}

/*
 * k_fft_z — AV_TX_FLOAT_FFT, power-of-two (ff_tx_fft + the split-radix codelets, libavutil/tx_template.c:540-749): the
 * complex input is read in order (coalesced 8-byte loads) and SCATTERED through the inverse of the reference's gather
 * permutation into the padded LDS work array, transformed there by the same flattened split-radix network as the MDCT,
 * and written out in order.  Forward and inverse differ by the permutation only.  16 B moved per complex sample.
 */
template <bool TL, bool WG = false>
__global__ __launch_bounds__(1024) void k_fft_z(TxDev d, const uint8_t *blob, int blob_bytes, const float *in, size_t in_pitch,
                                                float *out, size_t out_pitch, int nt, int waves_total)
{
    extern __shared__ __align__(16) uint8_t lds_raw[];
    const int wave = WG ? 0 : __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); /* WG: as in k_mdct_z */
    const int lane = WG ? (int)threadIdx.x : (int)(threadIdx.x & 63);
    const int TS = WG ? (int)blockDim.x : 64;
    /* TL: the context's tables are copied to LDS once per workgroup (blob_bytes > 0); otherwise they stay in global memory
     * (L2-resident) and blob_bytes is 0: large transforms, where the copy would cost the workgroup most of its waves */
    if (TL) {
        const uint4 *s4 = reinterpret_cast<const uint4 *>(blob);
        uint4 *l4 = reinterpret_cast<uint4 *>(lds_raw);
        for (int i = threadIdx.x; i < blob_bytes / 16; i += blockDim.x)
            l4[i] = s4[i];
        __syncthreads();
    }
    const uint8_t *const tbase = TL ? lds_raw : blob;
    const int *l_map = reinterpret_cast<const int *>(tbase + (reinterpret_cast<const uint8_t *>(d.map) - blob));
    const float *l_cos = reinterpret_cast<const float *>(tbase + (reinterpret_cast<const uint8_t *>(d.cos_tab) - blob));
    const uint32_t *l_sched = reinterpret_cast<const uint32_t *>(tbase + (reinterpret_cast<const uint8_t *>(d.sched) - blob));
    const uint16_t *l_b2 = reinterpret_cast<const uint16_t *>(tbase + (reinterpret_cast<const uint8_t *>(d.blocks2) - blob));
    const int n = d.n;
    float2 *z = reinterpret_cast<float2 *>(lds_raw + ((blob_bytes + 15) & ~15) + wave * tx_z_bytes(n));
    for (int t = WG ? (int)blockIdx.x : (int)(blockIdx.x * (blockDim.x >> 6)) + wave; t < nt; t += WG ? (int)gridDim.x : waves_total) {
        const float2 *in2 = reinterpret_cast<const float2 *>(reinterpret_cast<const uint8_t *>(in) + (size_t)t * in_pitch);
        float2 *out2 = reinterpret_cast<float2 *>(reinterpret_cast<uint8_t *>(out) + (size_t)t * out_pitch);
        for (int j = lane; j < n; j += TS)
            z[l_map[j]] = in2[j];
        tx_team_sync<WG>();
        tx_fft_lds<WG>(z, d, l_cos, l_sched, l_b2, lane);
        for (int i = lane; i < n; i += TS)
            out2[i] = z[TX_PAD(i)];
        tx_team_sync<WG>();
    }
}

/*
 * k_rdft — AV_TX_FLOAT_RDFT, power-of-two (ff_tx_rdft_r2c / _c2r, libavutil/tx_template.c:1601-1716): len reals <-> len/2 + 1
 * complex bins through the len/2-point complex FFT of k_fft_z plus the pass that separates / merges the even and odd halves.
 * r2c: the reals are read as len/2 complex samples and scattered into the work array, transformed, and bins i and len/2 - i
 * are produced together from the work array straight into global memory; c2r runs the same pass on the way in.  d.exp
 * holds the reference's table as floats: fact[8], tcos[len/4], tsin[len/4].
 */
/* RLG = 8 / 9 / 10: the len/2-point FFT runs the register-resident radix core (kernels/tx_radix_core.h; the work array in natural
 * order, results within the float tolerance); 0: the reference-order split-radix network (bit-identical) */
template <int INV, bool TL, int RLG = 0>
__global__ __launch_bounds__(1024) void k_rdft(TxDev d, const uint8_t *blob, int blob_bytes, const float *in, size_t in_pitch, float *out,
                                               size_t out_pitch, int nt, int waves_total, const float2 *wtab)
{
    extern __shared__ __align__(16) uint8_t lds_raw[];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    /* TL: the context's tables are copied to LDS once per workgroup (blob_bytes > 0); otherwise they stay in global memory
     * (L2-resident) and blob_bytes is 0: large transforms, where the copy would cost the workgroup most of its waves */
    if (TL) {
        const uint4 *s4 = reinterpret_cast<const uint4 *>(blob);
        uint4 *l4 = reinterpret_cast<uint4 *>(lds_raw);
        for (int i = threadIdx.x; i < blob_bytes / 16; i += blockDim.x)
            l4[i] = s4[i];
        __syncthreads();
    }
    const uint8_t *const tbase = TL ? lds_raw : blob;
    const int *l_map = reinterpret_cast<const int *>(tbase + (reinterpret_cast<const uint8_t *>(d.map) - blob));
    const float *fact = reinterpret_cast<const float *>(tbase + (reinterpret_cast<const uint8_t *>(d.exp) - blob));
    const float *l_cos = reinterpret_cast<const float *>(tbase + (reinterpret_cast<const uint8_t *>(d.cos_tab) - blob));
    const uint32_t *l_sched = reinterpret_cast<const uint32_t *>(tbase + (reinterpret_cast<const uint8_t *>(d.sched) - blob));
    const uint16_t *l_b2 = reinterpret_cast<const uint16_t *>(tbase + (reinterpret_cast<const uint8_t *>(d.blocks2) - blob));
    auto zi = [&](int j) { return RLG ? TX_PAD(j) : l_map[j]; };
    FrTw<RLG ? RLG : 8> rtw;
    if (RLG)
        fr_load_all_tw<RLG ? RLG : 8, INV>(rtw, wtab, lane);
    const int len2 = d.n, len4 = len2 >> 1;
    const float *tcos = fact + 8, *tsin = tcos + len4;
    const float f0 = fact[0], f1 = fact[1], f2 = fact[2], f3 = fact[3], f4 = fact[4], f5 = fact[5], f6 = fact[6], f7 = fact[7];
    float2 *z = reinterpret_cast<float2 *>(lds_raw + ((blob_bytes + 15) & ~15) + wave * tx_z_bytes(len2));
    /* bins i and len2 - i (0 < i < len4): the reference's loop body */
    auto pair = [&](int i, float2 a, float2 b, float2 &oa, float2 &ob) {
        const float t0r = f4 * (a.x + b.x), t0i = f5 * (a.y - b.y);
        const float t1r = f6 * (a.y + b.y), t1i = f7 * (a.x - b.x);
        const float c = tcos[i], sn = tsin[i];
        const float t2r = t1r * c - t1i * sn, t2i = t1r * sn + t1i * c;
        oa = make_float2(t0r + t2r, t2i - t0i);
        ob = make_float2(t0r - t2r, t2i + t0i);
    };
    for (int t = blockIdx.x * (blockDim.x >> 6) + wave; t < nt; t += waves_total) {
        const float2 *in2 = reinterpret_cast<const float2 *>(reinterpret_cast<const uint8_t *>(in) + (size_t)t * in_pitch);
        float2 *out2 = reinterpret_cast<float2 *>(reinterpret_cast<uint8_t *>(out) + (size_t)t * out_pitch);
        if (!INV) {
            for (int j = lane; j < len2; j += 64)
                z[zi(j)] = in2[j];
        } else {
            for (int i = lane; i <= len4; i += 64) {
                if (i == 0) {
                    const float re = in2[0].x, im = in2[len2].x; /* data[0].im = data[len2].re */
                    z[zi(0)] = make_float2(f0 * (re + im), f1 * (re - im));
                } else if (i == len4) {
                    const float2 v = in2[len4];
                    z[zi(len4)] = make_float2(f2 * v.x, f3 * v.y);
                } else {
                    float2 oa, ob;
                    pair(i, in2[i], in2[len2 - i], oa, ob);
                    z[zi(i)] = oa;
                    z[zi(len2 - i)] = ob;
                }
            }
        }
        tx_wave_sync();
        if constexpr (RLG != 0)
            fr_fft_lds<RLG, INV>(z, rtw, lane);
        else
            tx_fft_lds(z, d, l_cos, l_sched, l_b2, lane);
        if (!INV && d.half) {
            /* ff_tx_rdft_r2r / _r2i (tx_template.c:1718-1830, the len % 4 == 0 codelets): only the real / only the imaginary parts
             * of the bins, len/2 + 1 resp. len/2 floats.  The reference works in place over the FFT's output and reads every value
             * before it overwrites it, so this is the same function of the FFT output; bin len/4 enters the loop AFTER its two
             * scalings, tcos[len/4] is tsin[0] and tsin[len/4] the zero behind the table, and r2i's out[len/2 - 1] is the FFT's own
             * data[len/2 - 1].im (the copy loop reads a slot the loop never wrote). */
            float *o = reinterpret_cast<float *>(out2);
            const float2 d4 = z[TX_PAD(len4)];
            const float2 s4 = make_float2(f2 * d4.x, f3 * d4.y);
            for (int i = lane; i <= len4; i += 64) {
                if (i == 0) {
                    if (d.half == 1) {
                        const float2 v = z[TX_PAD(0)];
                        o[0] = f0 * (v.x + v.y);
                        o[len2] = f1 * (v.x - v.y);
                    } else {
                        o[len2 - 1] = z[TX_PAD(len2 - 1)].y;
                    }
                } else {
                    const float2 sf = i == len4 ? s4 : z[TX_PAD(i)], sl = i == len4 ? s4 : z[TX_PAD(len2 - i)];
                    const float t1 = f6 * (sf.y + sl.y), t2 = f7 * (sf.x - sl.x);
                    const float c = tcos[i], sn = tsin[i];
                    if (d.half == 1) {
                        const float t0 = f4 * (sf.x + sl.x);
                        const float t3 = t1 * c - t2 * sn;
                        o[i] = t0 + t3;
                        if (i < len4)
                            o[len2 - i] = t0 - t3;
                    } else {
                        const float t0 = f5 * (sf.y - sl.y);
                        const float t3 = t1 * sn + t2 * c;
                        o[i - 1] = t3 - t0;
                        if (i < len4)
                            o[len2 - 1 - i] = t0 + t3;
                    }
                }
            }
        } else if (!INV) {
            for (int i = lane; i <= len4; i += 64) {
                if (i == 0) {
                    const float2 v = z[TX_PAD(0)];
                    out2[0] = make_float2(f0 * (v.x + v.y), 0.0f);
                    out2[len2] = make_float2(f1 * (v.x - v.y), 0.0f); /* [0].im moves to the last bin, as convention requires */
                } else if (i == len4) {
                    const float2 v = z[TX_PAD(len4)];
                    out2[len4] = make_float2(f2 * v.x, f3 * v.y);
                } else {
                    float2 oa, ob;
                    pair(i, z[TX_PAD(i)], z[TX_PAD(len2 - i)], oa, ob);
                    out2[i] = oa;
                    out2[len2 - i] = ob;
                }
            }
        } else {
            for (int i = lane; i < len2; i += 64)
                out2[i] = z[TX_PAD(i)];
        }
        tx_wave_sync();
    }
}

/*
 * k_mdct_pfa — MDCT lengths 2 * 15 * 2^k (CELT 120..960, AAC-960 240 / 1920): ff_tx_mdct_pfa_15xM_fwd / _inv
 * (libavutil/tx_template.c:1425-1600), fft15 = 5 x fft3 + 3 x fft5 (:175-245,463-476), bit-identical floats.
 * A wave takes G transforms at once, G * m = 64: lane (g, i) runs sub-transform i of transform g —
 *   1. the fold / pre-twiddle runs in INPUT order with coalesced 8-byte loads (the reference twiddles point k with
 *      exp[k >> 1] wherever the map sends it, so the value depends on k alone) and parks the points in LDS;
 *   2. each lane gathers its 15 points through the Ruritanian input map (neighbouring lanes sit 15 points apart: an LDS
 *      gather, not an HBM one); after a wave barrier the same LDS bytes become the work array: the 15-point transform
 *      runs in registers and its outputs land at sub_map[i] + d * m;
 *   3. the 15 * G in-place m-point split-radix transforms are ONE flattened butterfly schedule over the wave's work
 *      array (the power-of-two kernels' tx_fft_lds, with the union of the arrays' butterfly lists);
 *   4. post-twiddle through the CRT output map, straight to global memory as 8-byte stores.
 */
/*
 * k_dct — AV_TX_FLOAT_DCT, power-of-two (ff_tx_dctII / ff_tx_dctIII, libavutil/tx_template.c:1832-2002): N reals <-> N reals
 * through the N-point RDFT above, i.e. the N/2-point complex FFT in LDS plus two passes.  One wave per transform.
 *   DCT-II:  fold in[i], in[N-1-i] into the sequence the RDFT takes (straight into the work array), FFT, the r2c pass, then per
 *            bin k the rotation by exp[] that yields out[2k] and the term t_k; the odd outputs are the reference's running sum
 *            out[N-1] = Re X[N/2], out[2k-1] = out[2k+1] + t_k, evaluated by one lane in exactly that order (the order of float
 *            additions is the result); the other waves of the workgroup hide it.  Outputs leave as coalesced float2.
 *   DCT-III: the bins the reference builds in place (each from in[2k-1], in[2k], in[2k+1]; bin 0 = in[0], Nyquist = 2 in[N-1])
 *            are formed on the fly inside the c2r pass, FFT, then the unfold out[i], out[N-1-i] from the work array.
 * d.exp holds fact[8], tcos[N/4], tsin[N/4] (the RDFT's) followed by the DCT's exp[N + N/2].  acc: N/2 + 1 floats per wave
 * behind the work array (DCT-II only).
 */
/* RLG = 8 / 9 / 10: the len/2-point FFT runs the register-resident radix core (kernels/tx_radix_core.h; the work array in natural
 * order, results within the float tolerance); 0: the reference-order split-radix network (bit-identical) */
template <int INV, bool TL, int RLG = 0>
__global__ __launch_bounds__(1024) void k_dct(TxDev d, const uint8_t *blob, int blob_bytes, const float *in, size_t in_pitch, float *out,
                                              size_t out_pitch, int nt, int waves_total, const float2 *wtab)
{
    extern __shared__ __align__(16) uint8_t lds_raw[];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    if (TL) {
        const uint4 *s4 = reinterpret_cast<const uint4 *>(blob);
        uint4 *l4 = reinterpret_cast<uint4 *>(lds_raw);
        for (int i = threadIdx.x; i < blob_bytes / 16; i += blockDim.x)
            l4[i] = s4[i];
        __syncthreads();
    }
    const uint8_t *const tbase = TL ? lds_raw : blob;
    const int *l_map = reinterpret_cast<const int *>(tbase + (reinterpret_cast<const uint8_t *>(d.map) - blob));
    const float *fact = reinterpret_cast<const float *>(tbase + (reinterpret_cast<const uint8_t *>(d.exp) - blob));
    const float *l_cos = reinterpret_cast<const float *>(tbase + (reinterpret_cast<const uint8_t *>(d.cos_tab) - blob));
    const uint32_t *l_sched = reinterpret_cast<const uint32_t *>(tbase + (reinterpret_cast<const uint8_t *>(d.sched) - blob));
    const uint16_t *l_b2 = reinterpret_cast<const uint16_t *>(tbase + (reinterpret_cast<const uint8_t *>(d.blocks2) - blob));
    auto zi = [&](int j) { return RLG ? TX_PAD(j) : l_map[j]; };
    FrTw<RLG ? RLG : 8> rtw;
    if (RLG)
        fr_load_all_tw<RLG ? RLG : 8, INV>(rtw, wtab, lane);
    const int len2 = d.n, len4 = len2 >> 1, N = 2 * len2;
    const float *tcos = fact + 8, *tsin = tcos + len4, *dexp = tsin + len4;
    const float f0 = fact[0], f1 = fact[1], f2 = fact[2], f3 = fact[3], f4 = fact[4], f5 = fact[5], f6 = fact[6], f7 = fact[7];
    const size_t per_wave = tx_z_bytes(len2) + (INV ? 0 : (((size_t)len2 + 1) * 4 + 15) & ~(size_t)15);
    uint8_t *area = lds_raw + ((blob_bytes + 15) & ~15) + wave * per_wave;
    float2 *z = reinterpret_cast<float2 *>(area);
    float *zf = reinterpret_cast<float *>(area);
    float *acc = reinterpret_cast<float *>(area + tx_z_bytes(len2));
    auto pair = [&](int i, float2 a, float2 b, float2 &oa, float2 &ob) { /* the RDFT's loop body, as in k_rdft */
        const float t0r = f4 * (a.x + b.x), t0i = f5 * (a.y - b.y);
        const float t1r = f6 * (a.y + b.y), t1i = f7 * (a.x - b.x);
        const float c = tcos[i], sn = tsin[i];
        const float t2r = t1r * c - t1i * sn, t2i = t1r * sn + t1i * c;
        oa = make_float2(t0r + t2r, t2i - t0i);
        ob = make_float2(t0r - t2r, t2i + t0i);
    };
    /* the workgroup's waves step together (the forward transform's running sums are evaluated for all of them at once, below):
     * wave w takes transform t0 + w; a wave past the end of the batch only keeps the barriers company */
    const int W = blockDim.x >> 6;
    for (int t0 = blockIdx.x * W; t0 < nt; t0 += waves_total) {
        const int t = t0 + wave;
        const bool active = t < nt;
        if (INV && !active)
            continue; /* (no barriers on the inverse path) */
        const float *x = reinterpret_cast<const float *>(reinterpret_cast<const uint8_t *>(in) + (size_t)t * in_pitch);
        float *y = reinterpret_cast<float *>(reinterpret_cast<uint8_t *>(out) + (size_t)t * out_pitch);
        if (!INV && active) {
            /* element c = (y[2c], y[2c+1]) and its mirror len2-1-c = (y[N-2-2c], y[N-1-2c]) from two coalesced float2 loads */
            const float2 *x2 = reinterpret_cast<const float2 *>(x);
            for (int c = lane; c < len4; c += 64) {
                const float2 a = x2[c], b = x2[len2 - 1 - c];
                const float s0 = dexp[N + 2 * c], s1 = dexp[N + 2 * c + 1];
                const float p1 = (a.x + b.y) * 0.5f, p2 = (a.x - b.y) * s0;
                const float q1 = (a.y + b.x) * 0.5f, q2 = (a.y - b.x) * s1;
                z[zi(c)] = make_float2(p1 + p2, q1 + q2);
                z[zi(len2 - 1 - c)] = make_float2(q1 - q2, p1 - p2);
            }
        } else if (INV) {
            /* bin k of the sequence ff_tx_dctIII hands the c2r transform */
            auto bin = [&](int k) {
                const int j = 2 * k;
                const float v1 = x[j], v2 = x[j - 1] - x[j + 1];
                const float e1 = dexp[N - j], e2 = dexp[j];
                return make_float2(e1 * v2 + e2 * v1, e1 * v1 - e2 * v2);
            };
            for (int i = lane; i <= len4; i += 64) {
                if (i == 0) {
                    const float re = x[0], im = 2 * x[N - 1];
                    z[zi(0)] = make_float2(f0 * (re + im), f1 * (re - im));
                } else if (i == len4) {
                    const float2 v = bin(len4);
                    z[zi(len4)] = make_float2(f2 * v.x, f3 * v.y);
                } else {
                    float2 oa, ob;
                    pair(i, bin(i), bin(len2 - i), oa, ob);
                    z[zi(i)] = oa;
                    z[zi(len2 - i)] = ob;
                }
            }
        }
        tx_wave_sync();
        if (RLG && !active)
            continue; /* (no barriers with the radix core: the running sums are a wave's own scan) */
        if constexpr (RLG != 0)
            fr_fft_lds<RLG, INV>(z, rtw, lane);
        else if (INV || active)
            tx_fft_lds(z, d, l_cos, l_sched, l_b2, lane);
        if (!INV) {
          if (active) {
            /* X[k] -> out[2k] (parked in the bin's own slot) and t_k (acc[k]); acc[len2] = Re X[N/2] */
            auto rot = [&](int k, float2 X) {
                const float e1 = dexp[N - 2 * k], e2 = dexp[2 * k];
                acc[k] = e1 * X.x - e2 * X.y;
                z[TX_PAD(k)].x = e1 * X.y + e2 * X.x;
            };
            for (int i = lane; i <= len4; i += 64) {
                if (i == 0) {
                    const float2 v = z[TX_PAD(0)];
                    z[TX_PAD(0)].x = dexp[0] * (f0 * (v.x + v.y));
                    acc[len2] = f1 * (v.x - v.y);
                } else if (i == len4) {
                    const float2 v = z[TX_PAD(len4)];
                    rot(len4, make_float2(f2 * v.x, f3 * v.y));
                } else {
                    float2 oa, ob;
                    pair(i, z[TX_PAD(i)], z[TX_PAD(len2 - i)], oa, ob);
                    rot(i, oa);
                    rot(len2 - i, ob);
                }
            }
          }
            /* acc[k] becomes out[2k - 1]: out[N - 1] = Re X[N/2], out[2k - 1] = out[2k + 1] + t_k — a chain of float additions whose
             * order is the result, so nothing inside a transform runs it in parallel.  Across transforms it does: lane w of the
             * workgroup's first wave walks the chain of wave w's transform, the W chains cost the instruction slots of one
             * (round 1 had every wave walk its own with one live lane: 138 M transforms/s at N = 1024 against 342 M/s for the
             * DCT-III, which has no such chain).  Eight terms per trip through registers. */
            if constexpr (RLG != 0) {
                /* the same sums as a suffix scan of the wave's own terms: lane l owns acc[l C .. l C + C - 1], adds up its chunk from the
                 * top, and takes the sum of the chunks above it (and of acc[len2]) from a shuffle scan.  A different order of float
                 * additions than the reference's chain: within the tolerance, not bit-identical. */
                tx_wave_sync();
                constexpr int C = (1 << RLG) / 64;
                float sfx[C];
#pragma unroll
                for (int i = 0; i < C; i++)
                    sfx[i] = acc[lane * C + i];
#pragma unroll
                for (int i = C - 2; i >= 0; i--)
                    sfx[i] += sfx[i + 1];
                float incl = sfx[0];
#pragma unroll
                for (int dd = 1; dd < 64; dd <<= 1) {
                    const float o = __shfl_down(incl, dd);
                    if (lane + dd < 64)
                        incl += o;
                }
                float above = __shfl_down(incl, 1);
                above = (lane == 63 ? 0.0f : above) + acc[len2];
                tx_wave_sync();
#pragma unroll
                for (int i = 0; i < C; i++)
                    acc[lane * C + i] = sfx[i] + above;
                tx_wave_sync();
            } else {
            __syncthreads();
            if (wave == 0 && lane < W && t0 + lane < nt) {
                float *a = reinterpret_cast<float *>(lds_raw + ((blob_bytes + 15) & ~15) + lane * per_wave + tx_z_bytes(len2));
                /* eight terms per trip: two 16-byte LDS reads (the next trip's, issued ahead), eight dependent adds each writing
                 * the sum in place of its term, two 16-byte writes (len2 >= 8: a[len2 - 8 .. len2 - 1] is 16-byte aligned) */
                float next = a[len2];
                int k = len2 - 1;
                if (len2 < 8) { /* N = 8 */
                    for (; k > 0; k--) {
                        next += a[k];
                        a[k] = next;
                    }
                } else {
                float4 lo = *reinterpret_cast<const float4 *>(a + k - 7), hi = *reinterpret_cast<const float4 *>(a + k - 3);
                for (; k >= 15; k -= 8) {
                    const float4 nlo = *reinterpret_cast<const float4 *>(a + k - 15), nhi = *reinterpret_cast<const float4 *>(a + k - 11);
                    hi.w = next + hi.w; hi.z = hi.w + hi.z; hi.y = hi.z + hi.y; hi.x = hi.y + hi.x;
                    lo.w = hi.x + lo.w; lo.z = lo.w + lo.z; lo.y = lo.z + lo.y; lo.x = lo.y + lo.x;
                    next = lo.x;
                    *reinterpret_cast<float4 *>(a + k - 3) = hi;
                    *reinterpret_cast<float4 *>(a + k - 7) = lo;
                    lo = nlo;
                    hi = nhi;
                }
                /* the last trip: a[0 .. 7]; a[0] takes no part (out[-1] does not exist) */
                hi.w = next + hi.w; hi.z = hi.w + hi.z; hi.y = hi.z + hi.y; hi.x = hi.y + hi.x;
                lo.w = hi.x + lo.w; lo.z = lo.w + lo.z; lo.y = lo.z + lo.y;
                *reinterpret_cast<float4 *>(a + 4) = hi;
                *reinterpret_cast<float4 *>(a) = lo;
                }
            }
            __syncthreads();
            }
            if (!active)
                continue;
            float2 *y2 = reinterpret_cast<float2 *>(y);
            for (int k = lane; k < len2; k += 64)
                y2[k] = make_float2(z[TX_PAD(k)].x, acc[k + 1]);
        } else {
            for (int i = lane; i < len2; i += 64) {
                const int j = N - 1 - i;
                const float a = zf[2 * TX_PAD(i >> 1) + (i & 1)], b = zf[2 * TX_PAD(j >> 1) + (j & 1)];
                const float t1 = a + b, t2 = (a - b) * dexp[N + i];
                y[i] = t1 + t2;
                y[j] = t1 - t2;
            }
        }
        tx_wave_sync();
    }
}

#define TXBF(x, y, a, b) do { x = (a) - (b); y = (a) + (b); } while (0)
#define TXCMUL(dre, dim, are, aim, bre, bim) do { (dre) = (are) * (bre) - (aim) * (bim); (dim) = (are) * (bim) + (aim) * (bre); } while (0)
#define TXSMUL(dre, dim, are, aim, bre, bim) do { (dre) = (are) * (bre) - (aim) * (bim); (dim) = (are) * (bim) - (aim) * (bre); } while (0)

struct TxTab53 { float t[12]; float t7[6]; float t9[8]; }; /* ff_tx_tab_53, ff_tx_tab_7, ff_tx_tab_9 (tx_template.c:92-130) */

/* fft3 (tx_template.c:175-209): in[0..2], out o0, o1, o2 */
__device__ __forceinline__ void tx_fft3(const TxTab53 &T, const float2 &i0, const float2 &i1, const float2 &i2, float2 &o0, float2 &o1,
                                        float2 &o2)
{
    float2 t0 = i0, t1, t2;
    TXBF(t1.x, t2.y, i1.y, i2.y);
    TXBF(t1.y, t2.x, i1.x, i2.x);
    o0.x = t0.x + t2.x;
    o0.y = t0.y + t2.y;
    t1.x = T.t[8] * t1.x;
    t1.y = T.t[9] * t1.y;
    t2.x = T.t[10] * t2.x;
    t2.y = T.t[10] * t2.y;
    o1.x = t0.x - t2.x + t1.x;
    o1.y = t0.y - t2.y - t1.y;
    o2.x = t0.x - t2.x - t1.x;
    o2.y = t0.y - t2.y + t1.y;
}

/* DECL_FFT5 (tx_template.c:211-245): in[0..4] -> o[0..4] = out[D0..D4] */
__device__ __forceinline__ void tx_fft5(const TxTab53 &T, const float2 (&in)[5], float2 (&o)[5])
{
    float2 dc = in[0], z0[4], t[6];
    TXBF(t[1].y, t[0].x, in[1].x, in[4].x);
    TXBF(t[1].x, t[0].y, in[1].y, in[4].y);
    TXBF(t[3].y, t[2].x, in[2].x, in[3].x);
    TXBF(t[3].x, t[2].y, in[2].y, in[3].y);
    o[0].x = dc.x + t[0].x + t[2].x;
    o[0].y = dc.y + t[0].y + t[2].y;
    TXSMUL(t[4].x, t[0].x, T.t[0], T.t[2], t[2].x, t[0].x);
    TXSMUL(t[4].y, t[0].y, T.t[0], T.t[2], t[2].y, t[0].y);
    TXCMUL(t[5].x, t[1].x, T.t[4], T.t[6], t[3].x, t[1].x);
    TXCMUL(t[5].y, t[1].y, T.t[4], T.t[6], t[3].y, t[1].y);
    TXBF(z0[0].x, z0[3].x, t[0].x, t[1].x);
    TXBF(z0[0].y, z0[3].y, t[0].y, t[1].y);
    TXBF(z0[2].x, z0[1].x, t[4].x, t[5].x);
    TXBF(z0[2].y, z0[1].y, t[4].y, t[5].y);
    o[1].x = dc.x + z0[3].x;
    o[1].y = dc.y + z0[0].y;
    o[2].x = dc.x + z0[2].x;
    o[2].y = dc.y + z0[1].y;
    o[3].x = dc.x + z0[1].x;
    o[3].y = dc.y + z0[2].y;
    o[4].x = dc.x + z0[0].x;
    o[4].y = dc.y + z0[3].y;
}

/* fft7 (tx_template.c:250-340, float branch) on the sums p[k] / differences q[k] of the mirrored inputs in[k+1], in[6-k]: output
 * pair (k, 7 - k) is dc + C_k -/+ S_k with C / S three-term expressions evaluated in the reference's term order; t7 = (re, im)
 * pairs: cs[k] = t7[2k], sn[k] = t7[2k+1] */
__device__ __forceinline__ void tx_fft7(const TxTab53 &T, const float2 (&in)[7], float2 (&o)[7])
{
    const float c0 = T.t7[0], c1 = T.t7[2], c2 = T.t7[4], s0 = T.t7[1], s1 = T.t7[3], s2 = T.t7[5];
    float2 p[3], q[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        p[k] = make_float2(in[k + 1].x + in[6 - k].x, in[k + 1].y + in[6 - k].y);
        q[k] = make_float2(in[k + 1].x - in[6 - k].x, in[k + 1].y - in[6 - k].y);
    }
    o[0] = make_float2(in[0].x + p[0].x + p[1].x + p[2].x, in[0].y + p[0].y + p[1].y + p[2].y);
    const float2 z0 = make_float2(c0 * p[0].x - c2 * p[2].x - c1 * p[1].x, c0 * p[0].y - c1 * p[1].y - c2 * p[2].y);
    const float2 z1 = make_float2(c0 * p[2].x - c1 * p[0].x - c2 * p[1].x, c0 * p[2].y - c1 * p[0].y - c2 * p[1].y);
    const float2 z2 = make_float2(c0 * p[1].x - c2 * p[0].x - c1 * p[2].x, c0 * p[1].y - c2 * p[0].y - c1 * p[2].y);
    /* a[k]: what pair k adds to / takes from z[k] (x: to the real part, y: to the imaginary part) */
    const float2 a0 = make_float2(s2 * q[2].y + s1 * q[1].y + s0 * q[0].y, s0 * q[0].x + s1 * q[1].x + s2 * q[2].x);
    const float2 a1 = make_float2(s0 * q[2].y + s2 * q[1].y - s1 * q[0].y, s2 * q[1].x + s0 * q[2].x - s1 * q[0].x);
    const float2 a2 = make_float2(s2 * q[0].y + s1 * q[2].y - s0 * q[1].y, s2 * q[0].x + s1 * q[2].x - s0 * q[1].x);
    const float dre = in[0].x, dim = in[0].y;
    o[1] = make_float2(dre + (z0.x + a0.x), dim + (z0.y - a0.y));
    o[6] = make_float2(dre + (z0.x - a0.x), dim + (z0.y + a0.y));
    o[2] = make_float2(dre + (z1.x - a1.x), dim + (z1.y + a1.y));
    o[5] = make_float2(dre + (z1.x + a1.x), dim + (z1.y - a1.y));
    o[3] = make_float2(dre + (z2.x + a2.x), dim + (z2.y - a2.y));
    o[4] = make_float2(dre + (z2.x - a2.x), dim + (z2.y + a2.y));
}

/* fft9 (tx_template.c:342-461, float branch); t9 = (re, im) pairs */
__device__ __forceinline__ void tx_fft9(const TxTab53 &T, const float2 (&in)[9], float2 (&o)[9])
{
    const float t0r = T.t9[0], t0i = T.t9[1], t1r = T.t9[2], t1i = T.t9[3], t2r = T.t9[4], t2i = T.t9[5], t3r = T.t9[6], t3i = T.t9[7];
    float2 p[4], q[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        p[k] = make_float2(in[k + 1].x + in[8 - k].x, in[k + 1].y + in[8 - k].y);
        q[k] = make_float2(in[k + 1].x - in[8 - k].x, in[k + 1].y - in[8 - k].y);
    }
    const float2 w0 = make_float2(p[0].x - p[3].x, p[0].y - p[3].y), w1 = make_float2(p[1].x - p[3].x, p[1].y - p[3].y);
    const float2 w2 = make_float2(q[0].x - q[3].x, q[0].y - q[3].y), w3 = make_float2(q[1].x + q[3].x, q[1].y + q[3].y);
    float2 z0 = make_float2(in[0].x + p[2].x, in[0].y + p[2].y);
    const float2 z1 = make_float2(p[0].x + p[1].x + p[3].x, p[0].y + p[1].y + p[3].y);
    o[0] = make_float2(z0.x + z1.x, z0.y + z1.y);
    float2 x[5], y[5];
    y[3] = make_float2(t0i * (q[0].x - q[1].x + q[3].x), t0i * (q[0].y - q[1].y + q[3].y));
    x[3] = make_float2(z0.x + t0r * z1.x, z0.y + t0r * z1.y);
    z0 = make_float2(in[0].x + t0r * p[2].x, in[0].y + t0r * p[2].y);
    x[1] = make_float2(t1r * w0.x + t2i * w1.x, t1r * w0.y + t2i * w1.y);
    x[2] = make_float2(t2i * w0.x - t3r * w1.x, t2i * w0.y - t3r * w1.y);
    y[1] = make_float2(t1i * w2.x + t2r * w3.x, t1i * w2.y + t2r * w3.y);
    y[2] = make_float2(t2r * w2.x - t3i * w3.x, t2r * w2.y - t3i * w3.y);
    y[0] = make_float2(t0i * q[2].x, t0i * q[2].y);
    x[4] = make_float2(x[1].x + x[2].x, x[1].y + x[2].y);
    y[4] = make_float2(y[1].x - y[2].x, y[1].y - y[2].y);
    x[1] = make_float2(z0.x + x[1].x, z0.y + x[1].y);
    y[1] = make_float2(y[0].x + y[1].x, y[0].y + y[1].y);
    x[2] = make_float2(z0.x + x[2].x, z0.y + x[2].y);
    y[2] = make_float2(y[2].x - y[0].x, y[2].y - y[0].y);
    x[4] = make_float2(z0.x - x[4].x, z0.y - x[4].y);
    y[4] = make_float2(y[0].x - y[4].x, y[0].y - y[4].y);
#pragma unroll
    for (int k = 1; k <= 4; k++) {
        o[k] = make_float2(x[k].x + y[k].y, x[k].y - y[k].x);
        o[9 - k] = make_float2(x[k].x - y[k].y, x[k].y + y[k].x);
    }
}

/* the F-point transform of a sub-transform's inputs; o[d] is the output the reference writes at out[d * stride] */
template <int F>
__device__ __forceinline__ void tx_fft_small(const TxTab53 &T, const float2 (&f)[F], float2 (&o)[F])
{
    if constexpr (F == 3) {
        tx_fft3(T, f[0], f[1], f[2], o[0], o[1], o[2]);
    } else if constexpr (F == 5) {
        tx_fft5(T, f, o);
    } else if constexpr (F == 7) {
        tx_fft7(T, f, o);
    } else if constexpr (F == 9) {
        tx_fft9(T, f, o);
    } else { /* fft15 = 5 x fft3 + fft5_m1 | _m2 | _m3 (tx_template.c:463-476) */
        constexpr int D15[15] = { 0, 6, 12, 3, 9, 10, 1, 7, 13, 4, 5, 11, 2, 8, 14 }; /* the three fft5s' output slots */
        float2 tmp[15];
#pragma unroll
        for (int i = 0; i < 5; i++)
            tx_fft3(T, f[3 * i], f[3 * i + 1], f[3 * i + 2], tmp[i], tmp[i + 5], tmp[i + 10]);
#pragma unroll
        for (int b = 0; b < 3; b++) {
            float2 r[5];
            const float2 (&ti)[5] = *reinterpret_cast<const float2 (*)[5]>(&tmp[5 * b]);
            tx_fft5(T, ti, r);
#pragma unroll
            for (int c = 0; c < 5; c++)
                o[D15[5 * b + c]] = r[c];
        }
    }
}

/* C: sub-transforms per lane (1 while m <= 64; 2 covers m = 128, 4 m = 256, with G = 1).  INV 0 / 1: the forward / inverse MDCT;
 * 2: the prime-factor FFT alone (ff_tx_fft_pfa, either direction: the maps carry it) */
template <int INV, int F, int C>
__global__ __launch_bounds__(1024) void k_mdct_pfa(TxDev d, TxPfa P, TxTab53 T, const uint8_t *blob, int blob_bytes, const float *in,
                                                   size_t in_pitch, float *out, size_t out_pitch, int nt, int waves_total)
{
    extern __shared__ __align__(16) uint8_t lds_raw[];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    {
        const uint4 *s4 = reinterpret_cast<const uint4 *>(blob);
        uint4 *l4 = reinterpret_cast<uint4 *>(lds_raw);
        for (int i = threadIdx.x; i < blob_bytes / 16; i += blockDim.x)
            l4[i] = s4[i];
    }
    __syncthreads();
    auto lp = [&](const void *p) { return lds_raw + (reinterpret_cast<const uint8_t *>(p) - blob); };
    const int *l_in = reinterpret_cast<const int *>(lp(P.in_map)), *l_out = reinterpret_cast<const int *>(lp(P.out_map));
    const int *l_sub = reinterpret_cast<const int *>(lp(P.sub_map));
    const float2 *l_exp = reinterpret_cast<const float2 *>(lp(d.exp));
    const float *l_cos = reinterpret_cast<const float *>(lp(d.cos_tab));
    const uint32_t *l_sched = reinterpret_cast<const uint32_t *>(lp(d.sched));
    const uint16_t *l_b2 = reinterpret_cast<const uint16_t *>(lp(d.blocks2));
    const int n1 = P.n1, m = P.m, G = P.G, q = n1 >> 1;
    uint8_t *mine = lds_raw + ((blob_bytes + 15) & ~15) + wave * tx_z_bytes(d.n);
    float2 *z = reinterpret_cast<float2 *>(mine);  /* the work array; before that, the folded / pre-twiddled points w[g][k >> 1] */
    const int g = lane >> d.lg, si = lane & (m - 1); /* m >= 64: g = 0, si = lane */

    for (int t0 = (blockIdx.x * (blockDim.x >> 6) + wave) * G; t0 < nt; t0 += waves_total * G) {
        const int ng = min(G, nt - t0);
        /* 1. fold / pre-twiddle in INPUT order (the value of point k depends on k alone: exp[k >> 1] and the samples around it),
         *    points i and n1-1-i together as in k_mdct_z: coalesced 8-byte loads, every input byte loaded once */
        if (INV == 2) { /* ff_tx_fft_pfa: the points are the input itself */
            for (int r = 0; r < ng; r++) {
                const float2 *in2 = reinterpret_cast<const float2 *>(reinterpret_cast<const uint8_t *>(in) + (size_t)(t0 + r) * in_pitch);
                float2 *w = z + r * n1;
                for (int i = lane; i < n1; i += 64)
                    w[i] = in2[i];
            }
        }
        for (int e = lane; INV != 2 && e < ng * q; e += 64) {
            const int r = (int)(((uint32_t)e * (uint32_t)P.magic_q) >> 24), i = e - r * q, j = n1 - 1 - i;
            const float2 *in2 = reinterpret_cast<const float2 *>(reinterpret_cast<const uint8_t *>(in) + (size_t)(t0 + r) * in_pitch);
            const float2 e0 = l_exp[i], e1 = l_exp[j];
            float2 *w = z + r * n1;
            if (!INV) {
                const float2 p1 = in2[q + i], p2 = in2[q - 1 - i], p3 = in2[3 * q + i], p4 = in2[3 * q - 1 - i];
                const float re0 = -p1.x + p2.y, im0 = -p3.x + -p4.y;
                const float re1 = -p4.x + -p3.y, im1 = p2.x + -p1.y;
                w[i] = make_float2(re0 * e0.y + im0 * e0.x, re0 * e0.x - im0 * e0.y);
                w[j] = make_float2(re1 * e1.y + im1 * e1.x, re1 * e1.x - im1 * e1.y);
            } else {
                const float2 f = in2[i], h = in2[j];
                w[i] = make_float2(h.y * e0.x - f.x * e0.y, h.y * e0.y + f.x * e0.x);
                w[j] = make_float2(f.y * e1.x - h.x * e1.y, f.y * e1.y + h.x * e1.x);
            }
        }
        tx_wave_sync();
        /* 2. this lane's F points per sub-transform, through the Ruritanian input map */
        float2 f[C][F];
#pragma unroll
        for (int c = 0; c < C; c++) {
            const int sc = si + 64 * c;
            if (g < ng && sc < m) {
                const float2 *w = z + g * n1;
#pragma unroll
                for (int j = 0; j < F; j++)
                    f[c][j] = w[l_in[sc * F + j]];
            }
        }
        tx_wave_sync(); /* every lane has its inputs: the same bytes become the work array */
#pragma unroll
        for (int c = 0; c < C; c++) {
            const int sc = si + 64 * c;
            if (g < ng && sc < m) {
                float2 o[F];
                tx_fft_small<F>(T, f[c], o);
                const int base = g * n1 + l_sub[sc];
#pragma unroll
                for (int dd = 0; dd < F; dd++) {
                    const int idx = base + dd * m;
                    z[TX_PAD(idx)] = o[dd];
                }
            }
        }
        tx_wave_sync();
        /* 3. the F G sub-transforms */
        tx_fft_lds(z, d, l_cos, l_sched, l_b2, lane);
        /* 4. post-twiddle (the FFT: the CRT output map alone, tx_template.c:1078-1079) */
        if (INV == 2) {
            for (int r = 0; r < ng; r++) {
                float2 *out2 = reinterpret_cast<float2 *>(reinterpret_cast<uint8_t *>(out) + (size_t)(t0 + r) * out_pitch);
                for (int i = lane; i < n1; i += 64)
                    out2[i] = z[TX_PAD(r * n1 + l_out[i])];
            }
        }
        for (int e = lane; INV != 2 && e < ng * q; e += 64) {
            const int r = (int)(((uint32_t)e * (uint32_t)P.magic_q) >> 24), i = e - r * q;
            const int i0 = q + i, i1 = q - i - 1;
            const float2 z0 = z[TX_PAD(r * n1 + l_out[i0])], z1 = z[TX_PAD(r * n1 + l_out[i1])];
            float2 *out2 = reinterpret_cast<float2 *>(reinterpret_cast<uint8_t *>(out) + (size_t)(t0 + r) * out_pitch);
            if (INV) {
                const float2 e0 = l_exp[i0], e1 = l_exp[i1];
                const float2 s1 = make_float2(z1.y, z1.x), s0 = make_float2(z0.y, z0.x);
                float a, b, c, h;
                TXCMUL(a, b, s1.x, s1.y, e1.y, e1.x); /* z[i1].re, z[i0].im */
                TXCMUL(c, h, s0.x, s0.y, e0.y, e0.x); /* z[i0].re, z[i1].im */
                out2[i1] = make_float2(a, h);
                out2[i0] = make_float2(c, b);
            } else {
                const float2 e0 = l_exp[i0], e1 = l_exp[i1];
                float a, b, c, h;
                TXCMUL(a, b, z0.x, z0.y, e0.y, e0.x); /* dst[2 i1 + 1], dst[2 i0] */
                TXCMUL(c, h, z1.x, z1.y, e1.y, e1.x); /* dst[2 i0 + 1], dst[2 i1] */
                out2[i1] = make_float2(h, a);
                out2[i0] = make_float2(b, c);
            }
        }
        tx_wave_sync();
    }
}
#undef TXBF
#undef TXCMUL
#undef TXSMUL

/*
 * AV_TX_FULL_IMDCT (ff_tx_mdct_inv_full, libavutil/tx_template.c:1391-1408): the half inverse was written to the middle half of
 * each 2 * len row; the outer quarters are its mirror images, the first one negated.  One thread per float2 of a quarter.
 */
__global__ __launch_bounds__(256) void k_imdct_mirror(float *out, size_t out_pitch, int len, int nt)
{
    const int q2 = len >> 2; /* float2 per quarter (a quarter = len / 2 floats) */
    const int i = blockIdx.x * 256 + threadIdx.x, t = blockIdx.y;
    if (i >= q2 || t >= nt)
        return;
    float2 *row = reinterpret_cast<float2 *>(reinterpret_cast<uint8_t *>(out) + (size_t)t * out_pitch);
    const int h2 = len >> 1;                       /* float2 per half */
    const float2 a = row[h2 - 1 - i];              /* dst[len - 2 - 2i], dst[len - 1 - 2i] */
    const float2 b = row[h2 + i];                  /* dst[len + 2i], dst[len + 2i + 1]     */
    row[i] = make_float2(-a.y, -a.x);              /* dst[2i] = -dst[len - 1 - 2i], dst[2i + 1] = -dst[len - 2 - 2i] */
    row[2 * h2 - 1 - i] = make_float2(b.y, b.x);   /* dst[2 len - 2 - 2i] = dst[len + 2i + 1], dst[2 len - 1 - 2i] = dst[len + 2i] */
}

/* ---- host: tables ------------------------------------------------------------------------------- */
static int sr_perm(int i, int len, int inv)
{
    len >>= 1;
    if (len <= 1)
        return i & 1;
    if (!(i & len))
        return sr_perm(i, len, inv) * 2;
    len >>= 1;
    return sr_perm(i, len, inv) * 4 + 1 - 2 * (!(i & len) ^ inv);
}

static void sr_schedule(int o, int n, int lg, std::vector<uint32_t> *lev, std::vector<uint16_t> *b2)
{
    if (n == 1)
        return;
    if (n == 2) {
        b2->push_back((uint16_t)TX_PAD(o));
        return;
    }
    const int q = n >> 2;
    sr_schedule(o, n >> 1, lg - 1, lev, b2);
    sr_schedule(o + 2 * q, q, lg - 2, lev, b2);
    sr_schedule(o + 3 * q, q, lg - 2, lev, b2);
    for (int k = 0; k < q; k++)
        lev[lg].push_back((uint32_t)TX_PAD(o + k) | ((uint32_t)k << 16));
}

extern "C" void ffhip_tx_uninit(FFHipTXContext **pctx)
{
    if (!pctx || !*pctx)
        return;
    FFHipTXContext *c = *pctx;
    FFHipDeviceGuard dg(c->device);
    if (c->wide)
        ffhip_txw_free(c->wide);
    if (c->dcst1)
        ffhip_dcst1_free(c->dcst1);
    if (c->dev)
        (void)hipFree(c->dev);
    if (c->wtab)
        (void)hipFree(c->wtab);
    if (c->stage)
        (void)hipFree(c->stage);
    delete c;
    *pctx = nullptr;
}

static void tx_single(FFHipTXContext *s, void *out, void *in, ptrdiff_t stride);

static int mulinv(int n, int m)
{
    n = n % m;
    for (int x = 1; x < m; x++)
        if (((n * x) % m) == 1)
            return x;
    return 0;
}

/* tables of the 15xM prime-factor MDCT (ff_tx_mdct_pfa_init, libavutil/tx_template.c:1425-1469; ff_tx_gen_compound_mapping with
 * opts == NULL, libavutil/tx.c:75-121; TX_EMBED_INPUT_PFA_MAP, tx_priv.h:275-284; ff_tx_mdct_gen_exp, tx_template.c:2107-2134) */
static int tx_init_pfa(FFHipTXContext *c, float scale_f, int F, bool is_fft)
{
    const int n1 = is_fft ? c->len : c->len >> 1, m = n1 / F, G = m < 64 ? 64 / m : 1, inv = c->inv;
    int lg = 0;
    while ((1 << lg) < m)
        lg++;
    std::vector<int> in_map(n1), out_map(n1), sub_map(m);
    for (int i = 0; i < m; i++)
        sub_map[-sr_perm(i, m, inv) & (m - 1)] = i; /* the sub-transform's SCATTER revtab */
    const int m_inv = mulinv(m, F), n_inv = mulinv(F, m);
    for (int j = 0; j < m; j++)
        for (int i = 0; i < F; i++) {
            in_map[j * F + i] = (i * m + j * F) % n1;
            out_map[(i * m * m_inv + j * F * n_inv) % n1] = i * m + j;
        }
    if (is_fft) {
        /* ff_tx_fft_pfa_init (tx_template.c:1032-1045): the compound map is generated for the forward direction and flattened
         * through the F-point codelet's own input map, which carries the direction: identity / reversed ACs for 3, 5, 7, 9
         * (ff_tx_gen_default_map, tx.c:525-542), the 3 x 5 map for 15 (ff_tx_gen_pfa_input_map, tx.c:44-72: for the inverse it is
         * the scatter form with its ACs reversed) */
        int fm[15];
        fm[0] = 0;
        for (int i = 1; i < F; i++)
            fm[i] = inv ? F - i : i;
        if (F == 15) {
            for (int a = 0; a < 5; a++)
                for (int b = 0; b < 3; b++) {
                    if (inv)
                        fm[(a * 3 + b * 5) % 15] = a * 3 + b;
                    else
                        fm[a * 3 + b] = (a * 3 + b * 5) % 15;
                }
            if (inv)
                for (int w = 1; w <= 7; w++)
                    std::swap(fm[w], fm[15 - w]);
        }
        for (int k = 0; k < n1; k += F) {
            int t[15];
            memcpy(t, &in_map[k], sizeof(int) * F);
            for (int i = 0; i < F; i++)
                in_map[k + i] = t[fm[i]];
        }
    }
    if (inv && !is_fft)
        for (int i = 0; i < m; i++) {
            int *p = &in_map[i * F + 1];
            for (int j = 0; j < (F - 1) >> 1; j++)
                std::swap(p[j], p[F - j - 2]);
        }
    if (F == 15 && !is_fft) /* the 15-point transform is itself 3 x 5: its input map is embedded (TX_EMBED_INPUT_PFA_MAP) */
        for (int k = 0; k < n1; k += 15) {
            int t[15];
            memcpy(t, &in_map[k], sizeof(t));
            for (int a = 0; a < 5; a++)
                for (int b = 0; b < 3; b++)
                    in_map[k + a * 3 + b] = t[(a * 3 + b * 5) % 15];
        }
    /* the natural-order table only: the reference's permuted copy for the inverse pre-twiddle is exp[map[i]], i.e. the
     * twiddle of input point k is exp[k >> 1] in both directions */
    std::vector<float2> ex(is_fft ? 2 : n1); /* an FFT has no twiddle table of its own */
    if (!is_fft) {
        const double sc = scale_f;
        const double theta = (sc < 0 ? n1 : 0) + 1.0 / 8.0, rt = sqrt(fabs(sc));
        for (int i = 0; i < n1; i++) {
            const double alpha = M_PI_2 * (i + theta) / n1;
            ex[i].x = (float)(cos(alpha) * rt);
            ex[i].y = (float)(sin(alpha) * rt);
        }
    }
    TxDev &d = c->d;
    memset(&d, 0, sizeof(d));
    d.n = G * n1; d.lg = lg;
    std::vector<float> cosv;
    for (int l = 2; l <= lg; l++) {
        const int mm = 1 << l;
        const double freq = 2 * M_PI / mm;
        d.cos_off[l] = (int)cosv.size();
        for (int i = 0; i < mm / 4; i++)
            cosv.push_back((float)cos(i * freq));
        cosv.push_back(0.0f);
    }
    /* one butterfly schedule for the wave's F G sub-transforms */
    std::vector<uint32_t> lev[16];
    std::vector<uint16_t> b2;
    for (int g = 0; g < G; g++)
        for (int a = 0; a < F; a++)
            sr_schedule(g * n1 + a * m, m, lg, lev, &b2);
    std::vector<uint32_t> sched;
    for (int l = 2; l <= lg; l++) {
        d.sched_off[l] = (int)sched.size();
        d.sched_cnt[l] = (int)lev[l].size();
        sched.insert(sched.end(), lev[l].begin(), lev[l].end());
        d.max_cnt = d.sched_cnt[l] > d.max_cnt ? d.sched_cnt[l] : d.max_cnt;
    }
    d.nblocks2 = (int)b2.size();
    d.ahead = 0;
    auto al = [](size_t v) { return (v + 15) & ~(size_t)15; };
    const size_t o_in = 0, o_out = al(o_in + (size_t)n1 * 4), o_sub = al(o_out + (size_t)n1 * 4), o_exp = al(o_sub + (size_t)m * 4);
    const size_t o_cos = al(o_exp + ex.size() * 8), o_sched = al(o_cos + cosv.size() * 4), o_b2 = al(o_sched + sched.size() * 4);
    const size_t total = al(o_b2 + b2.size() * 2 + 16);
    std::vector<uint8_t> blob(total, 0);
    memcpy(blob.data() + o_in, in_map.data(), (size_t)n1 * 4);
    memcpy(blob.data() + o_out, out_map.data(), (size_t)n1 * 4);
    memcpy(blob.data() + o_sub, sub_map.data(), (size_t)m * 4);
    memcpy(blob.data() + o_exp, ex.data(), ex.size() * 8);
    memcpy(blob.data() + o_cos, cosv.data(), cosv.size() * 4);
    memcpy(blob.data() + o_sched, sched.data(), sched.size() * 4);
    memcpy(blob.data() + o_b2, b2.data(), b2.size() * 2);
    if (hipMalloc(&c->dev, total) != hipSuccess || hipMemcpy(c->dev, blob.data(), total, hipMemcpyHostToDevice) != hipSuccess) {
        ffhip_set_error("ffhip_tx_init: table upload failed");
        return FFHIP_ENOMEM;
    }
    c->blob_bytes = total;
    const uint8_t *base = (const uint8_t *)c->dev;
    TxPfa &P = c->pfa;
    P.n1 = n1; P.m = m; P.G = G; P.F = F;
    P.magic_q = (int)(((1u << 24) + (uint32_t)(n1 / 2) - 1) / (uint32_t)(n1 / 2));
    P.fft = is_fft;
    P.in_map = (const int *)(base + o_in);
    P.out_map = (const int *)(base + o_out);
    P.sub_map = (const int *)(base + o_sub);
    d.map = P.in_map;
    d.exp = (const float2 *)(base + o_exp);
    d.cos_tab = (const float *)(base + o_cos);
    d.sched = (const uint32_t *)(base + o_sched);
    d.blocks2 = (const uint16_t *)(base + o_b2);
    return 0;
}

static void tx_single_wide(FFHipTXContext *s, void *out, void *in, ptrdiff_t stride);
static void tx_single_dcst1(FFHipTXContext *s, void *out, void *in, ptrdiff_t stride);

extern "C" int ffhip_tx_init(FFHipTXContext **pctx, ffhip_tx_fn *fn, int type, int inv, int len, const void *scale_,
                             uint64_t flags)
{
    static const float one = 1.0f;
    const bool any_fft = type == FFHIP_TX_FLOAT_FFT || type == FFHIP_TX_DOUBLE_FFT || type == FFHIP_TX_INT32_FFT;
    if (!pctx || (!scale_ && !any_fft))
        return FFHIP_EINVAL;
    *pctx = nullptr;
    if (type == FFHIP_TX_DOUBLE_FFT || type == FFHIP_TX_DOUBLE_MDCT || type == FFHIP_TX_INT32_FFT || type == FFHIP_TX_INT32_MDCT) {
        /* the scale is a double for the double types, a float for the int32 ones (SCALE_TYPE, tx_double.c / tx_int32.c) */
        const bool is_int = type == FFHIP_TX_INT32_FFT || type == FFHIP_TX_INT32_MDCT, is_mdct = !any_fft;
        const double sc = !scale_ ? 1.0 : is_int ? (double)*static_cast<const float *>(scale_) : *static_cast<const double *>(scale_);
        if (flags & (FFHIP_TX_FULL_IMDCT | FFHIP_TX_REAL_TO_REAL | FFHIP_TX_REAL_TO_IMAGINARY)) {
            ffhip_set_error("ffhip_tx_init: AV_TX_FULL_IMDCT and the half-complex RDFTs are float-only on the hip path");
            return FFHIP_ENOSYS;
        }
        FFHipTXContext *c = new (std::nothrow) FFHipTXContext();
        if (!c)
            return FFHIP_ENOMEM;
        const int r = ffhip_txw_create(&c->wide, is_int, is_mdct, inv, len, sc);
        if (r < 0) {
            delete c;
            return r;
        }
        c->device = ffhip_txw_device(c->wide);
        c->type = type; c->inv = !!inv; c->len = len; c->scale = (float)sc;
        *pctx = c;
        if (fn)
            *fn = tx_single_wide;
        return 0;
    }
    const float *scale = static_cast<const float *>(scale_);
    if (!scale)
        scale = &one; /* an FFT takes no scale (av_tx_init accepts NULL there) */
    if (type == FFHIP_TX_FLOAT_DCT_I || type == FFHIP_TX_FLOAT_DST_I) {
        if (inv) {
            /* ff_tx_dcstI_init (tx_template.c:2017-2021) doubles the length of an inverse context and halves its scale; its transform
             * then reads twice the samples the caller asked for — nothing a caller can use, left to the C code */
            ffhip_set_error("ffhip_tx_init: DCT-I / DST-I are forward transforms on the hip path (their own inverse up to the scale)");
            return FFHIP_ENOSYS;
        }
        if (flags & (FFHIP_TX_FULL_IMDCT | FFHIP_TX_REAL_TO_REAL | FFHIP_TX_REAL_TO_IMAGINARY)) {
            ffhip_set_error("ffhip_tx_init: AV_TX_FULL_IMDCT / AV_TX_REAL_TO_* do not apply to DCT-I / DST-I");
            return FFHIP_EINVAL;
        }
        FFHipTXContext *c = new (std::nothrow) FFHipTXContext();
        if (!c)
            return FFHIP_ENOMEM;
        const int r = ffhip_dcst1_create(&c->dcst1, type == FFHIP_TX_FLOAT_DST_I, len, *scale);
        if (r < 0) {
            delete c;
            return r;
        }
        c->device = ffhip_dcst1_device(c->dcst1);
        c->type = type; c->inv = 0; c->len = len; c->scale = *scale;
        *pctx = c;
        if (fn)
            *fn = tx_single_dcst1;
        return 0;
    }
    if (type != FFHIP_TX_FLOAT_MDCT && type != FFHIP_TX_FLOAT_FFT && type != FFHIP_TX_FLOAT_RDFT && type != FFHIP_TX_FLOAT_DCT) {
        ffhip_set_error("ffhip_tx_init: type %d is not on the hip path (float FFT / MDCT / RDFT / DCT-I/II/III / DST-I; double and int32 FFT / MDCT)", type);
        return FFHIP_ENOSYS;
    }
    const bool dct = type == FFHIP_TX_FLOAT_DCT;
    const bool rdft = type == FFHIP_TX_FLOAT_RDFT || dct; /* the DCT-II / -III run on the RDFT of the same length */
    if (dct && inv)
        len *= 2; /* ff_tx_dct_init (tx_template.c:1844-1848): the inverse is initialised with half its length ... */
    const float rscale = dct && inv ? *scale * 0.5f : *scale; /* ... and its RDFT with half the scale */
    int half = 0;
    if (type == FFHIP_TX_FLOAT_RDFT && (flags & (FFHIP_TX_REAL_TO_REAL | FFHIP_TX_REAL_TO_IMAGINARY))) {
        /* ff_tx_rdft_r2r / _r2i (tx_template.c:1718-1830) are FF_TX_FORWARD_ONLY */
        if (inv) {
            ffhip_set_error("ffhip_tx_init: AV_TX_REAL_TO_REAL / _IMAGINARY are forward-only");
            return FFHIP_EINVAL;
        }
        half = flags & FFHIP_TX_REAL_TO_REAL ? 1 : 2;
    }
    if (rdft && (len < 8 || len > 4096 || (len & (len - 1)))) {
        ffhip_set_error("ffhip_tx_init: %s len %d not a power of two in 8..4096", dct ? "DCT" : "RDFT", len);
        return FFHIP_EINVAL;
    }
    const bool fft = type == FFHIP_TX_FLOAT_FFT || rdft; /* the RDFT runs a len/2-point FFT */
    /* 2 * 15 * 2^k, k = 2..6: the lengths av_tx serves with ff_tx_mdct_pfa_15xM (CELT 120..960, AAC-960 240 / 1920) */
    if (rdft)
        len >>= 1;
    /* ... and 2 * F * 2^k for F = 9, 7, 5, 3 (ff_tx_mdct_pfa_<F>xM: 96 / 768-sample AAC frames, Siren's 320, ...), sub-transform
     * sizes 4..256 (4..64 for 15): the factorisation of len / 2 into an odd F and a power of two is unique, so this is the codelet
     * av_tx_init picks ("larger factors are generally better", libavutil/tx.c) */
    int pfa_f = 0;
    if (!fft) {
        static const int factors[5] = { 15, 9, 7, 5, 3 };
        for (int i = 0; i < 5 && !pfa_f; i++) {
            const int f2 = 2 * factors[i], m = len / f2;
            if (len % f2 == 0 && m >= 4 && m <= (factors[i] == 15 ? 64 : 256) && !(m & (m - 1)))
                pfa_f = factors[i];
        }
    } else if (type == FFHIP_TX_FLOAT_FFT) {
        /* F * 2^k complex points, F = 15, 9, 7, 5, 3: av_tx_init's ff_tx_fft_pfa over fft<F>_ns and the 2^k-point split-radix
         * codelet (libavutil/tx_template.c:948-1101; the odd part of len is the first factor ff_tx_decompose_length offers, so the
         * tree is the same one).  120 / 240 / 480 / 960 / 1920: the CELT / AAC-960 frame sizes as plain FFTs */
        static const int factors[5] = { 15, 9, 7, 5, 3 };
        for (int i = 0; i < 5 && !pfa_f; i++) {
            const int m = len / factors[i];
            if (len % factors[i] == 0 && m >= 4 && m <= (factors[i] == 15 ? 128 : 256) && !(m & (m - 1)))
                pfa_f = factors[i];
        }
    }
    const bool pfa = pfa_f != 0;
    /* powers of two: one wave per transform up to 2048 complex points, the whole workgroup on one transform above that (FFT
     * 4096..16384, MDCT 8192..32768); the RDFT / DCT kernels are wave-sized */
    const int fft_max = type == FFHIP_TX_FLOAT_FFT ? 16384 : 2048;
    if (!pfa && (fft ? (len < 4 || len > fft_max || (len & (len - 1))) : (len < 16 || len > 32768 || (len & (len - 1))))) {
        ffhip_set_error("ffhip_tx_init: len %d is neither a power of two in %s nor %s{3, 5, 7, 9} * 2^k (k = 2..8) nor %s15 * 2^k (k = 2..%d)", len,
                        fft ? (fft_max > 2048 ? "4..16384" : "4..2048") : "16..32768", fft ? "" : "2 * ", fft ? "" : "2 * ", fft ? 7 : 6);
        return FFHIP_EINVAL;
    }
    if (!ffhip_have_device())
        return FFHIP_ENOSYS;
    FFHipTXContext *c = new (std::nothrow) FFHipTXContext();
    if (!c)
        return FFHIP_ENOMEM;
    c->device = ffhip_current_device();
    c->type = type; c->inv = !!inv; c->len = rdft ? 2 * len : len; c->scale = *scale;
    c->full = type == FFHIP_TX_FLOAT_MDCT && inv && (flags & FFHIP_TX_FULL_IMDCT);
    c->half = half;
    if (pfa) {
        const int r = tx_init_pfa(c, *scale, pfa_f, type == FFHIP_TX_FLOAT_FFT);
        if (r < 0) {
            ffhip_tx_uninit(&c);
            return r;
        }
        *pctx = c;
        if (fn)
            *fn = tx_single;
        return 0;
    }
    const int n = fft ? len : len >> 1; /* complex size of the split-radix network */
    int lg = 0;
    while ((1 << lg) < n)
        lg++;
    /* permutation: forward asks for SCATTER, inverse for GATHER (tx_template.c:1231-1233) */
    /* (the reference's inverse GATHERs z[i] = f(in[map[i]]); we walk the input in order and scatter through the
     * inverse permutation, so both directions carry a scatter map here) */
    std::vector<int> map(n);
    for (int i = 0; i < n; i++) {
        const int p = -sr_perm(i, n, c->inv) & (n - 1);
        map[p] = i;
    }
    /* exp table (ff_tx_mdct_gen_exp) */
    std::vector<float2> ex(dct ? 4 + n / 2 + 3 * n / 2 : rdft ? 4 + n / 2 + 1 : fft ? 2 : n); /* RDFT: one zero behind tsin[], which r2r / r2i read at i = len/4 */ /* an FFT has no twiddle table of its own: keep its LDS blob small */
    if (rdft) {
        /* ff_tx_rdft_init (tx_template.c:1601-1655): fact[8], tcos[len/4], tsin[len/4], doubles stored as floats */
        const int rl = 2 * n, len4 = rl / 4;
        const double f = 2 * M_PI / rl, m = c->inv ? 2 * (double)rscale : (double)rscale;
        float *tab = reinterpret_cast<float *>(ex.data());
        tab[0] = (float)((c->inv ? 0.5 : 1.0) * m);
        tab[1] = (float)(c->inv ? 0.5 * m : 1.0 * m);
        tab[2] = (float)m;
        tab[3] = (float)-m;
        tab[4] = (float)((0.5 - 0.0) * m);
        tab[5] = half == 1 ? 1 / rscale : (float)((0.0 - 0.5) * m); /* r2r: 1 / s->scale_f */
        tab[6] = (float)((0.5 - c->inv) * m);
        tab[7] = (float)(-(0.5 - c->inv) * m);
        for (int i = 0; i < len4; i++) {
            tab[8 + i] = (float)cos(i * f);
            tab[8 + len4 + i] = (float)(cos(((rl - i * 4) / 4.0) * f) * (c->inv ? 1 : -1));
        }
        if (dct) {
            /* ff_tx_dct_init (tx_template.c:1856-1870): exp[rl] rotations, exp[rl + rl/2] the fold / unfold weights */
            float *de = tab + 8 + 2 * len4;
            const double freq = M_PI / (rl * 2);
            for (int i = 0; i < rl; i++)
                de[i] = (float)(cos(i * freq) * (!c->inv + 1));
            for (int i = 0; i < rl / 2; i++)
                de[rl + i] = c->inv ? (float)(0.5 / sin((2 * i + 1) * freq)) : (float)cos((rl - 2 * i - 1) * freq);
        }
    } else if (!fft) {
        const double sc = *scale;
        const double theta = (sc < 0 ? n : 0) + 1.0 / 8.0, rt = sqrt(fabs(sc));
        float2 *e = ex.data();
        for (int i = 0; i < n; i++) {
            const double alpha = M_PI_2 * (i + theta) / n;
            e[i].x = (float)(cos(alpha) * rt);
            e[i].y = (float)(sin(alpha) * rt);
        }
    }
    /* the map scatters into the padded work array */
    for (int i = 0; i < n; i++)
        map[i] = TX_PAD(map[i]);
    /* cosine tables per level (cos(2*pi*k/m), k <= m/4; the last entry is an exact 0) */
    std::vector<float> cosv;
    TxDev &d = c->d;
    memset(&d, 0, sizeof(d));
    d.n = n; d.lg = lg;
    d.half = c->half;
    for (int l = 2; l <= lg; l++) {
        const int m = 1 << l;
        const double freq = 2 * M_PI / m;
        d.cos_off[l] = (int)cosv.size();
        for (int i = 0; i < m / 4; i++)
            cosv.push_back((float)cos(i * freq));
        cosv.push_back(0.0f);
    }
    std::vector<uint32_t> lev[16];
    std::vector<uint16_t> b2;
    sr_schedule(0, n, lg, lev, &b2);
    std::vector<uint32_t> sched;
    for (int l = 2; l <= lg; l++) {
        d.sched_off[l] = (int)sched.size();
        d.sched_cnt[l] = (int)lev[l].size();
        sched.insert(sched.end(), lev[l].begin(), lev[l].end());
    }
    d.nblocks2 = (int)b2.size();
    for (int l = 2; l <= lg; l++)
        d.max_cnt = d.sched_cnt[l] > d.max_cnt ? d.sched_cnt[l] : d.max_cnt;
    {
        const char *ea = FFHIP_KNOB("FFHIP_TX_AHEAD");
        d.ahead = ea && ea[0] == '1'; /* measured slightly slower (350 vs 345 M transforms/s): opt-in */
    }
    /* one device allocation for all tables */
    size_t off_map = 0, off_exp, off_cos, off_sched, off_b2, total;
    off_exp = (off_map + map.size() * 4 + 15) & ~(size_t)15;
    off_cos = (off_exp + ex.size() * 8 + 15) & ~(size_t)15;
    off_sched = (off_cos + cosv.size() * 4 + 15) & ~(size_t)15;
    off_b2 = (off_sched + sched.size() * 4 + 15) & ~(size_t)15;
    total = (off_b2 + b2.size() * 2 + 16 + 15) & ~(size_t)15;
    std::vector<uint8_t> blob(total, 0);
    memcpy(blob.data() + off_map, map.data(), map.size() * 4);
    memcpy(blob.data() + off_exp, ex.data(), ex.size() * 8);
    memcpy(blob.data() + off_cos, cosv.data(), cosv.size() * 4);
    memcpy(blob.data() + off_sched, sched.data(), sched.size() * 4);
    memcpy(blob.data() + off_b2, b2.data(), b2.size() * 2);
    if (hipMalloc(&c->dev, total) != hipSuccess || hipMemcpy(c->dev, blob.data(), total, hipMemcpyHostToDevice) != hipSuccess) {
        ffhip_set_error("ffhip_tx_init: table upload failed");
        ffhip_tx_uninit(&c);
        return FFHIP_ENOMEM;
    }
    c->blob_bytes = total;
    uint8_t *base = (uint8_t *)c->dev;
    d.map = (const int *)(base + off_map);
    d.exp = (const float2 *)(base + off_exp);
    d.cos_tab = (const float *)(base + off_cos);
    d.sched = (const uint32_t *)(base + off_sched);
    d.blocks2 = (const uint16_t *)(base + off_b2);
    if (!(flags & FFHIP_TX_BITEXACT) && (type == FFHIP_TX_FLOAT_FFT ? ffhip_tx_radix_fft_ok(n) : ffhip_tx_radix_ok(n))) {
        std::vector<float2> w(n);
        for (int k = 0; k < n; k++) {
            const double a = 2 * M_PI * k / n;
            w[k] = make_float2((float)cos(a), (float)-sin(a));
        }
        if (hipMalloc(&c->wtab, n * sizeof(float2)) != hipSuccess ||
            hipMemcpy(c->wtab, w.data(), n * sizeof(float2), hipMemcpyHostToDevice) != hipSuccess) {
            ffhip_set_error("ffhip_tx_init: table upload failed");
            ffhip_tx_uninit(&c);
            return FFHIP_ENOMEM;
        }
    }
    *pctx = c;
    if (fn)
        *fn = tx_single;
    return 0;
}

static int tx_batch_half(FFHipTXContext *c, void *out, size_t out_pitch, const void *in, size_t in_pitch, ptrdiff_t stride, int nt,
                         void *stream);

extern "C" int ffhip_tx_batch_dev(FFHipTXContext *c, void *out, size_t out_pitch, const void *in, size_t in_pitch,
                                  ptrdiff_t stride, int nt, void *stream)
{
    if (!c || !out || !in || nt < 0 || (stride % (ptrdiff_t)sizeof(float)))
        return FFHIP_EINVAL;
    if (nt == 0)
        return 0;
    FFHipDeviceGuard dg(c->device);
    if (c->wide) {
        if (stride != (ptrdiff_t)ffhip_txw_elem_size(c->wide)) {
            ffhip_set_error("ffhip_tx: double / int32 batches are contiguous rows (stride == sizeof(sample))");
            return FFHIP_EINVAL;
        }
        return ffhip_txw_batch(c->wide, out, out_pitch, in, in_pitch, nt, (hipStream_t)stream);
    }
    if (c->dcst1) /* the input side is the strided one (ff_tx_dctI / ff_tx_dstI read src[i * stride]) */
        return ffhip_dcst1_batch(c->dcst1, (float *)out, out_pitch, (const float *)in, in_pitch, stride / (ptrdiff_t)sizeof(float), nt,
                                 (hipStream_t)stream);
    if (!c->full)
        return tx_batch_half(c, out, out_pitch, in, in_pitch, stride, nt, stream);
    /* full inverse: rows of 2 * len floats; the half transform goes to the middle, then the mirror pass */
    if (((uintptr_t)out | out_pitch) & 7) {
        ffhip_set_error("ffhip_tx: AV_TX_FULL_IMDCT batches need 8-byte aligned output rows");
        return FFHIP_EINVAL;
    }
    const int r = tx_batch_half(c, (float *)out + c->len / 2, out_pitch, in, in_pitch, stride, nt, stream);
    if (r < 0)
        return r;
    for (int t0 = 0; t0 < nt; t0 += 65535) {
        const int cnt = nt - t0 < 65535 ? nt - t0 : 65535;
        hipLaunchKernelGGL(k_imdct_mirror, dim3(cdiv(c->len / 4, 256), cnt), dim3(256), 0, (hipStream_t)stream,
                           (float *)((uint8_t *)out + (size_t)t0 * out_pitch), out_pitch, c->len, cnt);
    }
    LAUNCH_CHECK();
    return 0;
}

static int tx_batch_half(FFHipTXContext *c, void *out, size_t out_pitch, const void *in, size_t in_pitch, ptrdiff_t stride, int nt,
                         void *stream)
{
    const int n = c->d.n;
    if (!c->pfa.n1 && n > 2048) {
        /* 4096..16384 complex points: the work array (33..132 KiB padded) is the workgroup's, the whole workgroup runs each level
         * (k_fft_z / k_mdct_z with WG); tables stay in L2 */
        if ((c->type != FFHIP_TX_FLOAT_FFT && stride != (ptrdiff_t)sizeof(float)) || (((uintptr_t)in | in_pitch | (uintptr_t)out | out_pitch) & 7)) {
            ffhip_set_error("ffhip_tx: transforms above 2048 complex points need contiguous, 8-byte aligned rows");
            return FFHIP_EINVAL;
        }
        const size_t lds_z = tx_z_bytes(n);
        const int threads = n >= 8192 ? 1024 : 512;
        int cus = 256, dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
            cus = prop.multiProcessorCount;
        int per_cu = (int)((160 * 1024) / (((lds_z + 1279) / 1280) * 1280));
        if (per_cu * (threads / 64) > 32) per_cu = 32 / (threads / 64);
        if (per_cu < 1) per_cu = 1;
        const int blocks = nt < cus * per_cu ? nt : cus * per_cu;
        static FFHipPerDeviceOnce wg_attr;
        if (wg_attr.enter()) {
            (void)hipFuncSetAttribute((const void *)k_fft_z<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void *)k_mdct_z<0, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void *)k_mdct_z<1, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            wg_attr.leave(true);
        }
#define TX_LAUNCH_WG(K)                                                                                                               \
    hipLaunchKernelGGL((K), dim3(blocks), dim3(threads), lds_z, (hipStream_t)stream, c->d, (const uint8_t *)c->dev, 0,               \
                       (const float *)in, in_pitch, (float *)out, out_pitch, nt, blocks)
        {
            const char *er = FFHIP_KNOB("FFHIP_TX_RADIX");
            if (c->type == FFHIP_TX_FLOAT_FFT && c->wtab && !(er && er[0] == '0'))
                return ffhip_launch_fft_r(n, c->inv, c->wtab, (const float *)in, in_pitch, (float *)out, out_pitch, nt, (hipStream_t)stream);
        }
        if (c->type == FFHIP_TX_FLOAT_FFT)
            TX_LAUNCH_WG((k_fft_z<false, true>));
        else if (c->inv)
            TX_LAUNCH_WG((k_mdct_z<1, false, true>));
        else
            TX_LAUNCH_WG((k_mdct_z<0, false, true>));
#undef TX_LAUNCH_WG
        LAUNCH_CHECK();
        return 0;
    }
    if (!c->pfa.n1 && (c->type == FFHIP_TX_FLOAT_FFT || c->type == FFHIP_TX_FLOAT_RDFT || c->type == FFHIP_TX_FLOAT_DCT)) {
        /* complex in, complex out, contiguous (av_tx's FFT ignores `stride`); 8-byte aligned rows.  RDFT: len reals on one
         * side, len/2 + 1 complex bins on the other */
        if (((uintptr_t)in | in_pitch | (uintptr_t)out | out_pitch) & 7) {
            ffhip_set_error("ffhip_tx: FFT / RDFT batches need 8-byte aligned rows");
            return FFHIP_EINVAL;
        }
        /* tables in LDS while they are small next to the waves' work arrays; the big transforms (n = 2048: 50 KB of tables,
         * 17 KB per wave) keep them in L2 and spend the LDS on waves (FFHIP_TX_TABLDS=0/1 forces either) */
        const char *etl = FFHIP_KNOB("FFHIP_TX_TABLDS");
        const bool tl = etl ? etl[0] == '1' : c->blob_bytes <= 32 * 1024;
        const size_t blob_lds = tl ? (c->blob_bytes + 15) & ~(size_t)15 : 0;
        const int blob_arg = tl ? (int)c->blob_bytes : 0;
        /* the forward DCT keeps its running-sum terms behind each wave's work array */
        const size_t zw = tx_z_bytes(n) + (c->type == FFHIP_TX_FLOAT_DCT && !c->inv ? (((size_t)n + 1) * 4 + 15) & ~(size_t)15 : 0);
        /* the forward DCT's workgroups meet at two barriers per transform (the running sums): two or three smaller ones per CU
         * overlap one's chain with another's transforms */
        const char *ewpb = FFHIP_KNOB("FFHIP_DCT_WPB");
        int wpb = c->type == FFHIP_TX_FLOAT_DCT && !c->inv ? (ewpb ? atoi(ewpb) : 8) : 16;
        if (wpb < 1 || wpb > 16 || (wpb & (wpb - 1)))
            wpb = 8;
        size_t lds_z = blob_lds + zw * wpb;
        while (wpb > 1 && lds_z > 150 * 1024) {
            wpb >>= 1;
            lds_z = blob_lds + zw * wpb;
        }
        int cus = 256, dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
            cus = prop.multiProcessorCount;
        int per_cu = (int)((160 * 1024) / (((lds_z + 1279) / 1280) * 1280));
        if (per_cu * wpb > 32) per_cu = 32 / wpb;
        if (per_cu < 1) per_cu = 1;
        int blocks = cus * per_cu;
        if (blocks > (nt + wpb - 1) / wpb)
            blocks = (nt + wpb - 1) / wpb;
        int rlg = 0;
        {
            const char *er = FFHIP_KNOB("FFHIP_TX_RADIX");
            if (c->wtab && !(er && er[0] == '0') && ffhip_tx_radix_ok(n))
                rlg = c->d.lg;
        }
        static FFHipPerDeviceOnce fft_attr; /* function attributes are per device */
        if (fft_attr.enter()) {
            (void)hipFuncSetAttribute((const void *)k_fft_z<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void *)k_fft_z<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void *)k_rdft<0, true, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void *)k_rdft<0, true, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void *)k_rdft<0, true, 9>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void *)k_rdft<0, true, 10>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void *)k_rdft<1, true, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void *)k_rdft<1, true, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void *)k_rdft<1, true, 9>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void *)k_rdft<1, true, 10>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void *)k_rdft<0, false, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void *)k_rdft<0, false, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void *)k_rdft<0, false, 9>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void *)k_rdft<0, false, 10>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void *)k_rdft<1, false, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void *)k_rdft<1, false, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void *)k_rdft<1, false, 9>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void *)k_rdft<1, false, 10>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void *)k_dct<0, true, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void *)k_dct<0, true, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void *)k_dct<0, true, 9>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void *)k_dct<0, true, 10>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void *)k_dct<1, true, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void *)k_dct<1, true, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void *)k_dct<1, true, 9>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void *)k_dct<1, true, 10>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void *)k_dct<0, false, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void *)k_dct<0, false, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void *)k_dct<0, false, 9>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void *)k_dct<0, false, 10>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void *)k_dct<1, false, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void *)k_dct<1, false, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void *)k_dct<1, false, 9>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void *)k_dct<1, false, 10>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            fft_attr.leave(true);
        }
        if (c->type == FFHIP_TX_FLOAT_RDFT) {
#define TX_LAUNCH(K)                                                                                                                  \
    hipLaunchKernelGGL((K), dim3(blocks), dim3(64 * wpb), lds_z, (hipStream_t)stream, c->d, (const uint8_t *)c->dev, blob_arg,       \
                       (const float *)in, in_pitch, (float *)out, out_pitch, nt, blocks * wpb)
            /* (the split-radix kernels take the pointer too and ignore it) */
#define TX_LAUNCH_R(K)                                                                                                                \
    hipLaunchKernelGGL((K), dim3(blocks), dim3(64 * wpb), lds_z, (hipStream_t)stream, c->d, (const uint8_t *)c->dev, blob_arg,       \
                       (const float *)in, in_pitch, (float *)out, out_pitch, nt, blocks * wpb, (const float2 *)c->wtab)
#define TX_LAUNCH_RLG(KN, INV_)                                                                                                       \
    do {                                                                                                                              \
        switch (rlg) {                                                                                                                \
        case 8:  if (tl) TX_LAUNCH_R((KN<INV_, true, 8>)); else TX_LAUNCH_R((KN<INV_, false, 8>)); break;                            \
        case 9:  if (tl) TX_LAUNCH_R((KN<INV_, true, 9>)); else TX_LAUNCH_R((KN<INV_, false, 9>)); break;                            \
        case 10: if (tl) TX_LAUNCH_R((KN<INV_, true, 10>)); else TX_LAUNCH_R((KN<INV_, false, 10>)); break;                          \
        default: if (tl) TX_LAUNCH_R((KN<INV_, true, 0>)); else TX_LAUNCH_R((KN<INV_, false, 0>)); break;                            \
        }                                                                                                                             \
    } while (0)
            if (c->inv) TX_LAUNCH_RLG(k_rdft, 1); else TX_LAUNCH_RLG(k_rdft, 0);
            LAUNCH_CHECK();
            return 0;
        }
        if (c->type == FFHIP_TX_FLOAT_DCT) {
            if (c->inv) TX_LAUNCH_RLG(k_dct, 1); else TX_LAUNCH_RLG(k_dct, 0);
            LAUNCH_CHECK();
            return 0;
        }
        {
            const char *er = FFHIP_KNOB("FFHIP_TX_RADIX");
            if (c->wtab && !(er && er[0] == '0'))
                return ffhip_launch_fft_r(n, c->inv, c->wtab, (const float *)in, in_pitch, (float *)out, out_pitch, nt, (hipStream_t)stream);
        }
        if (tl) TX_LAUNCH((k_fft_z<true>)); else TX_LAUNCH((k_fft_z<false>));
        LAUNCH_CHECK();
        return 0;
    }
    if (c->pfa.n1) {
        const TxPfa &P = c->pfa;
        if ((!P.fft && stride != (ptrdiff_t)sizeof(float)) || (((uintptr_t)in | in_pitch | (uintptr_t)out | out_pitch) & 7)) {
            ffhip_set_error("ffhip_tx: the prime-factor lengths need contiguous, 8-byte aligned rows");
            return FFHIP_EINVAL;
        }
        const size_t area = tx_z_bytes(n); /* G * n1 points, padded: the parked inputs (unpadded) fit the same bytes */
        const size_t blob_al = (c->blob_bytes + 15) & ~(size_t)15;
        int wpb = 16;
        while (wpb > 1 && blob_al + area * wpb > 150 * 1024)
            wpb >>= 1;
        const size_t lds_p = blob_al + area * wpb;
        int cus = 256, dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
            cus = prop.multiProcessorCount;
        int per_cu = (int)((160 * 1024) / (((lds_p + 1279) / 1280) * 1280));
        if (per_cu * wpb > 32) per_cu = 32 / wpb;
        if (per_cu < 1) per_cu = 1;
        const int groups = (nt + P.G - 1) / P.G;
        int blocks = cus * per_cu;
        if (blocks > (groups + wpb - 1) / wpb)
            blocks = (groups + wpb - 1) / wpb;
        TxTab53 T;
        T.t[0] = T.t[1] = (float)cos(2 * M_PI / 5);
        T.t[2] = T.t[3] = (float)cos(2 * M_PI / 10);
        T.t[4] = T.t[5] = (float)sin(2 * M_PI / 5);
        T.t[6] = T.t[7] = (float)sin(2 * M_PI / 10);
        T.t[8] = T.t[9] = (float)cos(2 * M_PI / 12);
        T.t[10] = (float)cos(2 * M_PI / 6);
        T.t[11] = (float)cos(8 * M_PI / 6);
        T.t7[0] = (float)cos(2 * M_PI / 7);  T.t7[1] = (float)sin(2 * M_PI / 7);
        T.t7[2] = (float)sin(2 * M_PI / 28); T.t7[3] = (float)cos(2 * M_PI / 28);
        T.t7[4] = (float)cos(2 * M_PI / 14); T.t7[5] = (float)sin(2 * M_PI / 14);
        T.t9[0] = (float)cos(2 * M_PI / 3);  T.t9[1] = (float)sin(2 * M_PI / 3);
        T.t9[2] = (float)cos(2 * M_PI / 9);  T.t9[3] = (float)sin(2 * M_PI / 9);
        T.t9[4] = (float)cos(2 * M_PI / 36); T.t9[5] = (float)sin(2 * M_PI / 36);
        T.t9[6] = T.t9[2] + T.t9[5];
        T.t9[7] = T.t9[3] - T.t9[4];
        const int big = P.m > 64;
#define PFA_GO(INV_, F_, C_)                                                                                                               \
    do {                                                                                                                                   \
        static FFHipPerDeviceOnce attr;                                                                                                    \
        if (attr.enter()) {                                                                                                                \
            (void)hipFuncSetAttribute((const void *)k_mdct_pfa<INV_, F_, C_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);     \
            attr.leave(true);                                                                                                              \
        }                                                                                                                                  \
        hipLaunchKernelGGL((k_mdct_pfa<INV_, F_, C_>), dim3(blocks), dim3(64 * wpb), lds_p, (hipStream_t)stream, c->d, P, T,              \
                           (const uint8_t *)c->dev, (int)c->blob_bytes, (const float *)in, in_pitch, (float *)out, out_pitch, nt,         \
                           blocks * wpb);                                                                                                  \
    } while (0)
#define PFA_F(F_)                                                                                                                          \
    do {                                                                                                                                   \
        if (P.fft)       { if (big) PFA_GO(2, F_, 4); else PFA_GO(2, F_, 1); }                                                             \
        else if (c->inv) { if (big) PFA_GO(1, F_, 4); else PFA_GO(1, F_, 1); }                                                             \
        else             { if (big) PFA_GO(0, F_, 4); else PFA_GO(0, F_, 1); }                                                             \
    } while (0)
        switch (P.F) {
        case 3:  PFA_F(3); break;
        case 5:  PFA_F(5); break;
        case 7:  PFA_F(7); break;
        case 9:  PFA_F(9); break;
        default:
            if (P.fft) { if (big) PFA_GO(2, 15, 2); else PFA_GO(2, 15, 1); }
            else if (c->inv) PFA_GO(1, 15, 1);
            else PFA_GO(0, 15, 1);
            break;
        }
#undef PFA_F
#undef PFA_GO
        LAUNCH_CHECK();
        return 0;
    }
    const size_t per_wave = tx_z_bytes(n) + (size_t)n * 16;
    int wpb = (int)((60 * 1024) / per_wave);
    if (wpb > 4) wpb = 4;
    if (wpb < 1) {
        ffhip_set_error("ffhip_tx: len %d does not fit LDS", c->len);
        return FFHIP_EINVAL;
    }
    const ptrdiff_t es = stride / (ptrdiff_t)sizeof(float);
    const dim3 grid(cdiv(nt, wpb)), block(64 * wpb);
    /* FFT-level tables in LDS when they fit next to the waves' areas (FFHIP_TX_LDSTAB=0 keeps them in L2) */
    const char *et = FFHIP_KNOB("FFHIP_TX_LDSTAB");
    const size_t ftab_sz = (size_t)((const uint8_t *)c->d.blocks2 - (const uint8_t *)c->d.cos_tab); /* twiddles + butterfly lists */
    int ftab = 0;
    size_t lds = per_wave * wpb;
    /* measured (profiles/r01_sweep_tx.txt): 174 vs 149 M forward transforms/s with the level tables in LDS */
    if (!(et && et[0] == '0') && lds + ftab_sz <= 64 * 1024) {
        ftab = (int)ftab_sz;
        lds += ftab_sz;
    }
    /* contiguous 8-byte aligned batches: the staging-free kernel (FFHIP_TX_Z=0 selects the older ones) */
    {
        const char *ez = FFHIP_KNOB("FFHIP_TX_Z");
        const char *ew = FFHIP_KNOB("FFHIP_TX_WPB");
        int wpb = ew && atoi(ew) > 0 ? atoi(ew) : 16; /* waves per workgroup: 16 measured best (the table copy is shared) */
        if (wpb > 16) wpb = 16;
        const char *etl = FFHIP_KNOB("FFHIP_TX_TABLDS");
        const bool tl = etl ? etl[0] == '1' : c->blob_bytes <= 32 * 1024; /* as for the FFT: big transforms keep their tables in L2 */
        const size_t blob_lds = tl ? (c->blob_bytes + 15) & ~(size_t)15 : 0;
        const int blob_arg = tl ? (int)c->blob_bytes : 0;
        size_t lds_z = blob_lds + tx_z_bytes(n) * wpb;
        while (wpb > 1 && lds_z > 150 * 1024) {
            wpb >>= 1;
            lds_z = blob_lds + tx_z_bytes(n) * wpb;
        }
        if (es == 1 && !(((uintptr_t)in | in_pitch | (uintptr_t)out | out_pitch) & 7) && c->wtab) {
            const char *er = FFHIP_KNOB("FFHIP_TX_RADIX");
            if (!(er && er[0] == '0'))
                return ffhip_launch_mdct_r(n, c->inv, c->wtab, c->d.exp, (const float *)in, in_pitch, (float *)out, out_pitch, nt,
                                           (hipStream_t)stream);
        }
        if (es == 1 && !(((uintptr_t)in | in_pitch | (uintptr_t)out | out_pitch) & 7) && lds_z <= 150 * 1024 && !(ez && ez[0] == '0')) {
            int cus = 256, dev = 0;
            hipDeviceProp_t prop;
            if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
                cus = prop.multiProcessorCount;
            int per_cu = (int)((160 * 1024) / (((lds_z + 1279) / 1280) * 1280));
            if (per_cu * wpb > 32) per_cu = 32 / wpb;
            if (per_cu < 1) per_cu = 1;
            int blocks = cus * per_cu;
            if (blocks > (nt + wpb - 1) / wpb)
                blocks = (nt + wpb - 1) / wpb;
            static FFHipPerDeviceOnce attr_done;
            if (attr_done.enter()) {
                (void)hipFuncSetAttribute((const void *)k_mdct_z<0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                (void)hipFuncSetAttribute((const void *)k_mdct_z<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                (void)hipFuncSetAttribute((const void *)k_mdct_z<0, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                (void)hipFuncSetAttribute((const void *)k_mdct_z<1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                attr_done.leave(true);
            }
            if (c->inv) {
                if (tl) TX_LAUNCH((k_mdct_z<1, true>)); else TX_LAUNCH((k_mdct_z<1, false>));
            } else {
                if (tl) TX_LAUNCH((k_mdct_z<0, true>)); else TX_LAUNCH((k_mdct_z<0, false>));
            }
            LAUNCH_CHECK();
            return 0;
        }
    }
    const char *ev = FFHIP_KNOB("FFHIP_TX_PERSISTENT");
    const bool aligned = es == 1 && !(((uintptr_t)in | in_pitch | (uintptr_t)out | out_pitch) & 15);
    const size_t lds_p = ((c->blob_bytes + 15) & ~(size_t)15) + per_wave * 4;
    /* measured (profiles/r01_sweep_tx.txt, N = 1024): 180 vs 175 M forward and 247 vs 225 M inverse transforms/s against
     * the one-shot kernel with LDS level tables - the default for aligned contiguous batches; FFHIP_TX_PERSISTENT=0
     * selects the one-shot kernel */
    if (aligned && lds_p <= 64 * 1024 && !(ev && ev[0] == '0')) {
        int cus = 256, dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
            cus = prop.multiProcessorCount;
        int blocks = cus * (int)((160 * 1024) / lds_p);
        if (blocks > (nt + 3) / 4)
            blocks = (nt + 3) / 4;
        if (c->inv)
            hipLaunchKernelGGL((k_mdct_l<1>), dim3(blocks), dim3(256), lds_p, (hipStream_t)stream, c->d, (const uint8_t *)c->dev,
                               (int)c->blob_bytes, (const float *)in, in_pitch, (float *)out, out_pitch, nt, blocks * 4);
        else
            hipLaunchKernelGGL((k_mdct_l<0>), dim3(blocks), dim3(256), lds_p, (hipStream_t)stream, c->d, (const uint8_t *)c->dev,
                               (int)c->blob_bytes, (const float *)in, in_pitch, (float *)out, out_pitch, nt, blocks * 4);
        LAUNCH_CHECK();
        return 0;
    }
    if (!c->inv) {
        const int vin = !(((uintptr_t)in | in_pitch) & 15);
        const int vout = es == 1 && !(((uintptr_t)out | out_pitch) & 15);
        hipLaunchKernelGGL((k_mdct<0>), grid, block, lds, (hipStream_t)stream, c->d, (const float *)in, in_pitch, (float *)out,
                           out_pitch, es, nt, wpb, vin, vout, ftab);
    } else {
        const int vin = es == 1 && !(((uintptr_t)in | in_pitch) & 15);
        const int vout = !(((uintptr_t)out | out_pitch) & 15);
        hipLaunchKernelGGL((k_mdct<1>), grid, block, lds, (hipStream_t)stream, c->d, (const float *)in, in_pitch, (float *)out,
                           out_pitch, es, nt, wpb, vin, vout, ftab);
    }
    LAUNCH_CHECK();
    return 0;
}

/* av_tx_fn-shaped single transform with HOST pointers (libavutil/tx.h:151): stage, run, copy back */
static void tx_single(FFHipTXContext *s, void *out, void *in, ptrdiff_t stride)
{
    FFHipDeviceGuard dg(s->device);
    std::lock_guard<std::mutex> lk(s->mu);
    const int len = s->len;
    const bool rdft = s->type == FFHIP_TX_FLOAT_RDFT, dct = s->type == FFHIP_TX_FLOAT_DCT;
    const bool fft = s->type == FFHIP_TX_FLOAT_FFT || rdft || dct; /* contiguous on both sides */
    /* RDFT: len reals <-> len/2 + 1 complex bins; DCT: len reals <-> len reals (the reference's scribbles into its input and
     * behind the forward output, tx.h:100-102, are not reproduced) */
    /* half-complex RDFT: len/2 + 1 real resp. len/2 imaginary parts */
    const size_t in_elems = dct ? (size_t)len : rdft ? (size_t)(s->inv ? len + 2 : len) : fft ? (size_t)2 * len : s->inv ? (size_t)len : (size_t)2 * len;
    const size_t out_elems = s->half ? (size_t)(len / 2 + (s->half == 1)) : dct ? (size_t)len : rdft ? (size_t)(s->inv ? len : len + 2) : fft || s->full ? (size_t)2 * len : (size_t)len;
    const ptrdiff_t es = stride / (ptrdiff_t)sizeof(float);
    /* the strided side is packed on the host so that the device sees contiguous data */
    std::vector<float> hin(in_elems), hout(out_elems);
    const float *fi = (const float *)in;
    for (size_t i = 0; i < in_elems; i++)
        hin[i] = (s->inv && !fft) ? fi[(ptrdiff_t)i * es] : fi[i];
    const size_t need = (in_elems + out_elems) * sizeof(float) + 64;
    if (need > s->stage_sz) {
        if (s->stage)
            (void)hipFree(s->stage);
        s->stage = nullptr;
        s->stage_sz = 0;
        if (hipMalloc(&s->stage, need) != hipSuccess) {
            ffhip_set_error("ffhip_tx: staging allocation failed");
            return;
        }
        s->stage_sz = need;
    }
    float *din = (float *)s->stage, *dout = din + ((in_elems + 3) & ~(size_t)3);
    if (hipMemcpy(din, hin.data(), in_elems * sizeof(float), hipMemcpyHostToDevice) != hipSuccess)
        return;
    if (ffhip_tx_batch_dev(s, dout, ((out_elems * sizeof(float)) + 15) & ~(size_t)15, din, ((in_elems * sizeof(float)) + 15) & ~(size_t)15, sizeof(float), 1, 0) < 0)
        return;
    if (hipMemcpy(hout.data(), dout, out_elems * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess)
        return;
    float *fo = (float *)out;
    for (size_t i = 0; i < out_elems; i++)
        fo[(s->inv || fft) ? (ptrdiff_t)i : (ptrdiff_t)i * es] = hout[i];
}

/* DCT-I / DST-I: len reals in (strided), len reals out */
static void tx_single_dcst1(FFHipTXContext *s, void *out, void *in, ptrdiff_t stride)
{
    FFHipDeviceGuard dg(s->device);
    std::lock_guard<std::mutex> lk(s->mu);
    const size_t n = (size_t)s->len, row = (n * sizeof(float) + 15) & ~(size_t)15;
    std::vector<float> h(n);
    for (size_t i = 0; i < n; i++)
        h[i] = *(const float *)((const uint8_t *)in + (ptrdiff_t)i * stride);
    if (2 * row > s->stage_sz) {
        if (s->stage)
            (void)hipFree(s->stage);
        s->stage = nullptr;
        s->stage_sz = 0;
        if (hipMalloc(&s->stage, 2 * row) != hipSuccess) {
            ffhip_set_error("ffhip_tx: staging allocation failed");
            return;
        }
        s->stage_sz = 2 * row;
    }
    float *din = (float *)s->stage, *dout = (float *)((uint8_t *)s->stage + row);
    if (hipMemcpy(din, h.data(), n * sizeof(float), hipMemcpyHostToDevice) != hipSuccess)
        return;
    if (ffhip_dcst1_batch(s->dcst1, dout, row, din, row, 1, 1, 0) < 0)
        return;
    (void)hipMemcpy(out, dout, n * sizeof(float), hipMemcpyDeviceToHost);
}

/* the same for the double / int32 contexts: the strided side (forward MDCT: output, inverse: input; an FFT has none) packed on the host */
static void tx_single_wide(FFHipTXContext *s, void *out, void *in, ptrdiff_t stride)
{
    FFHipDeviceGuard dg(s->device);
    std::lock_guard<std::mutex> lk(s->mu);
    const size_t es = ffhip_txw_elem_size(s->wide), ni = ffhip_txw_in_elems(s->wide), no = ffhip_txw_out_elems(s->wide);
    const bool mdct = s->type == FFHIP_TX_DOUBLE_MDCT || s->type == FFHIP_TX_INT32_MDCT;
    std::vector<uint8_t> hin(ni * es), hout(no * es);
    for (size_t i = 0; i < ni; i++)
        memcpy(hin.data() + i * es, (const uint8_t *)in + ((mdct && s->inv) ? (ptrdiff_t)i * stride : (ptrdiff_t)(i * es)), es);
    const size_t in_b = (ni * es + 15) & ~(size_t)15, out_b = (no * es + 15) & ~(size_t)15, need = in_b + out_b;
    if (need > s->stage_sz) {
        if (s->stage)
            (void)hipFree(s->stage);
        s->stage = nullptr;
        s->stage_sz = 0;
        if (hipMalloc(&s->stage, need) != hipSuccess) {
            ffhip_set_error("ffhip_tx: staging allocation failed");
            return;
        }
        s->stage_sz = need;
    }
    uint8_t *din = (uint8_t *)s->stage, *dout = din + in_b;
    if (hipMemcpy(din, hin.data(), ni * es, hipMemcpyHostToDevice) != hipSuccess)
        return;
    if (ffhip_txw_batch(s->wide, dout, out_b, din, in_b, 1, 0) < 0)
        return;
    if (hipMemcpy(hout.data(), dout, no * es, hipMemcpyDeviceToHost) != hipSuccess)
        return;
    for (size_t i = 0; i < no; i++)
        memcpy((uint8_t *)out + ((mdct && !s->inv) ? (ptrdiff_t)i * stride : (ptrdiff_t)(i * es)), hout.data() + i * es, es);
}
