/*
 * tx_wide.hip — av_tx's sample types beyond float on the hip path: AV_TX_DOUBLE_FFT / _MDCT and AV_TX_INT32_FFT / _MDCT at
 * power-of-two lengths (libavutil/tx.h:48-69, tx_double.c, tx_int32.c = libavutil/tx_template.c compiled with
 * TX_DOUBLE / TX_INT32).  Users on the reference's side: the fixed-point audio codecs (ac3dec_fixed / ac3enc_fixed, the fixed AAC
 * decoder and its SBR, dcaenc) for int32, the audio filters for double.
 *
 * What differs from the float kernels of tx_api.hip is the arithmetic, not the network:
 *   int32   CMUL is a 64-bit product sum rounded to nearest, (accu + 2^30) >> 31 (tx_priv.h:117-126); BF wraps modulo 2^32 (:143-147);
 *           the MDCT's fold is (a + b + 32) >> 6 (:141); tables are RESCALE(x) = clip(llrintf(x * 2^31)) — the product is rounded to
 *           FLOAT first (:139) — so cos(0) is 2^31 - 1, not one, and a butterfly the reference writes WITHOUT a multiplication must
 *           not be given one: the size-4 codelet, k = 0 of the size-8 and size-16 codelets (BUTTERFLIES alone, tx_template.c:634-700),
 *           while ff_tx_fft_sr_combine (size >= 32, :540-566) multiplies at k = 0 too.
 *   double  the float operations on doubles; tables are the doubles the reference computes (RESCALE is the identity).
 * Integer sums are order-independent modulo 2^32, so only the products' roundings pin the int32 results; the double kernels keep
 * the reference's operation order (-ffp-contract=off).  Both are bit-identical to ff_tx_*_double_c / ff_tx_*_int32_c.
 *
 * One flattened split-radix network per context, as tx_api.hip builds it (size-2 blocks, then one butterfly list per level).  A
 * transform is a wave's (n <= 1024 complex points, four transforms per workgroup) or the workgroup's (above); the work array lives
 * in LDS, padded one element per 32; tables stay in L2.  These are the tails of the AVTXType surface: correct first, then as fast
 * as a straightforward LDS network gets — the float path's register-resident radix core has no exact integer twin.
 */
#include <math.h>
#include <new>
#include <stdint.h>
#include <string.h>
#include <vector>

#include "common.h"
#include "tx_kernels.h"

#define TXW_PAD(i) ((i) + ((i) >> 5))

template <typename T> struct TxwCpx { T re, im; };

struct TxwDev {
    int n, lg;                 /* complex points of the network and their log2 */
    const int *map;            /* n: padded work-array slot of input-order element j */
    const void *exp;           /* MDCT: n TxwCpx<T>, natural order */
    const void *cos_tab;       /* T, concatenated per level */
    const uint32_t *sched;     /* a0 (padded) | k << 16, concatenated per level */
    const uint16_t *blocks2;   /* padded offsets of the size-2 blocks */
    int nblocks2;
    int cos_off[20], sched_off[20], sched_cnt[20];
};

struct FFHipTxWide {
    int device, type, inv, len, is_int, is_mdct;
    TxwDev d;
    void *dev;
};

/* ---- the arithmetic of the two sample types ---------------------------------------------------------------------------------- */
__device__ __forceinline__ void txw_bf(double &x, double &y, double a, double b) { x = a - b; y = a + b; }
__device__ __forceinline__ void txw_bf(int &x, int &y, int a, int b)
{
    x = (int)((unsigned)a - (unsigned)b);
    y = (int)((unsigned)a + (unsigned)b);
}
/* CMUL (tx_priv.h:88-93 / :117-126) */
__device__ __forceinline__ void txw_cmul(double &dre, double &dim, double are, double aim, double bre, double bim)
{
    dre = are * bre - aim * bim;
    dim = are * bim + aim * bre;
}
__device__ __forceinline__ void txw_cmul(int &dre, int &dim, int are, int aim, int bre, int bim)
{
    long accu = (long)bre * are;
    accu -= (long)bim * aim;
    dre = (int)((accu + 0x40000000) >> 31);
    accu = (long)bim * are;
    accu += (long)bre * aim;
    dim = (int)((accu + 0x40000000) >> 31);
}
__device__ __forceinline__ double txw_fold(double a, double b) { return a + b; }
__device__ __forceinline__ int txw_fold(int a, int b) { return (int)((unsigned)a + (unsigned)b + 32u) >> 6; }
__device__ __forceinline__ double txw_neg(double a) { return -a; }
__device__ __forceinline__ int txw_neg(int a) { return (int)(0u - (unsigned)a); }

/* TRANSFORM / BUTTERFLIES (tx_template.c:512-538) on one butterfly; `mul` false: BUTTERFLIES alone with t1, t2 = a2 and t5, t6 = a3 */
template <typename T>
__device__ __forceinline__ void txw_butterfly(TxwCpx<T> &a0, TxwCpx<T> &a1, TxwCpx<T> &a2, TxwCpx<T> &a3, T wre, T wim, bool mul)
{
    T t1 = a2.re, t2 = a2.im, t5 = a3.re, t6 = a3.im, t3, t4;
    if (mul) {
        txw_cmul(t1, t2, a2.re, a2.im, wre, txw_neg(wim));
        txw_cmul(t5, t6, a3.re, a3.im, wre, wim);
    }
    const T r0 = a0.re, i0 = a0.im, r1 = a1.re, i1 = a1.im;
    txw_bf(t3, t5, t5, t1);
    txw_bf(a2.re, a0.re, r0, t5);
    txw_bf(a3.im, a1.im, i1, t3);
    txw_bf(t4, t6, t2, t6);
    txw_bf(a3.re, a1.re, r1, t4);
    txw_bf(a2.im, a0.im, i0, t6);
}

template <bool WG>
__device__ __forceinline__ void txw_sync()
{
    if (WG) {
        __syncthreads();
    } else {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

template <typename T, bool WG>
__device__ __forceinline__ void txw_fft_lds(TxwCpx<T> *z, const TxwDev &d, int lane, int TS)
{
    const T *cos_tab = static_cast<const T *>(d.cos_tab);
    for (int b = lane; b < d.nblocks2; b += TS) {
        const int o = d.blocks2[b];
        const TxwCpx<T> x = z[o], y = z[o + 1];
        TxwCpx<T> s, df;
        txw_bf(df.re, s.re, x.re, y.re);
        txw_bf(df.im, s.im, x.im, y.im);
        z[o] = s;
        z[o + 1] = df;
    }
    for (int l = 2; l <= d.lg; l++) {
        txw_sync<WG>();
        const int q = 1 << (l - 2);
        const int o1 = TXW_PAD(q), o2 = TXW_PAD(2 * q), o3 = TXW_PAD(3 * q);
        const T *tab = cos_tab + d.cos_off[l];
        const uint32_t *sc = d.sched + d.sched_off[l];
        for (int b = lane; b < d.sched_cnt[l]; b += TS) {
            const uint32_t e = sc[b];
            const int a0 = e & 0xFFFF, k = e >> 16;
            TxwCpx<T> v0 = z[a0], v1 = z[a0 + o1], v2 = z[a0 + o2], v3 = z[a0 + o3];
            txw_butterfly<T>(v0, v1, v2, v3, tab[k], tab[q - k], !(l <= 4 && k == 0));
            z[a0] = v0; z[a0 + o1] = v1; z[a0 + o2] = v2; z[a0 + o3] = v3;
        }
    }
    txw_sync<WG>();
}

/* ff_tx_fft (tx_template.c:735-749) / ff_tx_mdct_fwd / ff_tx_mdct_inv (:1272-1342) on contiguous rows */
template <typename T, int KIND /* 0 FFT, 1 forward MDCT, 2 inverse MDCT */, bool WG>
__global__ __launch_bounds__(1024) void k_txw(TxwDev d, const T *in, size_t in_pitch, T *out, size_t out_pitch, int nt, int teams_total)
{
    extern __shared__ __align__(16) uint8_t lds_raw[];
    const int wave = WG ? 0 : __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = WG ? (int)threadIdx.x : (int)(threadIdx.x & 63);
    const int TS = WG ? (int)blockDim.x : 64;
    const int n = d.n;
    TxwCpx<T> *z = reinterpret_cast<TxwCpx<T> *>(lds_raw) + (size_t)wave * (TXW_PAD(n) + 1);
    const TxwCpx<T> *ex = static_cast<const TxwCpx<T> *>(d.exp);
    for (int t = WG ? (int)blockIdx.x : (int)(blockIdx.x * (blockDim.x >> 6)) + wave; t < nt; t += teams_total) {
        const T *src = reinterpret_cast<const T *>(reinterpret_cast<const uint8_t *>(in) + (size_t)t * in_pitch);
        T *dst = reinterpret_cast<T *>(reinterpret_cast<uint8_t *>(out) + (size_t)t * out_pitch);
        if (KIND == 0) {
            const TxwCpx<T> *s2 = reinterpret_cast<const TxwCpx<T> *>(src);
            for (int j = lane; j < n; j += TS)
                z[d.map[j]] = s2[j];
        } else if (KIND == 1) {
            const int len2 = n, len3 = 3 * n;
            for (int i = lane; i < n; i += TS) {
                const int k = 2 * i;
                T re, im;
                if (k < len2) {
                    re = txw_fold(txw_neg(src[len2 + k]), src[len2 - 1 - k]);
                    im = txw_fold(txw_neg(src[len3 + k]), txw_neg(src[len3 - 1 - k]));
                } else {
                    re = txw_fold(txw_neg(src[len2 + k]), txw_neg(src[5 * len2 - 1 - k]));
                    im = txw_fold(src[k - len2], txw_neg(src[len3 - 1 - k]));
                }
                TxwCpx<T> v;
                txw_cmul(v.im, v.re, re, im, ex[i].re, ex[i].im);
                z[d.map[i]] = v;
            }
        } else {
            /* z[i] = CMUL3({ in2[-map[i]], in1[map[i]] }, exp[i]) with exp[i] = the natural table at map[i] / 2 (ff_tx_mdct_gen_exp's
             * pre_tab): walked in input order j = map[i] / 2, scattered through the inverse permutation */
            for (int j = lane; j < n; j += TS) {
                TxwCpx<T> v;
                txw_cmul(v.re, v.im, src[2 * n - 1 - 2 * j], src[2 * j], ex[j].re, ex[j].im);
                z[d.map[j]] = v;
            }
        }
        txw_sync<WG>();
        txw_fft_lds<T, WG>(z, d, lane, TS);
        if (KIND == 0) {
            TxwCpx<T> *o2 = reinterpret_cast<TxwCpx<T> *>(dst);
            for (int i = lane; i < n; i += TS)
                o2[i] = z[TXW_PAD(i)];
        } else {
            const int len4 = n >> 1;
            for (int i = lane; i < len4; i += TS) {
                const int i0 = len4 + i, i1 = len4 - i - 1;
                const TxwCpx<T> z1 = z[TXW_PAD(i1)], z0 = z[TXW_PAD(i0)], e0 = ex[i0], e1 = ex[i1];
                if (KIND == 1) {
                    T a, b, c, f;
                    txw_cmul(a, b, z0.re, z0.im, e0.im, e0.re); /* dst[2 i1 + 1], dst[2 i0] */
                    txw_cmul(c, f, z1.re, z1.im, e1.im, e1.re); /* dst[2 i0 + 1], dst[2 i1] */
                    dst[2 * i1 + 1] = a; dst[2 * i0] = b; dst[2 * i0 + 1] = c; dst[2 * i1] = f;
                } else {
                    T a, b, c, f;
                    txw_cmul(a, b, z1.im, z1.re, e1.im, e1.re); /* z[i1].re, z[i0].im */
                    txw_cmul(c, f, z0.im, z0.re, e0.im, e0.re); /* z[i0].re, z[i1].im */
                    dst[2 * i1] = a; dst[2 * i0 + 1] = b; dst[2 * i0] = c; dst[2 * i1 + 1] = f;
                }
            }
        }
        txw_sync<WG>();
    }
}

/* ---- host ---------------------------------------------------------------------------------------------------------------------- */
static int txw_sr_perm(int i, int len, int inv) /* split_radix_permutation, libavutil/tx.c:125-135 */
{
    len >>= 1;
    if (len <= 1)
        return i & 1;
    if (!(i & len))
        return txw_sr_perm(i, len, inv) * 2;
    len >>= 1;
    return txw_sr_perm(i, len, inv) * 4 + 1 - 2 * (!(i & len) ^ inv);
}

static void txw_schedule(int o, int n, int lg, std::vector<uint32_t> *lev, std::vector<uint16_t> *b2)
{
    if (n == 1)
        return;
    if (n == 2) {
        b2->push_back((uint16_t)TXW_PAD(o));
        return;
    }
    const int q = n >> 2;
    txw_schedule(o, n >> 1, lg - 1, lev, b2);
    txw_schedule(o + 2 * q, q, lg - 2, lev, b2);
    txw_schedule(o + 3 * q, q, lg - 2, lev, b2);
    for (int k = 0; k < q; k++)
        lev[lg].push_back((uint32_t)TXW_PAD(o + k) | ((uint32_t)k << 16));
}

/* RESCALE of TX_INT32 (tx_priv.h:139): the double product goes through llrintf, i.e. is rounded to float first */
static int32_t txw_rescale(double x)
{
    long long v = llrintf((float)(x * 2147483648.0));
    return (int32_t)(v < INT32_MIN ? INT32_MIN : v > INT32_MAX ? INT32_MAX : v);
}

static size_t txw_elem(const FFHipTxWide *w) { return w->is_int ? sizeof(int32_t) : sizeof(double); }

void ffhip_txw_free(FFHipTxWide *w)
{
    if (!w)
        return;
    FFHipDeviceGuard dg(w->device);
    if (w->dev)
        (void)hipFree(w->dev);
    delete w;
}

int ffhip_txw_max_points(int is_int) { return is_int ? 16384 : 8192; }

int ffhip_txw_create(FFHipTxWide **pw, int is_int, int is_mdct, int inv, int len, double scale)
{
    *pw = nullptr;
    const int n = is_mdct ? len >> 1 : len;
    if (n < 4 || n > ffhip_txw_max_points(is_int) || (n & (n - 1)) || (is_mdct && len < 16)) {
        ffhip_set_error("ffhip_tx_init: %s %s len %d: powers of two with 4..%d complex points only", is_int ? "int32" : "double",
                        is_mdct ? "MDCT" : "FFT", len, ffhip_txw_max_points(is_int));
        return FFHIP_EINVAL;
    }
    if (!ffhip_have_device())
        return FFHIP_ENOSYS;
    FFHipTxWide *w = new (std::nothrow) FFHipTxWide();
    if (!w)
        return FFHIP_ENOMEM;
    memset(w, 0, sizeof(*w));
    w->device = ffhip_current_device();
    w->inv = !!inv; w->len = len; w->is_int = is_int; w->is_mdct = is_mdct;
    int lg = 0;
    while ((1 << lg) < n)
        lg++;
    const size_t es = txw_elem(w);
    std::vector<int> map(n);
    for (int i = 0; i < n; i++)
        map[-txw_sr_perm(i, n, w->inv) & (n - 1)] = i;
    for (int i = 0; i < n; i++)
        map[i] = TXW_PAD(map[i]);
    /* ff_tx_mdct_gen_exp (tx_template.c:2107-2134); scale_d is the caller's float (int32: SCALE_TYPE float) or double */
    std::vector<uint8_t> ex(is_mdct ? (size_t)n * 2 * es : 16, 0);
    if (is_mdct) {
        const double theta = (scale < 0 ? n : 0) + 1.0 / 8.0, rt = sqrt(fabs(scale));
        for (int i = 0; i < n; i++) {
            const double alpha = M_PI_2 * (i + theta) / n;
            /* The reference writes cos(alpha) and sin(alpha) side by side; gcc turns such a pair into ONE sincos() call, and glibc's
             * sincos() cosine is not always cos()'s — n = 2048, i = 1452 differ in the last bit.  A double table keeps that bit, so
             * "the reference" is the gcc-built libavutil here (what distributions ship, and what oracle/_ref is): call sincos(). */
            double sn, cs;
            sincos(alpha, &sn, &cs);
            if (is_int) {
                reinterpret_cast<int32_t *>(ex.data())[2 * i] = txw_rescale(cs * rt);
                reinterpret_cast<int32_t *>(ex.data())[2 * i + 1] = txw_rescale(sn * rt);
            } else {
                reinterpret_cast<double *>(ex.data())[2 * i] = cs * rt;
                reinterpret_cast<double *>(ex.data())[2 * i + 1] = sn * rt;
            }
        }
    }
    /* ff_tx_init_tab_<m> (tx_template.c:69-79): cos(2 pi i / m), i < m / 4, then an exact 0 */
    TxwDev &d = w->d;
    d.n = n; d.lg = lg;
    std::vector<uint8_t> cosv;
    int ncos = 0;
    for (int l = 2; l <= lg; l++) {
        const int m = 1 << l;
        const double freq = 2 * M_PI / m;
        d.cos_off[l] = ncos;
        cosv.resize((size_t)(ncos + m / 4 + 1) * es, 0);
        for (int i = 0; i < m / 4; i++) {
            if (is_int)
                reinterpret_cast<int32_t *>(cosv.data())[ncos + i] = txw_rescale(cos(i * freq));
            else
                reinterpret_cast<double *>(cosv.data())[ncos + i] = cos(i * freq);
        }
        ncos += m / 4 + 1;
    }
    std::vector<uint32_t> lev[20];
    std::vector<uint16_t> b2;
    txw_schedule(0, n, lg, lev, &b2);
    std::vector<uint32_t> sched;
    for (int l = 2; l <= lg; l++) {
        d.sched_off[l] = (int)sched.size();
        d.sched_cnt[l] = (int)lev[l].size();
        sched.insert(sched.end(), lev[l].begin(), lev[l].end());
    }
    d.nblocks2 = (int)b2.size();
    size_t off_map = 0, off_exp, off_cos, off_sched, off_b2, total;
    off_exp = (off_map + map.size() * 4 + 15) & ~(size_t)15;
    off_cos = (off_exp + ex.size() + 15) & ~(size_t)15;
    off_sched = (off_cos + cosv.size() + 15) & ~(size_t)15;
    off_b2 = (off_sched + sched.size() * 4 + 15) & ~(size_t)15;
    total = (off_b2 + b2.size() * 2 + 16 + 15) & ~(size_t)15;
    std::vector<uint8_t> blob(total, 0);
    memcpy(blob.data() + off_map, map.data(), map.size() * 4);
    memcpy(blob.data() + off_exp, ex.data(), ex.size());
    memcpy(blob.data() + off_cos, cosv.data(), cosv.size());
    memcpy(blob.data() + off_sched, sched.data(), sched.size() * 4);
    memcpy(blob.data() + off_b2, b2.data(), b2.size() * 2);
    if (hipMalloc(&w->dev, total) != hipSuccess || hipMemcpy(w->dev, blob.data(), total, hipMemcpyHostToDevice) != hipSuccess) {
        ffhip_set_error("ffhip_tx_init: table upload failed");
        ffhip_txw_free(w);
        return FFHIP_ENOMEM;
    }
    uint8_t *base = (uint8_t *)w->dev;
    d.map = (const int *)(base + off_map);
    d.exp = base + off_exp;
    d.cos_tab = base + off_cos;
    d.sched = (const uint32_t *)(base + off_sched);
    d.blocks2 = (const uint16_t *)(base + off_b2);
    *pw = w;
    return 0;
}

/* elements of one transform's input / output row */
size_t ffhip_txw_in_elems(const FFHipTxWide *w) { return w->is_mdct ? (w->inv ? (size_t)w->len : (size_t)2 * w->len) : (size_t)2 * w->len; }
size_t ffhip_txw_out_elems(const FFHipTxWide *w) { return w->is_mdct ? (size_t)w->len : (size_t)2 * w->len; }
size_t ffhip_txw_elem_size(const FFHipTxWide *w) { return txw_elem(w); }
int ffhip_txw_device(const FFHipTxWide *w) { return w->device; }

template <typename T>
static int txw_launch(const FFHipTxWide *w, void *out, size_t out_pitch, const void *in, size_t in_pitch, int nt, hipStream_t stream)
{
    const int n = w->d.n;
    const size_t zb = ((size_t)TXW_PAD(n) + 1) * sizeof(TxwCpx<T>);
    const bool wg = n > 1024;
    const int threads = wg ? (n >= 8192 ? 1024 : 512) : 256;
    const size_t lds = wg ? zb : 4 * zb;
    int cus = 256, dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
        cus = prop.multiProcessorCount;
    int per_cu = (int)((160 * 1024) / (((lds + 1279) / 1280) * 1280));
    if (per_cu * (threads / 64) > 32) per_cu = 32 / (threads / 64);
    if (per_cu < 1) per_cu = 1;
    const int teams_per_block = wg ? 1 : 4;
    int blocks = cus * per_cu;
    if (blocks > cdiv(nt, teams_per_block))
        blocks = cdiv(nt, teams_per_block);
    const int kind = w->is_mdct ? 1 + w->inv : 0;
#define TXW_GO(KIND, WG)                                                                                                               \
    do {                                                                                                                               \
        static FFHipPerDeviceOnce attr;                                                                                                \
        if (attr.enter()) {                                                                                                            \
            (void)hipFuncSetAttribute((const void *)k_txw<T, KIND, WG>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);       \
            attr.leave(true);                                                                                                          \
        }                                                                                                                              \
        hipLaunchKernelGGL((k_txw<T, KIND, WG>), dim3(blocks), dim3(threads), lds, stream, w->d, (const T *)in, in_pitch, (T *)out,    \
                           out_pitch, nt, blocks * teams_per_block);                                                                   \
    } while (0)
    if (wg) {
        if (kind == 0) TXW_GO(0, true); else if (kind == 1) TXW_GO(1, true); else TXW_GO(2, true);
    } else {
        if (kind == 0) TXW_GO(0, false); else if (kind == 1) TXW_GO(1, false); else TXW_GO(2, false);
    }
#undef TXW_GO
    LAUNCH_CHECK();
    return 0;
}

int ffhip_txw_batch(const FFHipTxWide *w, void *out, size_t out_pitch, const void *in, size_t in_pitch, int nt, hipStream_t stream)
{
    const size_t es = txw_elem(w);
    if (((uintptr_t)in | in_pitch | (uintptr_t)out | out_pitch) & (2 * es - 1)) {
        ffhip_set_error("ffhip_tx: double / int32 batches need rows aligned to a complex sample (%d bytes)", (int)(2 * es));
        return FFHIP_EINVAL;
    }
    return w->is_int ? txw_launch<int>(w, out, out_pitch, in, in_pitch, nt, stream)
                     : txw_launch<double>(w, out, out_pitch, in, in_pitch, nt, stream);
}
