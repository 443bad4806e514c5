/*
 * tx_dcst1.hip — AV_TX_FLOAT_DCT_I / AV_TX_FLOAT_DST_I (libavutil/tx.h:107-128), forward, even lengths 4..1024 (round 6).
 *
 * What the reference computes (libavutil/tx_template.c:2006-2075): the input mirrored into 2 (N - 1) reals (DCT-I) or into 2 (N + 1)
 * reals with the odd symmetry (DST-I), then the half-complex RDFT of that length (ff_tx_rdft_r2r_mod2 / _r2i_mod2, :1718-1830: N even
 * makes the RDFT's length 2 mod 4), whose FFT has N -+ 1 points — an odd number, 63 = 7 x 9 for the one caller in the tree
 * (libavcodec/wmavoice.c:398-404, N = 64; 65 points run on ff_tx_fft_naive).  Every output is a fixed linear function of the N inputs.
 * The two outputs in the middle are not the textbook ones when *scale != 1: the mod-2 RDFT multiplies data[len4].re by its factor
 * before the pair (len4, len4 + 1) is read, and r2r hands out[len4 + 1] back through 1 / scale (tx_template.c:1752, :1806) — a caller
 * sees those values, so they are the specification here.
 *
 * On the device the transform IS that linear function: the N x N matrix is built once per context on the host by running the
 * reference's sequence of operations (mirror, the packed complex DFT of N -+ 1 points, the RDFT's post-pass with its in-place
 * aliasing and float-rounded factors) on the unit vectors in double precision; the kernel is the product in double precision (what
 * differs from the C code's floats is rounding alone: tolerance 2^-18 of a transform's largest output, as for the other float
 * transforms).  Lengths this short do not pay for a 63- or 65-point FFT network on a 64-wide machine; a product is what the matrix
 * cores are for:
 *   k_dcst1_m  v_mfma_f64_16x16x4_f64: a wave owns 32 transforms x 64 outputs (2 x 4 accumulator tiles of 16 x 16 doubles, 64 VGPRs).
 *              A = the inputs, lane (i = l % 16, q = l / 16) holding x[t0 + i][.] — the four k slots of a step are the columns
 *              16 jj + 4 q + s, so that a lane's float4 load serves four steps; B = the matrix as doubles, row 16 jj + 4 q + s, 16
 *              consecutive outputs per 16 lanes (128-byte row segments out of L1 / L2: every wave of the launch reads the same matrix);
 *              D comes back as row 4 v + l / 16, column l % 16 (tools/ubench/mfma_f64_layout.hip printed it) and is stored as floats,
 *              64 bytes per row and 16 lanes.  No LDS.
 *   k_dcst1    the first form (a lane per output, eight transforms per lane, v_fma_f64 from LDS inputs): 12.5 TFLOP/s at N = 64,
 *              kept for the measurement (FFHIP_DCST1_VALU=1).
 */
#include <math.h>
#include <new>
#include <stdint.h>
#include <vector>

#include "common.h"
#include "tx_kernels.h"

struct FFHipTxDcst1 {
    int device = 0;
    int n = 0, dst = 0;
    float *mat = nullptr; /* n x n floats, mat[j * n + k] = what input j contributes to output k */
    double *matd = nullptr; /* the same as doubles, jpad x kpad (rows to a multiple of 16, columns to a multiple of 64), zero padded */
    int jpad = 0, kpad = 0;
};

#define DC1_R 8 /* transforms per lane */

__global__ __launch_bounds__(256) void k_dcst1(const float *__restrict__ mat, int n, const float *in, size_t in_pitch, ptrdiff_t istride,
                                               float *out, size_t out_pitch, int nt, int groups)
{
    __shared__ float xs[8192];
    const int tt_all = groups * DC1_R;
    const long t0 = (long)blockIdx.x * tt_all;
    for (int i = threadIdx.x; i < tt_all * n; i += 256) {
        const int tt = i / n, j = i - tt * n;
        const long t = t0 + tt;
        xs[i] = t < nt ? *(const float *)((const uint8_t *)in + (size_t)t * in_pitch + (ptrdiff_t)j * istride * (ptrdiff_t)sizeof(float)) : 0.f;
    }
    __syncthreads();
    /* n < 256: 256 / n groups of n lanes, a group per eight transforms; otherwise one group whose lanes walk the outputs */
    const int g = groups > 1 ? (int)threadIdx.x / n : 0;
    if (g >= groups)
        return;
    const float *x = xs + g * DC1_R * n;
    for (int k = groups > 1 ? (int)threadIdx.x - g * n : (int)threadIdx.x; k < n; k += 256) {
        double acc[DC1_R];
#pragma unroll
        for (int r = 0; r < DC1_R; r++)
            acc[r] = 0.0;
        for (int j = 0; j < n; j++) {
            const double m = (double)mat[(size_t)j * n + k];
#pragma unroll
            for (int r = 0; r < DC1_R; r++)
                acc[r] += m * (double)x[r * n + j];
        }
#pragma unroll
        for (int r = 0; r < DC1_R; r++) {
            const long t = t0 + g * DC1_R + r;
            if (t < nt)
                *(float *)((uint8_t *)out + (size_t)t * out_pitch + (size_t)k * sizeof(float)) = (float)acc[r];
        }
    }
}


typedef double dc1_d4 __attribute__((ext_vector_type(4)));
typedef float dc1_f4 __attribute__((ext_vector_type(4)));

/* VEC: rows of the input are contiguous and 16-byte aligned, n a multiple of 4: a float4 per lane and sixteen columns */
template <bool VEC>
__global__ __launch_bounds__(256) void k_dcst1_m(const double *__restrict__ matd, int n, int jpad, int kpad, const float *in, size_t in_pitch,
                                                 ptrdiff_t istride, float *out, size_t out_pitch, int nt)
{
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int i = lane & 15, q = lane >> 4;
    const long t0 = ((long)blockIdx.x * 4 + wave) * 32;
    if (t0 >= nt)
        return;
    const int k0 = blockIdx.y * 64;
    const long ta = min(t0 + i, (long)nt - 1), tb = min(t0 + 16 + i, (long)nt - 1); /* rows past the batch: clamped loads, no stores */
    const uint8_t *ra = (const uint8_t *)in + (size_t)ta * in_pitch, *rb = (const uint8_t *)in + (size_t)tb * in_pitch;
    dc1_d4 acc[2][4];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int kt = 0; kt < 4; kt++)
            acc[a][kt] = (dc1_d4){ 0, 0, 0, 0 };
    const double *bp = matd + (size_t)(4 * q) * kpad + k0 + i;
    for (int jj = 0; jj < jpad; jj += 16) {
        const int j = jj + 4 * q;
        float xa[4], xb[4];
        if (VEC) {
            const dc1_f4 z = { 0, 0, 0, 0 };
            const dc1_f4 va = j < n ? *(const dc1_f4 *)(ra + (size_t)j * 4) : z, vb = j < n ? *(const dc1_f4 *)(rb + (size_t)j * 4) : z;
#pragma unroll
            for (int s = 0; s < 4; s++) { xa[s] = va[s]; xb[s] = vb[s]; }
        } else {
#pragma unroll
            for (int s = 0; s < 4; s++) {
                const bool ok = j + s < n;
                const ptrdiff_t o = (ptrdiff_t)(ok ? j + s : 0) * istride * (ptrdiff_t)sizeof(float);
                xa[s] = ok ? *(const float *)(ra + o) : 0.f;
                xb[s] = ok ? *(const float *)(rb + o) : 0.f;
            }
        }
#pragma unroll
        for (int s = 0; s < 4; s++) {
            const double *b = bp + (size_t)(jj + s) * kpad;
            const double b0 = b[0], b1 = b[16], b2 = b[32], b3 = b[48];
            const double a0 = (double)xa[s], a1 = (double)xb[s];
            acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
            acc[0][2] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b2, acc[0][2], 0, 0, 0);
            acc[1][2] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b2, acc[1][2], 0, 0, 0);
            acc[0][3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b3, acc[0][3], 0, 0, 0);
            acc[1][3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b3, acc[1][3], 0, 0, 0);
        }
    }
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int v = 0; v < 4; v++) {
            const long t = t0 + 16 * a + 4 * v + q;
            if (t >= nt)
                continue;
            float *o = (float *)((uint8_t *)out + (size_t)t * out_pitch);
#pragma unroll
            for (int kt = 0; kt < 4; kt++) {
                const int k = k0 + 16 * kt + i;
                if (k < n)
                    o[k] = (float)acc[a][kt][v];
            }
        }
}

/* ================================================================================================== */
/* host side: the matrix */

/* ff_tx_rdft_r2r_mod2 / _r2i_mod2 (tx_template.c:1718-1830) on one mirrored sequence, in double precision with the float-rounded
 * factors of ff_tx_rdft_init (:1601-1655, forward): buf holds the L reals in, the N outputs at its head out */
static void dcst1_half_rdft(std::vector<double> &buf, int L, bool r2r, float scale)
{
    const int len2 = L >> 1, len4 = L >> 2, al4 = (L + 3) / 4;
    const double f = 2 * M_PI / L, m = (double)scale;
    const double fact[8] = { (double)(float)(1.0 * m), (double)(float)(1.0 * m), (double)(float)m, (double)(float)-m, (double)(float)(0.5 * m),
                             r2r ? (double)(1 / scale) : (double)(float)(-0.5 * m), (double)(float)(0.5 * m), (double)(float)(-0.5 * m) };
    std::vector<double> tcos(al4), tsin(al4);
    for (int i = 0; i < al4; i++) {
        tcos[i] = (double)(float)cos(i * f);
        tsin[i] = (double)((float)cos(((L - i * 4) / 4.0) * f) * -1);
    }
    /* the FFT of len2 complex points (re, im) = (buf[2 j], buf[2 j + 1]), forward, no scale */
    std::vector<double> z(L + 2, 0.0);
    for (int j = 0; j < len2; j++) {
        const double re = buf[2 * j], im = buf[2 * j + 1];
        if (re == 0.0 && im == 0.0)
            continue; /* (a unit vector's mirror has two non-zero samples) */
        for (int k = 0; k < len2; k++) {
            const double a = -2 * M_PI * (double)((long)j * k % len2) / len2, c = cos(a), s = sin(a);
            z[2 * k] += re * c - im * s;
            z[2 * k + 1] += re * s + im * c;
        }
    }
    /* the post-pass, in place as the reference runs it (data[] and out[] are the same array) */
    double *d = z.data();
    double dc = d[0];
    d[0] = dc + d[1];
    dc = dc - d[1];
    d[0] = fact[0] * d[0];
    dc = fact[1] * dc;
    d[2 * len4] = fact[2] * d[2 * len4];
    double mid;
    {
        const double sfr = d[2 * len4], sfi = d[2 * len4 + 1], slr = d[2 * len4 + 2], sli = d[2 * len4 + 3];
        const double t0 = r2r ? fact[4] * (sfr + slr) : fact[5] * (sfi - sli);
        const double t1 = fact[6] * (sfi + sli), t2 = fact[7] * (sfr - slr);
        mid = r2r ? t0 - (t1 * tcos[len4] - t2 * tsin[len4]) : t0 + (t1 * tsin[len4] + t2 * tcos[len4]);
    }
    for (int i = 1; i <= len4; i++) {
        const double sfr = d[2 * i], sfi = d[2 * i + 1], slr = d[2 * (len2 - i)], sli = d[2 * (len2 - i) + 1];
        const double t0 = r2r ? fact[4] * (sfr + slr) : fact[5] * (sfi - sli);
        const double t1 = fact[6] * (sfi + sli), t2 = fact[7] * (sfr - slr);
        if (r2r) {
            const double t3 = t1 * tcos[i] - t2 * tsin[i];
            d[i] = t0 + t3;
            d[L - i] = t0 - t3;
        } else {
            const double t3 = t1 * tsin[i] + t2 * tcos[i];
            d[i - 1] = t3 - t0;
            d[L - i - 1] = t0 + t3;
        }
    }
    for (int i = 1; i < len4 + (r2r ? 0 : 1); i++)
        d[len2 - i] = d[L - i];
    if (r2r) {
        d[len2] = dc;
        d[len4 + 1] = mid * fact[5];
    } else {
        d[len4] = mid;
    }
    buf.swap(z);
}

int ffhip_dcst1_create(FFHipTxDcst1 **pw, int is_dst, int len, float scale)
{
    if (len < 4 || len > 1024 || (len & 1)) {
        ffhip_set_error("ffhip_tx_init: DCT-I / DST-I take even lengths 4..1024 on the hip path (len %d; av_tx_init refuses odd ones, "
                        "tx_template.c:2081)", len);
        return FFHIP_ENOSYS;
    }
    FFHipTxDcst1 *w = new (std::nothrow) FFHipTxDcst1();
    if (!w)
        return FFHIP_ENOMEM;
    w->n = len; w->dst = !!is_dst;
    (void)hipGetDevice(&w->device);
    const int n = len, ln = is_dst ? n + 1 : n - 1, L = 2 * ln;
    w->jpad = (n + 15) & ~15; w->kpad = (n + 63) & ~63;
    std::vector<float> mat((size_t)n * n);
    std::vector<double> matd((size_t)w->jpad * w->kpad, 0.0);
    std::vector<double> buf;
    for (int j = 0; j < n; j++) {
        buf.assign(L + 2, 0.0);
        if (!is_dst) {
            /* ff_tx_dctI (tx_template.c:2040-2056): tmp[i] = tmp[2 len - i] = src[i], i < len; tmp[len] = src[len] */
            if (j < ln) {
                buf[j] = 1.0;
                if (j)
                    buf[2 * ln - j] = 1.0;
            } else {
                buf[ln] = 1.0;
            }
        } else {
            /* ff_tx_dstI (:2058-2080): tmp[i] = -src[i - 1], tmp[2 len - i] = src[i - 1], 1 <= i < len; tmp[0] = tmp[len] = 0 */
            buf[j + 1] = -1.0;
            buf[2 * ln - (j + 1)] = 1.0;
        }
        dcst1_half_rdft(buf, L, !is_dst, scale);
        for (int k = 0; k < n; k++) {
            mat[(size_t)j * n + k] = (float)buf[k];
            matd[(size_t)j * w->kpad + k] = buf[k];
        }
    }
    if (hipMalloc(&w->mat, mat.size() * sizeof(float)) != hipSuccess ||
        hipMemcpy(w->mat, mat.data(), mat.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess ||
        hipMalloc(&w->matd, matd.size() * sizeof(double)) != hipSuccess ||
        hipMemcpy(w->matd, matd.data(), matd.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) {
        ffhip_set_error("ffhip_tx_init: table upload failed");
        ffhip_dcst1_free(w);
        return FFHIP_ENOMEM;
    }
    *pw = w;
    return 0;
}

void ffhip_dcst1_free(FFHipTxDcst1 *w)
{
    if (!w)
        return;
    if (w->mat)
        (void)hipFree(w->mat);
    if (w->matd)
        (void)hipFree(w->matd);
    delete w;
}

int ffhip_dcst1_device(const FFHipTxDcst1 *w) { return w->device; }
int ffhip_dcst1_len(const FFHipTxDcst1 *w) { return w->n; }

/* rows of n floats out (contiguous); the inputs of a row `istride` floats apart (av_tx_fn's stride, tx_template.c:2047) */
int ffhip_dcst1_batch(const FFHipTxDcst1 *w, float *out, size_t out_pitch, const float *in, size_t in_pitch, ptrdiff_t istride, int nt,
                      hipStream_t stream)
{
    if (nt <= 0)
        return 0;
    if (((uintptr_t)out | (uintptr_t)in | out_pitch | in_pitch) & 3) {
        ffhip_set_error("ffhip_tx: DCT-I / DST-I rows are 4-byte aligned");
        return FFHIP_EINVAL;
    }
    if (!FFHIP_KNOB("FFHIP_DCST1_VALU")) {
        const dim3 grid((unsigned)cdiv(nt, 128), (unsigned)(w->kpad / 64));
        if (istride == 1 && !(w->n & 3) && !(((uintptr_t)in | in_pitch) & 15))
            hipLaunchKernelGGL((k_dcst1_m<true>), grid, dim3(256), 0, stream, w->matd, w->n, w->jpad, w->kpad, in, in_pitch, istride, out, out_pitch, nt);
        else
            hipLaunchKernelGGL((k_dcst1_m<false>), grid, dim3(256), 0, stream, w->matd, w->n, w->jpad, w->kpad, in, in_pitch, istride, out, out_pitch, nt);
        LAUNCH_CHECK();
        return 0;
    }
    const int groups = w->n < 256 ? 256 / w->n : 1;
    const int per_wg = groups * DC1_R;
    hipLaunchKernelGGL(k_dcst1, dim3((unsigned)cdiv(nt, per_wg)), dim3(256), 0, stream, w->mat, w->n, in, in_pitch, istride, out, out_pitch,
                       nt, groups);
    LAUNCH_CHECK();
    return 0;
}
