/*
 * aac_api.hip — AACDecDSP.imdct_and_windowing of the float AAC decoder, 1024-sample frames, batched and HBM-resident
 * (libavcodec/aac/aacdec_dsp_template.c:325-387): inverse MDCT(s) -> window -> overlap-add without the
 * frame leaving the device.
 *
 * The reference keeps a per-channel overlap state `saved` and so looks sequential in time.  It is not: the new `saved` is a
 * function of the current frame's MDCT output buf[] alone, and out[] needs the previous frame's `saved` only.  So a run of frames
 * is three data-parallel steps —
 *   1. buf[f]   = one 1024-point or eight 128-point inverse MDCTs (ffhip_tx_batch_dev, one launch per run of equal kind);
 *   2. tail[f]  = the overlap state frame f leaves behind (k_aac_tail: windowed short-block overlaps; a long frame's tail is
 *                 simply the upper half of buf[f] and is read from there);
 *   3. out[f]   = overlap of tail[f - 1] (the caller's `saved` for the first frame) with buf[f]'s head under the previous frame's
 *                 window shape (k_aac_out);
 * then the last frame's tail is the caller's new `saved`.  Each output sample is one element of an AVFloatDSPContext
 * .vector_fmul_window call (libavutil/float_dsp.c:79-97) or a copy, evaluated with the same two products and one sum:
 * bit-identical.  HBM traffic per long frame: coeffs 4 KB in, buf 4 KB out, buf 4 KB + the previous buf's upper 2 KB in, out 4 KB
 * out: 18 KB (a short frame adds its 2 KB tail out and in and a second read of buf; +8 KB per frame when
 * a batch with many transient frames is first sorted by transform kind).
 */
#include <string.h>
#include <mutex>
#include <new>
#include <vector>

#include "kernels/common.h"

enum { AAC_ONLY_LONG, AAC_LONG_START, AAC_EIGHT_SHORT, AAC_LONG_STOP }; /* enum WindowSequence, libavcodec/aac.h:63-68 */

struct FFHipAacImdct {
    int device = 0; /* windows, work buffers and the MDCT tables live on this device; every call makes it current */
    FFHipTXContext *tx1024 = nullptr, *tx128 = nullptr, *tx_ltp = nullptr;
    float *ltp_in = nullptr;    /* [ltp_frames][2048]: the windowed predictions before the forward MDCT */
    size_t ltp_frames = 0;
    size_t last_n = 0;          /* the frames of the last imdct_and_windowing_batch_dev call: buf / tail / info / pos still hold them */
    int last_nch = 0;
    bool last_sorted = false;
    int L = 1024, in_short = 128; /* frame length (1024, 960, 768) and the distance of a short window's coefficients in a frame */
    float *win = nullptr;       /* [4][1024]: sine_<L>, sine_<L/8>, kbd_long_<L>, kbd_short_<L/8> */
    uint8_t *work = nullptr;    /* buf [n][1024] f32, tail [n][512] f32, sorted coeffs [n][1024] f32, pos [n] i32, info [n] u8 */
    size_t work_frames = 0;
    std::mutex mu;
};

__device__ __forceinline__ float4 aac_ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ float4 aac_ld4r(const float *p) /* p[3], p[2], p[1], p[0] */
{
    const float4 v = *reinterpret_cast<const float4 *>(p);
    return make_float4(v.w, v.z, v.y, v.x);
}

/* elements e..e+3 (e a multiple of 4) of vector_fmul_window(dst, src0, src1, win, len): four 16-byte loads, the descending
 * operands reversed in registers; per element the reference's two products and one sum */
__device__ __forceinline__ float4 aac_wov4(const float *src0, const float *src1, const float *win, int len, int e)
{
    if (e < len) {
        const float4 a = aac_ld4(src0 + e), b = aac_ld4r(src1 + len - 4 - e), wi = aac_ld4(win + e), wj = aac_ld4r(win + 2 * len - 4 - e);
        return make_float4(a.x * wj.x - b.x * wi.x, a.y * wj.y - b.y * wi.y, a.z * wj.z - b.z * wi.z, a.w * wj.w - b.w * wi.w);
    }
    const int t = 2 * len - 4 - e; /* the lowest of the four mirrored positions */
    const float4 a = aac_ld4r(src0 + t), b = aac_ld4(src1 + len - 4 - t), wi = aac_ld4r(win + t), wj = aac_ld4(win + e);
    return make_float4(a.x * wi.x + b.x * wj.x, a.y * wi.y + b.y * wj.y, a.z * wi.z + b.z * wj.z, a.w * wi.w + b.w * wj.w);
}

/* info byte: sequence | kb << 2 | previous sequence << 3 | previous kb << 5 */
/* pos: where frame f's buf[] lives when the frames were sorted by transform kind (nullptr: at f).  One thread = 4 samples. */
__global__ __launch_bounds__(128) void k_aac_tail(const float *buf, const int *pos, const uint8_t *info, const float *win, float *tail,
                                                  int first_final, int L)
{
    const int H = L >> 1, S = L >> 3, S2 = S >> 1, A = H - S2; /* 1024: 512, 128, 64, 448 */
    const int f = blockIdx.x;
    const float *b = buf + (size_t)(pos ? pos[f] : f) * L;
    const int in = info[f], seq = in & 3;
    /* a long frame's tail is the upper half of its buf[]: k_aac_out reads it there; only the batch's last frames (the state handed
     * back to the caller) are copied out */
    if (seq != AAC_EIGHT_SHORT && f < first_final)
        return;
    const float *swindow = win + ((in >> 2) & 1 ? 3 : 1) * 1024;
    const int s = 4 * threadIdx.x;
    if (s >= H)
        return;
    float4 v;
    if (seq != AAC_EIGHT_SHORT || s >= A) {
        v = aac_ld4(b + H + s); /* LONG_START's two copies (A + S2 samples) are this one range as well */
    } else if (s < S2) {
        v = aac_wov4(b + 3 * S + S2, b + 4 * S, swindow, S2, S2 + s);
    } else {
        const int q = (s - S2) / S, e = (s - S2) - q * S;
        v = aac_wov4(b + (4 + q) * S + S2, b + (5 + q) * S, swindow, S2, e);
    }
    *reinterpret_cast<float4 *>(tail + (size_t)f * 512 + s) = v;
}

__global__ __launch_bounds__(256) void k_aac_out(const float *buf, const int *pos, const uint8_t *info, const float *win, const float *tail,
                                                 const float *saved, int nch, float *out, int L)
{
    const int H = L >> 1, S = L >> 3, S2 = S >> 1, A = H - S2;
    const int f = blockIdx.x;
    const float *b = buf + (size_t)(pos ? pos[f] : f) * L;
    const int in = info[f], seq = in & 3, pseq = (in >> 3) & 3, pkb = (in >> 5) & 1;
    const float *sv = f < nch                    ? saved + (size_t)f * 512
                      : pseq != AAC_EIGHT_SHORT ? buf + (size_t)(pos ? pos[f - nch] : f - nch) * L + H
                                                : tail + (size_t)(f - nch) * 512;
    const float *swindow = win + ((in >> 2) & 1 ? 3 : 1) * 1024, *lwindow_prev = win + (pkb ? 2 : 0) * 1024, *swindow_prev = win + (pkb ? 3 : 1) * 1024;
    const bool long_long = (pseq == AAC_ONLY_LONG || pseq == AAC_LONG_STOP) && (seq == AAC_ONLY_LONG || seq == AAC_LONG_START);
    const int o = 4 * threadIdx.x; /* the region borders (1024: 448 / 576 / 960) are multiples of 4 at every frame length */
    if (o >= L)
        return;
    float4 v;
    if (long_long) {
        v = aac_wov4(sv, b, lwindow_prev, H, o);
    } else if (o < A) {
        v = aac_ld4(sv + o);
    } else if (o < A + S) {
        v = aac_wov4(sv + A, b, swindow_prev, S2, o - A);
    } else if (seq != AAC_EIGHT_SHORT) {
        v = aac_ld4(b + o - H);
    } else if (o < A + 4 * S) {
        const int q = (o - A - S) / S;
        v = aac_wov4(b + q * S + S2, b + (q + 1) * S, swindow, S2, (o - A - S) - q * S);
    } else {
        v = aac_wov4(b + 3 * S + S2, b + 4 * S, swindow, S2, o - (A + 4 * S));
    }
    *reinterpret_cast<float4 *>(out + (size_t)f * 1024 + o) = v;
}

/* sorted[pos[f]] = coeffs[f], compacted to L floats per frame (a short frame's eight windows, `in_short` apart in the caller's
 * frame, S apart here): 16 bytes per thread */
__global__ __launch_bounds__(256) void k_aac_gather(const float *coeffs, const int *pos, const uint8_t *info, float *sorted, int L, int in_short)
{
    const int f = blockIdx.x, i = 4 * threadIdx.x, S = L >> 3;
    if (i >= L)
        return;
    int src = i;
    if ((info[f] & 3) == AAC_EIGHT_SHORT) {
        const int w = i / S;
        src = w * in_short + (i - w * S);
    }
    *reinterpret_cast<float4 *>(sorted + (size_t)pos[f] * L + i) = aac_ld4(coeffs + (size_t)f * 1024 + src);
}

extern "C" void ffhip_aac_imdct_free(FFHipAacImdct **pc)
{
    FFHipDeviceGuard dg(pc && *pc ? (*pc)->device : -1);
    if (!pc || !*pc)
        return;
    FFHipAacImdct *c = *pc;
    ffhip_tx_uninit(&c->tx1024);
    ffhip_tx_uninit(&c->tx128);
    ffhip_tx_uninit(&c->tx_ltp);
    if (c->ltp_in) (void)hipFree(c->ltp_in);
    if (c->win) (void)hipFree(c->win);
    if (c->work) (void)hipFree(c->work);
    delete c;
    *pc = nullptr;
}

extern "C" int ffhip_aac_imdct_create_len(FFHipAacImdct **pc, int frame_len, const float *sine_long, const float *sine_short,
                                          const float *kbd_long, const float *kbd_short, float scale_long, float scale_short)
{
    if (!pc || !sine_long || !sine_short || !kbd_long || !kbd_short)
        return FFHIP_EINVAL;
    *pc = nullptr;
    if (frame_len != 1024 && frame_len != 960 && frame_len != 768) {
        ffhip_set_error("ffhip_aac_imdct_create: frame length %d (1024, 960, 768)", frame_len);
        return FFHIP_EINVAL;
    }
    if (!ffhip_have_device())
        return FFHIP_ENOSYS;
    FFHipAacImdct *c = new (std::nothrow) FFHipAacImdct();
    if (c)
        c->device = ffhip_current_device();
    if (!c)
        return FFHIP_ENOMEM;
    const int L = frame_len, S = L / 8;
    c->L = L;
    c->in_short = L == 768 ? 96 : 128; /* imdct_and_windowing_768 reads in + i * 96, the other two in + i * 128 */
    int r = ffhip_tx_init(&c->tx1024, nullptr, FFHIP_TX_FLOAT_MDCT, 1, L, &scale_long, FFHIP_TX_BITEXACT);
    if (r >= 0)
        r = ffhip_tx_init(&c->tx128, nullptr, FFHIP_TX_FLOAT_MDCT, 1, S, &scale_short, FFHIP_TX_BITEXACT);
    std::vector<float> w(4 * 1024, 0.0f);
    memcpy(&w[0], sine_long, L * sizeof(float));
    memcpy(&w[1024], sine_short, S * sizeof(float));
    memcpy(&w[2048], kbd_long, L * sizeof(float));
    memcpy(&w[3072], kbd_short, S * sizeof(float));
    if (r >= 0 && (hipMalloc(&c->win, w.size() * sizeof(float)) != hipSuccess ||
                   hipMemcpy(c->win, w.data(), w.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess)) {
        ffhip_set_error("ffhip_aac_imdct_create: window upload failed");
        r = FFHIP_ENOMEM;
    }
    if (r < 0) {
        ffhip_aac_imdct_free(&c);
        return r;
    }
    *pc = c;
    return 0;
}

extern "C" int ffhip_aac_imdct_create(FFHipAacImdct **pc, const float *sine_1024, const float *sine_128, const float *kbd_long_1024,
                                      const float *kbd_short_128, float scale_1024, float scale_128)
{
    return ffhip_aac_imdct_create_len(pc, 1024, sine_1024, sine_128, kbd_long_1024, kbd_short_128, scale_1024, scale_128);
}

extern "C" int ffhip_aac_imdct_and_windowing_batch_dev(FFHipAacImdct *c, const float *coeffs, float *out, float *saved,
                                                       const uint8_t *window_sequence, const uint8_t *use_kb_window,
                                                       const uint8_t *prev_sequence, const uint8_t *prev_kb_window, int nch, int nframes,
                                                       void *stream)
{
    FFHipDeviceGuard dg(c ? c->device : -1);
    if (!c || !coeffs || !out || !saved || !window_sequence || !use_kb_window || !prev_sequence || !prev_kb_window || nch <= 0 || nframes < 0)
        return FFHIP_EINVAL;
    const size_t n = (size_t)nch * nframes;
    if (!n)
        return 0;
    if (((uintptr_t)coeffs | (uintptr_t)out | (uintptr_t)saved) & 15) {
        ffhip_set_error("ffhip_aac_imdct_and_windowing: coeffs, out and saved must be 16-byte aligned");
        return FFHIP_EINVAL;
    }
    std::lock_guard<std::mutex> lk(c->mu);
    hipStream_t st = (hipStream_t)stream;
    std::vector<uint8_t> info(n);
    for (size_t i = 0; i < n; i++) {
        const int ps = i < (size_t)nch ? prev_sequence[i] : window_sequence[i - nch], pk = i < (size_t)nch ? prev_kb_window[i] : use_kb_window[i - nch];
        if (window_sequence[i] > 3 || ps > 3) {
            ffhip_set_error("ffhip_aac_imdct_and_windowing: window sequence %d outside 0..3", window_sequence[i] > 3 ? window_sequence[i] : ps);
            return FFHIP_EINVAL;
        }
        info[i] = (uint8_t)(window_sequence[i] | (use_kb_window[i] ? 4 : 0) | ps << 3 | (pk ? 32 : 0));
    }
    /* frames of one kind in memory order form runs, one transform launch each; a batch with many transients (runs) is instead
     * sorted by kind through one gather pass (+8 KB of traffic per frame) and transformed in two launches */
    size_t runs = 1, nlong = 0;
    for (size_t i = 0; i < n; i++) {
        nlong += window_sequence[i] != AAC_EIGHT_SHORT;
        runs += i && (window_sequence[i] == AAC_EIGHT_SHORT) != (window_sequence[i - 1] == AAC_EIGHT_SHORT);
    }
    const int L = c->L, S = L / 8;
    const bool sort = runs > 8 || L != 1024; /* the other frame lengths always compact their frames to L floats */
    const size_t per = 1024 * 4 + 512 * 4 + 1024 * 4 + 4 + 1; /* buf, tail, sorted coefficients, pos, info */
    if (n > c->work_frames) {
        if (c->work) {
            (void)hipStreamSynchronize(st); /* a previous call on this stream may still read the old block */
            (void)hipFree(c->work);
        }
        c->work = nullptr;
        c->work_frames = 0;
        c->last_n = 0;
        if (hipMalloc(&c->work, n * per + 64) != hipSuccess) {
            ffhip_set_error("ffhip_aac_imdct_and_windowing: %zu bytes of work space not available", n * per);
            return FFHIP_ENOMEM;
        }
        c->work_frames = n;
    }
    float *buf = (float *)c->work, *tail = buf + c->work_frames * 1024, *sorted = tail + c->work_frames * 512;
    int *dpos = (int *)(sorted + c->work_frames * 1024);
    uint8_t *dinfo = (uint8_t *)(dpos + c->work_frames);
    /* (pageable sources: the copies have been staged by the time the calls return, the vectors may go) */
    if (hipMemcpyAsync(dinfo, info.data(), n, hipMemcpyHostToDevice, st) != hipSuccess)
        return FFHIP_EINVAL;
    if (sort) {
        std::vector<int> pos(n);
        size_t il = 0, is = nlong;
        for (size_t i = 0; i < n; i++)
            pos[i] = (int)(window_sequence[i] != AAC_EIGHT_SHORT ? il++ : is++);
        if (hipMemcpyAsync(dpos, pos.data(), n * sizeof(int), hipMemcpyHostToDevice, st) != hipSuccess)
            return FFHIP_EINVAL;
        hipLaunchKernelGGL(k_aac_gather, dim3((unsigned)n), dim3(256), 0, st, coeffs, dpos, dinfo, sorted, L, c->in_short);
        int r = nlong ? ffhip_tx_batch_dev(c->tx1024, buf, (size_t)L * 4, sorted, (size_t)L * 4, sizeof(float), (int)nlong, stream) : 0;
        if (r >= 0 && n > nlong)
            r = ffhip_tx_batch_dev(c->tx128, buf + nlong * L, (size_t)S * 4, sorted + nlong * L, (size_t)S * 4, sizeof(float), (int)(n - nlong) * 8,
                                   stream);
        if (r < 0)
            return r;
    } else {
        dpos = nullptr;
        for (size_t i = 0; i < n;) {
            const bool is_short = window_sequence[i] == AAC_EIGHT_SHORT;
            size_t j = i + 1;
            while (j < n && (window_sequence[j] == AAC_EIGHT_SHORT) == is_short)
                j++;
            const int r = is_short ? ffhip_tx_batch_dev(c->tx128, buf + i * 1024, 512, coeffs + i * 1024, 512, sizeof(float), (int)(j - i) * 8, stream)
                                   : ffhip_tx_batch_dev(c->tx1024, buf + i * 1024, 4096, coeffs + i * 1024, 4096, sizeof(float), (int)(j - i), stream);
            if (r < 0)
                return r;
            i = j;
        }
    }
    hipLaunchKernelGGL(k_aac_tail, dim3((unsigned)n), dim3(128), 0, st, buf, dpos, dinfo, c->win, tail, (int)(n - nch), L);
    hipLaunchKernelGGL(k_aac_out, dim3((unsigned)n), dim3(256), 0, st, buf, dpos, dinfo, c->win, tail, saved, nch, out, L);
    LAUNCH_CHECK();
    if (hipMemcpy2DAsync(saved, 512 * sizeof(float), tail + (n - nch) * 512, 512 * sizeof(float), (size_t)(L / 2) * sizeof(float), (size_t)nch,
                         hipMemcpyDeviceToDevice, st) != hipSuccess)
        return FFHIP_EINVAL;
    c->last_n = n;
    c->last_nch = nch;
    c->last_sorted = sort;
    return 0;
}

/* One channel, one frame, host pointers: what a libavcodec/hip/aacdec_init.c wrapper of the reference member calls with
 * sce->coeffs, ics->window_sequence / use_kb_window, sce->saved, sce->output. */
extern "C" int ffhip_aac_imdct_and_windowing(FFHipAacImdct *c, const float *coeffs, const int window_sequence[2], const int use_kb_window[2],
                                             float *saved, float *out)
{
    FFHipDeviceGuard dg(c ? c->device : -1);
    if (!c || !coeffs || !window_sequence || !use_kb_window || !saved || !out)
        return FFHIP_EINVAL;
    std::lock_guard<std::mutex> lk(ffhip_scratch_mutex()); /* the arena is shared with every other host-pointer face */
    void *scratch;
    if (ffhip_scratch_reserve((1024 + 1024 + 512) * sizeof(float), &scratch) < 0)
        return FFHIP_ENOMEM;
    float *dco = (float *)scratch, *dout = dco + 1024, *dsv = dout + 1024;
    if (hipMemcpy(dco, coeffs, 1024 * sizeof(float), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(dsv, saved, (size_t)(c->L / 2) * sizeof(float), hipMemcpyHostToDevice) != hipSuccess)
        return FFHIP_EINVAL;
    const uint8_t seq = (uint8_t)window_sequence[0], kb = (uint8_t)use_kb_window[0], pseq = (uint8_t)window_sequence[1], pkb = (uint8_t)use_kb_window[1];
    const int r = ffhip_aac_imdct_and_windowing_batch_dev(c, dco, dout, dsv, &seq, &kb, &pseq, &pkb, 1, 1, nullptr);
    if (r < 0)
        return r;
    if (hipStreamSynchronize(nullptr) != hipSuccess || hipMemcpy(out, dout, (size_t)c->L * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(saved, dsv, (size_t)(c->L / 2) * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess)
        return FFHIP_EINVAL;
    return 0;
}

/* ---- AACDecDSP.apply_tns (aacdec_dsp_template.c:164-223) ------------------------------------------------------------------- */
static_assert(sizeof(FFHipAacTnsFilter) == 92, "FFHipAacTnsFilter is a 92-byte record");

/* The walk over windows and filters that turns the parsed TemporalNoiseShaping into filter ranges; host side (the fields come
 * out of the bitstream parser).  Returns the number of records written (<= 32). */
extern "C" int ffhip_aac_tns_filters(FFHipAacTnsFilter *out, int frame, const int n_filt[8], const int length[8][4], const int direction[8][4],
                                     const int order[8][4], const float coef[8][4][20], int num_windows, int num_swb,
                                     const uint16_t *swb_offset, int tns_max_bands, int max_sfb)
{
    if (!out || !n_filt || !length || !direction || !order || !coef || !swb_offset || num_windows < 1 || num_windows > 8)
        return FFHIP_EINVAL;
    const int lim = tns_max_bands < max_sfb ? tns_max_bands : max_sfb;
    int n = 0;
    if (!lim)
        return 0;
    for (int w = 0; w < num_windows; w++) {
        int lo = num_swb;
        for (int f = 0; f < n_filt[w] && f < 4; f++) {
            const int hi = lo;
            lo = hi > length[w][f] ? hi - length[w][f] : 0;
            const int ord = order[w][f];
            if (ord <= 0)
                continue;
            if (ord > 20) {
                ffhip_set_error("ffhip_aac_tns_filters: order %d above TNS_MAX_ORDER", ord);
                return FFHIP_EINVAL;
            }
            const int first = swb_offset[lo < lim ? lo : lim], last = swb_offset[hi < lim ? hi : lim];
            if (last <= first)
                continue;
            FFHipAacTnsFilter &r = out[n++];
            memset(&r, 0, sizeof(r));
            r.frame = frame;
            r.size = (int16_t)(last - first);
            r.inc = direction[w][f] ? -1 : 1;
            r.start = (int16_t)((direction[w][f] ? last - 1 : first) + w * 128);
            r.order = (uint8_t)ord;
            memcpy(r.coef, coef[w][f], ord * sizeof(float));
        }
    }
    return n;
}

/* One lane per filter: the recursion along frequency is serial, a batch has thousands of filters.  lpc[] and the history stay in
 * registers (everything is unrolled to TNS_MAX_ORDER with the order as a guard, so no array is indexed dynamically); products and
 * sums are separate operations in the reference's order.  The 64 filters of a wave live in 64 different frames, so a lane walking
 * its own range would touch a cache line per sample (measured: 2.9 GB of traffic for 0.37 GB of coefficients).  The ranges are
 * therefore moved in 64-sample chunks through LDS: the wave reads filter j's next 64 samples with one coalesced 256-byte access
 * (ascending or descending), lane l then filters row l, and the chunk goes back the same way. */
template <bool DECODE>
__global__ __launch_bounds__(64) void k_aac_tns(float *coeffs, const FFHipAacTnsFilter *filters, int n)
{
    __shared__ float chunk[64][65]; /* row pitch 65: lane l's walk along its row hits bank (l + k) % 32 */
    __shared__ long long s_off[64];
    __shared__ int s_inc[64], s_size[64];
    const int lane = threadIdx.x, g = blockIdx.x * 64 + lane;
    const FFHipAacTnsFilter *F = filters + (g < n ? g : 0);
    const int order = g < n ? F->order : 0, size = g < n ? F->size : 0;
    s_off[lane] = (long long)F->frame * 1024 + F->start;
    s_inc[lane] = F->inc;
    s_size[lane] = size;
    float lpc[20], hist[20];
    /* compute_lpc_coefs(coef, 0, order, lpc, 0, 0, 0, NULL), libavcodec/lpc_functions.h:54-103: the step-up recursion in place */
#pragma unroll
    for (int i = 0; i < 20; i++) {
        lpc[i] = 0.0f;
        hist[i] = 0.0f;
    }
#pragma unroll
    for (int i = 0; i < 20; i++) {
        if (i < order) {
            const float k = -F->coef[i];
            lpc[i] = k;
#pragma unroll
            for (int j = 0; j < (i + 1) >> 1; j++) {
                const float f = lpc[j], b = lpc[i - 1 - j];
                lpc[j] = f + k * b;
                lpc[i - 1 - j] = b + k * f;
            }
        }
    }
    int longest = size;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        const int o = __shfl_xor(longest, d);
        longest = o > longest ? o : longest;
    }
    __syncthreads();
    for (int c = 0; c < longest; c += 64) {
        for (int j = 0; j < 64; j++)
            if (c + lane < s_size[j])
                chunk[j][lane] = coeffs[s_off[j] + (long long)(c + lane) * s_inc[j]];
        __syncthreads();
        const int end = size - c < 64 ? size - c : 64;
        for (int k = 0; k < end; k++) {
            const int m = c + k, lim = m < order ? m : order;
            const float in = chunk[lane][k];
            float x = in;
#pragma unroll
            for (int i = 0; i < 20; i++)
                if (i < lim)
                    x = DECODE ? x - hist[i] * lpc[i] : x + hist[i] * lpc[i];
            chunk[lane][k] = x;
#pragma unroll
            for (int i = 19; i > 0; i--)
                hist[i] = hist[i - 1];
            hist[0] = DECODE ? x : in; /* the all-pole filter feeds back its outputs, the moving average remembers its inputs */
        }
        __syncthreads();
        for (int j = 0; j < 64; j++)
            if (c + lane < s_size[j])
                coeffs[s_off[j] + (long long)(c + lane) * s_inc[j]] = chunk[j][lane];
        __syncthreads();
    }
}

extern "C" int ffhip_aac_apply_tns_batch_dev(float *coeffs, const FFHipAacTnsFilter *filters, int nfilters, int decode, void *stream)
{
    if (!coeffs || !filters || nfilters < 0)
        return FFHIP_EINVAL;
    if (!ffhip_have_device())
        return FFHIP_ENOSYS;
    if (!nfilters)
        return 0;
    if (decode)
        hipLaunchKernelGGL(k_aac_tns<true>, dim3(cdiv(nfilters, 64)), dim3(64), 0, (hipStream_t)stream, coeffs, filters, nfilters);
    else
        hipLaunchKernelGGL(k_aac_tns<false>, dim3(cdiv(nfilters, 64)), dim3(64), 0, (hipStream_t)stream, coeffs, filters, nfilters);
    LAUNCH_CHECK();
    return 0;
}


/* ---- AACDecDSP.apply_mid_side_stereo / apply_intensity_stereo (aacdec_dsp_template.c:83-160) and apply_ltp's final add
 *      (:276-280): band ranges ------------------------------------------------------------------------------------------------ */
static_assert(sizeof(FFHipAacBandOp) == 20, "FFHipAacBandOp is a 20-byte record");
enum { AAC_NOISE_BT = 13, AAC_INTENSITY_BT2 = 14, AAC_INTENSITY_BT = 15 }; /* enum BandType, libavcodec/aac.h:66-78 */

static FFHipAacBandOp band_op(int kind, int frame0, int frame1, int start, int len, float scale)
{
    FFHipAacBandOp r;
    memset(&r, 0, sizeof(r));
    r.frame0 = frame0;
    r.frame1 = frame1;
    r.start = (int16_t)start;
    r.len = (int16_t)len;
    r.scale = scale;
    r.kind = (uint8_t)kind;
    return r;
}

static bool aac_groups_ok(int num_window_groups, const uint8_t *group_len, int max_sfb, const char *who)
{
    int windows = 0;
    for (int g = 0; g < num_window_groups && g < 8; g++)
        windows += group_len[g];
    if (num_window_groups < 1 || num_window_groups > 8 || windows > 8 || max_sfb < 0 || num_window_groups * max_sfb > 128) {
        ffhip_set_error("%s: %d window groups of %d windows, max_sfb %d", who, num_window_groups, windows, max_sfb);
        return false;
    }
    return true;
}

/* the walk of apply_mid_side_stereo; consecutive bands of a window that all qualify become one range (butterflies on adjacent
 * ranges are butterflies on their union).  At most 64 records (8 windows x (max_sfb + 1) / 2 runs). */
extern "C" int ffhip_aac_ms_bands(FFHipAacBandOp *out, int frame0, int frame1, int num_window_groups, const uint8_t *group_len, int max_sfb_ste,
                                  const uint8_t *ms_mask, const int *band_type0, const int *band_type1, const uint16_t *swb_offset)
{
    if (!out || !group_len || !ms_mask || !band_type0 || !band_type1 || !swb_offset ||
        !aac_groups_ok(num_window_groups, group_len, max_sfb_ste, "ffhip_aac_ms_bands"))
        return FFHIP_EINVAL;
    int n = 0, window0 = 0;
    for (int g = 0; g < num_window_groups; g++) {
        for (int sfb = 0; sfb < max_sfb_ste;) {
            auto on = [&](int b) {
                const int idx = g * max_sfb_ste + b;
                return ms_mask[idx] && band_type0[idx] < AAC_NOISE_BT && band_type1[idx] < AAC_NOISE_BT;
            };
            if (!on(sfb)) {
                sfb++;
                continue;
            }
            int end = sfb + 1;
            while (end < max_sfb_ste && on(end))
                end++;
            for (int w = 0; w < group_len[g]; w++)
                out[n++] = band_op(FFHIP_AAC_BAND_MS, frame0, frame1, (window0 + w) * 128 + swb_offset[sfb], swb_offset[end] - swb_offset[sfb], 0.0f);
            sfb = end;
        }
        window0 += group_len[g];
    }
    return n;
}

/* the walk of apply_intensity_stereo: one record per intensity band and window (<= 128) */
extern "C" int ffhip_aac_is_bands(FFHipAacBandOp *out, int frame0, int frame1, int num_window_groups, const uint8_t *group_len, int max_sfb,
                                  int ms_present, const uint8_t *ms_mask, const int *band_type1, const float *sf1, const uint16_t *swb_offset)
{
    if (!out || !group_len || !ms_mask || !band_type1 || !sf1 || !swb_offset ||
        !aac_groups_ok(num_window_groups, group_len, max_sfb, "ffhip_aac_is_bands"))
        return FFHIP_EINVAL;
    int n = 0, window0 = 0;
    for (int g = 0; g < num_window_groups; g++) {
        for (int sfb = 0; sfb < max_sfb; sfb++) {
            const int idx = g * max_sfb + sfb;
            if (band_type1[idx] != AAC_INTENSITY_BT && band_type1[idx] != AAC_INTENSITY_BT2)
                continue;
            int c = -1 + 2 * (band_type1[idx] - 14);
            if (ms_present)
                c *= 1 - 2 * ms_mask[idx];
            const float scale = c * sf1[idx];
            for (int w = 0; w < group_len[g]; w++)
                out[n++] = band_op(FFHIP_AAC_BAND_INTENSITY, frame0, frame1, (window0 + w) * 128 + swb_offset[sfb],
                                   swb_offset[sfb + 1] - swb_offset[sfb], scale);
        }
        window0 += group_len[g];
    }
    return n;
}

/* apply_ltp's last loop: coeffs[frame] += predFreq[pred_frame] on the used bands below min(max_sfb, MAX_LTP_LONG_SFB); runs of
 * used bands merged (<= 20 records) */
extern "C" int ffhip_aac_ltp_bands(FFHipAacBandOp *out, int frame, int pred_frame, int max_sfb, const int8_t *used, const uint16_t *swb_offset)
{
    if (!out || !used || !swb_offset || max_sfb < 0)
        return FFHIP_EINVAL;
    const int lim = max_sfb < 40 ? max_sfb : 40;
    int n = 0;
    for (int sfb = 0; sfb < lim;) {
        if (!used[sfb]) {
            sfb++;
            continue;
        }
        int end = sfb + 1;
        while (end < lim && used[end])
            end++;
        out[n++] = band_op(FFHIP_AAC_BAND_ADD, frame, pred_frame, swb_offset[sfb], swb_offset[end] - swb_offset[sfb], 0.0f);
        sfb = end;
    }
    return n;
}

/* one wave per record; each element is the reference's own one or two float operations */
__global__ __launch_bounds__(256) void k_aac_band_ops(float *a, float *b, const FFHipAacBandOp *ops, int n)
{
    const int o = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (o >= n)
        return;
    const FFHipAacBandOp op = ops[o];
    float *pa = a + (size_t)op.frame0 * 1024 + op.start, *pb = b + (size_t)op.frame1 * 1024 + op.start;
    for (int i = lane; i < op.len; i += 64) {
        const float x = pa[i], y = pb[i];
        if (op.kind == FFHIP_AAC_BAND_MS) {
            pa[i] = x + y;
            pb[i] = x - y;
        } else if (op.kind == FFHIP_AAC_BAND_INTENSITY) {
            pb[i] = x * op.scale;
        } else if (op.kind == FFHIP_AAC_BAND_FMAC) {
            pa[i] = x + op.scale * y; /* product, then sum: two roundings, as the C reference's dest += gain * src */
        } else {
            pa[i] = x + y;
        }
    }
}

extern "C" int ffhip_aac_band_ops_batch_dev(float *a, float *b, const FFHipAacBandOp *ops, int n, void *stream)
{
    if (!a || !b || !ops || n < 0)
        return FFHIP_EINVAL;
    if (!ffhip_have_device())
        return FFHIP_ENOSYS;
    if (!n)
        return 0;
    hipLaunchKernelGGL(k_aac_band_ops, dim3(cdiv(n, 4)), dim3(256), 0, (hipStream_t)stream, a, b, ops, n);
    LAUNCH_CHECK();
    return 0;
}

/* ---- AACDecDSP.apply_ltp's prediction (aacdec_dsp_template.c:252-272 with windowing_and_mdct_ltp, :225-247) ------------------- */
static_assert(sizeof(FFHipAacLtp) == 16, "FFHipAacLtp is a 16-byte record");

/* in[r][0..2047]: the delayed, scaled, windowed state; one thread = 4 samples (every region border is a multiple of 4) */
__global__ __launch_bounds__(256) void k_aac_ltp_window(const float *ltp_state, const FFHipAacLtp *recs, const float *win, float *in)
{
    const FFHipAacLtp r = recs[blockIdx.x];
    const float *st = ltp_state + (size_t)r.state * 3072;
    const int kb0 = r.kb & 1, kb1 = (r.kb >> 1) & 1;
    const float *lwindow = win + (kb0 ? 2 : 0) * 1024, *swindow = win + (kb0 ? 3 : 1) * 1024;
    const float *lwindow_prev = win + (kb1 ? 2 : 0) * 1024, *swindow_prev = win + (kb1 ? 3 : 1) * 1024;
    const int num_samples = r.lag < 1024 ? r.lag + 1024 : 2048;
    float *dst = in + (size_t)blockIdx.x * 2048;
    for (int s = 4 * threadIdx.x; s < 2048; s += 1024) {
        float v[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int i = s + k;
            float x = i < num_samples ? st[i + 2048 - r.lag] * r.coef : 0.0f;
            if (i < 1024) {
                if (r.seq0 != AAC_LONG_STOP)
                    x = x * lwindow_prev[i];
                else if (i < 448)
                    x = 0.0f;
                else if (i < 576)
                    x = x * swindow_prev[i - 448];
            } else {
                const int j = i - 1024;
                if (r.seq0 != AAC_LONG_START)
                    x = x * lwindow[1023 - j];
                else if (j >= 576)
                    x = 0.0f;
                else if (j >= 448)
                    x = x * swindow[127 - (j - 448)];
            }
            v[k] = x;
        }
        *reinterpret_cast<float4 *>(dst + s) = make_float4(v[0], v[1], v[2], v[3]);
    }
}

extern "C" int ffhip_aac_ltp_init(FFHipAacImdct *c, float scale_ltp)
{
    FFHipDeviceGuard dg(c ? c->device : -1);
    if (!c)
        return FFHIP_EINVAL;
    std::lock_guard<std::mutex> lk(c->mu);
    ffhip_tx_uninit(&c->tx_ltp);
    return ffhip_tx_init(&c->tx_ltp, nullptr, FFHIP_TX_FLOAT_MDCT, 0, 1024, &scale_ltp, FFHIP_TX_BITEXACT);
}

extern "C" int ffhip_aac_ltp_predict_batch_dev(FFHipAacImdct *c, const float *ltp_state, float *pred_freq, const FFHipAacLtp *recs, int n,
                                               void *stream)
{
    FFHipDeviceGuard dg(c ? c->device : -1);
    if (!c || !ltp_state || !pred_freq || !recs || n < 0)
        return FFHIP_EINVAL;
    if (!n)
        return 0;
    std::lock_guard<std::mutex> lk(c->mu);
    if (!c->tx_ltp) {
        ffhip_set_error("ffhip_aac_ltp_predict: ffhip_aac_ltp_init() has not created the forward MDCT");
        return FFHIP_EINVAL;
    }
    hipStream_t st = (hipStream_t)stream;
    if ((size_t)n > c->ltp_frames) {
        if (c->ltp_in) {
            (void)hipStreamSynchronize(st);
            (void)hipFree(c->ltp_in);
        }
        c->ltp_in = nullptr;
        c->ltp_frames = 0;
        if (hipMalloc(&c->ltp_in, (size_t)n * 2048 * sizeof(float)) != hipSuccess) {
            ffhip_set_error("ffhip_aac_ltp_predict: %zu bytes of work space not available", (size_t)n * 2048 * sizeof(float));
            return FFHIP_ENOMEM;
        }
        c->ltp_frames = n;
    }
    hipLaunchKernelGGL(k_aac_ltp_window, dim3(n), dim3(256), 0, st, ltp_state, recs, c->win, c->ltp_in);
    LAUNCH_CHECK();
    return ffhip_tx_batch_dev(c->tx_ltp, pred_freq, 4096, c->ltp_in, 8192, sizeof(float), n, stream);
}

/* ---- AACDecDSP.update_ltp (aacdec_dsp_template.c:287-320) on the frames the last imdct_and_windowing_batch_dev call ended with:
 *      their inverse-MDCT output (ac->buf_mdct) and overlap state are still in the context's work space -------------------------- */
__global__ __launch_bounds__(256) void k_aac_update_ltp(const float *buf, const int *pos, const uint8_t *info, const float *win, const float *tail,
                                                        int first, const float *out, float *ltp_state)
{
    const int ch = blockIdx.x, f = first + ch;
    const float *b = buf + (size_t)(pos ? pos[f] : f) * 1024, *sv = tail + (size_t)f * 512, *o = out + (size_t)ch * 1024;
    float *st = ltp_state + (size_t)ch * 3072;
    const int in = info[f], seq = in & 3, kb = (in >> 2) & 1;
    const float *lwindow = win + (kb ? 2 : 0) * 1024, *swindow = win + (kb ? 3 : 1) * 1024;
    const int s = 4 * threadIdx.x;
    float v[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int i = s + k;
        float x;
        if (seq == AAC_EIGHT_SHORT || seq == AAC_LONG_START) {
            if (i < 448)
                x = seq == AAC_EIGHT_SHORT ? sv[i] : b[512 + i];
            else if (i < 512)
                x = b[960 + (i - 448)] * swindow[64 + 63 - (i - 448)];
            else if (i < 576)
                x = b[1023 - (i - 512)] * swindow[63 - (i - 512)];
            else
                x = 0.0f;
        } else {
            x = i < 512 ? b[512 + i] * lwindow[512 + 511 - i] : b[1023 - (i - 512)] * lwindow[511 - (i - 512)];
        }
        v[k] = x;
    }
    const float4 mid = aac_ld4(st + 1024 + s), now = aac_ld4(o + s);
    *reinterpret_cast<float4 *>(st + s) = mid;
    *reinterpret_cast<float4 *>(st + 1024 + s) = now;
    *reinterpret_cast<float4 *>(st + 2048 + s) = make_float4(v[0], v[1], v[2], v[3]);
}

extern "C" int ffhip_aac_update_ltp_batch_dev(FFHipAacImdct *c, float *ltp_state, const float *out, int nch, void *stream)
{
    FFHipDeviceGuard dg(c ? c->device : -1);
    if (!c || !ltp_state || !out || nch <= 0)
        return FFHIP_EINVAL;
    if (((uintptr_t)ltp_state | (uintptr_t)out) & 15) {
        ffhip_set_error("ffhip_aac_update_ltp: ltp_state and out must be 16-byte aligned");
        return FFHIP_EINVAL;
    }
    std::lock_guard<std::mutex> lk(c->mu);
    if (c->L != 1024) {
        ffhip_set_error("ffhip_aac_update_ltp: long-term prediction runs on 1024-sample frames only (as apply_ltp / update_ltp do)");
        return FFHIP_EINVAL;
    }
    if (!c->last_n || c->last_nch != nch) {
        ffhip_set_error("ffhip_aac_update_ltp: follows an imdct_and_windowing_batch_dev call of the same %d channels on this context", nch);
        return FFHIP_EINVAL;
    }
    const float *buf = (const float *)c->work, *tail = buf + c->work_frames * 1024, *sorted = tail + c->work_frames * 512;
    const int *dpos = (const int *)(sorted + c->work_frames * 1024);
    const uint8_t *dinfo = (const uint8_t *)(dpos + c->work_frames);
    hipLaunchKernelGGL(k_aac_update_ltp, dim3(nch), dim3(256), 0, (hipStream_t)stream, buf, c->last_sorted ? dpos : nullptr, dinfo, c->win, tail,
                       (int)(c->last_n - nch), out, ltp_state);
    LAUNCH_CHECK();
    return 0;
}


/* ---- AAC-LD / AAC-ELD: AACDecDSP.imdct_and_windowing_ld / _eld (aacdec_dsp_template.c:516-602), float ------------------------------
 * One transform size per stream and no window sequences.  As with the 1024-sample member, what a frame leaves behind depends on
 * that frame's inverse MDCT alone (LD: its upper half; ELD: the whole of it, kept for three frames), so a run of frames is an MDCT
 * batch and one windowing pass over all frames; frames before the run come from the caller's `saved`. */
struct FFHipAacLd {
    int device = 0;
    FFHipTXContext *tx = nullptr;
    int eld = 0, n = 512;
    float *win = nullptr;     /* LD: sine_512 then sine_128; ELD: the 3.75 n window */
    float *work = nullptr;    /* buf [frames][n], ELD: + shuffled coefficients [frames][n], + new saved [nch][3 n] */
    uint8_t *info = nullptr;  /* LD: use_kb_window[1] per frame */
    size_t work_floats = 0, info_bytes = 0;
    std::mutex mu;
};

extern "C" void ffhip_aac_ld_free(FFHipAacLd **pc)
{
    FFHipDeviceGuard dg(pc && *pc ? (*pc)->device : -1);
    if (!pc || !*pc)
        return;
    FFHipAacLd *c = *pc;
    ffhip_tx_uninit(&c->tx);
    if (c->win) (void)hipFree(c->win);
    if (c->work) (void)hipFree(c->work);
    if (c->info) (void)hipFree(c->info);
    delete c;
    *pc = nullptr;
}

extern "C" int ffhip_aac_ld_create(FFHipAacLd **pc, int eld, int frame_len, const float *w0, const float *w1, float scale)
{
    if (!pc || !w0 || (!eld && !w1))
        return FFHIP_EINVAL;
    *pc = nullptr;
    if (!(frame_len == 512 || (eld && frame_len == 480))) {
        ffhip_set_error("ffhip_aac_ld_create: frame length %d (LD: 512; ELD: 512, 480)", frame_len);
        return FFHIP_EINVAL;
    }
    if (!ffhip_have_device())
        return FFHIP_ENOSYS;
    FFHipAacLd *c = new (std::nothrow) FFHipAacLd();
    if (c)
        c->device = ffhip_current_device();
    if (!c)
        return FFHIP_ENOMEM;
    c->eld = !!eld;
    c->n = frame_len;
    int r = ffhip_tx_init(&c->tx, nullptr, FFHIP_TX_FLOAT_MDCT, 1, frame_len, &scale, FFHIP_TX_BITEXACT);
    const size_t nw = eld ? (size_t)frame_len * 15 / 4 : 512 + 128;
    std::vector<float> w(nw);
    if (eld) {
        memcpy(w.data(), w0, nw * sizeof(float));
    } else {
        memcpy(w.data(), w0, 512 * sizeof(float));
        memcpy(w.data() + 512, w1, 128 * sizeof(float));
    }
    if (r >= 0 && (hipMalloc(&c->win, nw * sizeof(float)) != hipSuccess ||
                   hipMemcpy(c->win, w.data(), nw * sizeof(float), hipMemcpyHostToDevice) != hipSuccess)) {
        ffhip_set_error("ffhip_aac_ld_create: window upload failed");
        r = FFHIP_ENOMEM;
    }
    if (r < 0) {
        ffhip_aac_ld_free(&c);
        return r;
    }
    *pc = c;
    return 0;
}

/* LD: one thread = 4 samples of a frame's 512 */
__global__ __launch_bounds__(128) void k_aac_ld_out(const float *buf, const uint8_t *kbp, const float *win, const float *saved, int nch, float *out)
{
    const int f = blockIdx.x, o = 4 * threadIdx.x;
    const float *b = buf + (size_t)f * 512;
    const float *sv = f < nch ? saved + (size_t)f * 256 : buf + (size_t)(f - nch) * 512 + 256;
    float4 v;
    if (!kbp[f])
        v = aac_wov4(sv, b, win, 256, o);
    else if (o < 192)
        v = aac_ld4(sv + o);
    else if (o < 320)
        v = aac_wov4(sv + 192, b, win + 512, 64, o - 192); /* the low-overlap window: sine_128 across the middle */
    else
        v = aac_ld4(b + o - 256);
    *reinterpret_cast<float4 *>(out + (size_t)f * 1024 + o) = v;
}

/* ELD: the coefficient shuffle in front of the transform is new[k] = (k even ? -1 : +1) * old[n - 1 - k] (the reference's four
 * in-place swaps, aacdec_dsp_template.c:561-565, written out per element) */
__global__ __launch_bounds__(512) void k_aac_eld_shuffle(const float *coeffs, float *shuf, int n)
{
    const int f = blockIdx.x, k = threadIdx.x;
    if (k >= n)
        return;
    const float v = coeffs[(size_t)f * 1024 + n - 1 - k];
    shuf[(size_t)f * n + k] = (k & 1) ? v : -v;
}

/* history sample j of the frame `back` frames before frame f of its channel (back = 0: this frame), signs as the reference leaves
 * them in buf (-, +, -, +, ...): from the batch's transforms, or from the caller's saved (newest first) before the batch */
__device__ __forceinline__ float eld_hist(const float *buf, const float *saved, int n, int nch, int f, int back, int j)
{
    const int t = f / nch, ch = f - t * nch;
    if (t - back >= 0) {
        const float v = buf[(size_t)(f - back * nch) * n + j];
        return (j & 1) ? v : -v;
    }
    return saved[(size_t)ch * 3 * n + (size_t)(back - t - 1) * n + j];
}

__global__ __launch_bounds__(512) void k_aac_eld_out(const float *buf, const float *saved, const float *w, int n, int nch, float *out)
{
    const int f = blockIdx.x, o = threadIdx.x;
    if (o >= n)
        return;
    const int n2 = n >> 1, n4 = n >> 2;
    auto B = [&](int j) { return eld_hist(buf, saved, n, nch, f, 0, j); };
    auto S = [&](int idx) { return eld_hist(buf, saved, n, nch, f, 1 + idx / n, idx % n); }; /* the reference's saved[idx] */
    float v;
    if (o < n4) {
        const int i = o + n4;
        v = B(n2 - 1 - i) * w[i - n4] + S(i + n2) * w[i + n - n4] + -S(n + n2 - 1 - i) * w[i + 2 * n - n4] + -S(2 * n + n2 + i) * w[i + 3 * n - n4];
    } else if (o < n4 + n2) {
        const int i = o - n4;
        v = B(i) * w[i + n2 - n4] + -S(n - 1 - i) * w[i + n2 + n - n4] + -S(n + i) * w[i + n2 + 2 * n - n4] + S(2 * n + n - 1 - i) * w[i + n2 + 3 * n - n4];
    } else {
        const int i = o - n2 - n4;
        v = B(i + n2) * w[i + n - n4] + -S(n2 - 1 - i) * w[i + 2 * n - n4] + -S(n + n2 + i) * w[i + 3 * n - n4];
    }
    out[(size_t)f * 1024 + o] = v;
}

/* the history the batch leaves behind: slot k (newest first) = the frame k + 1 before the end */
__global__ __launch_bounds__(512) void k_aac_eld_save(const float *buf, const float *saved, int n, int nch, int nframes, float *nsaved)
{
    const int ch = blockIdx.x, k = blockIdx.y, j = threadIdx.x;
    if (j >= n)
        return;
    /* as seen from a (virtual) frame at time nframes: back = k + 1 */
    nsaved[(size_t)ch * 3 * n + (size_t)k * n + j] = eld_hist(buf, saved, n, nch, nframes * nch + ch, k + 1, j);
}

extern "C" int ffhip_aac_ld_batch_dev(FFHipAacLd *c, const float *coeffs, float *out, float *saved, const uint8_t *kb_prev, int nch, int nframes,
                                      void *stream)
{
    FFHipDeviceGuard dg(c ? c->device : -1);
    if (!c || !coeffs || !out || !saved || nch <= 0 || nframes < 0 || (!c->eld && !kb_prev))
        return FFHIP_EINVAL;
    const size_t nf = (size_t)nch * nframes;
    if (!nf)
        return 0;
    if (((uintptr_t)coeffs | (uintptr_t)out | (uintptr_t)saved) & 15) {
        ffhip_set_error("ffhip_aac_ld: coeffs, out and saved must be 16-byte aligned");
        return FFHIP_EINVAL;
    }
    std::lock_guard<std::mutex> lk(c->mu);
    hipStream_t st = (hipStream_t)stream;
    const int n = c->n;
    const size_t need = nf * n * (c->eld ? 2 : 1) + (c->eld ? (size_t)nch * 3 * n : 0);
    if (need > c->work_floats) {
        if (c->work) {
            (void)hipStreamSynchronize(st);
            (void)hipFree(c->work);
        }
        c->work = nullptr;
        c->work_floats = 0;
        if (hipMalloc(&c->work, need * sizeof(float)) != hipSuccess) {
            ffhip_set_error("ffhip_aac_ld: %zu bytes of work space not available", need * sizeof(float));
            return FFHIP_ENOMEM;
        }
        c->work_floats = need;
    }
    float *buf = c->work;
    if (!c->eld) {
        if (nf > c->info_bytes) {
            if (c->info) {
                (void)hipStreamSynchronize(st);
                (void)hipFree(c->info);
            }
            c->info = nullptr;
            c->info_bytes = 0;
            if (hipMalloc(&c->info, nf) != hipSuccess)
                return FFHIP_ENOMEM;
            c->info_bytes = nf;
        }
        if (hipMemcpyAsync(c->info, kb_prev, nf, hipMemcpyHostToDevice, st) != hipSuccess)
            return FFHIP_EINVAL;
        const int r = ffhip_tx_batch_dev(c->tx, buf, 2048, coeffs, 4096, sizeof(float), (int)nf, stream);
        if (r < 0)
            return r;
        hipLaunchKernelGGL(k_aac_ld_out, dim3((unsigned)nf), dim3(128), 0, st, buf, c->info, c->win, saved, nch, out);
        LAUNCH_CHECK();
        if (hipMemcpy2DAsync(saved, 256 * sizeof(float), buf + (nf - nch) * 512 + 256, 512 * sizeof(float), 256 * sizeof(float), (size_t)nch,
                             hipMemcpyDeviceToDevice, st) != hipSuccess)
            return FFHIP_EINVAL;
        return 0;
    }
    float *shuf = buf + nf * n, *nsaved = shuf + nf * n;
    hipLaunchKernelGGL(k_aac_eld_shuffle, dim3((unsigned)nf), dim3(512), 0, st, coeffs, shuf, n);
    const int r = ffhip_tx_batch_dev(c->tx, buf, (size_t)n * 4, shuf, (size_t)n * 4, sizeof(float), (int)nf, stream);
    if (r < 0)
        return r;
    hipLaunchKernelGGL(k_aac_eld_out, dim3((unsigned)nf), dim3(512), 0, st, buf, saved, c->win, n, nch, out);
    hipLaunchKernelGGL(k_aac_eld_save, dim3(nch, 3), dim3(512), 0, st, buf, saved, n, nch, nframes, nsaved);
    LAUNCH_CHECK();
    if (hipMemcpyAsync(saved, nsaved, (size_t)nch * 3 * n * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess)
        return FFHIP_EINVAL;
    return 0;
}


/* AACDecDSP.apply_dependent_coupling's walk (aacdec_float_coupling.h:42-71): one FMAC record per coupled band and window (<= 128);
 * gain = cce->coup.gain[index] (120 floats), band_type = cce->ch[0].band_type (ZERO_BT = 0 bands are skipped) */
extern "C" int ffhip_aac_coupling_bands(FFHipAacBandOp *out, int dest_frame, int src_frame, int num_window_groups, const uint8_t *group_len,
                                        int max_sfb, const int *band_type, const float *gain, const uint16_t *swb_offset)
{
    if (!out || !group_len || !band_type || !gain || !swb_offset || !aac_groups_ok(num_window_groups, group_len, max_sfb, "ffhip_aac_coupling_bands") ||
        num_window_groups * max_sfb > 120)
        return FFHIP_EINVAL;
    int n = 0, window0 = 0, idx = 0;
    for (int g = 0; g < num_window_groups; g++) {
        for (int i = 0; i < max_sfb; i++, idx++)
            if (band_type[idx] != 0)
                for (int w = 0; w < group_len[g]; w++)
                    out[n++] = band_op(FFHIP_AAC_BAND_FMAC, dest_frame, src_frame, (window0 + w) * 128 + swb_offset[i], swb_offset[i + 1] - swb_offset[i],
                                       gain[idx]);
        window0 += group_len[g];
    }
    return n;
}

/* ---- AACDecDSP.apply_prediction (AAC Main; aacdec_dsp_template.c:636-664, predict(): aacdec_float_prediction.h:35-85) -------------- */
static_assert(sizeof(FFHipAacPrediction) == 100, "FFHipAacPrediction is a 100-byte record");

extern "C" int ffhip_aac_prediction_record(FFHipAacPrediction *out, int channel, int frame, int is_long, int initialized, int predictor_present,
                                           const uint8_t *prediction_used, int pred_sfb_max, const uint16_t *swb_offset, int reset_group)
{
    if (!out || !prediction_used || !swb_offset || pred_sfb_max < 0 || pred_sfb_max > 41 || reset_group < 0 || reset_group > 30 ||
        swb_offset[pred_sfb_max] > 672) {
        ffhip_set_error("ffhip_aac_prediction_record: pred_sfb_max %d / reset group %d out of range", pred_sfb_max, reset_group);
        return FFHIP_EINVAL;
    }
    memset(out, 0, sizeof(*out));
    out->channel = channel;
    out->frame = frame;
    out->kmax = (int16_t)swb_offset[pred_sfb_max];
    out->flags = (uint8_t)((is_long ? FFHIP_AAC_PRED_LONG : 0) | (initialized ? 0 : FFHIP_AAC_PRED_RESET_FIRST));
    out->reset_group = (uint8_t)reset_group;
    if (predictor_present)
        for (int sfb = 0; sfb < pred_sfb_max; sfb++)
            if (prediction_used[sfb])
                for (int k = swb_offset[sfb]; k < swb_offset[sfb + 1]; k++)
                    out->enable[k >> 5] |= 1u << (k & 31);
    return 0;
}

__device__ __forceinline__ float pr_round(float f) { return __uint_as_float((__float_as_uint(f) + 0x00008000u) & 0xFFFF0000u); }
__device__ __forceinline__ float pr_even(float f)
{
    const uint32_t i = __float_as_uint(f);
    return __uint_as_float((i + 0x00007FFFu + (i & 1u)) & 0xFFFF0000u); /* the reference's `tmp.i & 0x00010000U >> 16` is tmp.i & 1 */
}
__device__ __forceinline__ float pr_trunc(float f) { return __uint_as_float(__float_as_uint(f) & 0xFFFF0000u); }

/* one thread per predictor: every coefficient below kmax has its own backward-adaptive state, nothing crosses coefficients */
__global__ __launch_bounds__(704) void k_aac_prediction(float *state, float *coeffs, const FFHipAacPrediction *recs)
{
    const FFHipAacPrediction &R = recs[blockIdx.x];
    const int k = threadIdx.x;
    if (k >= 672)
        return;
    float4 *sp = reinterpret_cast<float4 *>(state + ((size_t)R.channel * 672 + k) * 8);
    float4 A = sp[0];                       /* cor0 cor1 var0 var1 */
    float2 B = *reinterpret_cast<float2 *>(sp + 1); /* r0 r1 */
    auto reset = [&] { A = make_float4(0.0f, 0.0f, 1.0f, 1.0f); B = make_float2(0.0f, 0.0f); };
    if (R.flags & FFHIP_AAC_PRED_RESET_FIRST)
        reset();
    if (!(R.flags & FFHIP_AAC_PRED_LONG)) {
        reset();
    } else {
        if (k < R.kmax) {
            const float a = 0.953125f, alpha = 0.90625f;
            float *c = coeffs + (size_t)R.frame * 1024 + k;
            const float r0 = B.x, r1 = B.y, cor0 = A.x, cor1 = A.y, var0 = A.z, var1 = A.w;
            const float k1 = var0 > 1 ? cor0 * pr_even(a / var0) : 0;
            const float k2 = var1 > 1 ? cor1 * pr_even(a / var1) : 0;
            const float pv = pr_round(k1 * r0 + k2 * r1);
            float e0 = *c;
            if (R.enable[k >> 5] >> (k & 31) & 1u) {
                e0 = e0 + pv;
                *c = e0;
            }
            const float e1 = e0 - k1 * r0;
            A.y = pr_trunc(alpha * cor1 + r1 * e1);
            A.w = pr_trunc(alpha * var1 + 0.5f * (r1 * r1 + e1 * e1));
            A.x = pr_trunc(alpha * cor0 + r0 * e0);
            A.z = pr_trunc(alpha * var0 + 0.5f * (r0 * r0 + e0 * e0));
            B.y = pr_trunc(a * (r0 - k1 * e0));
            B.x = pr_trunc(a * e0);
        }
        if (R.reset_group && k % 30 == R.reset_group - 1)
            reset();
    }
    sp[0] = A;
    *reinterpret_cast<float2 *>(sp + 1) = B;
}

extern "C" int ffhip_aac_apply_prediction_batch_dev(float *predictor_state, float *coeffs, const FFHipAacPrediction *recs, int n, void *stream)
{
    if (!predictor_state || !coeffs || !recs || n < 0)
        return FFHIP_EINVAL;
    if ((uintptr_t)predictor_state & 15) {
        ffhip_set_error("ffhip_aac_apply_prediction: predictor_state must be 16-byte aligned");
        return FFHIP_EINVAL;
    }
    if (!ffhip_have_device())
        return FFHIP_ENOSYS;
    if (!n)
        return 0;
    hipLaunchKernelGGL(k_aac_prediction, dim3(n), dim3(704), 0, (hipStream_t)stream, predictor_state, coeffs, recs);
    LAUNCH_CHECK();
    return 0;
}
