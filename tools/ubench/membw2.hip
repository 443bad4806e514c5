// tools/ubench/membw2.hip — what a streaming kernel CAN reach on this box, by traffic mix and by how the accesses are issued (VERDICT r04
// weak #7 / next #5a: the library's probe k_membw issues one 16-byte access per loop iteration and copies at 4.7 TB/s where the guide
// measures 6.29 TB/s for a float4 copy).  Variants: U accesses of 16 B in flight per lane before the first dependent store (1, 2, 4, 8),
// plain / non-temporal stores, non-temporal loads, grids of 256 x {4, 8, 16} blocks of 256 threads, and the runtime's own hipMemcpyDtoDAsync.
// Mixes: copy (1 : 1), read-only, write-only, 1 : 2 (yuv420p -> rgb24: 1.5 B read, 3 B written per pixel), 1 : 4 (nv12 1080p -> 4K).
// Build: hipcc --offload-arch=gfx950 -O3 -o membw2 membw2.hip     Run: ./membw2 [GiB [reps [passes]]]    One JSON line per variant.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>

typedef uint32_t u4 __attribute__((ext_vector_type(4)));

// RD : WR = 1 : K over units of 16 B.  A lane-iteration reads U vectors and writes U * K; wave-instruction i of a block touches 4 KiB of
// consecutive memory (256 lanes x 16 B), the next instruction the next 4 KiB: exactly what a row-streaming kernel does.
// PAT 0: grid-stride (block b's iteration i touches chunk i * gridDim + b); PAT 1: every block streams through a private contiguous
// n / gridDim slice, slices numbered so that the blocks of one XCD (b % 8) hold one contiguous eighth of the buffer; PAT 2: private
// contiguous slices in launch order
template <int K, int U, int NTS, int NTL, int RD, int WR, int PAT = 0>
__global__ __launch_bounds__(256) void k_mix(const u4 *__restrict__ src, u4 *__restrict__ dst, size_t n_rd, uint32_t *sink)
{
    const size_t nthreads = (size_t)gridDim.x * 256;
    u4 acc = { 0, 0, 0, 0 };
    const size_t per_block = n_rd / gridDim.x / (256 * U) * (256 * U);
    const size_t bid = PAT == 1 ? (size_t)(blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3) : blockIdx.x;
    const size_t first = PAT ? bid * per_block : (size_t)blockIdx.x * 256 * U;
    const size_t last = PAT ? first + per_block : n_rd;
    const size_t step = PAT ? (size_t)256 * U : nthreads * U;
    for (size_t base = first; base + 256 * U <= last; base += step) {
        u4 v[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const size_t i = base + (size_t)u * 256 + threadIdx.x;
            if (RD)
                v[u] = NTL ? __builtin_nontemporal_load(src + i) : src[i];
            else
                v[u] = u4{ (uint32_t)i, 1, 2, 3 };
        }
        if (WR) {
#pragma unroll
            for (int u = 0; u < U; u++)
#pragma unroll
                for (int k = 0; k < K; k++) {
                    const size_t o = (base + (size_t)u * 256) * K + (size_t)k * 256 + threadIdx.x;
                    u4 w = v[u];
                    w.x += k;
                    if (NTS) __builtin_nontemporal_store(w, dst + o);
                    else dst[o] = w;
                }
        } else {
#pragma unroll
            for (int u = 0; u < U; u++)
                acc += v[u];
        }
    }
    if (!WR && acc.x + acc.y + acc.z + acc.w == 0x12345)
        sink[0] = 1;
}

static hipEvent_t e0, e1;
static int g_reps = 10, g_pass = 0;
template <int K, int U, int NTS, int NTL, int RD, int WR, int PAT = 0>
static void run(const char *mix, const u4 *s, u4 *d, size_t n_rd, int bpc, uint32_t *sink)
{
    const int blocks = 256 * bpc;
    const double moved = (double)n_rd * 16 * ((RD ? 1 : 0) + (WR ? K : 0));
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL((k_mix<K, U, NTS, NTL, RD, WR, PAT>), dim3(blocks), dim3(256), 0, 0, s, d, n_rd, sink);
    hipEventRecord(e0);
    const int reps = g_reps;
    for (int i = 0; i < reps; i++) hipLaunchKernelGGL((k_mix<K, U, NTS, NTL, RD, WR, PAT>), dim3(blocks), dim3(256), 0, 0, s, d, n_rd, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    printf("{\"pass\": %d, \"mix\": \"%s\", \"in_flight_16B\": %d, \"nt_store\": %d, \"nt_load\": %d, \"blocks_per_cu\": %d, \"pattern\": %d, \"ms\": %.4f, \"GB/s\": %.1f}\n", g_pass, mix, U, NTS, NTL, bpc,
           PAT, ms, moved / ms / 1e6);
    fflush(stdout);
}

#define SWEEP(K, RD, WR, name, n) do { \
    for (int b = 4; b <= 16; b *= 2) { \
        run<K, 1, 0, 0, RD, WR>(name, s, d, n, b, sink); run<K, 2, 0, 0, RD, WR>(name, s, d, n, b, sink); \
        run<K, 4, 0, 0, RD, WR>(name, s, d, n, b, sink); run<K, 8, 0, 0, RD, WR>(name, s, d, n, b, sink); } \
    run<K, 4, 1, 0, RD, WR>(name, s, d, n, 8, sink); run<K, 4, 1, 1, RD, WR>(name, s, d, n, 8, sink); run<K, 4, 0, 1, RD, WR>(name, s, d, n, 8, sink); \
    run<K, 2, 1, 0, RD, WR>(name, s, d, n, 8, sink); run<K, 8, 1, 1, RD, WR>(name, s, d, n, 16, sink); \
    for (int b = 4; b <= 16; b *= 2) { \
        run<K, 2, 0, 0, RD, WR, 1>(name, s, d, n, b, sink); run<K, 8, 0, 0, RD, WR, 1>(name, s, d, n, b, sink); \
        run<K, 2, 0, 0, RD, WR, 2>(name, s, d, n, b, sink); run<K, 8, 0, 0, RD, WR, 2>(name, s, d, n, b, sink); } } while (0)

int main(int argc, char **argv)
{
    const size_t gib = argc > 1 ? (size_t)atoi(argv[1]) : 2;
    const size_t bytes = gib << 30;    // of the LARGER side
    u4 *s, *d;
    uint32_t *sink;
    if (hipMalloc(&s, bytes) != hipSuccess || hipMalloc(&d, bytes) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) {
        fprintf(stderr, "hipMalloc failed\n");
        return 1;
    }
    hipMemset(s, 1, bytes);
    hipMemset(d, 2, bytes);
    hipEventCreate(&e0); hipEventCreate(&e1);
    const size_t n = bytes / 16;
    const int passes = argc > 3 ? atoi(argv[3]) : 1;
    if (argc > 2) g_reps = atoi(argv[2]);
    for (g_pass = 0; g_pass < passes; g_pass++) {
        SWEEP(1, 1, 1, "copy 1:1", n);
        SWEEP(1, 1, 0, "read", n);
        SWEEP(1, 0, 1, "write", n);
        SWEEP(2, 1, 1, "read1 write2", n / 2);
        SWEEP(4, 1, 1, "read1 write4", n / 4);
    }
    // the runtime's copy
    for (int i = 0; i < 3; i++) hipMemcpyDtoDAsync(d, s, bytes, 0);
    hipEventRecord(e0);
    for (int i = 0; i < 10; i++) hipMemcpyDtoDAsync(d, s, bytes, 0);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= 10;
    printf("{\"mix\": \"hipMemcpyDtoDAsync\", \"ms\": %.4f, \"GB/s\": %.1f}\n", ms, 2.0 * bytes / ms / 1e6);
    return 0;
}
