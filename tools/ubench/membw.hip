// tools/ubench/membw.hip — the achievable HBM roofs of this box by traffic mix: streaming read, streaming write, copy, and
// the scaler's 1 : 4 read : write mix, 16 bytes per lane, 1 KiB per wave-instruction, grid-stride over 256 x 4 x 8 waves.
// report the spec roof (8 TB/s) and the achievable one.  Build: hipcc --offload-arch=gfx950 -O3 -o membw membw.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>

typedef uint32_t u4 __attribute__((ext_vector_type(4)));

// MODE 0 read-only, 1 write-only, 2 copy, 3 read n/4 + write n (the 1080p -> 4K scaler's mix)
template <int MODE, int NT>
__global__ __launch_bounds__(256) void k_bw(const u4 *__restrict__ src, u4 *__restrict__ dst, size_t n16, uint32_t *sink)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    u4 acc = { 0, 0, 0, 0 };
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) {
        u4 v = { (uint32_t)i, 1, 2, 3 };
        if (MODE == 0 || MODE == 2)
            v = src[i];
        if (MODE == 3 && (i & 3) == 0)
            v = src[i >> 2];
        if (MODE == 0) {
            acc += v;
        } else {
            if (NT) __builtin_nontemporal_store(v, dst + i);
            else dst[i] = v;
        }
    }
    if (MODE == 0 && acc.x + acc.y + acc.z + acc.w == 0x12345)
        sink[0] = 1;
}

template <int MODE, int NT>
static void run(const char *name, const u4 *s, u4 *d, size_t bytes, double moved, uint32_t *sink)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * 8;
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL((k_bw<MODE, NT>), dim3(blocks), dim3(256), 0, 0, s, d, bytes / 16, sink);
    hipEventRecord(e0);
    const int reps = 20;
    for (int i = 0; i < reps; i++) hipLaunchKernelGGL((k_bw<MODE, NT>), dim3(blocks), dim3(256), 0, 0, s, d, bytes / 16, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    printf("{\"pattern\": \"%s\", \"bytes_moved\": %.0f, \"ms\": %.4f, \"GB/s\": %.1f}\n", name, moved, ms, moved / ms / 1e6);
}

int main(int argc, char **argv)
{
    const size_t bytes = (argc > 1 ? strtoull(argv[1], 0, 0) : 3072ull) << 20;
    u4 *a, *b;
    uint32_t *sink;
    if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&b, bytes) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess)
        return 1;
    hipMemset(a, 1, bytes); hipMemset(b, 2, bytes);
    run<0, 0>("read", a, b, bytes, (double)bytes, sink);
    run<1, 0>("write", a, b, bytes, (double)bytes, sink);
    run<1, 1>("write_nt", a, b, bytes, (double)bytes, sink);
    run<2, 0>("copy", a, b, bytes, 2.0 * bytes, sink);
    run<2, 1>("copy_nt", a, b, bytes, 2.0 * bytes, sink);
    run<3, 0>("read1_write4", a, b, bytes, 1.25 * bytes, sink);
    run<3, 1>("read1_write4_nt", a, b, bytes, 1.25 * bytes, sink);
    return 0;
}
