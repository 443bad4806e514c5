// tools/ubench/mfma_i8_rate.hip — how fast does v_mfma_i32_16x16x64_i8 issue, (a) accumulating in place, (b) with C and D apart as the
// SATD search uses it, (c) the latter with four v_sad_u32 between two MFMAs.  One workgroup of 256 threads per CU x 4, long loop.
// hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_i8_rate tools/ubench/mfma_i8_rate.hip && /tmp/mfma_i8_rate
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef int i4 __attribute__((ext_vector_type(4)));
#define REP 64
template <int VAR>
__global__ __launch_bounds__(256) void k(const i4 *in, int *out, int iters)
{
    const int l = threadIdx.x & 63;
    i4 a = in[l], b = in[64 + l], c0 = in[128 + l], c1 = c0 + 1, c2 = c0 + 2, c3 = c0 + 3;
    uint32_t s = 0;
    const int kb = 1 << 20;
    for (int it = 0; it < iters; it++) {
        if (VAR == 0) {
            asm volatile("v_mfma_i32_16x16x64_i8 %0, %4, %5, %0\n\tv_mfma_i32_16x16x64_i8 %1, %4, %5, %1\n\t"
                         "v_mfma_i32_16x16x64_i8 %2, %4, %5, %2\n\tv_mfma_i32_16x16x64_i8 %3, %4, %5, %3\n\t"
                         : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a), "v"(b));
        } else if (VAR == 1) {
            asm volatile("v_mfma_i32_16x16x64_i8 v[112:115], %4, %5, %0\n\tv_mfma_i32_16x16x64_i8 v[116:119], %4, %5, %1\n\t"
                         "v_mfma_i32_16x16x64_i8 v[120:123], %4, %5, %2\n\tv_mfma_i32_16x16x64_i8 v[124:127], %4, %5, %3\n\t"
                         : : "v"(c0), "v"(c1), "v"(c2), "v"(c3), "v"(a), "v"(b)
                         : "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127");
        } else if (VAR == 2) {
#define SD(r) "v_sad_u32 %0, v" #r ", %7, %0\n\t"
            asm volatile("v_mfma_i32_16x16x64_i8 v[112:115], %5, %6, %1\n\t" SD(116) SD(117) SD(118) SD(119)
                         "v_mfma_i32_16x16x64_i8 v[116:119], %5, %6, %2\n\t" SD(120) SD(121) SD(122) SD(123)
                         "v_mfma_i32_16x16x64_i8 v[120:123], %5, %6, %3\n\t" SD(124) SD(125) SD(126) SD(127)
                         "v_mfma_i32_16x16x64_i8 v[124:127], %5, %6, %4\n\t" SD(112) SD(113) SD(114) SD(115)
                         : "+v"(s) : "v"(c0), "v"(c1), "v"(c2), "v"(c3), "v"(a), "v"(b), "s"(kb)
                         : "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127");
        } else {
            asm volatile(SD(116) SD(117) SD(118) SD(119) SD(120) SD(121) SD(122) SD(123) SD(124) SD(125) SD(126) SD(127) SD(112) SD(113) SD(114) SD(115)
                         : "+v"(s) : "v"(c0), "v"(c1), "v"(c2), "v"(c3), "v"(a), "v"(b), "s"(kb)
                         : "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127");
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = c0.x + c1.y + c2.z + c3.w + (int)s;
}
template <int VAR>
static void run(const char *name, const i4 *in, int *out, int wgs)
{
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<VAR>, dim3(wgs), dim3(256), 0, 0, in, out, 1000);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<VAR>, dim3(wgs), dim3(256), 0, 0, in, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    // per SIMD: (wgs * 4 waves / 1024 SIMDs) waves, each iters * 4 MFMAs
    const double per_simd = (double)wgs * 4 / 1024 * iters * 4;
    printf("%-34s wgs %5d: %.3f ms, %.2f ns per 4-instruction group slot per SIMD (= %.1f cycles at 2.4 GHz)\n", name, wgs, ms, ms * 1e6 / per_simd,
           ms * 1e6 / per_simd * 2.4);
}
int main()
{
    i4 *in; int *out;
    hipMalloc(&in, 4096); hipMemset(in, 1, 4096); hipMalloc(&out, 4096 * 256 * 4);
    for (int wgs : { 256, 1024 }) {
        run<0>("mfma, accumulate in place", in, out, wgs);
        run<1>("mfma, C and D apart", in, out, wgs);
        run<2>("mfma C/D apart + 4 v_sad_u32", in, out, wgs);
        run<3>("4 v_sad_u32 only (x4)", in, out, wgs);
    }
    return 0;
}
