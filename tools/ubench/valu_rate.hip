// tools/ubench/valu_rate.hip — issue rate of the integer VALU instructions the DSP kernels lean on, gfx950.
// Each kernel runs N iterations of 16 INDEPENDENT chains of one instruction (so dependent-issue latency
// is hidden) on every SIMD of the chip with 4 waves/SIMD; result = wave-instructions per cycle per SIMD
// relative to v_add_u32.  Build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define CHAINS 16
#define ITERS 4096

#define KERNEL(NAME, ASM)                                                                         \
    __global__ __launch_bounds__(256) void k_##NAME(uint32_t *out, uint32_t seed)                 \
    {                                                                                             \
        uint32_t a[CHAINS], b = seed + threadIdx.x, c = seed * 3 + 1;                             \
        for (int i = 0; i < CHAINS; i++) a[i] = seed + i * 7 + threadIdx.x;                       \
        for (int it = 0; it < ITERS; it++) {                                                      \
            _Pragma("unroll") for (int i = 0; i < CHAINS; i++) asm volatile(ASM : "+v"(a[i]) : "v"(b), "v"(c)); \
        }                                                                                         \
        uint32_t s = 0;                                                                           \
        for (int i = 0; i < CHAINS; i++) s += a[i];                                               \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                           \
    }

KERNEL(add_u32, "v_add_u32 %0, %0, %1")
KERNEL(dot2c_i32_i16, "v_dot2c_i32_i16 %0, %1, %2")
KERNEL(dot4c_i32_i8, "v_dot4c_i32_i8 %0, %1, %2")
KERNEL(dot4_u32_u8, "v_dot4_u32_u8 %0, %1, %2, %0")
KERNEL(perm_b32, "v_perm_b32 %0, %0, %1, %2")
KERNEL(ashr_pk_u8, "v_ashr_pk_u8_i32 %0, %0, %1, 19")
KERNEL(mad_u32_u24, "v_mad_u32_u24 %0, %1, %2, %0")
KERNEL(mad_i32_i24, "v_mad_i32_i24 %0, %1, %2, %0")
KERNEL(mad_i32_i16, "v_mad_i32_i16 %0, %1, %2, %0")
KERNEL(mul_lo_u32, "v_mul_lo_u32 %0, %0, %1")
KERNEL(sad_u8, "v_sad_u8 %0, %1, %2, %0")
KERNEL(msad_u8, "v_msad_u8 %0, %1, %2, %0")
KERNEL(sad_u16, "v_sad_u16 %0, %1, %2, %0")
KERNEL(pk_add_u16, "v_pk_add_u16 %0, %0, %1")
KERNEL(pk_sub_i16, "v_pk_sub_i16 %0, %0, %1")
KERNEL(pk_max_i16, "v_pk_max_i16 %0, %0, %1")
KERNEL(pk_mad_i16, "v_pk_mad_i16 %0, %1, %2, %0")
KERNEL(pk_mul_lo_u16, "v_pk_mul_lo_u16 %0, %0, %1")
KERNEL(med3_i32, "v_med3_i32 %0, %0, %1, %2")
KERNEL(min_i32, "v_min_i32 %0, %0, %1")
KERNEL(ashrrev, "v_ashrrev_i32 %0, 7, %0")
KERNEL(lshl_add, "v_lshl_add_u32 %0, %0, 3, %1")
KERNEL(fma_f32, "v_fma_f32 %0, %1, %2, %0")

KERNEL(bfe_u32, "v_bfe_u32 %0, %0, 8, 8")
KERNEL(mov_b32, "v_mov_b32 %0, %1")
KERNEL(cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
KERNEL(alignbyte, "v_alignbyte_b32 %0, %0, %1, 2")

struct Item { const char *name; void (*fn)(uint32_t *, uint32_t); };

int main()
{
    uint32_t *out;
    const int blocks = 256 * 4, threads = 256; // 4 blocks x 4 waves per CU = 4 waves per SIMD
    hipMalloc(&out, (size_t)blocks * threads * 4);
    Item items[] = {
#define I(N) { #N, k_##N }
        I(add_u32), I(dot2c_i32_i16), I(dot4c_i32_i8), I(dot4_u32_u8), I(perm_b32), I(ashr_pk_u8), I(mad_u32_u24),
        I(mad_i32_i24), I(mad_i32_i16), I(mul_lo_u32), I(sad_u8), I(msad_u8), I(sad_u16), I(pk_add_u16),
        I(pk_sub_i16), I(pk_max_i16), I(pk_mad_i16), I(pk_mul_lo_u16), I(med3_i32), I(min_i32), I(ashrrev), I(lshl_add),
        I(fma_f32), I(bfe_u32), I(mov_b32), I(cndmask), I(alignbyte),
    };
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    double base = 0;
    for (auto &it : items) {
        hipLaunchKernelGGL(it.fn, dim3(blocks), dim3(threads), 0, 0, out, 1u);
        hipEventRecord(e0);
        hipLaunchKernelGGL(it.fn, dim3(blocks), dim3(threads), 0, 0, out, 2u);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double winst = (double)blocks * 4 * CHAINS * ITERS; // wave-instructions
        const double per_simd_per_ns = winst / 1024.0 / (ms * 1e6);
        if (!base) base = per_simd_per_ns;
        printf("%-16s %8.3f ms  %6.3f wave-instr/ns/SIMD  rel %5.2f  (%.1f G lane-ops/s chip)\n", it.name, ms, per_simd_per_ns,
               per_simd_per_ns / base, winst * 64 / (ms * 1e-3) / 1e9);
    }
    return 0;
}
