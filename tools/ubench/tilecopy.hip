// tools/ubench/tilecopy.hip — what the memory pipeline does with 16x16-byte tiles (the block-MC access shape): the same bytes moved as
//   P0  one tile per wave, one dword per lane: a wave-instruction touches 16 rows x 16 B          (k_h264_qpel_l's store shape)
//   P1  four x-adjacent tiles per wave, 16 B per lane: 16 rows x 64 B per instruction
//   P2  four x-adjacent tiles per wave, dword per lane, four passes: 4 rows x 64 B per instruction
//   P3  P0's stores, the loads as the 21-row x 8-dword footprint of a displaced block (three dword loads per lane)
//   P4  P3 with the footprint passed through LDS and re-read (the kernel's skeleton without arithmetic)
//   P5  one 16-byte load per lane covers the footprint (lane = row, half: 21 rows x 32 B in ONE instruction), LDS, P0's stores
//   P6  P5's loads for four tiles, their outputs gathered in LDS and stored as 16 rows x 64 B (one 16 B store per lane)
// 32 planes of 3840x2160 inside 3904-byte pitched planes.  Build: hipcc --offload-arch=gfx950 -O3 -o tilecopy tilecopy.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>

typedef uint32_t u4 __attribute__((ext_vector_type(4)));
static const int W = 3840, H = 2160, PAD = 32, STRIDE = W + 2 * PAD, ROWS = H + 2 * PAD, MBW = W / 16, MBH = H / 16;

__device__ __forceinline__ size_t tile_off(int t, int *dx, int *dy)
{
    const int per = MBW * MBH, p = t / per, r = t - p * per, my = r / MBW, mx = r - my * MBW;
    const uint32_t h = (uint32_t)t * 2654435761u;
    *dx = (int)(h >> 8 & 31) - 16;
    *dy = (int)(h >> 16 & 31) - 16;
    return ((size_t)p * ROWS + PAD + my * 16) * STRIDE + PAD + mx * 16;
}

template <int P>
__global__ __launch_bounds__(256) void k_tile(uint8_t *dst, const uint8_t *src, int ntiles)
{
    __shared__ uint32_t lds[4][21 * 8];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int dx, dy;
    if (P == 0) {
        const int t = blockIdx.x * 4 + wave;
        if (t >= ntiles) return;
        const size_t o = tile_off(t, &dx, &dy) + (size_t)(lane >> 2) * STRIDE + 4 * (lane & 3);
        *(uint32_t *)(dst + o) = *(const uint32_t *)(src + o);
    } else if (P == 1) {
        const int t = (blockIdx.x * 4 + wave) * 4;
        if (t >= ntiles) return;
        const size_t o = tile_off(t, &dx, &dy) + (size_t)(lane >> 2) * STRIDE + 16 * (lane & 3);
        *(u4 *)(dst + o) = *(const u4 *)(src + o);
    } else if (P == 2) {
        const int t = (blockIdx.x * 4 + wave) * 4;
        if (t >= ntiles) return;
        const size_t o = tile_off(t, &dx, &dy) + (size_t)(lane >> 4) * STRIDE + 4 * (lane & 15);
        uint32_t v[4];
#pragma unroll
        for (int p = 0; p < 4; p++) v[p] = *(const uint32_t *)(src + o + (size_t)(4 * p) * STRIDE);
#pragma unroll
        for (int p = 0; p < 4; p++) *(uint32_t *)(dst + o + (size_t)(4 * p) * STRIDE) = v[p];
    } else {
        const int t = blockIdx.x * 4 + wave;
        if (t >= ntiles) return;
        const size_t o = tile_off(t, &dx, &dy);
        const uint8_t *s0 = src + o + (ptrdiff_t)(dy - 2) * STRIDE + dx - 2;
        const uint32_t sh = (uint32_t)((uintptr_t)s0 & 3);
        const uint8_t *sa = s0 - sh;
        uint32_t f[3];
#pragma unroll
        for (int i = 0; i < 3; i++) {
            const int u = lane + 64 * i, r = u >> 3, j = u & 7;
            f[i] = (r < 21 && j < 7) ? *(const uint32_t *)(sa + (size_t)r * STRIDE + 4 * j) : 0;
        }
        uint32_t out;
        if (P == 3) {
            out = f[0] ^ f[1] ^ f[2];
        } else {
#pragma unroll
            for (int i = 0; i < 3; i++)
                if (lane + 64 * i < 21 * 8) lds[wave][lane + 64 * i] = f[i];
            __builtin_amdgcn_wave_barrier();
            const uint32_t *p = lds[wave] + ((lane >> 2) + 2) * 8 + (lane & 3) + ((sh + 2) >> 2);
            out = __builtin_amdgcn_alignbyte(p[1], p[0], (sh + 2) & 3);
        }
        *(uint32_t *)(dst + o + (size_t)(lane >> 2) * STRIDE + 4 * (lane & 3)) = out;
    }
}

template <int P>
__global__ __launch_bounds__(256) void k_tile_w(uint8_t *dst, const uint8_t *src, int ntiles)
{
    __shared__ uint32_t lds[4][4][21 * 8];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int NT = P == 6 ? 4 : 1;
    const int t0 = (blockIdx.x * 4 + wave) * NT;
    if (t0 >= ntiles) return;
    size_t o[4];
    uint32_t sh[4];
    u4 f[4];
#pragma unroll
    for (int k = 0; k < NT; k++) {
        int dx, dy;
        o[k] = tile_off(t0 + k < ntiles ? t0 + k : ntiles - 1, &dx, &dy);
        const uint8_t *s0 = src + o[k] + (ptrdiff_t)(dy - 2) * STRIDE + dx - 2;
        sh[k] = (uint32_t)((uintptr_t)s0 & 3);
        const uint8_t *sa = s0 - sh[k];
        f[k] = lane < 42 ? *(const u4 *)(sa + (size_t)(lane >> 1) * STRIDE + 16 * (lane & 1)) : (u4){ 0, 0, 0, 0 };
    }
#pragma unroll
    for (int k = 0; k < NT; k++)
        if (lane < 42) *(u4 *)(lds[wave][k] + (lane >> 1) * 8 + 4 * (lane & 1)) = f[k];
    __builtin_amdgcn_wave_barrier();
    if (P == 5) {
        const uint32_t *p = lds[wave][0] + ((lane >> 2) + 2) * 8 + (lane & 3) + ((sh[0] + 2) >> 2);
        const uint32_t out = __builtin_amdgcn_alignbyte(p[1], p[0], (sh[0] + 2) & 3);
        *(uint32_t *)(dst + o[0] + (size_t)(lane >> 2) * STRIDE + 4 * (lane & 3)) = out;
    } else {
        /* lane (row y, tile c) assembles the 16 bytes of its row of tile c and stores them: x-adjacent tiles make 64 B rows */
        const int y = lane >> 2, c = lane & 3;
        const uint32_t s = c == 0 ? sh[0] : c == 1 ? sh[1] : c == 2 ? sh[2] : sh[3];
        const size_t oc = c == 0 ? o[0] : c == 1 ? o[1] : c == 2 ? o[2] : o[3];
        const uint32_t *p = lds[wave][c] + (y + 2) * 8 + ((s + 2) >> 2);
        u4 out;
        out.x = __builtin_amdgcn_alignbyte(p[1], p[0], (s + 2) & 3);
        out.y = __builtin_amdgcn_alignbyte(p[2], p[1], (s + 2) & 3);
        out.z = __builtin_amdgcn_alignbyte(p[3], p[2], (s + 2) & 3);
        out.w = __builtin_amdgcn_alignbyte(p[4], p[3], (s + 2) & 3);
        if (t0 + c < ntiles)
            *(u4 *)(dst + oc + (size_t)y * STRIDE) = out;
    }
}

// P7: the staged strip (round 5).  A workgroup takes 16 x-adjacent tiles of one macroblock row; the union of their displaced footprints
// (rows y0 - R - 2 .. y0 + 15 + R + 3, columns x0 - R - 2 .. x0 + 255 + R + 3) is staged ONCE into LDS with coalesced 16-byte row loads
// (a wave-instruction covers whole rows of the strip: the "plain rows" shape), then every wave serves its four tiles from LDS and stores
// them as P6 does.  R = the largest displacement (16 here: tile_off's).  P8: the same with a strip of 8 tiles per workgroup of 128 threads.
template <int R, int NTW>
__global__ __launch_bounds__(NTW * 16) void k_tile_s(uint8_t *dst, const uint8_t *src, int ntiles)
{
    constexpr int ROWS_S = 16 + 2 * R + 5, WB = NTW * 16 + 2 * R + 5, CH = (WB + 15 + 15) / 16, PITCH = CH * 4 + 1; /* dwords; +1: bank spread */
    __shared__ uint32_t lds[ROWS_S * PITCH];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int t0 = blockIdx.x * NTW;
    if (t0 >= ntiles) return;
    int dx0, dy0;
    const size_t o0 = tile_off(t0, &dx0, &dy0);   /* the strip's first tile: same macroblock row for all NTW (MBW % NTW == 0) */
    const uint8_t *s0 = src + o0 - (ptrdiff_t)(R + 2) * STRIDE - (R + 2);
    const uint32_t sh0 = (uint32_t)((uintptr_t)s0 & 15);
    const uint8_t *sa = s0 - sh0;
    for (int u = threadIdx.x; u < ROWS_S * CH; u += NTW * 16) {
        const int r = u / CH, c = u - r * CH;
        const u4 v = *(const u4 *)(sa + (size_t)r * STRIDE + 16 * c);
        uint32_t *q = lds + r * PITCH + 4 * c;
        q[0] = v.x; q[1] = v.y; q[2] = v.z; q[3] = v.w;
    }
    __syncthreads();
    const int y = lane >> 2, c = lane & 3, t = t0 + 4 * wave + c;
    int dx, dy;
    const size_t oc = tile_off(t < ntiles ? t : ntiles - 1, &dx, &dy);
    /* byte position of (row y, column 0) of tile t's displaced source inside the staged strip */
    const uint32_t bx = sh0 + (R + 2) + 16 * (4 * wave + c) + dx, by = (R + 2) + dy + y;
    const uint32_t *p = lds + by * PITCH + (bx >> 2);
    u4 out;
    out.x = __builtin_amdgcn_alignbyte(p[1], p[0], bx & 3);
    out.y = __builtin_amdgcn_alignbyte(p[2], p[1], bx & 3);
    out.z = __builtin_amdgcn_alignbyte(p[3], p[2], bx & 3);
    out.w = __builtin_amdgcn_alignbyte(p[4], p[3], bx & 3);
    if (t < ntiles)
        *(u4 *)(dst + oc + (size_t)y * STRIDE) = out;
}

template <int R, int NTW>
static void run_s(const char *name, uint8_t *d, const uint8_t *s, int ntiles)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = (ntiles + NTW - 1) / NTW, reps = 10;
    for (int i = 0; i < 2; i++) hipLaunchKernelGGL((k_tile_s<R, NTW>), dim3(blocks), dim3(NTW * 16), 0, 0, d, s, ntiles);
    hipEventRecord(e0);
    for (int i = 0; i < reps; i++) hipLaunchKernelGGL((k_tile_s<R, NTW>), dim3(blocks), dim3(NTW * 16), 0, 0, d, s, ntiles);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    printf("{\"pattern\": \"%s\", \"tiles\": %d, \"ms\": %.4f, \"GB/s\": %.1f}\n", name, ntiles, ms, 512.0 * ntiles / ms / 1e6);
}

template <int P>
static void run(const char *name, uint8_t *d, const uint8_t *s, int ntiles)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int per_wg = P == 1 || P == 2 || P == 6 ? 16 : 4, blocks = (ntiles + per_wg - 1) / per_wg;
    const int reps = 10;
    if (P >= 5) {
        for (int i = 0; i < 2; i++) hipLaunchKernelGGL((k_tile_w<P>), dim3(blocks), dim3(256), 0, 0, d, s, ntiles);
        hipEventRecord(e0);
        for (int i = 0; i < reps; i++) hipLaunchKernelGGL((k_tile_w<P>), dim3(blocks), dim3(256), 0, 0, d, s, ntiles);
    } else {
        for (int i = 0; i < 2; i++) hipLaunchKernelGGL((k_tile<P < 5 ? P : 0>), dim3(blocks), dim3(256), 0, 0, d, s, ntiles);
        hipEventRecord(e0);
        for (int i = 0; i < reps; i++) hipLaunchKernelGGL((k_tile<P < 5 ? P : 0>), dim3(blocks), dim3(256), 0, 0, d, s, ntiles);
    }
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    printf("{\"pattern\": \"%s\", \"tiles\": %d, \"ms\": %.4f, \"GB/s\": %.1f}\n", name, ntiles, ms, 512.0 * ntiles / ms / 1e6);
}

int main(int argc, char **argv)
{
    const int planes = argc > 1 ? atoi(argv[1]) : 32, ntiles = planes * MBW * MBH;
    const size_t bytes = (size_t)planes * ROWS * STRIDE + 4096;
    uint8_t *a, *b;
    if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&b, bytes) != hipSuccess)
        return 1;
    hipMemset(a, 1, bytes); hipMemset(b, 2, bytes);
    run<0>("P0 tile/wave, dword/lane (16 rows x 16 B)", b, a, ntiles);
    run<1>("P1 4 tiles/wave, 16 B/lane (16 rows x 64 B)", b, a, ntiles);
    run<2>("P2 4 tiles/wave, dword/lane x 4 passes (4 rows x 64 B)", b, a, ntiles);
    run<3>("P3 footprint loads (21 rows x 28 B), P0 stores", b, a, ntiles);
    run<4>("P4 P3 through LDS", b, a, ntiles);
    run<5>("P5 footprint as one 16 B load per lane (21 rows x 32 B), LDS, P0 stores", b, a, ntiles);
    run<6>("P6 P5 loads x 4 tiles, LDS, one 16 B store per lane (16 rows x 64 B)", b, a, ntiles);
    run_s<16, 16>("P7 staged strip of 16 tiles (53 rows x 304 B coalesced into LDS), tiles served from LDS, P6 stores", b, a, ntiles);
    run_s<16, 8>("P8 staged strip of 8 tiles, 128 threads", b, a, ntiles);
    run_s<24, 16>("P7 staged for displacements up to 24 (69 rows x 320 B; the data still moves by +-16)", b, a, ntiles);
    run<6>("P6 again", b, a, ntiles);
    return 0;
}
