// mfma_f64_layout.hip — where v_mfma_f64_16x16x4_f64 keeps its operands (round 6, kernels/tx_dcst1.hip): prints, for D = A x B with
// A[i][k] = 1000 i + k-dependent one-hots, the (row, col) every lane's four result registers hold.
//   hipcc -O2 --offload-arch=gfx950 tools/ubench/mfma_f64_layout.hip -o tools/ubench/mfma_f64_layout
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4 __attribute__((ext_vector_type(4)));

// A[i][k] = (i + 1), B[k][j] = (k == 0) * (j + 1) * 100  ->  D[i][j] = (i + 1) * (j + 1) * 100, assuming lane l feeds A[l % 16][l / 16], B[l / 16][l % 16]
__global__ void probe(double *out, int variant)
{
    const int l = threadIdx.x;
    double a, b;
    if (variant == 0) { a = (l % 16) + 1; b = (l / 16 == 0) ? ((l % 16) + 1) * 100.0 : 0.0; }
    else { /* k-dependence: A[i][k] = (k == 2), B[k][j] = k * 1000 + j  ->  D[i][j] = 2000 + j when k index = l / 16 on both sides */
        a = (l / 16 == 2) ? 1.0 : 0.0; b = (l / 16) * 1000.0 + (l % 16); }
    d4 c = { 0, 0, 0, 0 };
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int v = 0; v < 4; v++)
        out[l * 4 + v] = c[v];
}

int main()
{
    double *d, h[256];
    hipMalloc(&d, sizeof(h));
    for (int variant = 0; variant < 2; variant++) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, variant);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        if (variant == 0) {
            for (int l = 0; l < 64; l += 5)
                for (int v = 0; v < 4; v++) {
                    const int p = (int)(h[l * 4 + v] / 100.0 + 0.5);
                    // p = (i + 1) * (j + 1): with j = l % 16 assumed, i + 1 = p / (j + 1)
                    printf("lane %2d reg %d: value %6.0f -> row %d if col = lane %% 16 = %d\n", l, v, h[l * 4 + v], p / (l % 16 + 1) - 1, l % 16);
                }
        } else {
            printf("k check (expect 2000 + lane %% 16): lane 0 %g, lane 17 %g, lane 63 %g\n", h[0], h[17 * 4], h[63 * 4 + 3]);
        }
    }
    return 0;
}
