/* tx_kernels.h — launchers of kernels/tx_radix.hip (the register-resident FFT / MDCT of 256 .. 1024 complex points) */
#ifndef FFHIP_TX_KERNELS_H
#define FFHIP_TX_KERNELS_H

#include <hip/hip_runtime.h>
#include <stddef.h>

/* n complex points are served by the radix kernels */
bool ffhip_tx_radix_ok(int n);
/* ... the FFT alone also at 2048 .. 16384 (a team of n / 16 threads per transform) */
bool ffhip_tx_radix_fft_ok(int n);

/* wtab: n entries exp(-2 pi i k / n) on the device.  Rows 8-byte aligned and contiguous. */
int ffhip_launch_fft_r(int n, int inv, const float2 *wtab, const float *in, size_t in_pitch, float *out, size_t out_pitch, int nt,
                       hipStream_t stream);
/* exptab: the context's ff_tx_mdct_gen_exp table (n entries, natural order, scale folded in) */
int ffhip_launch_mdct_r(int n, int inv, const float2 *wtab, const float2 *exptab, const float *in, size_t in_pitch, float *out,
                        size_t out_pitch, int nt, hipStream_t stream);

/* kernels/tx_wide.hip: AV_TX_DOUBLE_* / AV_TX_INT32_* FFT and MDCT at power-of-two lengths */
struct FFHipTxWide;
int    ffhip_txw_create(FFHipTxWide **w, int is_int, int is_mdct, int inv, int len, double scale);
void   ffhip_txw_free(FFHipTxWide *w);
int    ffhip_txw_batch(const FFHipTxWide *w, void *out, size_t out_pitch, const void *in, size_t in_pitch, int nt, hipStream_t stream);
size_t ffhip_txw_in_elems(const FFHipTxWide *w);
size_t ffhip_txw_out_elems(const FFHipTxWide *w);
size_t ffhip_txw_elem_size(const FFHipTxWide *w);
int    ffhip_txw_device(const FFHipTxWide *w);

/* kernels/tx_dcst1.hip: AV_TX_FLOAT_DCT_I / AV_TX_FLOAT_DST_I, forward, even lengths 4..1024 */
struct FFHipTxDcst1;
int  ffhip_dcst1_create(FFHipTxDcst1 **w, int is_dst, int len, float scale);
void ffhip_dcst1_free(FFHipTxDcst1 *w);
int  ffhip_dcst1_batch(const FFHipTxDcst1 *w, float *out, size_t out_pitch, const float *in, size_t in_pitch, ptrdiff_t istride, int nt,
                       hipStream_t stream);
int  ffhip_dcst1_device(const FFHipTxDcst1 *w);
int  ffhip_dcst1_len(const FFHipTxDcst1 *w);

#endif
