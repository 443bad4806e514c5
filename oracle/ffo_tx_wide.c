/*
 * ffo_tx_wide.c — CPU restatement of av_tx's double and int32 FFT / MDCT at power-of-two lengths (AV_TX_DOUBLE_FFT / _MDCT,
 * AV_TX_INT32_FFT / _MDCT: libavutil/tx.h:48-69, tx_double.c, tx_int32.c).  TEST INFRASTRUCTURE ONLY: imported by tests/, never by the
 * product.  Pinned bit for bit to the reference compiled in place (tests/test_oracle_vs_ref_tx_wide.py) and through the committed
 * vectors of tests/golden/tx_wide.npz.  Written as the reference's recursion with its hard-coded 4 / 8 / 16-point codelets — not as
 * the flattened network libffhip runs — so that the two forms check each other.
 */
#define _GNU_SOURCE /* sincos() */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "ffo.h"

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

#define W_INT 0
#include "ffo_tx_wide_tmpl.h"
#undef W_INT
#define W_INT 1
#include "ffo_tx_wide_tmpl.h"
#undef W_INT

/* len complex samples in, len out (contiguous); is_int: int32_t pairs, else double pairs */
void ffo_txw_fft_run(int is_int, int inv, int len, void *out, const void *in)
{
    if (is_int)
        fft_run_i32(inv, len, out, in);
    else
        fft_run_f64(inv, len, out, in);
}

/* forward: 2 * len samples in, len out; inverse: len in, len out (the half-window form); contiguous */
void ffo_txw_mdct_run(int is_int, int inv, int len, double scale, void *out, const void *in)
{
    if (is_int)
        mdct_run_i32(inv, len, scale, out, in);
    else
        mdct_run_f64(inv, len, scale, out, in);
}
