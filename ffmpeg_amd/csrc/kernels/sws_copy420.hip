/*
 * sws_copy420.hip — 4:2:0 between its planar and semi-planar layouts at the same size (round 5): NV12 / NV21 <-> yuv420p, NV12 <-> NV21.
 *
 * What the reference runs for these is a copy (nv12ToPlanarWrapper / planarToNv12Wrapper / nv24-style interleaves,
 * libswscale/swscale_unscaled.c:142-235: the luma plane copied, the chroma bytes dealt out or woven together); through the scaler — which is
 * how this library reaches it: four one-tap banks — it is hScale8To15_c with the tap 16384 (s << 7, swscale.c:128-142) and
 * yuv2plane1_8_c with the constant dither 64 ((s * 128 + 64) >> 7 = s, output.c:468-486): the same bytes.  The column walker computed
 * that with its whole apparatus at 0.29 of HBM; this kernel streams: a lane owns 16 bytes of a destination row, reads the 16 or 32
 * source bytes they come from and shuffles them with v_perm_b32.
 *
 * A row's last lane takes the row's LAST 16 bytes whatever their alignment (it overlaps its neighbour with identical bytes): no byte
 * tails, nothing read or written past a row.  Rows of fewer than 16 bytes keep the older kernels.
 */
#include "common.h"
#include "sws_kernels.h"

typedef uint32_t c4_u4 __attribute__((ext_vector_type(4)));
typedef uint32_t c4_u2 __attribute__((ext_vector_type(2)));
typedef c4_u4 __attribute__((aligned(1))) c4_u4a;
typedef c4_u2 __attribute__((aligned(1))) c4_u2a;
typedef const uint8_t __attribute__((address_space(1))) *c4_gcp;
typedef uint8_t __attribute__((address_space(1))) *c4_gp;
typedef const c4_u4a __attribute__((address_space(1))) *c4_gc4;
typedef const c4_u2a __attribute__((address_space(1))) *c4_gc2;
typedef c4_u4a __attribute__((address_space(1))) *c4_g4;

/* job kinds: what the 16 destination bytes at x are made of */
enum { C4_COPY = 0, C4_PICK = 1 /* one channel out of (a, b) pairs: src bytes 2x + k */, C4_WEAVE = 2 /* (a, b) pairs from two planes */,
       C4_SWAP = 3 /* (a, b) -> (b, a) */ };

__global__ __launch_bounds__(256) void k_sws_copy420(FFHipCopy420Args A)
{
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    const uint32_t gw = blockIdx.x * 4u + (uint32_t)wave;
    if (gw >= (uint32_t)A.units_per_frame * (uint32_t)A.nframes)
        return;
    const int f = (int)(gw / (uint32_t)A.units_per_frame);
    const int u = (int)(gw - (uint32_t)f * (uint32_t)A.units_per_frame);
    int j = 0;
    if (A.njobs > 1 && u >= A.job[1].unit_begin) j = 1;
    if (A.njobs > 2 && u >= A.job[2].unit_begin) j = 2;
    const FFHipCopy420Job &J = A.job[j];
    const int local = u - J.unit_begin;
    const int row = local / J.ncb, cb = local - row * J.ncb;
    const int x = 16 * (cb * 64 + lane);
    if (x >= J.wbytes)
        return;
    const int xs = min(x, J.wbytes - 16);
    c4_gcp s0 = (c4_gcp)(J.src[0] + (size_t)f * J.sfp[0] + (ptrdiff_t)row * J.sstride[0]);
    c4_u4 o;
    if (J.kind == C4_COPY || J.kind == C4_SWAP) { /* uniform */
        o = *(c4_gc4)(s0 + (uint32_t)xs);
        if (J.kind == C4_SWAP) {
            o.x = __builtin_amdgcn_perm(o.x, o.x, 0x02030001u); o.y = __builtin_amdgcn_perm(o.y, o.y, 0x02030001u);
            o.z = __builtin_amdgcn_perm(o.z, o.z, 0x02030001u); o.w = __builtin_amdgcn_perm(o.w, o.w, 0x02030001u);
        }
    } else if (J.kind == C4_PICK) {
        const c4_u4 a = *(c4_gc4)(s0 + 2u * (uint32_t)xs), b = *(c4_gc4)(s0 + 2u * (uint32_t)xs + 16u);
        const uint32_t sel = J.k ? 0x07050301u : 0x06040200u;
        o.x = __builtin_amdgcn_perm(a.y, a.x, sel); o.y = __builtin_amdgcn_perm(a.w, a.z, sel);
        o.z = __builtin_amdgcn_perm(b.y, b.x, sel); o.w = __builtin_amdgcn_perm(b.w, b.z, sel);
    } else {
        /* pairs xs / 2 .. xs / 2 + 7: eight bytes of each plane; src[0] fills the even destination bytes */
        c4_gcp s1 = (c4_gcp)(J.src[1] + (size_t)f * J.sfp[1] + (ptrdiff_t)row * J.sstride[1]);
        const c4_u2 a = *(c4_gc2)(s0 + (uint32_t)(xs >> 1)), b = *(c4_gc2)(s1 + (uint32_t)(xs >> 1));
        o.x = __builtin_amdgcn_perm(b.x, a.x, 0x05010400u); o.y = __builtin_amdgcn_perm(b.x, a.x, 0x07030602u);
        o.z = __builtin_amdgcn_perm(b.y, a.y, 0x05010400u); o.w = __builtin_amdgcn_perm(b.y, a.y, 0x07030602u);
    }
    c4_gp d = (c4_gp)(J.dst + (size_t)f * J.dfp + (ptrdiff_t)row * J.dstride);
    *(c4_g4)(d + (uint32_t)xs) = o;
}

int ffhip_launch_copy420(FFHipCopy420Args &A, hipStream_t stream)
{
    if (A.nframes <= 0)
        return 0;
    int u = 0;
    for (int i = 0; i < A.njobs; i++) {
        FFHipCopy420Job &j = A.job[i];
        if (j.wbytes < 16 || j.rows <= 0 || (j.kind == C4_WEAVE && (j.wbytes & 1))) {
            ffhip_set_error("ffhip_sws: the 4:2:0 layout kernel takes rows of 16 bytes or more (job %d: %d x %d)", i, j.wbytes, j.rows);
            return FFHIP_EINVAL;
        }
        j.ncb = cdiv(cdiv(j.wbytes, 16), 64);
        j.unit_begin = u;
        u += j.ncb * j.rows;
    }
    A.units_per_frame = u;
    const long long waves = (long long)u * A.nframes;
    if (waves >= (1LL << 31)) {
        ffhip_set_error("ffhip_sws: batch too large for one launch (%lld waves)", waves);
        return FFHIP_EINVAL;
    }
    hipLaunchKernelGGL(k_sws_copy420, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, stream, A);
    LAUNCH_CHECK();
    return 0;
}
