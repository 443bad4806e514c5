/* sws_uops.hip — the "hip" SwsOpBackend of libswscale's format layer.
 *
 * libswscale hands a backend a list of micro-ops (SwsUOpList, libswscale/uops.h:262-297): one read, a few per-pixel operations on
 * four component registers, one write.  The C backend chains one function per micro-op over blocks of 32 pixels
 * (uops_backend.c:104-131); the x86 backend chains asm kernels; the Vulkan backend writes SPIR-V.  On gfx950 a list becomes ONE
 * kernel: this file writes its HIP text (every micro-op contributes the statements of its definition in uops_tmpl.c, on the same C
 * types, so integer promotion / wrap-around / float evaluation order are the C backend's by construction), compiles it with hiprtc
 * (-ffp-contract=off as uops_backend.c:24-35 demands of a bit-exact backend; ~25 ms, cached by program text) and launches it with
 * one thread per 4 (8 for 1-bit formats) horizontally adjacent pixels: planar u8 moves as dwords, packed rgb24 as dwordx3, f32
 * planes as dwordx4, and the registers of a pixel never leave VGPRs between read and write.  Constants (matrices, clamps, shifts)
 * are immediates of the generated text; filter banks and dither matrices are device buffers.
 *
 * Nothing here computes on the CPU: the void host face answers a device failure through the function the caller's C backend
 * compiled for the same list (ffhip_sws_uops_set_fallback), like every other host face of this library.
 */
#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>

#include <fcntl.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <memory>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "kernels/common.h"
#include "kernels/shim_arena.h"

namespace {

constexpr int MAXDATA = 8;

/* the kernel's argument block; the generated text declares the same struct */
struct KArgs {
    const uint8_t *in[4];
    uint8_t *out[4];
    long in_stride[4];               /* SwsOpExec.in_stride: distance of the tap rows of a vertical filter */
    long in_step[4], out_step[4];    /* what one processed line advances the C backend's pointers by */
    long in_pitch[4], out_pitch[4];  /* picture f of a batch */
    const int32_t *rowtab;           /* READ_PLANAR_FV: source lines skipped before line r of this call (prefix sums of in_bump_y) */
    const int32_t *offx;             /* READ_PLANAR_FH: SwsOpExec.in_offset_x */
    const void *data[MAXDATA];
    int32_t x0, y0, npx, ny;
};

const char *const TY[5] = { "", "u8", "u16", "u32", "f32" };
const int TSIZE[5] = { 0, 1, 2, 4, 4 };

struct Plan {
    int V = 4, block_size = 1;
    int adv_in[4] = { 0, 0, 0, 0 }, adv_out[4] = { 0, 0, 0, 0 }; /* bits a pixel advances plane i by (what iter->in / out move) */
    bool palette = false;
    int fh_size = 0, fh_elem = 0, fh_mask = 0, fv_size = 0;       /* taps / bytes per tap / planes of the filtered read */
    std::vector<std::vector<uint8_t>> data;                       /* device buffers the text refers to as a.data[k] */
    std::string src;
};

struct Fmt {
    std::string s;
    void operator()(const char *fmt, ...) __attribute__((format(printf, 2, 3)))
    {
        va_list ap, aq;
        va_start(ap, fmt);
        va_copy(aq, ap);
        const int n = vsnprintf(nullptr, 0, fmt, ap);
        va_end(ap);
        if (n > 0) {
            const size_t at = s.size();
            s.resize(at + n + 1);
            vsnprintf(&s[at], n + 1, fmt, aq);
            s.resize(at + n);
        }
        va_end(aq);
    }
};

std::string lit(int type, FFHipSwsPixel v)
{
    char b[64];
    switch (type) {
    case FFHIP_SWS_PIXEL_U8:  snprintf(b, sizeof b, "(u8)%uu", v.u8); break;
    case FFHIP_SWS_PIXEL_U16: snprintf(b, sizeof b, "(u16)%uu", v.u16); break;
    case FFHIP_SWS_PIXEL_U32: snprintf(b, sizeof b, "(u32)%uu", v.u32); break;
    default:                  snprintf(b, sizeof b, "__builtin_bit_cast(f32, 0x%08xu)", v.u32); break;
    }
    return b;
}
const char *vmax(int type) { return type == FFHIP_SWS_PIXEL_U8 ? "(u8)0xFFu" : type == FFHIP_SWS_PIXEL_U16 ? "(u16)0xFFFFu" : "(u32)0xFFFFFFFFu"; }

int add_data(Plan &pl, const void *p, size_t bytes)
{
    if ((int)pl.data.size() >= MAXDATA)
        return -1;
    const uint8_t *b = static_cast<const uint8_t *>(p);
    pl.data.emplace_back(b, b + bytes);
    return (int)pl.data.size() - 1;
}

#define FOR_MASK(c) for (int c = 0; c < 4; c++) if (u.mask >> c & 1)
#define BAD(...) do { ffhip_set_error(__VA_ARGS__); return FFHIP_EINVAL; } while (0)
#define NOTSUP(...) do { ffhip_set_error(__VA_ARGS__); return FFHIP_ENOTSUP; } while (0)

/* ---- the text of one list ---------------------------------------------------------------------------------------------------- */
int generate(const FFHipSwsUOp *uops, int n_uops, Plan &pl)
{
    if (!uops || n_uops < 1 || n_uops > 64)
        BAD("sws uops: %d micro-ops", n_uops);
    bool bits = false;
    for (int i = 0; i < n_uops; i++) {
        const FFHipSwsUOp &u = uops[i];
        if (u.type < FFHIP_SWS_PIXEL_U8 || u.type > FFHIP_SWS_PIXEL_F32)
            BAD("sws uops: micro-op %d has pixel type %d", i, u.type);
        const bool rd = u.uop >= FFHIP_SWS_UOP_READ_PLANAR && u.uop <= FFHIP_SWS_UOP_READ_PALETTE;
        const bool wr = u.uop >= FFHIP_SWS_UOP_WRITE_PLANAR && u.uop <= FFHIP_SWS_UOP_WRITE_BIT;
        if (rd && i != 0)
            BAD("sws uops: a read at position %d", i);
        if (wr != (i == n_uops - 1))
            BAD("sws uops: the write must be the last micro-op, and only that");
        if (u.uop == FFHIP_SWS_UOP_READ_BIT || u.uop == FFHIP_SWS_UOP_WRITE_BIT)
            bits = true, pl.block_size = 8;
        if ((u.uop == FFHIP_SWS_UOP_READ_NIBBLE || u.uop == FFHIP_SWS_UOP_WRITE_NIBBLE) && pl.block_size < 2)
            pl.block_size = 2;
    }
    const int V = pl.V = bits ? 8 : 4;
    Fmt o;
    o("typedef unsigned char u8; typedef unsigned short u16; typedef unsigned int u32; typedef float f32;\n"
      "typedef int i32; typedef long long i64;\n"
      "struct KArgs { const u8 *in[4]; u8 *out[4]; long in_stride[4], in_step[4], out_step[4], in_pitch[4], out_pitch[4];\n"
      "               const i32 *rowtab; const i32 *offx; const void *data[%d]; i32 x0, y0, npx, ny; };\n"
      "template <class T> __device__ __forceinline__ T get(u32 r) { return (T)r; }\n"
      "template <> __device__ __forceinline__ f32 get<f32>(u32 r) { return __builtin_bit_cast(f32, r); }\n"
      "template <class T> __device__ __forceinline__ u32 put(T v) { return (u32)v; }\n"
      "template <> __device__ __forceinline__ u32 put<f32>(f32 v) { return __builtin_bit_cast(u32, v); }\n"
      "#define V %d\n"
      "#define EACH for (int i = 0; i < V; i++)\n"
      "typedef u32 u32v __attribute__((ext_vector_type(V)));\n"
      "/* FULL: all V pixels of the thread exist; XAL: the thread's first pixel is a multiple of V in the picture */\n"
      "template <bool FULL, bool XAL> __device__ __forceinline__ void body(const KArgs &a, const u8 *__restrict__ in0,\n"
      "    const u8 *__restrict__ in1, const u8 *__restrict__ in2, const u8 *__restrict__ in3, u8 *__restrict__ out0,\n"
      "    u8 *__restrict__ out1, u8 *__restrict__ out2, u8 *__restrict__ out3, const int p, const int n, const int xabs, const int yabs)\n{\n"
      "    u32 r[4][V] = {}; u32 tmp[V] = {};\n"
      "    (void)tmp; (void)xabs; (void)yabs; (void)in0; (void)in1; (void)in2; (void)in3; (void)out0; (void)out1; (void)out2; (void)out3;\n",
      MAXDATA, V);

    for (int k = 0; k < n_uops; k++) {
        const FFHipSwsUOp &u = uops[k];
        const char *T = TY[u.type];
        const int ts = TSIZE[u.type];
        o("    /* %d */\n", k);
        switch (u.uop) {
        /* ---- reads: registers <- planes (uops_tmpl.c:66-96, 127-190, 247-350) ---- */
        case FFHIP_SWS_UOP_READ_PLANAR:
            FOR_MASK(c) {
                pl.adv_in[c] = 8 * ts;
                o("    { %s v[V]; const u8 *s = in%d + (long)p * %d;\n"
                  "      if (FULL) __builtin_memcpy(v, s, sizeof v); else EACH if (i < n) __builtin_memcpy(&v[i], s + i * %d, %d); else v[i] = 0;\n"
                  "      EACH r[%d][i] = put<%s>(v[i]); }\n", T, c, ts, ts, ts, c, T);
            }
            break;
        case FFHIP_SWS_UOP_READ_PACKED: {
            const int el = (u.mask & 8) ? 4 : (u.mask & 4) ? 3 : (u.mask & 2) ? 2 : 1;
            pl.adv_in[0] = 8 * ts * el;
            o("    { %s v[V * %d]; const u8 *s = in0 + (long)p * %d;\n"
              "      if (FULL) __builtin_memcpy(v, s, sizeof v); else for (int i = 0; i < V * %d; i++) if (i < n * %d) __builtin_memcpy(&v[i], s + i * %d, %d); else v[i] = 0;\n",
              T, el, ts * el, el, el, ts, ts);
            FOR_MASK(c) if (c < el) o("      EACH r[%d][i] = put<%s>(v[%d * i + %d]);\n", c, T, el, c);
            o("    }\n");
            break;
        }
        case FFHIP_SWS_UOP_READ_NIBBLE:
            if (u.type != FFHIP_SWS_PIXEL_U8)
                BAD("sws uops: 4-bit read of a %s plane", T);
            pl.adv_in[0] = 4;
            o("    EACH if (i < n) { const u8 b = in0[(p + i) >> 1]; r[0][i] = (i & 1) ? (b & 0xF) : (b >> 4); }\n");
            break;
        case FFHIP_SWS_UOP_READ_BIT:
            if (u.type != FFHIP_SWS_PIXEL_U8)
                BAD("sws uops: 1-bit read of a %s plane", T);
            pl.adv_in[0] = 1;
            o("    EACH if (i < n) { const u8 b = in0[(p + i) >> 3]; r[0][i] = (b >> (7 - (i & 7))) & 1; }\n");
            break;
        case FFHIP_SWS_UOP_READ_PALETTE:
            if (u.type != FFHIP_SWS_PIXEL_U8)
                BAD("sws uops: palette read of a %s plane", T);
            pl.adv_in[0] = 8;
            pl.palette = true;
            o("    EACH if (i < n) { const u8 *e = in1 + 4 * (int)in0[p + i]; r[0][i] = e[0]; r[1][i] = e[1]; r[2][i] = e[2]; r[3][i] = e[3]; }\n");
            break;
        case FFHIP_SWS_UOP_READ_PLANAR_FH: {
            const FFHipSwsFilterWeights *f = u.data.kernel;
            if (u.par.filter.type != FFHIP_SWS_PIXEL_F32)
                NOTSUP("sws uops: horizontal filter stored as type %d", u.par.filter.type);
            if (!f || f->filter_size < 1 || !f->weights || f->dst_size < 1)
                BAD("sws uops: horizontal filter without a kernel");
            const int d = add_data(pl, f->weights, (size_t)f->dst_size * f->filter_size * sizeof(int));
            if (d < 0)
                BAD("sws uops: too many data buffers");
            pl.fh_size = f->filter_size;
            pl.fh_elem = ts;
            pl.fh_mask = u.mask;
            /* uops_tmpl.c:316-350: integer taps into int32 (u8) / int64 (u16, u32) / float (f32), one multiply by 1 / SWS_FILTER_SCALE */
            const char *acc = u.type == FFHIP_SWS_PIXEL_U8 ? "i32" : u.type == FFHIP_SWS_PIXEL_F32 ? "f32" : "i64";
            o("    EACH if (i < n) { const i32 *wt = (const i32 *)a.data[%d] + (long)%d * (xabs + i); const i32 off = a.offx[xabs + i];\n", d, f->filter_size);
            FOR_MASK(c) {
                o("      { const u8 *s = in%d + off; %s acc = 0;\n"
                  "        for (int j = 0; j < %d; j++) { %s t; __builtin_memcpy(&t, s + j * %d, %d); acc += wt[j] * t; }\n"
                  "        r[%d][i] = put<f32>((f32)acc * %s); }\n",
                  c, acc, f->filter_size, T, ts, ts, c, "__builtin_bit_cast(f32, 0x38800000u)" /* 1.0f / 16384 */);
            }
            o("    }\n");
            break;
        }
        case FFHIP_SWS_UOP_READ_PLANAR_FV: {
            const FFHipSwsFilterWeights *f = u.data.kernel;
            if (u.par.filter.type != FFHIP_SWS_PIXEL_F32)
                NOTSUP("sws uops: vertical filter stored as type %d", u.par.filter.type);
            if (!f || f->filter_size < 1 || !f->weights || f->dst_size < 1)
                BAD("sws uops: vertical filter without a kernel");
            /* uops_tmpl.c:247-265: the weights become floats once, at setup */
            std::vector<float> w((size_t)f->dst_size * f->filter_size);
            for (size_t i = 0; i < w.size(); i++)
                w[i] = (float)f->weights[i] / FFHIP_SWS_FILTER_SCALE;
            const int d = add_data(pl, w.data(), w.size() * sizeof(float));
            if (d < 0)
                BAD("sws uops: too many data buffers");
            pl.fv_size = f->filter_size;
            o("    { const f32 *wt = (const f32 *)a.data[%d] + (long)%d * yabs;\n", d, f->filter_size);
            FOR_MASK(c) {
                pl.adv_in[c] = 8 * ts;
                o("      { f32 acc[V]; EACH acc[i] = 0.0f; const u8 *s = in%d + (long)p * %d;\n"
                  "        for (int j = 0; j < %d; j++, s += a.in_stride[%d]) { const f32 w = wt[j]; %s v[V];\n"
                  "          if (FULL) __builtin_memcpy(v, s, sizeof v); else EACH if (i < n) __builtin_memcpy(&v[i], s + i * %d, %d); else v[i] = 0;\n"
                  "          EACH acc[i] += w * v[i]; }\n"
                  "        EACH r[%d][i] = put<f32>(acc[i]); }\n", c, ts, f->filter_size, c, T, ts, ts, c);
            }
            o("    }\n");
            break;
        }
        /* ---- writes (uops_tmpl.c:98-125, 192-222) ---- */
        case FFHIP_SWS_UOP_WRITE_PLANAR:
            FOR_MASK(c) {
                pl.adv_out[c] = 8 * ts;
                o("    { %s v[V]; EACH v[i] = get<%s>(r[%d][i]); u8 *d = out%d + (long)p * %d;\n"
                  "      if (FULL) __builtin_memcpy(d, v, sizeof v); else EACH if (i < n) __builtin_memcpy(d + i * %d, &v[i], %d); }\n",
                  T, T, c, c, ts, ts, ts);
            }
            break;
        case FFHIP_SWS_UOP_WRITE_PACKED: {
            const int el = (u.mask & 8) ? 4 : (u.mask & 4) ? 3 : (u.mask & 2) ? 2 : 1;
            pl.adv_out[0] = 8 * ts * el;
            o("    { %s v[V * %d]; u8 *d = out0 + (long)p * %d;\n", T, el, ts * el);
            bool all = true;
            for (int c = 0; c < el; c++)
                if (u.mask >> c & 1)
                    o("      EACH v[%d * i + %d] = get<%s>(r[%d][i]);\n", el, c, T, c);
                else
                    all = false;
            if (all)
                o("      if (FULL) __builtin_memcpy(d, v, sizeof v); else for (int i = 0; i < V * %d; i++) if (i < n * %d) __builtin_memcpy(d + i * %d, &v[i], %d); }\n",
                  el, el, ts, ts);
            else /* if (X) out0[elems * i + 0] = ...: the unmasked elements of a pixel keep what the buffer held */
                for (int c = 0; c < el; c++)
                    if (u.mask >> c & 1)
                        o("      EACH if (i < n) __builtin_memcpy(d + (%d * i + %d) * %d, &v[%d * i + %d], %d);\n", el, c, ts, el, c, ts);
            if (!all)
                o("    }\n");
            break;
        }
        case FFHIP_SWS_UOP_WRITE_NIBBLE:
            if (u.type != FFHIP_SWS_PIXEL_U8)
                BAD("sws uops: 4-bit write of a %s plane", T);
            pl.adv_out[0] = 4;
            o("    for (int i = 0; i < V; i += 2) if (i < n) out0[(p + i) >> 1] = (u8)(get<u8>(r[0][i]) << 4 | get<u8>(r[0][i + 1]));\n");
            break;
        case FFHIP_SWS_UOP_WRITE_BIT:
            if (u.type != FFHIP_SWS_PIXEL_U8)
                BAD("sws uops: 1-bit write of a %s plane", T);
            pl.adv_out[0] = 1;
            o("    if (n > 0) { u32 b = 0; EACH b |= (u32)get<u8>(r[0][i]) << (7 - i); out0[p >> 3] = (u8)b; }\n");
            break;
        /* ---- register moves (uops_tmpl.c:357-407): sequential, register -1 is the temporary ---- */
        case FFHIP_SWS_UOP_PERMUTE:
        case FFHIP_SWS_UOP_COPY:
            if (u.par.move.num_moves < 0 || u.par.move.num_moves > 6)
                BAD("sws uops: %d moves", u.par.move.num_moves);
            for (int m = 0; m < u.par.move.num_moves; m++) {
                const int d = u.par.move.dst[m], s = u.par.move.src[m];
                if (d < -1 || d > 3 || s < -1 || s > 3)
                    BAD("sws uops: move %d -> %d", s, d);
                char dn[16], sn[16];
                d < 0 ? snprintf(dn, sizeof dn, "tmp") : snprintf(dn, sizeof dn, "r[%d]", d);
                s < 0 ? snprintf(sn, sizeof sn, "tmp") : snprintf(sn, sizeof sn, "r[%d]", s);
                o("    EACH %s[i] = %s[i];\n", dn, sn);
            }
            break;
        /* ---- conversions and bit manipulation (uops_tmpl.c:417-560) ---- */
        case FFHIP_SWS_UOP_SWAP_BYTES:
            if (u.type != FFHIP_SWS_PIXEL_U16 && u.type != FFHIP_SWS_PIXEL_U32)
                BAD("sws uops: byte swap of %s", T);
            FOR_MASK(c) o("    EACH r[%d][i] = put<%s>(%s(get<%s>(r[%d][i])));\n", c, T, ts == 2 ? "__builtin_bswap16" : "__builtin_bswap32", T, c);
            break;
        case FFHIP_SWS_UOP_EXPAND_BIT:
            if (u.type == FFHIP_SWS_PIXEL_F32)
                BAD("sws uops: bit expansion of f32");
            FOR_MASK(c) o("    EACH r[%d][i] = put<%s>(get<%s>(r[%d][i]) ? %s : (%s)0);\n", c, T, T, c, vmax(u.type), T);
            break;
        case FFHIP_SWS_UOP_EXPAND_PAIR:
            if (u.type != FFHIP_SWS_PIXEL_U8)
                BAD("sws uops: pair expansion of %s", T);
            FOR_MASK(c) o("    EACH { const u8 x = get<u8>(r[%d][i]); r[%d][i] = put<u16>((u16)(x << 8 | x)); }\n", c, c);
            break;
        case FFHIP_SWS_UOP_EXPAND_QUAD:
            if (u.type != FFHIP_SWS_PIXEL_U8)
                BAD("sws uops: quad expansion of %s", T);
            FOR_MASK(c) o("    EACH { const u8 x = get<u8>(r[%d][i]); r[%d][i] = put<u32>((u32)x << 24 | x << 16 | x << 8 | x); }\n", c, c);
            break;
        case FFHIP_SWS_UOP_TO_U8:
        case FFHIP_SWS_UOP_TO_U16:
        case FFHIP_SWS_UOP_TO_U32:
        case FFHIP_SWS_UOP_TO_F32: {
            const char *D = TY[FFHIP_SWS_PIXEL_U8 + (u.uop - FFHIP_SWS_UOP_TO_U8)];
            FOR_MASK(c) o("    EACH r[%d][i] = put<%s>((%s)get<%s>(r[%d][i]));\n", c, D, D, T, c);
            break;
        }
        case FFHIP_SWS_UOP_LSHIFT:
        case FFHIP_SWS_UOP_RSHIFT:
            if (u.type == FFHIP_SWS_PIXEL_F32)
                BAD("sws uops: shift of f32");
            FOR_MASK(c) o("    EACH { %s x = get<%s>(r[%d][i]); x %s= %d; r[%d][i] = put<%s>(x); }\n", T, T, c,
                          u.uop == FFHIP_SWS_UOP_LSHIFT ? "<<" : ">>", u.par.shift.amount, c, T);
            break;
        case FFHIP_SWS_UOP_UNPACK:
        case FFHIP_SWS_UOP_PACK: {
            if (u.type == FFHIP_SWS_PIXEL_F32)
                BAD("sws uops: (un)pack of f32");
            const uint8_t *b = u.par.pack.pattern;
            const int sh[4] = { b[3] + b[2] + b[1], b[3] + b[2], b[3], 0 };
            if (u.uop == FFHIP_SWS_UOP_UNPACK) {
                o("    EACH { const %s val = get<%s>(r[0][i]);\n", T, T);
                FOR_MASK(c) o("      r[%d][i] = put<%s>((%s)((val >> %d) & (%s)((1 << %d) - 1)));\n", c, T, T, sh[c], T, b[c]);
                o("    }\n");
            } else {
                o("    EACH { %s val = 0;\n", T);
                FOR_MASK(c) o("      val |= get<%s>(r[%d][i]) << %d;\n", T, c, sh[c]);
                o("      r[0][i] = put<%s>(val); }\n", T);
            }
            break;
        }
        case FFHIP_SWS_UOP_CLEAR:
            FOR_MASK(c) {
                std::string v = (u.par.clear.one >> c & 1) && u.type != FFHIP_SWS_PIXEL_F32 ? std::string(vmax(u.type))
                              : (u.par.clear.zero >> c & 1) ? std::string("(") + T + ")0" : lit(u.type, u.data.vec4[c]);
                o("    EACH r[%d][i] = put<%s>(%s);\n", c, T, v.c_str());
            }
            break;
        /* ---- arithmetic (uops_tmpl.c:640-700): compound assignment on the pixel type ---- */
        case FFHIP_SWS_UOP_SCALE:
            FOR_MASK(c) o("    EACH { %s x = get<%s>(r[%d][i]); x *= %s; r[%d][i] = put<%s>(x); }\n", T, T, c, lit(u.type, u.data.scalar).c_str(), c, T);
            break;
        case FFHIP_SWS_UOP_ADD:
            FOR_MASK(c) o("    EACH { %s x = get<%s>(r[%d][i]); x += %s; r[%d][i] = put<%s>(x); }\n", T, T, c, lit(u.type, u.data.vec4[c]).c_str(), c, T);
            break;
        case FFHIP_SWS_UOP_MIN:
            FOR_MASK(c) o("    EACH { const %s x = get<%s>(r[%d][i]), k = %s; r[%d][i] = put<%s>(x > k ? k : x); }\n", T, T, c, lit(u.type, u.data.vec4[c]).c_str(), c, T);
            break;
        case FFHIP_SWS_UOP_MAX:
            FOR_MASK(c) o("    EACH { const %s x = get<%s>(r[%d][i]), k = %s; r[%d][i] = put<%s>(x > k ? x : k); }\n", T, T, c, lit(u.type, u.data.vec4[c]).c_str(), c, T);
            break;
        case FFHIP_SWS_UOP_DITHER: {
            const int lg = u.par.dither.size_log2, size = 1 << lg;
            if (lg > 8 || !u.data.ptr)
                BAD("sws uops: dither matrix of 2^%d", lg);
            int mo = 0;
            for (int c = 0; c < 4; c++) /* ff_sws_dither_height (uops.c:232-238): over all four, masked or not */
                mo = u.par.dither.y_offset[c] > mo ? u.par.dither.y_offset[c] : mo;
            const int d = add_data(pl, u.data.ptr, (size_t)size * (size + mo) * sizeof(FFHipSwsPixel));
            if (d < 0)
                BAD("sws uops: too many data buffers");
            /* uops_tmpl.c:737-765: row (y & (size - 1)) + y_offset[c], column x & (size - 1) */
            /* the V columns of a thread are adjacent and, when the thread starts on a multiple of V, do not wrap: one vector load */
            o("    { const u32 *m = (const u32 *)a.data[%d] + (yabs & %d) * %d;\n", d, size - 1, size);
            FOR_MASK(c) {
                o("      { u32 dv[V]; const u32 *mr = m + %d;\n", u.par.dither.y_offset[c] * size);
                if (size >= V)
                    o("        if (XAL) { const u32v t = *(const u32v *)(mr + (xabs & %d)); EACH dv[i] = t[i]; } else\n", size - 1);
                o("        EACH dv[i] = mr[(xabs + i) & %d];\n", size - 1);
                o("        EACH { %s x = get<%s>(r[%d][i]); x += get<%s>(dv[i]); r[%d][i] = put<%s>(x); } }\n", T, T, c, T, c, T);
            }
            o("    }\n");
            break;
        }
        case FFHIP_SWS_UOP_LINEAR: {
            /* uops_tmpl.c:795-830: every row starts from its offset and adds the products left to right, each product rounded on
             * its own; coefficients flagged one / zero are not multiplied / not added */
            o("    EACH { const %s c0 = get<%s>(r[0][i]), c1 = get<%s>(r[1][i]), c2 = get<%s>(r[2][i]), c3 = get<%s>(r[3][i]);\n", T, T, T, T, T);
            FOR_MASK(c) {
                const uint32_t one = u.par.lin.one, zero = u.par.lin.zero;
                o("      { %s acc = %s;\n", T, (zero >> (5 * c + 4) & 1) ? (std::string("(") + T + ")0").c_str() : lit(u.type, u.data.mat4[c][4]).c_str());
                for (int j = 0; j < 4; j++) {
                    if (zero >> (5 * c + j) & 1)
                        continue;
                    if (one >> (5 * c + j) & 1)
                        o("        acc += c%d;\n", j);
                    else
                        o("        acc += %s * c%d;\n", lit(u.type, u.data.mat4[c][j]).c_str(), j);
                }
                o("        r[%d][i] = put<%s>(acc); }\n", c, T);
            }
            o("    }\n");
            break;
        }
        case FFHIP_SWS_UOP_LUT_3D:
        case FFHIP_SWS_UOP_RW_SHUFFLE:
        case FFHIP_SWS_UOP_LINEAR_FMA:
        case FFHIP_SWS_UOP_READ_PLANAR_FV_FMA:
            NOTSUP("sws uops: micro-op %d is not taken by the hip backend", u.uop);
        default:
            BAD("sws uops: unknown micro-op %d", u.uop);
        }
    }
    /* one thread: V adjacent pixels of one line; 64 x 4 threads: 256 pixels of 4 lines (a wave is one line: the line number, the
     * plane pointers of the line and the dither row live in SGPRs); z: the picture of a batch */
    int used_in = 0, used_out = 0;
    for (int c = 0; c < 4; c++) {
        used_in |= (pl.adv_in[c] || (pl.fh_mask >> c & 1)) << c;
        used_out |= (pl.adv_out[c] != 0) << c;
    }
    if (pl.palette)
        used_in |= 2;
    o("}\n"
      "extern \"C\" __global__ __launch_bounds__(256) void sws_uops(const KArgs a)\n{\n"
      "    const int p = (blockIdx.x * 64 + threadIdx.x) * V;\n"
      "    if (p >= a.npx) return;\n"
      "    const int n = a.npx - p < V ? a.npx - p : V;\n"
      "    const bool xal = !(a.x0 & (V - 1));\n"
      "    for (int rr = __builtin_amdgcn_readfirstlane(blockIdx.y * 4 + threadIdx.y); rr < a.ny; rr += gridDim.y * 4) {\n"
      "        const long skip = a.rowtab ? a.rowtab[rr] : 0; (void)skip;\n");
    for (int c = 0; c < 4; c++) {
        if (used_in >> c & 1)
            o("        const u8 *in%d = a.in[%d] + blockIdx.z * a.in_pitch[%d] + rr * a.in_step[%d] + skip * a.in_stride[%d];\n", c, c, c, c, c);
        else
            o("        const u8 *in%d = nullptr;\n", c);
        if (used_out >> c & 1)
            o("        u8 *out%d = a.out[%d] + blockIdx.z * a.out_pitch[%d] + rr * a.out_step[%d];\n", c, c, c, c);
        else
            o("        u8 *out%d = nullptr;\n", c);
    }
    o("        const int xabs = a.x0 + p, yabs = a.y0 + rr;\n"
      "#define BODY(F, X) body<F, X>(a, in0, in1, in2, in3, out0, out1, out2, out3, p, n, xabs, yabs)\n"
      "        if (n == V) { if (xal) BODY(true, true); else BODY(true, false); }\n"
      "        else        { if (xal) BODY(false, true); else BODY(false, false); }\n"
      "    }\n"
      "}\n");
    pl.src = std::move(o.s);
    return 0;
}

/* ---- hiprtc + module cache --------------------------------------------------------------------------------------------------- */
struct Program {
    std::vector<char> code;          /* the code object: per architecture, shared by every device of that architecture */
    hipModule_t mod[64] = {};        /* loaded per device, on the device's first use of the program */
    hipFunction_t fn[64] = {};
};
std::mutex g_cache_mutex;
std::unordered_map<std::string, std::shared_ptr<Program>> g_cache; /* by arch + program text; programs live as long as the process */

/* the architecture the current device wants ("gfx950"; feature suffixes such as ":sramecc+:xnack-" dropped) */
std::string device_arch(void)
{
    hipDeviceProp_t pr;
    if (ffhip_have_device() && hipGetDeviceProperties(&pr, ffhip_current_device()) == hipSuccess && pr.gcnArchName[0]) {
        std::string a(pr.gcnArchName);
        const size_t c = a.find(':');
        return c == std::string::npos ? a : a.substr(0, c);
    }
    return "gfx950"; /* no device: ffhip_sws_uops_check() only wants to know that the list compiles */
}

/*
 * The on-disk cache of code objects (round 4): a new op list costs a ~25 ms hiprtc compile once per MACHINE, not once per process.
 * Directory: what the application passed to ffhip_sws_uops_set_cache_dir() (off until it does: the library itself reads no
 * environment variable; the FFmpeg-side backend picks $XDG_CACHE_HOME/ffhip or $HOME/.cache/ffhip, integration/swscale_hw_hip.c).
 * A file is named by a 128-bit FNV-1a hash of its key — this file's format tag, the hiprtc version, the architecture and the
 * program text — and carries the whole key in front of the code object, so a hash collision or a file of another compiler version
 * reads as a miss; files are written to a temporary name and renamed, so a concurrent reader sees a whole file or none.
 * Trust (round 5): the directory must be a real directory owned by the effective user and closed to group / others (else the cache
 * stays off), files are opened O_NOFOLLOW and must be the user's own regular files, temporaries come from mkstemp(), and the header
 * carries a hash of the code object that is verified before the code is handed to the runtime.
 */
std::string g_disk_dir; /* guarded by g_cache_mutex; empty = off (the default: the library reads no environment variable) */

std::string disk_key(const std::string &arch, const std::string &src)
{
    int maj = 0, min = 0;
    (void)hiprtcVersion(&maj, &min);
    char head[96];
    snprintf(head, sizeof(head), "ffhip-uops-2 hiprtc %d.%d %s\n", maj, min, arch.c_str());
    return head + src;
}

std::string disk_path(const std::string &dir, const std::string &key)
{
    uint64_t h[2] = { 0xcbf29ce484222325ull, 0x84222325cbf29ce4ull };
    for (unsigned char c : key) {
        h[0] = (h[0] ^ c) * 0x100000001b3ull;
        h[1] = (h[1] ^ (c + 0x9e)) * 0x100000001b3ull + (h[0] >> 29);
    }
    char name[64];
    snprintf(name, sizeof(name), "/%016llx%016llx.hsaco", (unsigned long long)h[0], (unsigned long long)h[1]);
    return dir + name;
}

/* 64-bit FNV-1a of the code object: stored in the file's header, verified on load (a truncated, bit-rotten or tampered-with file
 * reads as a miss — the code then comes from hiprtc as if the cache were cold) */
uint64_t disk_sum(const char *p, size_t n)
{
    uint64_t h = 0xcbf29ce484222325ull;
    for (size_t i = 0; i < n; i++)
        h = (h ^ (unsigned char)p[i]) * 0x100000001b3ull;
    return h;
}

/* The directory is trusted only when it is OURS: a real directory (no symlink in the last component), owned by the effective user, not
 * writable by group or others.  Anything else — a shared /tmp-style location, a directory someone else made first — leaves the cache
 * off: code objects loaded from it run with the process's access to all of its device memory. */
bool disk_dir_trusted(const std::string &dir)
{
    struct stat st;
    if (lstat(dir.c_str(), &st) != 0 || !S_ISDIR(st.st_mode))
        return false;
    return st.st_uid == geteuid() && !(st.st_mode & (S_IWGRP | S_IWOTH));
}

bool disk_load(const std::string &dir, const std::string &path, const std::string &key, std::vector<char> *code)
{
    if (!disk_dir_trusted(dir))
        return false;
    const int fd = open(path.c_str(), O_RDONLY | O_NOFOLLOW | O_CLOEXEC);
    if (fd < 0)
        return false;
    struct stat st;
    if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode) || st.st_uid != geteuid()) {
        close(fd);
        return false;
    }
    FILE *f = fdopen(fd, "rb");
    if (!f) {
        close(fd);
        return false;
    }
    bool ok = false;
    uint64_t klen = 0, clen = 0, sum = 0;
    if (fread(&klen, 8, 1, f) == 1 && fread(&clen, 8, 1, f) == 1 && fread(&sum, 8, 1, f) == 1 && klen == key.size() && clen > 0 && clen < (1u << 28)) {
        std::string k(klen, '\0');
        code->resize(clen);
        ok = fread(&k[0], 1, klen, f) == klen && k == key && fread(code->data(), 1, clen, f) == clen && fgetc(f) == EOF &&
             disk_sum(code->data(), clen) == sum;
    }
    fclose(f);
    if (!ok)
        code->clear();
    return ok;
}

void disk_store(const std::string &dir, const std::string &path, const std::string &key, const std::vector<char> &code)
{
    /* mkdir -p of the last two components (…/.cache may not exist yet); failures just leave the cache cold.  mkdir(0700) does nothing
     * to a directory that exists already: whether it is ours is checked afterwards */
    const size_t cut = dir.rfind('/');
    if (cut != std::string::npos && cut > 0)
        (void)mkdir(dir.substr(0, cut).c_str(), 0700);
    (void)mkdir(dir.c_str(), 0700);
    if (!disk_dir_trusted(dir))
        return;
    /* a fresh, exclusively created temporary file in that directory (no predictable name, no symlink followed), then rename */
    std::string t = dir + "/.ffhip-XXXXXX";
    const int fd = mkstemp(&t[0]);
    if (fd < 0)
        return;
    FILE *f = fdopen(fd, "wb");
    if (!f) {
        close(fd);
        (void)remove(t.c_str());
        return;
    }
    const uint64_t klen = key.size(), clen = code.size(), sum = disk_sum(code.data(), code.size());
    const bool ok = fwrite(&klen, 8, 1, f) == 1 && fwrite(&clen, 8, 1, f) == 1 && fwrite(&sum, 8, 1, f) == 1 &&
                    fwrite(key.data(), 1, klen, f) == klen && fwrite(code.data(), 1, clen, f) == clen;
    if (fclose(f) != 0 || !ok || rename(t.c_str(), path.c_str()) != 0)
        (void)remove(t.c_str());
}

std::atomic<long> g_compiles{0}, g_disk_hits{0};

int build(const std::string &src, std::shared_ptr<Program> *out, bool load)
{
    const std::string arch = device_arch();
    const std::string key = arch + "\n" + src;
    std::lock_guard<std::mutex> lk(g_cache_mutex);
    std::shared_ptr<Program> &pr = g_cache[key];
    const std::string ddir = pr ? std::string() : g_disk_dir;
    const std::string dkey = ddir.empty() ? std::string() : disk_key(arch, src);
    const std::string dpath = ddir.empty() ? std::string() : disk_path(ddir, dkey);
    if (!pr && !ddir.empty()) {
        auto np = std::make_shared<Program>();
        if (disk_load(ddir, dpath, dkey, &np->code)) {
            g_disk_hits++;
            pr = np;
        }
    }
    if (!pr) {
        g_compiles++;
        hiprtcProgram prog;
        if (hiprtcCreateProgram(&prog, src.c_str(), "sws_uops.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS) {
            ffhip_set_error("hiprtcCreateProgram failed");
            g_cache.erase(key);
            return FFHIP_EIO;
        }
        const std::string archopt = "--offload-arch=" + arch;
        const char *opts[] = { archopt.c_str(), "-O3", "-ffp-contract=off", "-std=c++17" };
        const hiprtcResult rc = hiprtcCompileProgram(prog, 4, opts);
        if (rc != HIPRTC_SUCCESS) {
            size_t ls = 0;
            hiprtcGetProgramLogSize(prog, &ls);
            std::string log(ls + 1, '\0');
            hiprtcGetProgramLog(prog, &log[0]);
            ffhip_set_error("hiprtc: %s\n%.1500s", hiprtcGetErrorString(rc), log.c_str());
            hiprtcDestroyProgram(&prog);
            g_cache.erase(key);
            return FFHIP_EIO;
        }
        auto np = std::make_shared<Program>();
        size_t cs = 0;
        hiprtcGetCodeSize(prog, &cs);
        np->code.resize(cs);
        hiprtcGetCode(prog, np->code.data());
        hiprtcDestroyProgram(&prog);
        pr = np;
        if (!ddir.empty())
            disk_store(ddir, dpath, dkey, np->code);
    }
    if (load) {
        if (!ffhip_have_device())
            return FFHIP_ENOSYS;
        const int d = ffhip_current_device();
        if (!pr->fn[d]) {
            HIP_TRY(hipModuleLoadData(&pr->mod[d], pr->code.data()));
            HIP_TRY(hipModuleGetFunction(&pr->fn[d], pr->mod[d], "sws_uops"));
        }
    }
    *out = pr;
    return 0;
}

} // namespace

/* a small table a launch reads (line skips / filter offsets).  Content-addressed and never overwritten: launches of the same
 * compiled list on different streams with different tables each keep reading their own copy. */
struct DevTable {
    std::vector<int32_t> host;
    int32_t *dev = nullptr;
};
struct FFHipSwsUOps {
    int device = 0; /* the list's constants, tables and loaded module live here; every call makes it current */
    Plan plan;
    std::shared_ptr<Program> prog;
    void *data[MAXDATA] = {};
    FFHipSwsOpFunc fb_func = nullptr;
    const void *fb_priv = nullptr;
    std::mutex tab_mutex;
    std::vector<std::unique_ptr<DevTable>> tabs;
};

extern "C" int ffhip_sws_uops_source(const FFHipSwsUOp *uops, int num_uops, char *buf, size_t size)
{
    Plan pl;
    const int r = generate(uops, num_uops, pl);
    if (r < 0)
        return r;
    if (buf && size) {
        const size_t n = pl.src.size() < size - 1 ? pl.src.size() : size - 1;
        memcpy(buf, pl.src.data(), n);
        buf[n] = 0;
    }
    return (int)pl.src.size();
}

extern "C" int ffhip_sws_uops_check(const FFHipSwsUOp *uops, int num_uops)
{
    Plan pl;
    const int r = generate(uops, num_uops, pl);
    if (r < 0)
        return r;
    std::shared_ptr<Program> pr;
    return build(pl.src, &pr, false);
}

extern "C" int ffhip_sws_uops_set_cache_dir(const char *dir)
{
    std::lock_guard<std::mutex> lk(g_cache_mutex);
    g_disk_dir = dir ? dir : "";
    return 0;
}

extern "C" void ffhip_sws_uops_cache_stats(long *compiles, long *disk_hits)
{
    if (compiles)
        *compiles = g_compiles.load();
    if (disk_hits)
        *disk_hits = g_disk_hits.load();
}

extern "C" void ffhip_sws_uops_free(FFHipSwsUOps **pp)
{
    if (!pp || !*pp)
        return;
    FFHipSwsUOps *p = *pp;
    FFHipDeviceGuard dg(p->device);
    for (void *d : p->data)
        if (d)
            (void)hipFree(d);
    for (auto &t : p->tabs)
        if (t->dev)
            (void)hipFree(t->dev);
    delete p;
    *pp = nullptr;
}

extern "C" int ffhip_sws_uops_compile(const FFHipSwsUOp *uops, int num_uops, FFHipSwsUOps **out)
{
    if (!out)
        return FFHIP_EINVAL;
    *out = nullptr;
    std::unique_ptr<FFHipSwsUOps> p(new (std::nothrow) FFHipSwsUOps);
    if (!p)
        return FFHIP_ENOMEM;
    int r = generate(uops, num_uops, p->plan);
    if (r < 0)
        return r;
    if (!ffhip_have_device()) {
        ffhip_set_error("sws uops: no HIP device");
        return FFHIP_ENOSYS;
    }
    p->device = ffhip_current_device();
    if ((r = build(p->plan.src, &p->prog, true)) < 0)
        return r;
    FFHipSwsUOps *raw = p.release();
    for (size_t i = 0; i < raw->plan.data.size(); i++) {
        const std::vector<uint8_t> &b = raw->plan.data[i];
        if (hipMalloc(&raw->data[i], b.size() ? b.size() : 4) != hipSuccess ||
            hipMemcpy(raw->data[i], b.data(), b.size(), hipMemcpyHostToDevice) != hipSuccess) {
            ffhip_set_error("sws uops: uploading %zu bytes of constants failed", b.size());
            ffhip_sws_uops_free(&raw);
            return FFHIP_ENOMEM;
        }
    }
    *out = raw;
    return 0;
}

extern "C" int ffhip_sws_uops_block_size(const FFHipSwsUOps *p) { return p ? p->plan.block_size : FFHIP_EINVAL; }

extern "C" void ffhip_sws_uops_set_fallback(FFHipSwsUOps *p, FFHipSwsOpFunc func, const void *priv)
{
    if (p) {
        p->fb_func = func;
        p->fb_priv = priv;
    }
}

namespace {

/* fills everything of KArgs that does not depend on where the planes live */
int geometry(FFHipSwsUOps *p, const FFHipSwsOpExec *e, int bx_start, int y_start, int bx_end, int y_end, KArgs &a, std::vector<int32_t> &rowtab)
{
    const Plan &pl = p->plan;
    if (!e || bx_end < bx_start || y_end < y_start || bx_start < 0)
        BAD("sws uops: blocks %d..%d, lines %d..%d", bx_start, bx_end, y_start, y_end);
    memset(&a, 0, sizeof a);
    a.x0 = bx_start * pl.block_size;
    a.npx = (bx_end - bx_start) * pl.block_size;
    a.y0 = y_start;
    a.ny = y_end - y_start;
    for (int i = 0; i < 4; i++) {
        a.in_stride[i] = e->in_stride[i];
        a.in_step[i]  = e->in_bump[i]  + ((long)a.npx * pl.adv_in[i]  >> 3);
        a.out_step[i] = e->out_bump[i] + ((long)a.npx * pl.adv_out[i] >> 3);
    }
    rowtab.clear();
    if (pl.fv_size && e->in_bump_y) { /* uops_backend.c:122-127: after line y the pointers move in_bump_y[y] lines further */
        rowtab.resize(a.ny);
        int32_t acc = 0;
        for (int r = 0; r < a.ny; r++) {
            rowtab[r] = acc;
            acc += e->in_bump_y[y_start + r];
        }
    }
    if (pl.fh_size && !e->in_offset_x)
        BAD("sws uops: a horizontally filtered read needs SwsOpExec.in_offset_x");
    return 0;
}

int launch(FFHipSwsUOps *p, KArgs &a, int nframes, hipStream_t st)
{
    for (int i = 0; i < MAXDATA; i++)
        a.data[i] = p->data[i];
    if (a.npx <= 0 || a.ny <= 0 || nframes <= 0)
        return 0;
    const int groups = cdiv(a.npx, p->plan.V);
    /* lines per thread: a wave that converts 256 pixels and retires is mostly launch and address set-up; 8 lines per thread keeps
     * >= 4 workgroups per CU in flight on pictures from SD up, smaller jobs fall back to fewer lines */
    static const int rows_env = FFHIP_KNOB("FFHIP_UOPS_ROWS") ? atoi(FFHIP_KNOB("FFHIP_UOPS_ROWS")) : 0;
    int rows = rows_env > 0 ? rows_env : 8;
    while (rows > 1 && (long)cdiv(groups, 64) * cdiv(a.ny, 4 * rows) * nframes < 2048)
        rows >>= 1;
    unsigned gy = cdiv(a.ny, 4 * rows);
    if (gy > 16384)
        gy = 16384;
    size_t sz = sizeof a;
    void *cfg[] = { HIP_LAUNCH_PARAM_BUFFER_POINTER, &a, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END };
    HIP_TRY(hipModuleLaunchKernel(p->prog->fn[p->device], cdiv(groups, 64), gy, nframes, 64, 4, 1, 0, st, nullptr, cfg));
    return 0;
}

/* the device copy of a table with this content (under p->tab_mutex).  A new content gets a NEW allocation — tables in use by
 * launches in flight on any stream are never touched; past 32 distinct tables the device is drained once and the set restarts. */
int table(FFHipSwsUOps *p, const int32_t *src, size_t n, hipStream_t st, int32_t **dev)
{
    for (auto &t : p->tabs)
        if (t->host.size() == n && !memcmp(t->host.data(), src, n * sizeof(int32_t))) {
            *dev = t->dev;
            return 0;
        }
    if (p->tabs.size() >= 32) {
        HIP_TRY(hipDeviceSynchronize());
        for (auto &t : p->tabs)
            (void)hipFree(t->dev);
        p->tabs.clear();
    }
    std::unique_ptr<DevTable> t(new DevTable);
    t->host.assign(src, src + n);
    HIP_TRY(hipMalloc(&t->dev, (n ? n : 1) * sizeof(int32_t)));
    const hipError_t e = hipMemcpyAsync(t->dev, t->host.data(), n * sizeof(int32_t), hipMemcpyHostToDevice, st);
    if (e != hipSuccess) {
        (void)hipFree(t->dev);
        ffhip_set_error("sws uops: table upload failed: %s", hipGetErrorString(e));
        return FFHIP_EIO;
    }
    *dev = t->dev;
    p->tabs.push_back(std::move(t));
    return 0;
}

} // namespace

extern "C" int ffhip_sws_uops_run_dev(FFHipSwsUOps *p, const FFHipSwsOpExec *e, int bx_start, int y_start, int bx_end, int y_end,
                                      int nframes, const ptrdiff_t *in_frame_pitch, const ptrdiff_t *out_frame_pitch, void *stream)
{
    if (!p || !p->prog || !p->prog->fn[p->device])
        return FFHIP_EINVAL;
    FFHipDeviceGuard dg(p->device);
    if (nframes < 1 || (nframes > 1 && (!in_frame_pitch || !out_frame_pitch)))
        BAD("sws uops: %d pictures without frame pitches", nframes);
    hipStream_t st = static_cast<hipStream_t>(stream);
    KArgs a;
    std::vector<int32_t> rowtab;
    int r = geometry(p, e, bx_start, y_start, bx_end, y_end, a, rowtab);
    if (r < 0)
        return r;
    std::lock_guard<std::mutex> lk(p->tab_mutex);
    if (!rowtab.empty()) {
        int32_t *d = nullptr;
        if ((r = table(p, rowtab.data(), rowtab.size(), st, &d)) < 0)
            return r;
        a.rowtab = d;
    }
    if (p->plan.fh_size) {
        int32_t *d = nullptr;
        if ((r = table(p, e->in_offset_x, (size_t)a.x0 + a.npx, st, &d)) < 0)
            return r;
        a.offx = d;
    }
    for (int i = 0; i < 4; i++) {
        a.in[i] = e->in[i];
        a.out[i] = e->out[i];
        a.in_pitch[i] = in_frame_pitch ? in_frame_pitch[i] : 0;
        a.out_pitch[i] = out_frame_pitch ? out_frame_pitch[i] : 0;
    }
    return launch(p, a, nframes, st);
}

/* ---- SwsOpFunc on host memory ------------------------------------------------------------------------------------------------ */
namespace {

struct Span { long lo, hi; };   /* bytes of a plane a call touches, relative to exec->in[i] / out[i] */

bool host_run(FFHipSwsUOps *p, const FFHipSwsOpExec *e, int bx_start, int y_start, int bx_end, int y_end)
{
    const Plan &pl = p->plan;
    FFHipDeviceGuard dg(p->device);
    KArgs a;
    std::vector<int32_t> rowtab;
    if (geometry(p, e, bx_start, y_start, bx_end, y_end, a, rowtab) < 0)
        return false;
    if (a.npx <= 0 || a.ny <= 0)
        return true;
    /* per plane: bytes per line and the lines, as the C backend walks them */
    long in_row[4] = {}, out_row[4] = {}, in_lines[4] = {};
    const long last_skip = rowtab.empty() ? 0 : rowtab.back();
    for (int i = 0; i < 4; i++) {
        if (pl.adv_in[i])
            in_row[i] = ((long)a.npx * pl.adv_in[i] + 7) >> 3;
        out_row[i] = ((long)a.npx * pl.adv_out[i] + 7) >> 3;
    }
    std::vector<int32_t> offx;
    if (pl.fh_size) { /* the filtered read does not move the pointers: a line spans the taps of the samples of this call */
        long hi = 0;
        for (int x = a.x0; x < a.x0 + a.npx; x++)
            hi = e->in_offset_x[x] + (long)pl.fh_size * pl.fh_elem > hi ? e->in_offset_x[x] + (long)pl.fh_size * pl.fh_elem : hi;
        for (int i = 0; i < 4; i++)
            in_row[i] = (pl.fh_mask >> i & 1) ? hi : 0;
    }
    /* negative steps (flipped pictures) and steps smaller than a line are the caller's business in the C backend too; the staging
     * below needs lines that do not overlap */
    size_t need = 0;
    long in_off[4], out_off[4];
    for (int i = 0; i < 4; i++) {
        in_lines[i] = in_row[i] ? a.ny : 0;
        in_off[i] = (long)need;
        if (in_row[i]) {
            if (pl.fv_size) { /* tap rows are in_stride apart; the last line of the call reaches last_skip + ny - 1 + taps - 1 */
                if (a.in_step[i] != a.in_stride[i])
                    return false;
                in_lines[i] = a.ny + last_skip + pl.fv_size - 1;
            }
            if (a.in_step[i] < in_row[i] && in_lines[i] > 1)
                return false;
            need += (size_t)(in_row[i] + 15 & ~15L) * in_lines[i];
        }
    }
    size_t pal_off = 0;
    if (pl.palette) {
        pal_off = need;
        need += 1024;
    }
    for (int i = 0; i < 4; i++) {
        out_off[i] = (long)need;
        if (out_row[i]) {
            if (a.out_step[i] < out_row[i] && a.ny > 1)
                return false;
            need += (size_t)(out_row[i] + 15 & ~15L) * a.ny;
        }
    }
    const size_t tabs_off = need;
    need += (rowtab.size() + (pl.fh_size ? (size_t)a.x0 + a.npx : 0)) * sizeof(int32_t) + 64;

    Arena A(need);
    if (!A.ok)
        return false;
    for (int i = 0; i < 4; i++) {
        if (!in_row[i])
            continue;
        const long pitch = in_row[i] + 15 & ~15L;
        if (hipMemcpy2D(A.buf + in_off[i], pitch, e->in[i], a.in_step[i], in_row[i], in_lines[i], hipMemcpyHostToDevice) != hipSuccess)
            return false;
        a.in[i] = A.buf + in_off[i];
        a.in_step[i] = pitch;
        a.in_stride[i] = pitch;
    }
    if (pl.palette) {
        if (hipMemcpy(A.buf + pal_off, e->in[1], 1024, hipMemcpyHostToDevice) != hipSuccess)
            return false;
        a.in[1] = A.buf + pal_off;
        a.in_step[1] = a.in_stride[1] = 0;
    }
    for (int i = 0; i < 4; i++) {
        if (!out_row[i])
            continue;
        const long pitch = out_row[i] + 15 & ~15L;
        /* a write that leaves elements of a pixel alone (packed write with a partial mask) needs what the buffer held */
        if (hipMemcpy2D(A.buf + out_off[i], pitch, e->out[i], a.out_step[i], out_row[i], a.ny, hipMemcpyHostToDevice) != hipSuccess)
            return false;
        a.out[i] = A.buf + out_off[i];
        a.out_step[i] = pitch;
    }
    int32_t *tabs = reinterpret_cast<int32_t *>(A.buf + (tabs_off + 63 & ~(size_t)63));
    if (!rowtab.empty()) {
        if (hipMemcpy(tabs, rowtab.data(), rowtab.size() * sizeof(int32_t), hipMemcpyHostToDevice) != hipSuccess)
            return false;
        a.rowtab = tabs;
        tabs += rowtab.size();
    }
    if (pl.fh_size) {
        if (hipMemcpy(tabs, e->in_offset_x, ((size_t)a.x0 + a.npx) * sizeof(int32_t), hipMemcpyHostToDevice) != hipSuccess)
            return false;
        a.offx = tabs;
    }
    if (launch(p, a, 1, 0) < 0 || !A.down())
        return false;
    for (int i = 0; i < 4; i++) {
        if (!out_row[i])
            continue;
        const long pitch = out_row[i] + 15 & ~15L;
        const uint8_t *s = A.host(A.buf + out_off[i]);
        uint8_t *d = e->out[i];
        for (int r = 0; r < a.ny; r++, s += pitch, d += e->out_bump[i] + ((long)a.npx * pl.adv_out[i] >> 3))
            memcpy(d, s, out_row[i]);
    }
    return true;
}

} // namespace

extern "C" void ffhip_sws_uops_func(const FFHipSwsOpExec *exec, const void *priv, int bx_start, int y_start, int bx_end, int y_end)
{
    FFHipSwsUOps *p = const_cast<FFHipSwsUOps *>(static_cast<const FFHipSwsUOps *>(priv));
    if (p && host_run(p, exec, bx_start, y_start, bx_end, y_end))
        return;
    const bool have = p && p->fb_func;
    shim_note("sws_uops_func", have);
    if (have)
        p->fb_func(exec, p->fb_priv, bx_start, y_start, bx_end, y_end);
}
