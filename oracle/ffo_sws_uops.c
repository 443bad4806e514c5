/*
 * ffo_sws_uops.c — CPU restatement of libswscale's micro-op semantics.
 *
 * TEST INFRASTRUCTURE ONLY (see ffo.h).  One pixel at a time, four component registers, the list interpreted in order: what
 * libswscale/uops_tmpl.c defines per block of 32 pixels and libswscale/uops_backend.c:104-131 walks over a slice.  Every case
 * cites the template function it restates.  Compiled with -ffp-contract=off like the reference backend (uops_backend.c:24-35).
 * Pinned against backend_c itself (oracle/_ref) in tests/test_oracle_vs_ref.py: per micro-op on checkasm's shapes
 * (tests/checkasm/sw_ops.c) and end to end as the backend of the reference's own graph.
 *
 * The structs are the boundary's (include/ffhip.h: layout-identical to SwsUOp / SwsOpExec), and so are the five entry points: the
 * oracle can stand where libffhip stands under the reference's dispatch layer.
 */
#include <stdlib.h>
#include <string.h>

#include "../include/ffhip.h"
#include "ffo.h"

typedef struct FfoSwsUOps {
    FFHipSwsUOp *uops;
    int n;
    int block_size;
} FfoSwsUOps;

void ffo_sws_uops_free(FfoSwsUOps **pp);

/* test knob: report (and walk) blocks of this many pixels instead of the smallest legal block, to drive the caller's dispatcher
 * down the paths a 32-pixel backend takes */
static int force_block;
void ffo_sws_uops_force_block(int pixels) { force_block = pixels; }

static int px_size(int type) { return type == FFHIP_SWS_PIXEL_U8 ? 1 : type == FFHIP_SWS_PIXEL_U16 ? 2 : 4; }

static int dither_height(const FFHipSwsUOp *u)
{   /* ff_sws_dither_height (uops.c:232-238): the matrix is padded by the largest row offset */
    int mo = 0;
    for (int c = 0; c < 4; c++)
        if (u->par.dither.y_offset[c] > mo)
            mo = u->par.dither.y_offset[c];
    return (1 << u->par.dither.size_log2) + mo;
}

int ffo_sws_uops_compile(const FFHipSwsUOp *uops, int n, FfoSwsUOps **out)
{
    FfoSwsUOps *p = calloc(1, sizeof(*p));
    if (!p || n < 1)
        return -12;
    p->uops = calloc(n, sizeof(*uops));
    memcpy(p->uops, uops, n * sizeof(*uops));
    p->n = n;
    p->block_size = 1;
    for (int i = 0; i < n; i++) {
        FFHipSwsUOp *u = &p->uops[i];
        switch (u->uop) {
        case FFHIP_SWS_UOP_READ_BIT: case FFHIP_SWS_UOP_WRITE_BIT: p->block_size = 8; break;
        case FFHIP_SWS_UOP_READ_NIBBLE: case FFHIP_SWS_UOP_WRITE_NIBBLE: if (p->block_size < 2) p->block_size = 2; break;
        case FFHIP_SWS_UOP_READ_PLANAR_FH: case FFHIP_SWS_UOP_READ_PLANAR_FV: {   /* the list owns a copy of the kernel */
            const FFHipSwsFilterWeights *f = u->data.kernel;
            FFHipSwsFilterWeights *k = malloc(sizeof(*k));
            *k = *f;
            k->weights = malloc(sizeof(int) * f->dst_size * f->filter_size);
            memcpy(k->weights, f->weights, sizeof(int) * f->dst_size * f->filter_size);
            k->offsets = NULL;
            u->data.kernel = k;
            break;
        }
        case FFHIP_SWS_UOP_DITHER: {
            const size_t bytes = sizeof(FFHipSwsPixel) * (1 << u->par.dither.size_log2) * dither_height(u);
            FFHipSwsPixel *m = malloc(bytes);
            memcpy(m, u->data.ptr, bytes);
            u->data.ptr = m;
            break;
        }
        case FFHIP_SWS_UOP_LUT_3D: case FFHIP_SWS_UOP_RW_SHUFFLE: case FFHIP_SWS_UOP_LINEAR_FMA: case FFHIP_SWS_UOP_READ_PLANAR_FV_FMA:
            p->n = i;   /* what was copied so far is what free releases */
            ffo_sws_uops_free(&p);
            return -95;
        }
    }
    if (force_block)
        p->block_size = force_block;
    *out = p;
    return 0;
}

void ffo_sws_uops_free(FfoSwsUOps **pp)
{
    FfoSwsUOps *p = pp ? *pp : NULL;
    if (!p)
        return;
    for (int i = 0; i < p->n; i++) {
        FFHipSwsUOp *u = &p->uops[i];
        if (u->uop == FFHIP_SWS_UOP_READ_PLANAR_FH || u->uop == FFHIP_SWS_UOP_READ_PLANAR_FV) {
            free(u->data.kernel->weights);
            free(u->data.kernel);
        } else if (u->uop == FFHIP_SWS_UOP_DITHER) {
            free(u->data.ptr);
        }
    }
    free(p->uops);
    free(p);
    *pp = NULL;
}

int ffo_sws_uops_block_size(const FfoSwsUOps *p) { return p->block_size; }
void ffo_sws_uops_set_fallback(FfoSwsUOps *p, FFHipSwsOpFunc f, const void *priv) { (void)p; (void)f; (void)priv; }

/* a register holds the raw bits of a pixel_t; these are the views of it */
typedef union Reg { uint8_t u8; uint16_t u16; uint32_t u32; float f32; } Reg;

/* the per-type micro-ops; `px` is the union member, `T` the C type of pixel_t, `I` inter_t of the horizontal filter
 * (uops_tmpl.c:26-60) */
#define TYPED_OPS(NAME, T, px, I, MAXV)                                                                                     \
static void ops_##NAME(const FFHipSwsUOp *u, Reg r[4], Reg *tmp, const uint8_t *const in[4], const FFHipSwsOpExec *e,        \
                       long p, int xabs, int yabs)                                                                           \
{                                                                                                                            \
    (void)tmp; (void)in; (void)e; (void)p; (void)xabs; (void)yabs;                                                            \
    const int m = u->mask;                                                                                                   \
    switch (u->uop) {                                                                                                        \
    case FFHIP_SWS_UOP_READ_PLANAR:                          /* read_planar, uops_tmpl.c:66-81 */                            \
        for (int c = 0; c < 4; c++) if (m >> c & 1) memcpy(&r[c].px, in[c] + p * sizeof(T), sizeof(T));                       \
        break;                                                                                                               \
    case FFHIP_SWS_UOP_READ_PACKED: {                        /* read_packed, :83-96 */                                       \
        const int el = (m & 8) ? 4 : (m & 4) ? 3 : (m & 2) ? 2 : 1;                                                          \
        for (int c = 0; c < el; c++) if (m >> c & 1) memcpy(&r[c].px, in[0] + (p * el + c) * sizeof(T), sizeof(T));           \
        break;                                                                                                               \
    }                                                                                                                        \
    case FFHIP_SWS_UOP_READ_PLANAR_FH: {                     /* read_planar_fh, :316-350 */                                  \
        const FFHipSwsFilterWeights *f = u->data.kernel;                                                                     \
        const int *w = f->weights + (long)f->filter_size * xabs;                                                             \
        const float scale = 1.0f / FFHIP_SWS_FILTER_SCALE;                                                                   \
        for (int c = 0; c < 4; c++) if (m >> c & 1) {                                                                        \
            const uint8_t *s = in[c] + e->in_offset_x[xabs];                                                                 \
            I acc = 0;                                                                                                       \
            for (int j = 0; j < f->filter_size; j++) { T t; memcpy(&t, s + j * sizeof(T), sizeof(T)); acc += w[j] * t; }      \
            r[c].f32 = (float)acc * scale;                                                                                   \
        }                                                                                                                    \
        break;                                                                                                               \
    }                                                                                                                        \
    case FFHIP_SWS_UOP_READ_PLANAR_FV: {                     /* setup_filter_v + read_planar_fv, :247-297 */                 \
        const FFHipSwsFilterWeights *f = u->data.kernel;                                                                     \
        const int *w = f->weights + (long)f->filter_size * yabs;                                                             \
        for (int c = 0; c < 4; c++) if (m >> c & 1) {                                                                        \
            const uint8_t *s = in[c] + p * sizeof(T);                                                                        \
            float acc = 0.0f;                                                                                                \
            for (int j = 0; j < f->filter_size; j++, s += e->in_stride[c]) {                                                 \
                const float weight = (float)w[j] / FFHIP_SWS_FILTER_SCALE;                                                   \
                T t; memcpy(&t, s, sizeof(T));                                                                               \
                acc += weight * t;                                                                                           \
            }                                                                                                                \
            r[c].f32 = acc;                                                                                                  \
        }                                                                                                                    \
        break;                                                                                                               \
    }                                                                                                                        \
    case FFHIP_SWS_UOP_PERMUTE: case FFHIP_SWS_UOP_COPY:     /* permute / copy, :357-407: sequential moves, -1 = temporary */ \
        for (int n = 0; n < u->par.move.num_moves; n++) {                                                                    \
            Reg *d = u->par.move.dst[n] < 0 ? tmp : &r[u->par.move.dst[n]];                                                  \
            *d = u->par.move.src[n] < 0 ? *tmp : r[u->par.move.src[n]];                                                      \
        }                                                                                                                    \
        break;                                                                                                               \
    case FFHIP_SWS_UOP_TO_U8:  for (int c = 0; c < 4; c++) if (m >> c & 1) { const T x = r[c].px; r[c].u32 = 0; r[c].u8  = x; } break; /* DECL_CAST, :417-440 */ \
    case FFHIP_SWS_UOP_TO_U16: for (int c = 0; c < 4; c++) if (m >> c & 1) { const T x = r[c].px; r[c].u32 = 0; r[c].u16 = x; } break; \
    case FFHIP_SWS_UOP_TO_U32: for (int c = 0; c < 4; c++) if (m >> c & 1) { const T x = r[c].px; r[c].u32 = x; } break;      \
    case FFHIP_SWS_UOP_TO_F32: for (int c = 0; c < 4; c++) if (m >> c & 1) { const T x = r[c].px; r[c].f32 = x; } break;      \
    case FFHIP_SWS_UOP_SCALE:  for (int c = 0; c < 4; c++) if (m >> c & 1) r[c].px *= u->data.scalar.px; break;              /* :640-653 */ \
    case FFHIP_SWS_UOP_ADD:    for (int c = 0; c < 4; c++) if (m >> c & 1) r[c].px += u->data.vec4[c].px; break;             /* :655-666 */ \
    case FFHIP_SWS_UOP_MIN:    for (int c = 0; c < 4; c++) if (m >> c & 1) { const T k = u->data.vec4[c].px; r[c].px = r[c].px > k ? k : r[c].px; } break; /* FFMIN, :668-679 */ \
    case FFHIP_SWS_UOP_MAX:    for (int c = 0; c < 4; c++) if (m >> c & 1) { const T k = u->data.vec4[c].px; r[c].px = r[c].px > k ? r[c].px : k; } break; /* FFMAX, :681-692 */ \
    case FFHIP_SWS_UOP_CLEAR:                                /* clear, :614-634 */                                           \
        for (int c = 0; c < 4; c++) if (m >> c & 1) {                                                                        \
            const T k = (MAXV && (u->par.clear.one >> c & 1)) ? (T)MAXV : (u->par.clear.zero >> c & 1) ? (T)0 : u->data.vec4[c].px; \
            r[c].u32 = 0; r[c].px = k;                                                                                       \
        }                                                                                                                    \
        break;                                                                                                               \
    case FFHIP_SWS_UOP_DITHER: {                             /* dither, :737-765 */                                          \
        const int size = 1 << u->par.dither.size_log2;                                                                       \
        for (int c = 0; c < 4; c++) if (m >> c & 1)                                                                          \
            r[c].px += u->data.ptr[((yabs & (size - 1)) + u->par.dither.y_offset[c]) * size + (xabs & (size - 1))].px;       \
        break;                                                                                                               \
    }                                                                                                                        \
    case FFHIP_SWS_UOP_LINEAR: {                             /* linear, :795-830 */                                          \
        const T v[4] = { r[0].px, r[1].px, r[2].px, r[3].px };                                                               \
        for (int c = 0; c < 4; c++) if (m >> c & 1) {                                                                        \
            T acc = (u->par.lin.zero >> (5 * c + 4) & 1) ? (T)0 : u->data.mat4[c][4].px;                                     \
            for (int j = 0; j < 4; j++) {                                                                                    \
                if (u->par.lin.zero >> (5 * c + j) & 1) continue;                                                            \
                if (u->par.lin.one >> (5 * c + j) & 1) acc += v[j]; else acc += u->data.mat4[c][j].px * v[j];                \
            }                                                                                                                \
            r[c].px = acc;                                                                                                   \
        }                                                                                                                    \
        break;                                                                                                               \
    }                                                                                                                        \
    default: break;                                                                                                          \
    }                                                                                                                        \
}

/* integer-only micro-ops (uops_tmpl.c:446-612) */
#define INT_OPS(NAME, T, px, MAXV, SWAP)                                                                                     \
static void iops_##NAME(const FFHipSwsUOp *u, Reg r[4])                                                                      \
{                                                                                                                            \
    const int m = u->mask;                                                                                                   \
    const uint8_t *b = u->par.pack.pattern;                                                                                  \
    const int sh[4] = { b[3] + b[2] + b[1], b[3] + b[2], b[3], 0 };                                                          \
    switch (u->uop) {                                                                                                        \
    case FFHIP_SWS_UOP_LSHIFT: for (int c = 0; c < 4; c++) if (m >> c & 1) r[c].px <<= u->par.shift.amount; break;            \
    case FFHIP_SWS_UOP_RSHIFT: for (int c = 0; c < 4; c++) if (m >> c & 1) r[c].px >>= u->par.shift.amount; break;            \
    case FFHIP_SWS_UOP_SWAP_BYTES: for (int c = 0; c < 4; c++) if (m >> c & 1) r[c].px = SWAP(r[c].px); break;               \
    case FFHIP_SWS_UOP_EXPAND_BIT: for (int c = 0; c < 4; c++) if (m >> c & 1) r[c].px = r[c].px ? (T)MAXV : 0; break;        \
    case FFHIP_SWS_UOP_UNPACK: {                                                                                             \
        const T val = r[0].px;                                                                                               \
        for (int c = 0; c < 4; c++) if (m >> c & 1) { r[c].u32 = 0; r[c].px = (val >> sh[c]) & (T)((1 << b[c]) - 1); }       \
        break;                                                                                                               \
    }                                                                                                                        \
    case FFHIP_SWS_UOP_PACK: {                                                                                               \
        T val = 0;                                                                                                           \
        for (int c = 0; c < 4; c++) if (m >> c & 1) val |= r[c].px << sh[c];                                                 \
        r[0].px = val;                                                                                                       \
        break;                                                                                                               \
    }                                                                                                                        \
    default: break;                                                                                                          \
    }                                                                                                                        \
}

static uint8_t  noswap8(uint8_t x) { return x; }
static uint16_t swap16(uint16_t x) { return (uint16_t)(x << 8 | x >> 8); }
static uint32_t swap32(uint32_t x) { return __builtin_bswap32(x); }

TYPED_OPS(u8,  uint8_t,  u8,  int32_t, 0xFFu)
TYPED_OPS(u16, uint16_t, u16, int64_t, 0xFFFFu)
TYPED_OPS(u32, uint32_t, u32, int64_t, 0xFFFFFFFFu)
TYPED_OPS(f32, float,    f32, float,   0)
INT_OPS(u8,  uint8_t,  u8,  0xFFu,       noswap8)
INT_OPS(u16, uint16_t, u16, 0xFFFFu,     swap16)
INT_OPS(u32, uint32_t, u32, 0xFFFFFFFFu, swap32)

/* bits one pixel advances plane i of the read / the write by: what the template functions add to iter->in / iter->out */
static void advances(const FfoSwsUOps *p, int adv_in[4], int adv_out[4])
{
    memset(adv_in, 0, 4 * sizeof(int));
    memset(adv_out, 0, 4 * sizeof(int));
    const FFHipSwsUOp *rd = &p->uops[0], *wr = &p->uops[p->n - 1];
    const int rs = 8 * px_size(rd->type), ws = 8 * px_size(wr->type);
    const int rel = (rd->mask & 8) ? 4 : (rd->mask & 4) ? 3 : (rd->mask & 2) ? 2 : 1;
    const int wel = (wr->mask & 8) ? 4 : (wr->mask & 4) ? 3 : (wr->mask & 2) ? 2 : 1;
    switch (rd->uop) {
    case FFHIP_SWS_UOP_READ_PLANAR: case FFHIP_SWS_UOP_READ_PLANAR_FV:
        for (int c = 0; c < 4; c++) if (rd->mask >> c & 1) adv_in[c] = rs;
        break;
    case FFHIP_SWS_UOP_READ_PACKED:  adv_in[0] = rs * rel; break;
    case FFHIP_SWS_UOP_READ_NIBBLE:  adv_in[0] = 4; break;
    case FFHIP_SWS_UOP_READ_BIT:     adv_in[0] = 1; break;
    case FFHIP_SWS_UOP_READ_PALETTE: adv_in[0] = 8; break;
    default: break;                                       /* READ_PLANAR_FH leaves the pointers alone */
    }
    switch (wr->uop) {
    case FFHIP_SWS_UOP_WRITE_PLANAR:
        for (int c = 0; c < 4; c++) if (wr->mask >> c & 1) adv_out[c] = ws;
        break;
    case FFHIP_SWS_UOP_WRITE_PACKED: adv_out[0] = ws * wel; break;
    case FFHIP_SWS_UOP_WRITE_NIBBLE: adv_out[0] = 4; break;
    case FFHIP_SWS_UOP_WRITE_BIT:    adv_out[0] = 1; break;
    default: break;
    }
}

/* SwsOpFunc: process() of uops_backend.c:104-131 with the block loop opened up into pixels */
void ffo_sws_uops_func(const FFHipSwsOpExec *e, const void *priv, int bx_start, int y_start, int bx_end, int y_end)
{
    const FfoSwsUOps *P = priv;
    int adv_in[4], adv_out[4];
    advances(P, adv_in, adv_out);
    const long npx = (long)(bx_end - bx_start) * P->block_size;
    const int x0 = bx_start * P->block_size;
    const uint8_t *in[4];
    uint8_t *out[4];
    for (int i = 0; i < 4; i++) {
        in[i] = e->in[i];
        out[i] = e->out[i];
    }
    for (int y = y_start; y < y_end; y++) {
        for (long p = 0; p < npx; p++) {
            Reg r[4] = { { .u32 = 0 }, { .u32 = 0 }, { .u32 = 0 }, { .u32 = 0 } }, tmp = { .u32 = 0 };
            for (int k = 0; k < P->n; k++) {
                const FFHipSwsUOp *u = &P->uops[k];
                const int m = u->mask;
                switch (u->uop) {
                case FFHIP_SWS_UOP_READ_NIBBLE: { const uint8_t b = in[0][p >> 1]; r[0].u32 = (p & 1) ? (b & 0xF) : (b >> 4); continue; }   /* :150-163 */
                case FFHIP_SWS_UOP_READ_BIT:    { const uint8_t b = in[0][p >> 3]; r[0].u32 = (b >> (7 - (p & 7))) & 1; continue; }          /* :127-148 */
                case FFHIP_SWS_UOP_READ_PALETTE: {                                                                                         /* :165-182 */
                    const uint8_t *v = in[1] + 4 * in[0][p];
                    for (int c = 0; c < 4; c++) r[c].u32 = v[c];
                    continue;
                }
                case FFHIP_SWS_UOP_EXPAND_PAIR: for (int c = 0; c < 4; c++) if (m >> c & 1) { const uint8_t x = r[c].u8; r[c].u32 = 0; r[c].u16 = x << 8 | x; } continue; /* :528-541 */
                case FFHIP_SWS_UOP_EXPAND_QUAD: for (int c = 0; c < 4; c++) if (m >> c & 1) { const uint8_t x = r[c].u8; r[c].u32 = (uint32_t)x << 24 | x << 16 | x << 8 | x; } continue; /* :543-556 */
                case FFHIP_SWS_UOP_WRITE_PLANAR:                                                                                           /* :98-113 */
                    for (int c = 0; c < 4; c++) if (m >> c & 1) memcpy(out[c] + p * px_size(u->type), &r[c], px_size(u->type));
                    continue;
                case FFHIP_SWS_UOP_WRITE_PACKED: {                                                                                         /* :115-125 */
                    const int el = (m & 8) ? 4 : (m & 4) ? 3 : (m & 2) ? 2 : 1, s = px_size(u->type);
                    for (int c = 0; c < el; c++) if (m >> c & 1) memcpy(out[0] + (p * el + c) * s, &r[c], s);
                    continue;
                }
                case FFHIP_SWS_UOP_WRITE_NIBBLE:                                                                                           /* :210-222 */
                    if (p & 1) out[0][p >> 1] = (out[0][p >> 1] & 0xF0) | r[0].u8; else out[0][p >> 1] = (uint8_t)(r[0].u8 << 4);
                    continue;
                case FFHIP_SWS_UOP_WRITE_BIT:                                                                                              /* :192-208 */
                    if (!(p & 7)) out[0][p >> 3] = 0;
                    out[0][p >> 3] |= r[0].u8 << (7 - (p & 7));
                    continue;
                }
                switch (u->type) {
                case FFHIP_SWS_PIXEL_U8:  ops_u8(u, r, &tmp, in, e, p, x0 + (int)p, y);  iops_u8(u, r);  break;
                case FFHIP_SWS_PIXEL_U16: ops_u16(u, r, &tmp, in, e, p, x0 + (int)p, y); iops_u16(u, r); break;
                case FFHIP_SWS_PIXEL_U32: ops_u32(u, r, &tmp, in, e, p, x0 + (int)p, y); iops_u32(u, r); break;
                default:                  ops_f32(u, r, &tmp, in, e, p, x0 + (int)p, y); break;
                }
            }
        }
        const int y_bump = e->in_bump_y ? e->in_bump_y[y] : 0;
        for (int i = 0; i < 4; i++) {
            in[i]  += e->in_bump[i] + (npx * adv_in[i] >> 3) + y_bump * e->in_stride[i];
            out[i] += e->out_bump[i] + (npx * adv_out[i] >> 3);
        }
    }
}
