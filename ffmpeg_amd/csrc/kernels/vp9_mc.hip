/*
 * vp9_mc.hip — VP9 motion compensation, 8 bits, batched: VP9DSPContext.mc[size][filter][avg][!!mx][!!my]
 * (libavcodec/vp9dsp_template.c:1966-2293): the three 8-tap filter sets (each pass clip_u8((sum + 64) >> 7), the 2-D form through
 * 8-bit temporaries of rows -3..h+3), bilinear (a + ((m (b - a) + 8) >> 4)), full-pel copies; put and avg ((dst + v + 1) >> 1).
 *
 * Same shape as k_hevc_mc (hevc_mc.hip): one wave per block, a lane owns 4 adjacent samples of a row; the horizontal pass is two
 * v_dot4_i32_i8 per sample on bytes biased by -128 (every tap set sums to 128: the bias is 128 * 128 seeded into the accumulator),
 * windows cut with v_alignbyte from one 12-byte load; the rows the vertical pass needs — filtered and clipped, or raw — sit in
 * wave-private LDS as int16 pairs of vertically adjacent rows, 4 or 5 v_dot2_i32_i16 per sample; 16-column tiles keep the plane
 * small enough for 8 waves per SIMD.  Rows are read in whole dwords: a block's last group reads 1 byte beyond the right margin
 * the reference needs.
 */
#include "common.h"
#include "h264_kernels.h"

static_assert(sizeof(FFHipVp9McBlock) == 16, "FFHipVp9McBlock is a 16-byte record");
typedef short vm_s2 __attribute__((ext_vector_type(2)));

/* generated from ff_vp9_subpel_filters (libavcodec/vp9dsp.c:32-86), [filter 0 smooth / 1 regular / 2 sharp][m]: the 8 taps as two
 * dwords of int8 for v_dot4_i32_i8 (m = 0 is never used: full-pel positions are copies; its 128 would not fit) */
__constant__ __attribute__((aligned(16))) uint32_t vp9_h8[48][2] = {
    { 0x00000000u, 0x00000000u },    { 0x4020fffdu, 0x00fd0126u },    { 0x3f1dfefeu, 0x00fd0229u },    { 0x3f1afefeu, 0x00fc042bu },
    { 0x3e18fdfeu, 0x00fc052eu },    { 0x3c15fdfeu, 0x00fc0731u },    { 0x3b12fcffu, 0x00fc0933u },    { 0x3910fcffu, 0xfffc0c35u },
    { 0x370efcffu, 0xfffc0e37u },    { 0x350cfcffu, 0xfffc1039u },    { 0x3309fc00u, 0xfffc123bu },    { 0x3107fc00u, 0xfefd153cu },
    { 0x2e05fc00u, 0xfefd183eu },    { 0x2b04fc00u, 0xfefe1a3fu },    { 0x2902fd00u, 0xfefe1d3fu },    { 0x2601fd00u, 0xfdff2040u },
    { 0x00000000u, 0x00000000u },    { 0x7efb0100u, 0x0001fd08u },    { 0x7af603ffu, 0x0002fa12u },    { 0x76f304ffu, 0xff03f71bu },
    { 0x70f004ffu, 0xff04f525u },    { 0x69ee05ffu, 0xff04f230u },    { 0x61ed05ffu, 0xff05f03au },    { 0x58ed06ffu, 0xff05ee44u },
    { 0x4eed06ffu, 0xff06ed4eu },    { 0x44ee05ffu, 0xff06ed58u },    { 0x3af005ffu, 0xff05ed61u },    { 0x30f204ffu, 0xff05ee69u },
    { 0x25f504ffu, 0xff04f070u },    { 0x1bf703ffu, 0xff04f376u },    { 0x12fa0200u, 0xff03f67au },    { 0x08fd0100u, 0x0001fb7eu },
    { 0x00000000u, 0x00000000u },    { 0x7ff903ffu, 0x0001fd08u },    { 0x7df305feu, 0xff03fa11u },    { 0x79ef07fdu, 0xfe05f61bu },
    { 0x73ec09fcu, 0xfe06f325u },    { 0x6ce90afcu, 0xfd08f030u },    { 0x64e80afcu, 0xfd09ed3bu },    { 0x5ae80bfcu, 0xfc0aeb46u },
    { 0x50e90bfcu, 0xfc0be950u },    { 0x46eb0afcu, 0xfc0be85au },    { 0x3bed09fdu, 0xfc0ae864u },    { 0x30f008fdu, 0xfc0ae96cu },
    { 0x25f306feu, 0xfc09ec73u },    { 0x1bf605feu, 0xfd07ef79u },    { 0x11fa03ffu, 0xfe05f37du },    { 0x08fd0100u, 0xff03f97fu },
};
/* the same taps as v_dot2_i32_i16 operands over row pairs: [..][0..4] first row even (c0,c1)..(c6,c7)(0,0); [5..9] odd (0,c0)(c1,c2)..(c7,0) */
__constant__ __attribute__((aligned(16))) uint32_t vp9_v2[48][12] = {
    { 0x00000000u, 0x00000000u, 0x00000000u, 0x00000000u, 0x00000000u, 0x00000000u, 0x00000000u, 0x00000000u, 0x00000000u, 0x00000000u, 0x00000000u, 0x00000000u },
    { 0xfffffffdu, 0x00400020u, 0x00010026u, 0x0000fffdu, 0x00000000u, 0xfffd0000u, 0x0020ffffu, 0x00260040u, 0xfffd0001u, 0x00000000u, 0x00000000u, 0x00000000u },
    { 0xfffefffeu, 0x003f001du, 0x00020029u, 0x0000fffdu, 0x00000000u, 0xfffe0000u, 0x001dfffeu, 0x0029003fu, 0xfffd0002u, 0x00000000u, 0x00000000u, 0x00000000u },
    { 0xfffefffeu, 0x003f001au, 0x0004002bu, 0x0000fffcu, 0x00000000u, 0xfffe0000u, 0x001afffeu, 0x002b003fu, 0xfffc0004u, 0x00000000u, 0x00000000u, 0x00000000u },
    { 0xfffdfffeu, 0x003e0018u, 0x0005002eu, 0x0000fffcu, 0x00000000u, 0xfffe0000u, 0x0018fffdu, 0x002e003eu, 0xfffc0005u, 0x00000000u, 0x00000000u, 0x00000000u },
    { 0xfffdfffeu, 0x003c0015u, 0x00070031u, 0x0000fffcu, 0x00000000u, 0xfffe0000u, 0x0015fffdu, 0x0031003cu, 0xfffc0007u, 0x00000000u, 0x00000000u, 0x00000000u },
    { 0xfffcffffu, 0x003b0012u, 0x00090033u, 0x0000fffcu, 0x00000000u, 0xffff0000u, 0x0012fffcu, 0x0033003bu, 0xfffc0009u, 0x00000000u, 0x00000000u, 0x00000000u },
    { 0xfffcffffu, 0x00390010u, 0x000c0035u, 0xfffffffcu, 0x00000000u, 0xffff0000u, 0x0010fffcu, 0x00350039u, 0xfffc000cu, 0x0000ffffu, 0x00000000u, 0x00000000u },
    { 0xfffcffffu, 0x0037000eu, 0x000e0037u, 0xfffffffcu, 0x00000000u, 0xffff0000u, 0x000efffcu, 0x00370037u, 0xfffc000eu, 0x0000ffffu, 0x00000000u, 0x00000000u },
    { 0xfffcffffu, 0x0035000cu, 0x00100039u, 0xfffffffcu, 0x00000000u, 0xffff0000u, 0x000cfffcu, 0x00390035u, 0xfffc0010u, 0x0000ffffu, 0x00000000u, 0x00000000u },
    { 0xfffc0000u, 0x00330009u, 0x0012003bu, 0xfffffffcu, 0x00000000u, 0x00000000u, 0x0009fffcu, 0x003b0033u, 0xfffc0012u, 0x0000ffffu, 0x00000000u, 0x00000000u },
    { 0xfffc0000u, 0x00310007u, 0x0015003cu, 0xfffefffdu, 0x00000000u, 0x00000000u, 0x0007fffcu, 0x003c0031u, 0xfffd0015u, 0x0000fffeu, 0x00000000u, 0x00000000u },
    { 0xfffc0000u, 0x002e0005u, 0x0018003eu, 0xfffefffdu, 0x00000000u, 0x00000000u, 0x0005fffcu, 0x003e002eu, 0xfffd0018u, 0x0000fffeu, 0x00000000u, 0x00000000u },
    { 0xfffc0000u, 0x002b0004u, 0x001a003fu, 0xfffefffeu, 0x00000000u, 0x00000000u, 0x0004fffcu, 0x003f002bu, 0xfffe001au, 0x0000fffeu, 0x00000000u, 0x00000000u },
    { 0xfffd0000u, 0x00290002u, 0x001d003fu, 0xfffefffeu, 0x00000000u, 0x00000000u, 0x0002fffdu, 0x003f0029u, 0xfffe001du, 0x0000fffeu, 0x00000000u, 0x00000000u },
    { 0xfffd0000u, 0x00260001u, 0x00200040u, 0xfffdffffu, 0x00000000u, 0x00000000u, 0x0001fffdu, 0x00400026u, 0xffff0020u, 0x0000fffdu, 0x00000000u, 0x00000000u },
    { 0x00000000u, 0x00000000u, 0x00000000u, 0x00000000u, 0x00000000u, 0x00000000u, 0x00000000u, 0x00000000u, 0x00000000u, 0x00000000u, 0x00000000u, 0x00000000u },
    { 0x00010000u, 0x007efffbu, 0xfffd0008u, 0x00000001u, 0x00000000u, 0x00000000u, 0xfffb0001u, 0x0008007eu, 0x0001fffdu, 0x00000000u, 0x00000000u, 0x00000000u },
    { 0x0003ffffu, 0x007afff6u, 0xfffa0012u, 0x00000002u, 0x00000000u, 0xffff0000u, 0xfff60003u, 0x0012007au, 0x0002fffau, 0x00000000u, 0x00000000u, 0x00000000u },
    { 0x0004ffffu, 0x0076fff3u, 0xfff7001bu, 0xffff0003u, 0x00000000u, 0xffff0000u, 0xfff30004u, 0x001b0076u, 0x0003fff7u, 0x0000ffffu, 0x00000000u, 0x00000000u },
    { 0x0004ffffu, 0x0070fff0u, 0xfff50025u, 0xffff0004u, 0x00000000u, 0xffff0000u, 0xfff00004u, 0x00250070u, 0x0004fff5u, 0x0000ffffu, 0x00000000u, 0x00000000u },
    { 0x0005ffffu, 0x0069ffeeu, 0xfff20030u, 0xffff0004u, 0x00000000u, 0xffff0000u, 0xffee0005u, 0x00300069u, 0x0004fff2u, 0x0000ffffu, 0x00000000u, 0x00000000u },
    { 0x0005ffffu, 0x0061ffedu, 0xfff0003au, 0xffff0005u, 0x00000000u, 0xffff0000u, 0xffed0005u, 0x003a0061u, 0x0005fff0u, 0x0000ffffu, 0x00000000u, 0x00000000u },
    { 0x0006ffffu, 0x0058ffedu, 0xffee0044u, 0xffff0005u, 0x00000000u, 0xffff0000u, 0xffed0006u, 0x00440058u, 0x0005ffeeu, 0x0000ffffu, 0x00000000u, 0x00000000u },
    { 0x0006ffffu, 0x004effedu, 0xffed004eu, 0xffff0006u, 0x00000000u, 0xffff0000u, 0xffed0006u, 0x004e004eu, 0x0006ffedu, 0x0000ffffu, 0x00000000u, 0x00000000u },
    { 0x0005ffffu, 0x0044ffeeu, 0xffed0058u, 0xffff0006u, 0x00000000u, 0xffff0000u, 0xffee0005u, 0x00580044u, 0x0006ffedu, 0x0000ffffu, 0x00000000u, 0x00000000u },
    { 0x0005ffffu, 0x003afff0u, 0xffed0061u, 0xffff0005u, 0x00000000u, 0xffff0000u, 0xfff00005u, 0x0061003au, 0x0005ffedu, 0x0000ffffu, 0x00000000u, 0x00000000u },
    { 0x0004ffffu, 0x0030fff2u, 0xffee0069u, 0xffff0005u, 0x00000000u, 0xffff0000u, 0xfff20004u, 0x00690030u, 0x0005ffeeu, 0x0000ffffu, 0x00000000u, 0x00000000u },
    { 0x0004ffffu, 0x0025fff5u, 0xfff00070u, 0xffff0004u, 0x00000000u, 0xffff0000u, 0xfff50004u, 0x00700025u, 0x0004fff0u, 0x0000ffffu, 0x00000000u, 0x00000000u },
    { 0x0003ffffu, 0x001bfff7u, 0xfff30076u, 0xffff0004u, 0x00000000u, 0xffff0000u, 0xfff70003u, 0x0076001bu, 0x0004fff3u, 0x0000ffffu, 0x00000000u, 0x00000000u },
    { 0x00020000u, 0x0012fffau, 0xfff6007au, 0xffff0003u, 0x00000000u, 0x00000000u, 0xfffa0002u, 0x007a0012u, 0x0003fff6u, 0x0000ffffu, 0x00000000u, 0x00000000u },
    { 0x00010000u, 0x0008fffdu, 0xfffb007eu, 0x00000001u, 0x00000000u, 0x00000000u, 0xfffd0001u, 0x007e0008u, 0x0001fffbu, 0x00000000u, 0x00000000u, 0x00000000u },
    { 0x00000000u, 0x00000000u, 0x00000000u, 0x00000000u, 0x00000000u, 0x00000000u, 0x00000000u, 0x00000000u, 0x00000000u, 0x00000000u, 0x00000000u, 0x00000000u },
    { 0x0003ffffu, 0x007ffff9u, 0xfffd0008u, 0x00000001u, 0x00000000u, 0xffff0000u, 0xfff90003u, 0x0008007fu, 0x0001fffdu, 0x00000000u, 0x00000000u, 0x00000000u },
    { 0x0005fffeu, 0x007dfff3u, 0xfffa0011u, 0xffff0003u, 0x00000000u, 0xfffe0000u, 0xfff30005u, 0x0011007du, 0x0003fffau, 0x0000ffffu, 0x00000000u, 0x00000000u },
    { 0x0007fffdu, 0x0079ffefu, 0xfff6001bu, 0xfffe0005u, 0x00000000u, 0xfffd0000u, 0xffef0007u, 0x001b0079u, 0x0005fff6u, 0x0000fffeu, 0x00000000u, 0x00000000u },
    { 0x0009fffcu, 0x0073ffecu, 0xfff30025u, 0xfffe0006u, 0x00000000u, 0xfffc0000u, 0xffec0009u, 0x00250073u, 0x0006fff3u, 0x0000fffeu, 0x00000000u, 0x00000000u },
    { 0x000afffcu, 0x006cffe9u, 0xfff00030u, 0xfffd0008u, 0x00000000u, 0xfffc0000u, 0xffe9000au, 0x0030006cu, 0x0008fff0u, 0x0000fffdu, 0x00000000u, 0x00000000u },
    { 0x000afffcu, 0x0064ffe8u, 0xffed003bu, 0xfffd0009u, 0x00000000u, 0xfffc0000u, 0xffe8000au, 0x003b0064u, 0x0009ffedu, 0x0000fffdu, 0x00000000u, 0x00000000u },
    { 0x000bfffcu, 0x005affe8u, 0xffeb0046u, 0xfffc000au, 0x00000000u, 0xfffc0000u, 0xffe8000bu, 0x0046005au, 0x000affebu, 0x0000fffcu, 0x00000000u, 0x00000000u },
    { 0x000bfffcu, 0x0050ffe9u, 0xffe90050u, 0xfffc000bu, 0x00000000u, 0xfffc0000u, 0xffe9000bu, 0x00500050u, 0x000bffe9u, 0x0000fffcu, 0x00000000u, 0x00000000u },
    { 0x000afffcu, 0x0046ffebu, 0xffe8005au, 0xfffc000bu, 0x00000000u, 0xfffc0000u, 0xffeb000au, 0x005a0046u, 0x000bffe8u, 0x0000fffcu, 0x00000000u, 0x00000000u },
    { 0x0009fffdu, 0x003bffedu, 0xffe80064u, 0xfffc000au, 0x00000000u, 0xfffd0000u, 0xffed0009u, 0x0064003bu, 0x000affe8u, 0x0000fffcu, 0x00000000u, 0x00000000u },
    { 0x0008fffdu, 0x0030fff0u, 0xffe9006cu, 0xfffc000au, 0x00000000u, 0xfffd0000u, 0xfff00008u, 0x006c0030u, 0x000affe9u, 0x0000fffcu, 0x00000000u, 0x00000000u },
    { 0x0006fffeu, 0x0025fff3u, 0xffec0073u, 0xfffc0009u, 0x00000000u, 0xfffe0000u, 0xfff30006u, 0x00730025u, 0x0009ffecu, 0x0000fffcu, 0x00000000u, 0x00000000u },
    { 0x0005fffeu, 0x001bfff6u, 0xffef0079u, 0xfffd0007u, 0x00000000u, 0xfffe0000u, 0xfff60005u, 0x0079001bu, 0x0007ffefu, 0x0000fffdu, 0x00000000u, 0x00000000u },
    { 0x0003ffffu, 0x0011fffau, 0xfff3007du, 0xfffe0005u, 0x00000000u, 0xffff0000u, 0xfffa0003u, 0x007d0011u, 0x0005fff3u, 0x0000fffeu, 0x00000000u, 0x00000000u },
    { 0x00010000u, 0x0008fffdu, 0xfff9007fu, 0xffff0003u, 0x00000000u, 0x00000000u, 0xfffd0001u, 0x007f0008u, 0x0003fff9u, 0x0000ffffu, 0x00000000u, 0x00000000u },
};

constexpr int VM_TW = 16, VM_PITCH = 20, VM_PAIRS = 37; /* tile width, dwords per row pair, (64 + 7 + 1) / 2 + 1 row pairs */

__device__ __forceinline__ void vm_wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

/* four horizontally filtered, rounded and clipped samples: p = the first window's first byte (x0 - 3), any alignment */
__device__ __forceinline__ void vm_hrow4(const uint8_t *p, int clo, int chi, int (&o)[4])
{
    const uint32_t *q = reinterpret_cast<const uint32_t *>(p);
    const uint32_t d0 = q[0] ^ 0x80808080u, d1 = q[1] ^ 0x80808080u, d2 = q[2] ^ 0x80808080u;
    constexpr int seed = 128 * 128 + 64;
    int s[4];
    s[0] = __builtin_amdgcn_sdot4((int)d1, chi, __builtin_amdgcn_sdot4((int)d0, clo, seed, false), false);
    s[1] = __builtin_amdgcn_sdot4((int)__builtin_amdgcn_alignbyte(d2, d1, 1), chi,
                                  __builtin_amdgcn_sdot4((int)__builtin_amdgcn_alignbyte(d1, d0, 1), clo, seed, false), false);
    s[2] = __builtin_amdgcn_sdot4((int)__builtin_amdgcn_alignbyte(d2, d1, 2), chi,
                                  __builtin_amdgcn_sdot4((int)__builtin_amdgcn_alignbyte(d1, d0, 2), clo, seed, false), false);
    s[3] = __builtin_amdgcn_sdot4((int)__builtin_amdgcn_alignbyte(d2, d1, 3), chi,
                                  __builtin_amdgcn_sdot4((int)__builtin_amdgcn_alignbyte(d1, d0, 3), clo, seed, false), false);
#pragma unroll
    for (int j = 0; j < 4; j++)
        o[j] = clip_u8(s[j] >> 7);
}

/* SKIP16: the 16 x 16 blocks of the batch are k_vp9_mc_m's (below); a launch of a few thousand workgroups walks the batch for the rest */
template <bool SKIP16>
__global__ __launch_bounds__(256) void k_vp9_mc(uint8_t *dst, ptrdiff_t dststride, const uint8_t *src, ptrdiff_t srcstride,
                                                const FFHipVp9McBlock *blocks, int n)
{
    __shared__ uint32_t tmp_all[4][VM_PAIRS * VM_PITCH];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    /* SKIP16: the wave looks at 64 records at a time — a lane each — and works through the ones that are left to this kernel */
    for (int c0 = SKIP16 ? blockIdx.x * 64 : blockIdx.x * 4 + wave; c0 < n; c0 += SKIP16 ? (int)gridDim.x * 64 : n) {
    unsigned long todo = 1;
    int turn = 0;
    if (SKIP16) {
        const int bl = c0 + lane;
        const FFHipVp9McBlock &rr = blocks[min(bl, n - 1)];
        todo = __ballot(bl < n && !(rr.width == 16 && rr.height == 16));
    }
    while (todo) {
    const int b = SKIP16 ? c0 + (int)__builtin_ctzl(todo) : c0;
    todo &= todo - 1;
    if (SKIP16 && (turn++ & 3) != wave)
        continue; /* the four waves of the workgroup look at the same 64 records and take turns */
    const FFHipVp9McBlock k = blocks[b];
    const int w = __builtin_amdgcn_readfirstlane((int)k.width), h = __builtin_amdgcn_readfirstlane((int)k.height);
    const int filter = __builtin_amdgcn_readfirstlane((int)k.filter) & 3;
    const int mx = __builtin_amdgcn_readfirstlane((int)k.mx) & 15, my = __builtin_amdgcn_readfirstlane((int)k.my) & 15;
    const bool avg = __builtin_amdgcn_readfirstlane((int)k.avg) != 0;
    const uint8_t *s = src + __builtin_amdgcn_readfirstlane(k.src_offset);
    uint8_t *d0 = dst + __builtin_amdgcn_readfirstlane(k.dst_offset);
    const bool d_al = ((reinterpret_cast<uintptr_t>(d0) | (uintptr_t)dststride) & 3) == 0;
    uint32_t *tmp = tmp_all[wave];
    const bool bil = filter == 3;
    const int fi = (bil ? 0 : filter) * 16;

    /* put / avg of samples x0..x0+3 of row y */
    auto emit = [&](int y, int x0, const int (&v)[4]) {
        uint8_t *d = d0 + (ptrdiff_t)y * dststride + x0;
        uint32_t out = (uint32_t)v[0] | (uint32_t)v[1] << 8 | (uint32_t)v[2] << 16 | (uint32_t)v[3] << 24;
        if (d_al) {
            if (avg) {
                const uint32_t o = *reinterpret_cast<const uint32_t *>(d);
                out = (o | out) - (((o ^ out) & 0xfefefefeu) >> 1); /* (a + b + 1) >> 1 on four bytes */
            }
            *reinterpret_cast<uint32_t *>(d) = out;
        } else {
#pragma unroll
            for (int j = 0; j < 4; j++)
                d[j] = (uint8_t)(avg ? (d[j] + v[j] + 1) >> 1 : v[j]);
        }
    };
    auto bilin = [](int m, int a, int b2) { return a + ((m * (b2 - a) + 8) >> 4); };

    if (!my) {
        /* horizontal only (or a copy): no temporaries */
        const int ng = w >> 2, lg = __builtin_ctz(ng); /* 1, 2, 4, 8 or 16 groups per row */
        const int clo = (int)vp9_h8[fi + mx][0], chi = (int)vp9_h8[fi + mx][1];
        for (int i = lane; i < h * ng; i += 64) {
            const int y = i >> lg, xg = i & (ng - 1);
            const uint8_t *p = s + (ptrdiff_t)y * srcstride + 4 * xg;
            int o[4];
            if (!mx) {
                const uint32_t v = *reinterpret_cast<const uint32_t *>(p);
                o[0] = v & 255; o[1] = (v >> 8) & 255; o[2] = (v >> 16) & 255; o[3] = v >> 24;
            } else if (bil) {
                const uint32_t v = *reinterpret_cast<const uint32_t *>(p);
                const int e4 = p[4];
                const int a0 = v & 255, a1 = (v >> 8) & 255, a2 = (v >> 16) & 255, a3 = v >> 24;
                o[0] = bilin(mx, a0, a1); o[1] = bilin(mx, a1, a2); o[2] = bilin(mx, a2, a3); o[3] = bilin(mx, a3, e4);
            } else {
                vm_hrow4(p - 3, clo, chi, o);
            }
            emit(y, 4 * xg, o);
        }
        continue;
    }
    const int before = bil ? 0 : 3, rows = bil ? h + 1 : h + 7;
    const int clo = (int)vp9_h8[fi + mx][0], chi = (int)vp9_h8[fi + mx][1];
    uint32_t ce[5], co[5];
#pragma unroll
    for (int q = 0; q < 5; q++) {
        ce[q] = vp9_v2[fi + my][q];
        co[q] = vp9_v2[fi + my][5 + q];
    }
    for (int tx = 0; tx < w; tx += VM_TW) {
        const int ngt = min((w - tx) >> 2, VM_TW / 4), sh = ngt == 4 ? 2 : ngt == 2 ? 1 : 0; /* 4, 2 or 1 groups per row */
        /* rows -before .. of the tile, horizontally filtered (or raw), into the row-pair plane */
        for (int i = lane; i < rows * ngt; i += 64) {
            const int r = i >> sh, xg = i & (ngt - 1);
            const uint8_t *p = s + (ptrdiff_t)(r - before) * srcstride + tx + 4 * xg;
            int o[4];
            if (!mx) {
                const uint32_t v = *reinterpret_cast<const uint32_t *>(p);
                o[0] = v & 255; o[1] = (v >> 8) & 255; o[2] = (v >> 16) & 255; o[3] = v >> 24;
            } else if (bil) {
                const uint32_t v = *reinterpret_cast<const uint32_t *>(p);
                const int e4 = p[4];
                const int a0 = v & 255, a1 = (v >> 8) & 255, a2 = (v >> 16) & 255, a3 = v >> 24;
                o[0] = bilin(mx, a0, a1); o[1] = bilin(mx, a1, a2); o[2] = bilin(mx, a2, a3); o[3] = bilin(mx, a3, e4);
            } else {
                vm_hrow4(p - 3, clo, chi, o);
            }
            int16_t *t16 = reinterpret_cast<int16_t *>(tmp + (r >> 1) * VM_PITCH + 4 * xg) + (r & 1);
            t16[0] = (int16_t)o[0]; t16[2] = (int16_t)o[1]; t16[4] = (int16_t)o[2]; t16[6] = (int16_t)o[3];
        }
        vm_wave_sync();
        for (int i = lane; i < h * ngt; i += 64) {
            const int y = i >> sh, xg = i & (ngt - 1);
            int v[4];
            if (bil) {
                const int16_t *ta = reinterpret_cast<const int16_t *>(tmp + (y >> 1) * VM_PITCH + 4 * xg) + (y & 1);
                const int16_t *tb = reinterpret_cast<const int16_t *>(tmp + ((y + 1) >> 1) * VM_PITCH + 4 * xg) + ((y + 1) & 1);
#pragma unroll
                for (int j = 0; j < 4; j++)
                    v[j] = bilin(my, ta[2 * j], tb[2 * j]);
            } else {
                const uint4 *t = reinterpret_cast<const uint4 *>(tmp + (y >> 1) * VM_PITCH + 4 * xg);
                const bool odd = y & 1;
                int acc[4] = { 64, 64, 64, 64 };
#pragma unroll
                for (int q = 0; q < 5; q++) {
                    const uint4 u = t[q * (VM_PITCH / 4)];
                    const vm_s2 c = __builtin_bit_cast(vm_s2, odd ? co[q] : ce[q]);
                    acc[0] = __builtin_amdgcn_sdot2(__builtin_bit_cast(vm_s2, u.x), c, acc[0], false);
                    acc[1] = __builtin_amdgcn_sdot2(__builtin_bit_cast(vm_s2, u.y), c, acc[1], false);
                    acc[2] = __builtin_amdgcn_sdot2(__builtin_bit_cast(vm_s2, u.z), c, acc[2], false);
                    acc[3] = __builtin_amdgcn_sdot2(__builtin_bit_cast(vm_s2, u.w), c, acc[3], false);
                }
#pragma unroll
                for (int j = 0; j < 4; j++)
                    v[j] = clip_u8(acc[j] >> 7);
            }
            emit(y, tx + 4 * xg, v);
        }
        vm_wave_sync(); /* the next tile overwrites the plane */
    }
    }
    }
}

/* ================================================================================================================================== */
/*
 * k_vp9_mc_m — the 16 x 16 blocks of a batch on the MATRIX CORES: k_hevc_qpel_m's two-stage form (hevc_qpel_m.hip) with VP9's taps.
 * A VP9 pass ends in clip_u8((sum + 64) >> 7) — the 2-D form goes through 8-BIT temporaries (vp9dsp_template.c:2036-2076) — so the
 * second stage's B operand is the first stage's clipped bytes: one product per stage and row block, no 16-bit split.  The rounding
 * constant and the 128 * sum(taps) that undoes the ^0x80 ride in the accumulator; v_ashr_pk_u8_i32 shifts, clips and packs two sums.
 * The banded operands are not tables (3 filter sets x 15 fractions): a lane cuts its eight bytes out of the block's packed tap row
 * (vp9_h8) with a 64-bit shift.  Bilinear blocks (a + ((m (b - a) + 8) >> 4) == ((16 - m) a + m b + 8) >> 4, exactly) are the taps
 * (16 - m, m) at positions 3 and 4 of the same band with >> 4; full-pel directions hand the samples through an identity band.
 * avg: (dst + v + 1) >> 1 on the packed row.
 */
typedef int vq_i4 __attribute__((ext_vector_type(4)));
typedef uint32_t vq_u4 __attribute__((ext_vector_type(4)));

/* bytes [s, s + 4) of the 8-byte tap row T, zero outside it */
__device__ __forceinline__ uint32_t vq_win32(unsigned long T, int s)
{
    if (s <= -4 || s >= 8)
        return 0;
    return s >= 0 ? (uint32_t)(T >> (8 * s)) : (uint32_t)(T << (8 * -s));
}
__device__ __forceinline__ long vq_long(uint32_t lo, uint32_t hi) { return (long)(((unsigned long)hi << 32) | lo); }
__device__ __forceinline__ vq_i4 vq_splat(int v) { return (vq_i4){ v, v, v, v }; }
/* clip_u8(a >> SH) .. clip_u8(d >> SH) as bytes 0..3.  The builtin, not inline assembly: the operands come straight out of an MFMA and the
 * compiler counts that hazard's wait states only for instructions it knows; v_perm takes the two low halves (what v_ashr_pk_u8_i32 leaves
 * in the upper half of its destination is not zero: common.h) */
template <int SH>
__device__ __forceinline__ uint32_t vq_pack(int a, int b, int c, int d)
{
    const uint32_t lo = __builtin_amdgcn_ashr_pk_u8_i32(a, b, SH), hi = __builtin_amdgcn_ashr_pk_u8_i32(c, d, SH);
    return __builtin_amdgcn_perm(hi, lo, 0x05040100u);
}
__device__ __forceinline__ uint32_t vq_transpose(uint32_t c, uint32_t selT1, uint32_t selT2)
{
    const uint32_t t1 = (uint32_t)__builtin_amdgcn_mov_dpp((int)c, 0xB1, 0xf, 0xf, true);     /* quad_perm [1,0,3,2] */
    const uint32_t c1 = __builtin_amdgcn_perm(t1, c, selT1);
    const uint32_t t2 = (uint32_t)__builtin_amdgcn_mov_dpp((int)c1, 0x4E, 0xf, 0xf, true);    /* quad_perm [2,3,0,1] */
    return __builtin_amdgcn_perm(t2, c1, selT2);
}
__device__ __forceinline__ uint32_t vq_rnd_avg4(uint32_t a, uint32_t b) { return (a | b) - (((a ^ b) & 0xFEFEFEFEu) >> 1); }

__global__ __launch_bounds__(256) void k_vp9_mc_m(uint8_t *dst, ptrdiff_t dststride, const uint8_t *src, ptrdiff_t srcstride,
                                                  const FFHipVp9McBlock *blocks, int n, int per_xcd)
{
    __shared__ __align__(16) uint32_t rawp[4][24 * 12];
    __shared__ __align__(16) uint32_t obp[4][4 * 64];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    const int wg = per_xcd ? ((int)blockIdx.x & 7) * per_xcd + ((int)blockIdx.x >> 3) : (int)blockIdx.x;
    const int b0 = (wg * 4 + wave) * 4;
    if (b0 >= n)
        return;
    uint32_t *raw = rawp[wave], *ob = obp[wave];
    const int fr0 = (lane * 171) >> 9, fc0 = lane - 3 * fr0;          /* chunk `lane`: footprint row lane / 3, 16-byte chunk lane % 3 */
    const int fr1 = (64 + lane) / 3, fc1 = 64 + lane - 3 * fr1;        /* chunk 64 + lane (lanes 0..4: rows 21, 22) */
    const uint32_t selT1 = (lane & 1) ? 0x03070105u : 0x06020400u, selT2 = (lane & 2) ? 0x03020706u : 0x05040100u;
    const int g = lane >> 4, m = lane & 15;
    const int ry = 4 * g + (lane & 3), rxg = (lane >> 2) & 3;          /* the row and 4-sample group this lane owns after the transposition */
    const long K80 = (long)0x8080808080808080ull;

    int Gmx[4], Gmy[4], Gfi[4], Gdoff[4];
    bool Gel[4], Gavg[4], Gbil[4], tile = true;
    uint32_t Gsh16[4];
    vq_u4 Gf0[4], Gf1[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const FFHipVp9McBlock rec = blocks[min(b0 + k, n - 1)];
        const int w = __builtin_amdgcn_readfirstlane((int)rec.width), h = __builtin_amdgcn_readfirstlane((int)rec.height);
        const int filter = __builtin_amdgcn_readfirstlane((int)rec.filter) & 3;
        Gbil[k] = filter == 3;
        Gfi[k] = (Gbil[k] ? 0 : filter) * 16;
        Gmx[k] = __builtin_amdgcn_readfirstlane((int)rec.mx) & 15;
        Gmy[k] = __builtin_amdgcn_readfirstlane((int)rec.my) & 15;
        Gavg[k] = __builtin_amdgcn_readfirstlane((int)rec.avg) != 0;
        Gdoff[k] = __builtin_amdgcn_readfirstlane(rec.dst_offset);
        Gel[k] = b0 + k < n && w == 16 && h == 16;
        tile = tile && Gel[k] && !((reinterpret_cast<uintptr_t>(dst) + (uintptr_t)(intptr_t)Gdoff[k]) & 3);
        const uint8_t *s0 = src + __builtin_amdgcn_readfirstlane(rec.src_offset) - 3 - 3 * srcstride;
        Gsh16[k] = (uint32_t)(reinterpret_cast<uintptr_t>(s0) & 15);
        const uint8_t *sa = s0 - Gsh16[k];
        /* the part of the footprint the position reads: 8 taps want all 23 rows / columns, two taps one more than the block, none the block */
        const int r_lo = Gmy[k] && !Gbil[k] ? 0 : 3, r_hi = Gmy[k] ? (Gbil[k] ? 19 : 22) : 18;
        const int c_lo = (int)Gsh16[k] + (Gmx[k] && !Gbil[k] ? 0 : 3), c_hi = (int)Gsh16[k] + (Gmx[k] ? (Gbil[k] ? 19 : 22) : 18);
        const bool want0 = Gel[k] && fr0 >= r_lo && fr0 <= r_hi && 16 * fc0 + 15 >= c_lo && 16 * fc0 <= c_hi;
        const bool want1 = Gel[k] && lane < 5 && fr1 >= r_lo && fr1 <= r_hi && 16 * fc1 + 15 >= c_lo && 16 * fc1 <= c_hi;
        Gf0[k] = want0 ? *reinterpret_cast<const vq_u4 *>(sa + (ptrdiff_t)fr0 * srcstride + 16 * fc0) : (vq_u4){ 0, 0, 0, 0 };
        Gf1[k] = want1 ? *reinterpret_cast<const vq_u4 *>(sa + (ptrdiff_t)fr1 * srcstride + 16 * fc1) : (vq_u4){ 0, 0, 0, 0 };
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
        if (!Gel[k])
            continue; /* k_vp9_mc<true> takes it */
        *reinterpret_cast<vq_u4 *>(raw + fr0 * 12 + 4 * fc0) = Gf0[k];
        if (lane < 5)
            *reinterpret_cast<vq_u4 *>(raw + fr1 * 12 + 4 * fc1) = Gf1[k];
        __builtin_amdgcn_wave_barrier();
        const int mx = Gmx[k], my = Gmy[k];
        const bool bil = Gbil[k];
        const uint32_t sh = Gsh16[k] & 3;
        const uint32_t *r0p = raw + (Gsh16[k] >> 2);   /* the dword that holds footprint byte 0 of row 0 */
        uint32_t out;
        if (mx | my) {
            /* the packed tap rows of the two directions: eight int8 of the set, (16 - m, m) at taps 3 and 4, or the identity */
            const unsigned long Tx = !mx ? 0x01000000ul : bil ? ((unsigned long)(16 - mx) << 24 | (unsigned long)mx << 32)
                                                             : ((unsigned long)vp9_h8[Gfi[k] + mx][1] << 32 | vp9_h8[Gfi[k] + mx][0]);
            const unsigned long Ty = bil ? ((unsigned long)(16 - my) << 24 | (unsigned long)my << 32)
                                         : ((unsigned long)vp9_h8[Gfi[k] + my][1] << 32 | vp9_h8[Gfi[k] + my][0]);
            const int seed = bil ? 128 * 16 + 8 : 128 * 128 + 64;     /* 128 * sum(taps) + the pass's rounding constant */
            /* stage 1: B[k = 8g + j][n] = tap[k - n]; my = 0: the lanes feed footprint rows 3 .. 18 and the C layout is the output block */
            const long cTh = vq_long(vq_win32(Tx, 8 * g - m), vq_win32(Tx, 8 * g - m + 4));
            const uint32_t *pa = r0p + (my ? m : m + 3) * 12 + 2 * g;
            const uint32_t a0 = pa[0], a1 = pa[1], a2 = pa[2];
            const long fa = vq_long(__builtin_amdgcn_alignbyte(a1, a0, sh), __builtin_amdgcn_alignbyte(a2, a1, sh)) ^ K80;
            const vq_i4 h0 = __builtin_amdgcn_mfma_i32_16x16x32_i8(fa, cTh, vq_splat(mx ? seed : 128), 0, 0, 0);
            uint32_t c;
            if (!my) {
                c = bil ? vq_pack<4>(h0.x, h0.y, h0.z, h0.w) : vq_pack<7>(h0.x, h0.y, h0.z, h0.w);
            } else {
                const uint32_t *pb = r0p + min(16 + m, 22) * 12 + 2 * g;
                const uint32_t b0_ = pb[0], b1_ = pb[1], b2_ = pb[2];
                const long fb = vq_long(__builtin_amdgcn_alignbyte(b1_, b0_, sh), __builtin_amdgcn_alignbyte(b2_, b1_, sh)) ^ K80;
                const vq_i4 h1 = __builtin_amdgcn_mfma_i32_16x16x32_i8(fb, cTh, vq_splat(mx ? seed : 128), 0, 0, 0);
                /* the 8-bit temporaries (or the raw samples) of column n ARE the stage-2 K slots of this lane: slot 8g + j holds footprint
                 * row 4g + j (j < 4) or 16 + 4g + j - 4; A[y][slot] = tap[row - y] */
                uint32_t xlo, xhi;
                if (!mx) {
                    xlo = vq_pack<0>(h0.x, h0.y, h0.z, h0.w); xhi = vq_pack<0>(h1.x, h1.y, h1.z, h1.w);
                } else if (bil) {
                    xlo = vq_pack<4>(h0.x, h0.y, h0.z, h0.w); xhi = vq_pack<4>(h1.x, h1.y, h1.z, h1.w);
                } else {
                    xlo = vq_pack<7>(h0.x, h0.y, h0.z, h0.w); xhi = vq_pack<7>(h1.x, h1.y, h1.z, h1.w);
                }
                const long bv = vq_long(xlo, xhi) ^ K80;
                const long cTv = vq_long(vq_win32(Ty, 4 * g - m), vq_win32(Ty, 16 + 4 * g - m));
                const vq_i4 vv = __builtin_amdgcn_mfma_i32_16x16x32_i8(cTv, bv, vq_splat(seed), 0, 0, 0);
                c = bil ? vq_pack<4>(vv.x, vv.y, vv.z, vv.w) : vq_pack<7>(vv.x, vv.y, vv.z, vv.w);
            }
            out = vq_transpose(c, selT1, selT2);
        } else {
            const uint32_t o = sh + 3;
            const uint32_t *pf = r0p + (ry + 3) * 12 + rxg + (o >> 2);
            out = __builtin_amdgcn_alignbyte(pf[1], pf[0], o & 3);
        }
        if (tile) {
            ob[64 * k + 4 * ry + rxg] = out;
        } else {
            uint8_t *d = dst + Gdoff[k] + (ptrdiff_t)ry * dststride + 4 * rxg;
            if (!((reinterpret_cast<uintptr_t>(d)) & 3)) {
                uint32_t *dw = reinterpret_cast<uint32_t *>(d);
                *dw = Gavg[k] ? vq_rnd_avg4(*dw, out) : out;
            } else {
                for (int i = 0; i < 4; i++) {
                    const uint32_t v = (out >> (8 * i)) & 0xFF;
                    d[i] = (uint8_t)(Gavg[k] ? (d[i] + v + 1) >> 1 : v);
                }
            }
        }
        __builtin_amdgcn_wave_barrier(); /* the next block overwrites the plane */
    }
    if (tile) {
        const int y = lane >> 2, c = lane & 3;
        vq_u4 o = *reinterpret_cast<const vq_u4 *>(ob + 64 * c + 4 * y);
        const int doff = c == 0 ? Gdoff[0] : c == 1 ? Gdoff[1] : c == 2 ? Gdoff[2] : Gdoff[3];
        const bool avg = c == 0 ? Gavg[0] : c == 1 ? Gavg[1] : c == 2 ? Gavg[2] : Gavg[3];
        vq_u4 *dp = reinterpret_cast<vq_u4 *>(dst + doff + (ptrdiff_t)y * dststride);
        if (avg) {
            const vq_u4 old = *dp;
            o.x = vq_rnd_avg4(old.x, o.x); o.y = vq_rnd_avg4(old.y, o.y); o.z = vq_rnd_avg4(old.z, o.z); o.w = vq_rnd_avg4(old.w, o.w);
        }
        *dp = o;
    }
}

int ffhip_launch_vp9_mc(uint8_t *dst, ptrdiff_t dststride, const uint8_t *src, ptrdiff_t srcstride, const FFHipVp9McBlock *blocks, int n,
                        hipStream_t stream)
{
    if (n <= 0)
        return 0;
    const char *em = FFHIP_KNOB("FFHIP_VP9_MC_M"); /* measured variant: 0 = without the matrix-core kernel */
    if (!(em && em[0] == '0') && !(srcstride & 15) && !(dststride & 3)) {
        /* the 16 x 16 blocks on the matrix cores (aligned 16-byte footprint chunks: the source stride must keep a row's alignment),
         * everything else in a second launch that skips those */
        const int per_xcd = cdiv(cdiv(n, 16), 8);
        hipLaunchKernelGGL(k_vp9_mc_m, dim3(8 * per_xcd), dim3(256), 0, stream, dst, dststride, src, srcstride, blocks, n, per_xcd);
        hipLaunchKernelGGL(k_vp9_mc<true>, dim3(min(cdiv(n, 64), 32768)), dim3(256), 0, stream, dst, dststride, src, srcstride, blocks, n);
    } else {
        hipLaunchKernelGGL(k_vp9_mc<false>, dim3(cdiv(n, 4)), dim3(256), 0, stream, dst, dststride, src, srcstride, blocks, n);
    }
    LAUNCH_CHECK();
    return 0;
}

/* the taps as int8 per (filter, fraction) for the scaled kernel's lane-varying fractions (fraction 0 = {0,0,0,128,..} is a copy and
 * never read) */
__constant__ int8_t vp9_t8[48][8] = {
    { 0, 0, 0, 0, 0, 0, 0, 0 },
    { -3, -1, 32, 64, 38, 1, -3, 0 },
    { -2, -2, 29, 63, 41, 2, -3, 0 },
    { -2, -2, 26, 63, 43, 4, -4, 0 },
    { -2, -3, 24, 62, 46, 5, -4, 0 },
    { -2, -3, 21, 60, 49, 7, -4, 0 },
    { -1, -4, 18, 59, 51, 9, -4, 0 },
    { -1, -4, 16, 57, 53, 12, -4, -1 },
    { -1, -4, 14, 55, 55, 14, -4, -1 },
    { -1, -4, 12, 53, 57, 16, -4, -1 },
    { 0, -4, 9, 51, 59, 18, -4, -1 },
    { 0, -4, 7, 49, 60, 21, -3, -2 },
    { 0, -4, 5, 46, 62, 24, -3, -2 },
    { 0, -4, 4, 43, 63, 26, -2, -2 },
    { 0, -3, 2, 41, 63, 29, -2, -2 },
    { 0, -3, 1, 38, 64, 32, -1, -3 },
    { 0, 0, 0, 0, 0, 0, 0, 0 },
    { 0, 1, -5, 126, 8, -3, 1, 0 },
    { -1, 3, -10, 122, 18, -6, 2, 0 },
    { -1, 4, -13, 118, 27, -9, 3, -1 },
    { -1, 4, -16, 112, 37, -11, 4, -1 },
    { -1, 5, -18, 105, 48, -14, 4, -1 },
    { -1, 5, -19, 97, 58, -16, 5, -1 },
    { -1, 6, -19, 88, 68, -18, 5, -1 },
    { -1, 6, -19, 78, 78, -19, 6, -1 },
    { -1, 5, -18, 68, 88, -19, 6, -1 },
    { -1, 5, -16, 58, 97, -19, 5, -1 },
    { -1, 4, -14, 48, 105, -18, 5, -1 },
    { -1, 4, -11, 37, 112, -16, 4, -1 },
    { -1, 3, -9, 27, 118, -13, 4, -1 },
    { 0, 2, -6, 18, 122, -10, 3, -1 },
    { 0, 1, -3, 8, 126, -5, 1, 0 },
    { 0, 0, 0, 0, 0, 0, 0, 0 },
    { -1, 3, -7, 127, 8, -3, 1, 0 },
    { -2, 5, -13, 125, 17, -6, 3, -1 },
    { -3, 7, -17, 121, 27, -10, 5, -2 },
    { -4, 9, -20, 115, 37, -13, 6, -2 },
    { -4, 10, -23, 108, 48, -16, 8, -3 },
    { -4, 10, -24, 100, 59, -19, 9, -3 },
    { -4, 11, -24, 90, 70, -21, 10, -4 },
    { -4, 11, -23, 80, 80, -23, 11, -4 },
    { -4, 10, -21, 70, 90, -24, 11, -4 },
    { -3, 9, -19, 59, 100, -24, 10, -4 },
    { -3, 8, -16, 48, 108, -23, 10, -4 },
    { -2, 6, -13, 37, 115, -20, 9, -4 },
    { -2, 5, -10, 27, 121, -17, 7, -3 },
    { -1, 3, -6, 17, 125, -13, 5, -2 },
    { 0, 1, -3, 8, 127, -7, 3, -1 },
};

/*
 * Scaled motion compensation: VP9DSPContext.smc[size][filter][avg] (vp9dsp_template.c:2362-2540).  The reference picture has
 * another size: output x samples around column (mx + x dx) >> 4 with the taps of fraction (mx + x dx) & 15, output y around row
 * (my + y dy) >> 4.  Neighbouring outputs no longer share windows or taps, so this is a plain gather: one wave per block, the
 * horizontally filtered 8-bit temporaries (up to 135 rows) in wave-private LDS, eight multiply-adds per sample and pass.  Rare in
 * streams (reference scaling), kept simple.
 */
static_assert(sizeof(FFHipVp9ScaledBlock) == 16, "FFHipVp9ScaledBlock is a 16-byte record");

/* PIX = uint8_t / uint16_t (temporaries are samples of that depth, clipped to (1 << bd) - 1); WPB waves (blocks) per workgroup: the
 * 135-row temporaries of 16-bit samples do not fit four to an LDS allocation.  UNSCALED: records are FFHipVp9McBlock (dx = dy = 16) —
 * the plain mc[][][][][] table above 8 bits runs here too: at a step of 16 the scaled form IS the unscaled one (a zero fraction's tap
 * set is the identity), with one difference that matters to a decoder: no margin row / column is read on an axis whose fraction is 0. */
template <typename PIX, int WPB, bool UNSCALED>
__global__ __launch_bounds__(64 * WPB) void k_vp9_smc(uint8_t *dst, ptrdiff_t dststride, const uint8_t *src, ptrdiff_t srcstride,
                                                      const FFHipVp9ScaledBlock *blocks, int n, int bd)
{
    __shared__ PIX tmp_all[WPB][135 * 64];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int b = blockIdx.x * WPB + wave;
    if (b >= n)
        return;
    const FFHipVp9ScaledBlock k = blocks[b];
    const int w = k.width, h = k.height, filter = k.filter & 3, mx = k.mx & 15, my = k.my & 15, dx = UNSCALED ? 16 : k.dx, dy = UNSCALED ? 16 : k.dy;
    const bool avg = k.avg != 0, bil = filter == 3;
    const int maxv = (1 << bd) - 1;
    const bool flat_v = UNSCALED && my == 0; /* the vertical pass is the identity: rows 0 .. h - 1 only */
    const int before = (bil || flat_v) ? 0 : 3, rows = flat_v ? h : (((h - 1) * dy + my) >> 4) + (bil ? 2 : 8);
    const ptrdiff_t sst = srcstride / (ptrdiff_t)sizeof(PIX);
    const PIX *s = reinterpret_cast<const PIX *>(src + k.src_offset);
    PIX *tmp = tmp_all[wave];
    const int lgw = __builtin_ctz(w);
    auto tap = [&](int m, const PIX *p, ptrdiff_t step) {
        if (!m)
            return (int)p[0];
        if (bil)
            return (int)p[0] + ((m * ((int)p[step] - (int)p[0]) + 8) >> 4);
        const int8_t *f = vp9_t8[filter * 16 + m];
        int sum = 64;
#pragma unroll
        for (int t = 0; t < 8; t++)
            sum += f[t] * p[(t - 3) * step];
        return min(max(sum >> 7, 0), maxv);
    };
    for (int i = lane; i < rows * w; i += 64) {
        const int r = i >> lgw, x = i & (w - 1), pos = mx + x * dx;
        tmp[r * 64 + x] = (PIX)tap(pos & 15, s + (ptrdiff_t)(r - before) * sst + (pos >> 4), 1);
    }
    vm_wave_sync();
    for (int i = lane; i < h * w; i += 64) {
        const int y = i >> lgw, x = i & (w - 1), pos = my + y * dy;
        const int v = tap(pos & 15, tmp + ((pos >> 4) + before) * 64 + x, 64);
        PIX *d = reinterpret_cast<PIX *>(dst + k.dst_offset + (ptrdiff_t)y * dststride) + x;
        *d = (PIX)(avg ? (*d + v + 1) >> 1 : v);
    }
}

int ffhip_launch_vp9_smc(uint8_t *dst, ptrdiff_t dststride, const uint8_t *src, ptrdiff_t srcstride, const FFHipVp9ScaledBlock *blocks, int n,
                         hipStream_t stream)
{
    return ffhip_launch_vp9_smc_bd(8, dst, dststride, src, srcstride, blocks, n, stream);
}

static bool vp9_hbd_ok(int bd, const void *a, const void *b, ptrdiff_t sa, ptrdiff_t sb)
{
    return (bd == 10 || bd == 12) && !(((uintptr_t)a | (uintptr_t)b | (size_t)sa | (size_t)sb) & 1);
}

int ffhip_launch_vp9_smc_bd(int bd, uint8_t *dst, ptrdiff_t dststride, const uint8_t *src, ptrdiff_t srcstride, const FFHipVp9ScaledBlock *blocks,
                            int n, hipStream_t stream)
{
    if (n <= 0)
        return 0;
    if (bd == 8)
        hipLaunchKernelGGL((k_vp9_smc<uint8_t, 4, false>), dim3(cdiv(n, 4)), dim3(256), 0, stream, dst, dststride, src, srcstride, blocks, n, 8);
    else if (vp9_hbd_ok(bd, dst, src, dststride, srcstride))
        hipLaunchKernelGGL((k_vp9_smc<uint16_t, 2, false>), dim3(cdiv(n, 2)), dim3(128), 0, stream, dst, dststride, src, srcstride, blocks, n, bd);
    else {
        ffhip_set_error("ffhip_vp9_scaled_mc: bit depth %d (8, 10, 12) / 16-bit planes must be 2-byte aligned", bd);
        return FFHIP_EINVAL;
    }
    LAUNCH_CHECK();
    return 0;
}

int ffhip_launch_vp9_mc_bd(int bd, uint8_t *dst, ptrdiff_t dststride, const uint8_t *src, ptrdiff_t srcstride, const FFHipVp9McBlock *blocks, int n,
                           hipStream_t stream)
{
    if (bd == 8)
        return ffhip_launch_vp9_mc(dst, dststride, src, srcstride, blocks, n, stream);
    if (n <= 0)
        return 0;
    if (!vp9_hbd_ok(bd, dst, src, dststride, srcstride)) {
        ffhip_set_error("ffhip_vp9_mc: bit depth %d (8, 10, 12) / 16-bit planes must be 2-byte aligned", bd);
        return FFHIP_EINVAL;
    }
    static_assert(sizeof(FFHipVp9McBlock) == sizeof(FFHipVp9ScaledBlock), "the two records share their layout up to `avg`");
    hipLaunchKernelGGL((k_vp9_smc<uint16_t, 2, true>), dim3(cdiv(n, 2)), dim3(128), 0, stream, dst, dststride, src, srcstride,
                       reinterpret_cast<const FFHipVp9ScaledBlock *>(blocks), n, bd);
    LAUNCH_CHECK();
    return 0;
}
