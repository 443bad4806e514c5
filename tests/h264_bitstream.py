"""A small H.264 CAVLC bitstream WRITER (test infrastructure): decides syntax elements, writes them as ITU-T H.264 clause 7.3 lays them out,
and knows nothing about decoding — no prediction, no motion-vector derivation, no reconstruction.  The reference's whole decoder
(h264dec.c, h264_slice.c, h264_cavlc.c, h264_mvpred.h, fill_decode_caches(), fill_filter_caches() ... compiled where they lie into
oracle/_ref/libffref_h264dec.so) turns the streams into per-macroblock state; the tests then compare its plain decode with its decode
through the `hip` recorder.  What the writer must track is only what CAVLC itself conditions on: the number of coefficients of the
neighbouring blocks (nC of coeff_token, 9.2.1), neighbour availability per slice, the Intra4x4 mode predictor (8.3.1.1), and which
Intra modes are legal at picture / slice borders.

Syntax covered: SPS (High / High 10, 4:2:0, frame_mbs_only or not, mb_adaptive_frame_field), PPS (CAVLC, deblocking control), slice
headers (I / P, IDR or not, field_pic / bottom_field, num_ref_idx override, disable_deblocking_filter_idc 0 / 1 / 2 with offsets, several
slices per picture), macroblocks: I_PCM, Intra16x16 (all four modes where legal, any cbp), Intra4x4 (all nine modes where legal),
P_L0_16x16 / 16x8 / 8x16 / P_8x8 (all four sub-types) with ref_idx and mvd, P_Skip runs, mb_qp_delta, residual_block_cavlc with
trailing ones, level escapes, total_zeros and run_before, MBAFF's mb_field_decoding_flag.  pic_order_cnt_type 2 (output order = decode
order), sliding-window reference marking.

Round 6: B slices — B_Direct_16x16, B_Skip, B_L0 / L1 / Bi 16x16, the eighteen 16x8 / 8x16 combinations, B_8x8 with all thirteen
sub-macroblock types (B_Direct_8x8 included), direct_spatial_mv_pred_flag 0 / 1, pic_order_cnt_type 0 with pictures decoded out of output
order (VUI bitstream_restriction carries the reorder depth), non-reference B pictures; PPS weighted_pred_flag / weighted_bipred_idc 1 with a
pred_weight_table( ) per slice, weighted_bipred_idc 2 (implicit); transform_8x8_mode_flag: Intra8x8 (I_NxN with transform_size_8x8_flag,
prev_intra8x8_pred_mode) and transform_size_8x8_flag on inter macroblocks where 7.3.5 allows it; High 4:2:2 (chroma_format_idc 2: 2x4
chroma DC with its own coeff_token / total_zeros tables, eight chroma AC blocks per plane).  Still syntax only: which vectors, references,
weights and direct modes come out of it is h264_mvpred.h's, h264_direct.c's, h264_parse.c's and h264_slice.c's business.
"""
import numpy as np

import h264_vlc_tables as T


class BitWriter:
    def __init__(self):
        self.acc = 0
        self.n = 0

    def u(self, nbits, v):
        assert 0 <= v < (1 << nbits), (nbits, v)
        self.acc = (self.acc << nbits) | v
        self.n += nbits

    def ue(self, v):
        assert v >= 0
        k = (v + 1).bit_length()
        self.u(2 * k - 1, v + 1)

    def se(self, v):
        self.ue(2 * v - 1 if v > 0 else -2 * v)

    def te(self, v, vmax):
        if vmax > 1:
            self.ue(v)
        else:
            self.u(1, 1 - v)

    def code(self, length, bits):
        assert length > 0
        self.u(length, bits)

    def align_zero(self):
        while self.n & 7:
            self.u(1, 0)

    def trailing(self):
        self.u(1, 1)
        self.align_zero()

    def bytes(self):
        assert self.n % 8 == 0
        return self.acc.to_bytes(self.n // 8, "big") if self.n else b""


def nal(ref_idc, unit_type, rbsp):
    out = bytearray(b"\x00\x00\x00\x01")
    out.append((ref_idc << 5) | unit_type)
    zeros = 0
    for b in rbsp:
        if zeros >= 2 and b <= 3:
            out.append(3)            # emulation_prevention_three_byte
            zeros = 0
        out.append(b)
        zeros = zeros + 1 if b == 0 else 0
    return bytes(out)


# blkIdx (the standard's z-order of 4x4 luma blocks) -> (x, y) in 4x4 units inside the macroblock
def blk_xy(b):
    return ((b >> 2) & 1) * 2 + (b & 1), (b >> 3) * 2 + ((b >> 1) & 1)


def residual_block(bw, coefs, nC, max_num):
    """residual_block_cavlc (7.3.5.3.2 / 9.2): `coefs` in scan order (len == max_num).  nC: the coeff_token table selector (-1: chroma DC).
    Returns total_coeff."""
    nz = [(i, c) for i, c in enumerate(coefs) if c]
    total = len(nz)
    t1 = 0
    for _, c in reversed(nz):
        if abs(c) == 1 and t1 < 3:
            t1 += 1
        else:
            break
    if nC == -1:
        bw.code(T.chroma_dc_coeff_token_len[4 * total + t1], T.chroma_dc_coeff_token_bits[4 * total + t1])
    elif nC == -2:     # 4:2:2 chroma DC (Table 9-5, nC == -2)
        bw.code(T.chroma422_dc_coeff_token_len[4 * total + t1], T.chroma422_dc_coeff_token_bits[4 * total + t1])
    else:
        tab = 0 if nC < 2 else 1 if nC < 4 else 2 if nC < 8 else 3
        bw.code(T.coeff_token_len[tab][4 * total + t1], T.coeff_token_bits[tab][4 * total + t1])
    if not total:
        return 0
    suffix_len = 1 if total > 10 and t1 < 3 else 0
    for k, (_, c) in enumerate(reversed(nz)):
        if k < t1:
            bw.u(1, int(c < 0))
            continue
        code = 2 * c - 2 if c > 0 else -2 * c - 1
        if k == t1 and t1 < 3:
            code -= 2
        if suffix_len == 0:
            if code < 14:
                bw.u(code + 1, 1)
            elif code < 30:
                bw.u(15, 1)
                bw.u(4, code - 14)
            else:
                assert code - 30 < 4096
                bw.u(16, 1)
                bw.u(12, code - 30)
        else:
            if code < (15 << suffix_len):
                bw.u((code >> suffix_len) + 1, 1)
                bw.u(suffix_len, code & ((1 << suffix_len) - 1))
            else:
                assert code - (15 << suffix_len) < 4096
                bw.u(16, 1)
                bw.u(12, code - (15 << suffix_len))
        if suffix_len == 0:
            suffix_len = 1
        if abs(c) > (3 << (suffix_len - 1)) and suffix_len < 6:
            suffix_len += 1
    if total < max_num:
        tz = nz[-1][0] + 1 - total
        if max_num == 4:
            bw.code(T.chroma_dc_total_zeros_len[total - 1][tz], T.chroma_dc_total_zeros_bits[total - 1][tz])
        elif max_num == 8:   # Table 9-9 (b)
            bw.code(T.chroma422_dc_total_zeros_len[total - 1][tz], T.chroma422_dc_total_zeros_bits[total - 1][tz])
        else:
            bw.code(T.total_zeros_len[total - 1][tz], T.total_zeros_bits[total - 1][tz])
        left = tz
        for k in range(total - 1, 0, -1):
            if left <= 0:
                break
            run = nz[k][0] - nz[k - 1][0] - 1
            tab = min(left, 7) - 1
            bw.code(T.run_len[tab][run], T.run_bits[tab][run])
            left -= run
    return total


INTER_CBP_CODE = {cbp: k for k, cbp in enumerate(T.golomb_to_inter_cbp)}
INTRA_CBP_CODE = {cbp: k for k, cbp in enumerate(T.golomb_to_intra4x4_cbp)}


class Params:
    """What a stream is made with."""

    def __init__(self, mb_w=6, mb_h=5, bit_depth=8, frame_mbs_only=1, mbaff=0, num_ref_frames=3, init_qp=26, chroma_qp_offset=0,
                 seed=1, chroma_format=1, t8x8=0, weighted_pred=0, weighted_bipred=0, poc_type=2, reorder=0, lossless=0):
        self.mb_w, self.mb_h, self.bit_depth = mb_w, mb_h, bit_depth
        self.frame_mbs_only, self.mbaff = frame_mbs_only, mbaff
        self.num_ref_frames, self.init_qp, self.chroma_qp_offset = num_ref_frames, init_qp, chroma_qp_offset
        self.log2_max_frame_num = 8
        self.seed = seed
        # round 6: chroma_format_idc (1 / 2), PPS transform_8x8_mode_flag, weighted_pred_flag, weighted_bipred_idc, pic_order_cnt_type
        # (2: output order = decode order; 0: pic_order_cnt_lsb per picture, `reorder` = VUI max_num_reorder_frames)
        self.chroma_format, self.t8x8, self.weighted_pred, self.weighted_bipred = chroma_format, t8x8, weighted_pred, weighted_bipred
        self.poc_type, self.reorder, self.log2_max_poc_lsb = poc_type, reorder, 8
        # lossless: qpprime_y_zero_transform_bypass_flag = 1 — macroblocks with QP'Y = 0 are decoded with the transform bypassed; 1: a High
        # profile (the residual is added as it is), 2: profile_idc 244, High 4:4:4 Predictive (vertically / horizontally predicted intra blocks
        # are DPCM-coded on top of that).  The running QP stays in 0 .. 6 (init_qp 3) so that both kinds of macroblock turn up in a slice.
        self.lossless = lossless
        if lossless:
            self.init_qp = 3
        assert frame_mbs_only or mb_h % 2 == 0
        assert chroma_format in (1, 2) and poc_type in (0, 2)


def sps_nal(p):
    bw = BitWriter()
    profile = 244 if p.lossless == 2 else 122 if p.chroma_format == 2 else 100 if p.bit_depth == 8 else 110
    bw.u(8, profile)
    bw.u(8, 0)                      # constraint flags + reserved
    bw.u(8, 40)                     # level_idc
    bw.ue(0)                        # seq_parameter_set_id
    bw.ue(p.chroma_format)          # chroma_format_idc
    bw.ue(p.bit_depth - 8)
    bw.ue(p.bit_depth - 8)
    bw.u(1, 1 if p.lossless else 0) # qpprime_y_zero_transform_bypass_flag
    bw.u(1, 0)                      # seq_scaling_matrix_present_flag
    bw.ue(p.log2_max_frame_num - 4)
    bw.ue(p.poc_type)               # pic_order_cnt_type
    if p.poc_type == 0:
        bw.ue(p.log2_max_poc_lsb - 4)
    bw.ue(p.num_ref_frames)
    bw.u(1, 0)                      # gaps_in_frame_num_value_allowed_flag
    bw.ue(p.mb_w - 1)
    bw.ue((p.mb_h if p.frame_mbs_only else p.mb_h // 2) - 1)
    bw.u(1, p.frame_mbs_only)
    if not p.frame_mbs_only:
        bw.u(1, p.mbaff)
    bw.u(1, 1)                      # direct_8x8_inference_flag
    bw.u(1, 0)                      # frame_cropping_flag
    if p.poc_type == 0:
        # VUI with bitstream_restriction_flag only: max_num_reorder_frames tells the decoder how many pictures to hold back (E.1.1)
        bw.u(1, 1)                  # vui_parameters_present_flag
        bw.u(4, 0)                  # aspect_ratio_info / overscan_info / video_signal_type / chroma_loc_info present flags
        bw.u(1, 0)                  # timing_info_present_flag
        bw.u(2, 0)                  # nal_hrd / vcl_hrd parameters present
        bw.u(1, 0)                  # pic_struct_present_flag
        bw.u(1, 1)                  # bitstream_restriction_flag
        bw.u(1, 1)                  # motion_vectors_over_pic_boundaries_flag
        bw.ue(2)                    # max_bytes_per_pic_denom
        bw.ue(1)                    # max_bits_per_mb_denom
        bw.ue(16)                   # log2_max_mv_length_horizontal
        bw.ue(16)                   # log2_max_mv_length_vertical
        bw.ue(p.reorder)            # max_num_reorder_frames
        bw.ue(max(p.num_ref_frames, p.reorder))   # max_dec_frame_buffering
    else:
        bw.u(1, 0)                  # vui_parameters_present_flag
    bw.trailing()
    return nal(3, 7, bw.bytes())


def pps_nal(p, num_ref_default=1):
    bw = BitWriter()
    bw.ue(0)
    bw.ue(0)
    bw.u(1, 0)                      # entropy_coding_mode_flag: CAVLC
    bw.u(1, 0)                      # bottom_field_pic_order_in_frame_present_flag
    bw.ue(0)                        # num_slice_groups_minus1
    bw.ue(num_ref_default - 1)
    bw.ue(0)
    bw.u(1, p.weighted_pred)        # weighted_pred_flag
    bw.u(2, p.weighted_bipred)      # weighted_bipred_idc
    bw.se(p.init_qp - 26)
    bw.se(0)
    bw.se(p.chroma_qp_offset)
    bw.u(1, 1)                      # deblocking_filter_control_present_flag
    bw.u(1, 0)                      # constrained_intra_pred_flag
    bw.u(1, 0)                      # redundant_pic_cnt_present_flag
    if p.t8x8:
        bw.u(1, 1)                  # transform_8x8_mode_flag
        bw.u(1, 0)                  # pic_scaling_matrix_present_flag
        bw.se(p.chroma_qp_offset)   # second_chroma_qp_index_offset
    bw.trailing()
    return nal(3, 8, bw.bytes())


I4_DC = 2


class Picture:
    """CAVLC context of the picture being written (frame or field): per 4x4-block coefficient counts, per-macroblock slice number and
    kind, Intra4x4 modes."""

    def __init__(self, mb_w, mb_h, chroma_format=1):
        self.mb_w, self.mb_h = mb_w, mb_h
        self.cper_y = 2 * chroma_format                    # 4x4 chroma blocks per macroblock, down: 2 (4:2:0) / 4 (4:2:2); across: 2
        ch = self.cper_y * mb_h
        self.tc = [np.zeros((4 * mb_h, 4 * mb_w), np.int32), np.zeros((ch, 2 * mb_w), np.int32), np.zeros((ch, 2 * mb_w), np.int32)]
        self.slice_of = -np.ones((mb_h, mb_w), np.int32)
        self.kind = np.zeros((mb_h, mb_w), np.int32)       # 0 inter / skip, 1 Intra4x4, 2 other intra
        self.i4mode = np.full((4 * mb_h, 4 * mb_w), I4_DC, np.int32)


class StreamWriter:
    """Access units of a random but legal stream.  pictures: a list of dicts
        { 'type': 'I' | 'P', 'slices': [first_mb, ...] (first_mb of every slice, ascending, starting with 0), 'deblock': [(idc, a, b), ...]
          one per slice, 'field': None | 'top' | 'bottom', 'num_ref': active reference count for P slices }"""

    def __init__(self, params):
        self.p = params
        self.rng = np.random.default_rng(params.seed)
        self.frame_num = 0
        self.idr_id = 0
        self.small = bool(params.mbaff)
        self.stats = {"pcm": 0, "i16": 0, "i4": 0, "p16": 0, "p168": 0, "p88": 0, "skip": 0, "coded_blocks": 0, "escapes": 0,
                      "i8": 0, "t8x8_inter": 0, "b_direct16": 0, "b_16": 0, "b_168": 0, "b_88": 0, "b_direct8": 0, "b_skip": 0, "b_bi": 0,
                      "wp_slices": 0}

    # ---- helpers -------------------------------------------------------------------------------------------------------------------
    def avail(self, pic, mx, my, sl):
        return 0 <= mx < pic.mb_w and 0 <= my < pic.mb_h and pic.slice_of[my, mx] == sl

    def nC(self, pic, plane, bx, by, sl):
        """9.2.1: from the left (A) and upper (B) neighbouring blocks' total_coeff; a block is available with its macroblock"""
        per = 4 if plane == 0 else 2
        pery = 4 if plane == 0 else pic.cper_y
        tc = pic.tc[plane]
        a = b = None
        if bx > 0 and self.avail(pic, (bx - 1) // per, by // pery, sl):
            a = tc[by, bx - 1]
        if by > 0 and self.avail(pic, bx // per, (by - 1) // pery, sl):
            b = tc[by - 1, bx]
        if a is not None and b is not None:
            return (a + b + 1) >> 1
        return a if a is not None else b if b is not None else 0

    def rand_coefs(self, n, density, big=0):
        """n levels in scan order; big: the largest magnitude an escape-coded level may take (0: none).  self.small (MBAFF): at most one
        coefficient per block, so that nC of every neighbouring block is 0 or 1 whatever the neighbour derivation"""
        r = self.rng
        c = np.zeros(n, np.int64)
        k = r.random()
        if k < 0.25:
            return c.tolist()
        cnt = 1 if k < 0.45 or self.small else int(r.integers(1, max(2, int(n * density)) + 1))
        idx = r.choice(n, size=min(cnt, n), replace=False)
        mags = r.geometric(0.45, size=len(idx))
        if big >= 8 and r.random() < 0.3:
            mags[0] = int(r.integers(8, big + 1))   # level_prefix 14 / 15 escapes
            self.stats["escapes"] += 1
        c[idx] = mags * r.choice([-1, 1], size=len(idx))
        return c.tolist()

    def write_luma_block(self, bw, pic, mx, my, blk, sl, coefs, max_num):
        x, y = blk_xy(blk)
        bx, by = 4 * mx + x, 4 * my + y
        t = residual_block(bw, coefs, self.nC(pic, 0, bx, by, sl), max_num)
        pic.tc[0][by, bx] = t
        self.stats["coded_blocks"] += t > 0

    def write_residual(self, bw, pic, mx, my, sl, i16, cbp, big, qp=26):
        """residual( ) for 4:2:0 (7.3.5.3): luma DC of Intra16x16, the luma blocks of the coded 8x8 quadrants, chroma DC, chroma AC.
        Levels stay where the dequantised coefficient (about 4 * 2^(qp / 6) per level step) is far inside the legal range."""
        density = 0.35
        big = min(400, int(500 / 2 ** (qp / 6.0))) if big else 0
        if i16:
            residual_block(bw, self.rand_coefs(16, 0.3), self.nC(pic, 0, 4 * mx, 4 * my, sl), 16)
        for b in range(16):
            if cbp & (1 << (b >> 2)):
                n = 15 if i16 else 16
                self.write_luma_block(bw, pic, mx, my, b, sl, self.rand_coefs(n, density, big), n)
        if cbp & 0x30 and pic.cper_y == 4:
            for _ in range(2):                  # 4:2:2: 2x4 chroma DC, nC = -2 (7.3.5.3: 4 * NumC8x8 coefficients)
                residual_block(bw, self.rand_coefs(8, 0.6), -2, 8)
        elif cbp & 0x30:
            for _ in range(2):
                residual_block(bw, self.rand_coefs(4, 0.8), -1, 4)
        if cbp & 0x20:
            for pl in (1, 2):
                for b in range(2 * pic.cper_y):   # chroma4x4BlkIdx: raster over a 2-wide column of blocks (6.4.7)
                    bx, by = 2 * mx + (b & 1), pic.cper_y * my + (b >> 1)
                    pic.tc[pl][by, bx] = residual_block(bw, self.rand_coefs(15, 0.2), self.nC(pic, pl, bx, by, sl), 15)

    def qp_delta(self, st):
        """keeps the running QP inside a band where random levels stay far from the coefficient range"""
        lo, hi = (0, 6) if self.p.lossless else (self.p.init_qp - 10, self.p.init_qp + 8)
        d = int(self.rng.integers(-3, 4))
        if self.p.lossless and self.rng.random() < 0.4:
            d = -st["qp"]               # QP'Y = 0: this macroblock (and those after it, until the next delta) bypasses the transform
        if not lo <= st["qp"] + d <= hi:
            d = 0
        st["qp"] += d
        return d

    # ---- macroblocks -----------------------------------------------------------------------------------------------------------------
    def mbaff_neighbours(self, pic, mx, my, sl, pair_field):
        """(left, top, topleft) availability of a macroblock of an MBAFF frame (6.4.12.2): the neighbours are macroblocks of the left /
        upper / upper-left PAIR — except for the bottom frame macroblock of a pair, whose upper neighbour is the pair's own top macroblock
        and whose upper-left one lies in the left pair.  WHICH macroblock of the pair it is, is the decoder's business."""
        p = my >> 1

        def pair(x, q):
            return 0 <= x < pic.mb_w and 0 <= q and pic.slice_of[2 * q, x] == sl
        left = pair(mx - 1, p)
        if (my & 1) and not pair_field:
            return left, True, left
        return left, pair(mx, p - 1), pair(mx - 1, p - 1)

    def mbaff_i4_isolated(self, pic, mx, my):
        """Intra4x4 / Intra8x8 in an MBAFF frame: the mode predictor takes the modes of neighbouring blocks in the left / upper pair or the
        pair's other macroblock, chosen by the frame / field kinds of both pairs; a macroblock none of whose possible neighbours is I_NxN
        sees DC whichever it is (8.3.1.1), and the writer needs no more."""
        p = my >> 1
        for x, q in ((mx - 1, p), (mx, p - 1), (mx, p)):
            for b in (0, 1):
                if 0 <= x < pic.mb_w and q >= 0 and (x, 2 * q + b) != (mx, my) and pic.kind[2 * q + b, x] == 1:
                    return False
        return True

    def intra_mb(self, bw, pic, mx, my, sl, st, type_offset, allow_i4=True, neigh=None):
        """neigh: (left, top, topleft) of a macroblock of an MBAFF frame (mbaff_neighbours()); an I_NxN macroblock there is isolated
        (mbaff_i4_isolated()): every block outside it counts as DC where its macroblock is available"""
        r = self.rng
        left, top = self.avail(pic, mx - 1, my, sl), self.avail(pic, mx, my - 1, sl)
        topleft = self.avail(pic, mx - 1, my - 1, sl)
        if neigh is not None:
            left, top, topleft = neigh
        self._neigh = neigh
        k = r.random()
        if k < 0.08 and not self.small:
            # I_PCM: pcm_alignment_zero_bit, 256 + 2 x 64 samples of bit_depth bits
            bw.ue(type_offset + 25)
            bw.align_zero()
            for v in r.integers(0, 1 << self.p.bit_depth, 256 + 64 * pic.cper_y):
                bw.u(self.p.bit_depth, int(v))
            pic.tc[0][4 * my:4 * my + 4, 4 * mx:4 * mx + 4] = 16
            for pl in (1, 2):
                pic.tc[pl][pic.cper_y * my:pic.cper_y * (my + 1), 2 * mx:2 * mx + 2] = 16
            pic.kind[my, mx] = 2
            self.stats["pcm"] += 1
            return
        chroma_modes = [0] + ([1] if left else []) + ([2] if top else []) + ([3] if left and top and topleft else [])
        if k < 0.55 or not allow_i4:
            modes = [2] + ([0] if top else []) + ([1] if left else []) + ([3] if left and top and topleft else [])
            mode = int(r.choice(modes))
            cbp_l = 15 if r.random() < 0.5 else 0
            cbp_c = int(r.integers(0, 3))
            bw.ue(type_offset + 1 + mode + 4 * cbp_c + (12 if cbp_l else 0))
            bw.ue(int(r.choice(chroma_modes)))
            bw.se(self.qp_delta(st))
            pic.kind[my, mx] = 2
            self.write_residual(bw, pic, mx, my, sl, True, cbp_l | (cbp_c << 4), False, st["qp"])
            self.stats["i16"] += 1
            return
        # I_NxN: Intra4x4 prediction, or (transform_8x8_mode_flag) Intra8x8 with transform_size_8x8_flag = 1
        bw.ue(type_offset + 0)
        if self.p.t8x8:
            i8 = int(r.random() < 0.5)
            bw.u(1, i8)                         # transform_size_8x8_flag
            if i8:
                self.intra8x8_modes(bw, pic, mx, my, sl, left, top, topleft)
                pic.kind[my, mx] = 1
                bw.ue(int(r.choice(chroma_modes)))
                cbp = int(r.integers(0, 48))
                bw.ue(INTRA_CBP_CODE[cbp])
                if cbp:
                    bw.se(self.qp_delta(st))
                self.write_residual(bw, pic, mx, my, sl, False, cbp, True, st["qp"])
                self.stats["i8"] += 1
                return
        for b in range(16):
            x, y = blk_xy(b)
            bx, by = 4 * mx + x, 4 * my + y
            has_l = x > 0 or left
            has_t = y > 0 or top
            # 0 vertical, 3 diagonal down-left, 7 vertical-left read the row above; 1 horizontal, 8 horizontal-up the column to the left;
            # 4, 5, 6 (diagonal down-right, vertical-right, horizontal-down) both and the top-left sample, which at the macroblock's own
            # corner belongs to the top-left macroblock
            legal = [I4_DC] + ([1, 8] if has_l else []) + ([0, 3, 7] if has_t else [])
            if has_l and has_t and not (x == 0 and y == 0 and not topleft):
                legal += [4, 5, 6]
            want = int(r.choice(legal))
            # 8.3.1.1: the predictor is min(mode A, mode B); DC when a neighbouring macroblock is unavailable; DC stands for a neighbour
            # that is not Intra4x4 coded
            def mode_of(nbx, nby):
                nmx, nmy = nbx // 4, nby // 4
                if self._neigh is not None and (nmx, nmy) != (mx, my):
                    return I4_DC if (left if nbx < 4 * mx else top) else None
                if not self.avail(pic, nmx, nmy, sl) and (nmx, nmy) != (mx, my):
                    return None
                if (nmx, nmy) != (mx, my) and pic.kind[nmy, nmx] != 1:
                    return I4_DC
                return int(pic.i4mode[nby, nbx])
            ma = mode_of(bx - 1, by) if bx > 0 else None
            mb_ = mode_of(bx, by - 1) if by > 0 else None
            pred = I4_DC if ma is None or mb_ is None else min(ma, mb_)
            if want == pred:
                bw.u(1, 1)
            else:
                bw.u(1, 0)
                bw.u(3, want if want < pred else want - 1)
            pic.i4mode[by, bx] = want
        pic.kind[my, mx] = 1
        bw.ue(int(r.choice(chroma_modes)))
        cbp = int(r.integers(0, 48))
        bw.ue(INTRA_CBP_CODE[cbp])
        if cbp:
            bw.se(self.qp_delta(st))
        self.write_residual(bw, pic, mx, my, sl, False, cbp, True, st["qp"])
        self.stats["i4"] += 1

    def intra8x8_modes(self, bw, pic, mx, my, sl, left, top, topleft):
        """prev_intra8x8_pred_mode_flag / rem_intra8x8_pred_mode for the four 8x8 blocks (7.3.5.1, 8.3.2.1): the predictor is min(mode A,
        mode B) of the neighbouring 8x8 blocks — for an Intra4x4 neighbour macroblock the modes of ITS 4x4 blocks next to the edge (index
        8x8 * 4 + 1 / + 2), which is the 4x4 block left of / above this block's first 4x4 block — with the Intra4x4 rules for unavailable
        and not-I_NxN neighbours.  pic.i4mode keeps an 8x8 block's mode in its four 4x4 entries."""
        r = self.rng
        for b8 in range(4):
            x, y = 2 * (b8 & 1), 2 * (b8 >> 1)
            bx, by = 4 * mx + x, 4 * my + y
            has_l = x > 0 or left
            has_t = y > 0 or top
            legal = [I4_DC] + ([1, 8] if has_l else []) + ([0, 3, 7] if has_t else [])
            if has_l and has_t and not (x == 0 and y == 0 and not topleft):
                legal += [4, 5, 6]
            want = int(r.choice(legal))

            def mode_of(nbx, nby):
                nmx, nmy = nbx // 4, nby // 4
                if self._neigh is not None and (nmx, nmy) != (mx, my):
                    return I4_DC if (left if nbx < 4 * mx else top) else None
                if not self.avail(pic, nmx, nmy, sl) and (nmx, nmy) != (mx, my):
                    return None
                if (nmx, nmy) != (mx, my) and pic.kind[nmy, nmx] != 1:
                    return I4_DC
                return int(pic.i4mode[nby, nbx])
            ma = mode_of(bx - 1, by) if bx > 0 else None
            mb_ = mode_of(bx, by - 1) if by > 0 else None
            pred = I4_DC if ma is None or mb_ is None else min(ma, mb_)
            if want == pred:
                bw.u(1, 1)
            else:
                bw.u(1, 0)
                bw.u(3, want if want < pred else want - 1)
            pic.i4mode[by:by + 2, bx:bx + 2] = want

    def mvd(self, bw):
        r = self.rng
        for _ in range(2):
            k = r.random()
            v = 0 if k < 0.3 else int(r.integers(-12, 13)) if k < 0.85 else int(r.integers(-90, 91)) if k < 0.97 else int(r.integers(-400, 401))
            bw.se(v)

    def inter_mb(self, bw, pic, mx, my, sl, st, num_ref):
        """num_ref: the range of ref_idx for this macroblock (a field macroblock of an MBAFF frame: twice the slice's count, 7.4.5.1)"""
        r = self.rng
        k = r.random()
        def ref():
            if num_ref > 1:
                bw.te(int(r.integers(0, num_ref)), num_ref - 1)
        if k < 0.35:
            bw.ue(0)                # P_L0_16x16
            ref()
            self.mvd(bw)
            self.stats["p16"] += 1
        elif k < 0.6:
            bw.ue(1 + int(r.integers(0, 2)))   # P_L0_L0_16x8 / 8x16
            ref(); ref()
            self.mvd(bw); self.mvd(bw)
            self.stats["p168"] += 1
        else:
            ref0 = num_ref > 1 and r.random() < 0.25
            bw.ue(4 if ref0 else 3)            # P_8x8ref0 / P_8x8
            subs = [int(v) for v in r.integers(0, 4, 4)]
            if self.p.t8x8 and r.random() < 0.4:
                subs = [0, 0, 0, 0]            # four 8x8 sub-macroblocks: transform_size_8x8_flag may follow
            for s in subs:
                bw.ue(s)
            if not ref0:
                for _ in range(4):
                    ref()
            for s in subs:
                for _ in range((1, 2, 2, 4)[s]):
                    self.mvd(bw)
            self.stats["p88"] += 1
            no_sub8 = all(v == 0 for v in subs)
        self.inter_tail(bw, pic, mx, my, sl, st, k < 0.6 or no_sub8)

    def inter_tail(self, bw, pic, mx, my, sl, st, t8_allowed):
        """coded_block_pattern, transform_size_8x8_flag where 7.3.5 has it (coded luma, transform_8x8_mode_flag, no sub-macroblock
        partition smaller than 8x8; direct needs direct_8x8_inference_flag, which the SPS sets), mb_qp_delta, residual"""
        r = self.rng
        cbp = int(r.integers(0, 48)) if r.random() < 0.7 else 0
        bw.ue(INTER_CBP_CODE[cbp])
        if self.p.t8x8 and (cbp & 15) and t8_allowed:
            t8 = int(r.random() < 0.6)
            bw.u(1, t8)
            self.stats["t8x8_inter"] += t8
        if cbp:
            bw.se(self.qp_delta(st))
        pic.kind[my, mx] = 0
        self.write_residual(bw, pic, mx, my, sl, False, cbp, True, st["qp"])

    # Table 7-14 mb_type 4..21: the prediction of the two partitions (the 16x8 / 8x16 shape does not change the syntax)
    B_PARTS = {4: "00", 5: "00", 6: "11", 7: "11", 8: "01", 9: "01", 10: "10", 11: "10", 12: "0b", 13: "0b", 14: "1b", 15: "1b", 16: "b0", 17: "b0",
               18: "b1", 19: "b1", 20: "bb", 21: "bb"}
    # Table 7-18 sub_mb_type 1..12: (prediction, sub-macroblock partitions)
    B_SUB = {1: ("0", 1), 2: ("1", 1), 3: ("b", 1), 4: ("0", 2), 5: ("0", 2), 6: ("1", 2), 7: ("1", 2), 8: ("b", 2), 9: ("b", 2), 10: ("0", 4),
             11: ("1", 4), 12: ("b", 4)}

    def inter_mb_b(self, bw, pic, mx, my, sl, st, nref):
        """a B macroblock (mb_type 0..22): ref_idx_l0 of every partition that uses list 0, then ref_idx_l1, then mvd_l0, then mvd_l1
        (7.3.5.1 / 7.3.5.2).  nref = (list 0, list 1) ref_idx ranges."""
        r = self.rng
        k = r.random()

        def ref(lst):
            if nref[lst] > 1:
                bw.te(int(r.integers(0, nref[lst])), nref[lst] - 1)
        if k < 0.15:
            bw.ue(0)                                  # B_Direct_16x16: no mb_pred( )
            self.stats["b_direct16"] += 1
            self.inter_tail(bw, pic, mx, my, sl, st, True)
            return
        if k < 0.4:
            t = int(r.integers(1, 4))                 # B_L0_16x16 / B_L1_16x16 / B_Bi_16x16
            bw.ue(t)
            parts = ["0", "1", "b"][t - 1]
            self.stats["b_16"] += 1
        elif k < 0.7:
            t = int(r.integers(4, 22))
            bw.ue(t)
            parts = self.B_PARTS[t]
            self.stats["b_168"] += 1
        else:
            bw.ue(22)                                 # B_8x8
            subs = [int(v) for v in r.integers(0, 13, 4)]
            if self.p.t8x8 and r.random() < 0.5:
                subs = [int(v) for v in r.integers(0, 4, 4)]   # 8x8 and direct sub-macroblocks only: transform_size_8x8_flag may follow
            for sm in subs:
                bw.ue(sm)
            info = [self.B_SUB.get(sm) for sm in subs]     # None: B_Direct_8x8
            for lst, ch in ((0, "0"), (1, "1")):
                for i in info:
                    if i and i[0] in (ch, "b"):
                        ref(lst)
            for lst, ch in ((0, "0"), (1, "1")):
                for i in info:
                    if i and i[0] in (ch, "b"):
                        for _ in range(i[1]):
                            self.mvd(bw)
            self.stats["b_88"] += 1
            self.stats["b_direct8"] += sum(1 for i in info if i is None)
            self.stats["b_bi"] += any(i and i[0] == "b" for i in info)
            self.inter_tail(bw, pic, mx, my, sl, st, all(sm < 4 for sm in subs))
            return
        for lst, ch in ((0, "0"), (1, "1")):
            for pp in parts:
                if pp in (ch, "b"):
                    ref(lst)
        for lst, ch in ((0, "0"), (1, "1")):
            for pp in parts:
                if pp in (ch, "b"):
                    self.mvd(bw)
        self.stats["b_bi"] += "b" in parts
        self.inter_tail(bw, pic, mx, my, sl, st, True)

    def pred_weight_table(self, bw, counts):
        """pred_weight_table( ) (7.3.3.2): one entry per active reference of every list; weights / offsets drawn so that any pair of
        weights of the two lists stays inside the bi-predictive constraint (-128 <= w0 + w1 <= 127 at logWD 5 or 6)"""
        r = self.rng
        ld, cd = int(r.integers(4, 7)), int(r.integers(4, 7))
        bw.ue(ld)
        bw.ue(cd)                                     # chroma_format_idc != 0
        for n in counts:
            for _ in range(n):
                lf = int(r.random() < 0.7)
                bw.u(1, lf)
                if lf:
                    bw.se(int(r.integers(-10, 61)))
                    bw.se(int(r.integers(-20, 21)))
                cf = int(r.random() < 0.6)
                bw.u(1, cf)
                if cf:
                    for _ in range(2):
                        bw.se(int(r.integers(-10, 61)))
                        bw.se(int(r.integers(-20, 21)))
        self.stats["wp_slices"] += 1

    # ---- slices and pictures ---------------------------------------------------------------------------------------------------------
    def slice_nal(self, pic, desc, sl, first_mb, end_mb, idr, field):
        p, r = self.p, self.rng
        ptype = desc["type"]
        bw = BitWriter()
        mbaff = p.mbaff and not field
        bw.ue(first_mb >> mbaff)    # MBAFF: in units of macroblock pairs
        bw.ue({"I": 2, "P": 0, "B": 1}[ptype])
        bw.ue(0)
        bw.u(p.log2_max_frame_num, self.frame_num)
        if not p.frame_mbs_only:
            bw.u(1, 1 if field else 0)
            if field:
                bw.u(1, int(field == "bottom"))
        if idr:
            bw.ue(self.idr_id)
        if p.poc_type == 0:
            bw.u(p.log2_max_poc_lsb, desc["poc"] % (1 << p.log2_max_poc_lsb))   # pic_order_cnt_lsb
        num_ref = desc.get("num_ref", 1)
        num_ref_l1 = desc.get("num_ref_l1", 1)
        if ptype == "B":
            bw.u(1, desc.get("direct_spatial", 1))   # direct_spatial_mv_pred_flag
        if ptype == "P":
            bw.u(1, 1)              # num_ref_idx_active_override_flag
            bw.ue(num_ref - 1)
            bw.u(1, 0)              # ref_pic_list_modification_flag_l0
        elif ptype == "B":
            bw.u(1, 1)
            bw.ue(num_ref - 1)
            bw.ue(num_ref_l1 - 1)
            bw.u(1, 0)              # ref_pic_list_modification_flag_l0
            bw.u(1, 0)              # ref_pic_list_modification_flag_l1
        if (ptype == "P" and p.weighted_pred) or (ptype == "B" and p.weighted_bipred == 1):
            self.pred_weight_table(bw, [num_ref] + ([num_ref_l1] if ptype == "B" else []))
        if not desc.get("ref", True):
            pass                    # nal_ref_idc == 0: no dec_ref_pic_marking( )
        elif idr:
            bw.u(1, 0)              # no_output_of_prior_pics_flag
            bw.u(1, 0)              # long_term_reference_flag
        else:
            bw.u(1, 0)              # adaptive_ref_pic_marking_mode_flag: sliding window
        qp_delta = int(r.integers(-3, 4))
        bw.se(qp_delta)
        idc, a, b = desc["deblock"][sl]
        bw.ue(idc)
        if idc != 1:
            bw.se(a)
            bw.se(b)
        st = {"qp": p.init_qp + qp_delta}
        # slice_data( )
        skip_run = 0
        n_mbs = pic.mb_w * pic.mb_h
        addr_list = list(range(first_mb, end_mb))
        prev_skipped = False
        pair_field = 0
        i = 0
        while i < len(addr_list):
            addr = addr_list[i]
            if mbaff:
                # macroblock pairs: addresses 2k (top) and 2k + 1 (bottom) of pair k, pairs in raster order
                pair = addr >> 1
                mx, my = pair % pic.mb_w, 2 * (pair // pic.mb_w) + (addr & 1)
            else:
                mx, my = addr % pic.mb_w, addr // pic.mb_w
            pic.slice_of[my, mx] = sl
            skip = ptype != "I" and r.random() < 0.22
            if skip:
                skip_run += 1
                pic.kind[my, mx] = 0
                self.stats["b_skip" if ptype == "B" else "skip"] += 1
                prev_skipped = True
                i += 1
                continue
            if ptype != "I":
                bw.ue(skip_run)
                skip_run = 0
            if mbaff and ((addr & 1) == 0 or prev_skipped):
                # mb_field_decoding_flag (7.3.4): with the top macroblock of a pair, or with the bottom one when the top was skipped
                pair_field = int(r.random() < 0.5)
                bw.u(1, pair_field)
            prev_skipped = False
            neigh = self.mbaff_neighbours(pic, mx, my, sl, pair_field) if mbaff else None
            allow_i4 = self.mbaff_i4_isolated(pic, mx, my) if mbaff else True
            if ptype == "I":
                self.intra_mb(bw, pic, mx, my, sl, st, 0, allow_i4=allow_i4, neigh=neigh)
            elif r.random() < (0.12 if ptype == "B" else 0.2):
                self.intra_mb(bw, pic, mx, my, sl, st, 23 if ptype == "B" else 5, allow_i4=allow_i4, neigh=neigh)
            elif ptype == "B":
                # (a field macroblock pair of an MBAFF frame: twice the slice's reference counts, 7.4.5.1)
                self.inter_mb_b(bw, pic, mx, my, sl, st, (num_ref * 2, num_ref_l1 * 2) if mbaff and pair_field else (num_ref, num_ref_l1))
            else:
                self.inter_mb(bw, pic, mx, my, sl, st, num_ref * 2 if mbaff and pair_field else num_ref)
            i += 1
        if skip_run:
            bw.ue(skip_run)
        bw.trailing()
        return nal(1 if desc.get("ref", True) else 0, 5 if idr else 1, bw.bytes())

    def picture(self, desc, idr=False):
        """one access unit (a frame, or one field) as bytes"""
        p = self.p
        field = desc.get("field")
        mb_h = p.mb_h // 2 if field else p.mb_h
        pic = Picture(p.mb_w, mb_h, p.chroma_format)
        out = b""
        firsts = list(desc["slices"]) + [p.mb_w * mb_h]
        for sl in range(len(desc["slices"])):
            out += self.slice_nal(pic, desc, sl, firsts[sl], firsts[sl + 1], idr, field)
        return out

    def stream(self, pictures):
        """list of access units; the first carries SPS + PPS and is an IDR picture"""
        aus = []
        for n, d in enumerate(pictures):
            idr = n == 0
            au = (sps_nal(self.p) + pps_nal(self.p)) if idr else b""
            au += self.picture(d, idr)
            aus.append(au)
            # frame_num (7.4.3): counts reference pictures — a non-reference picture carries the value the next reference picture will,
            # and leaves it there; the second field of a frame shares its first field's
            if d.get("ref", True) and (not d.get("field") or d.get("second_field")):
                self.frame_num = (self.frame_num + 1) % (1 << self.p.log2_max_frame_num)
        return aus
