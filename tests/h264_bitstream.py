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
                 seed=1):
        self.mb_w, self.mb_h, self.bit_depth = mb_w, mb_h, bit_depth
        self.frame_mbs_only, self.mbaff = frame_mbs_only, mbaff
        self.num_ref_frames, self.init_qp, self.chroma_qp_offset = num_ref_frames, init_qp, chroma_qp_offset
        self.log2_max_frame_num = 8
        self.seed = seed
        assert frame_mbs_only or mb_h % 2 == 0


def sps_nal(p):
    bw = BitWriter()
    profile = 100 if p.bit_depth == 8 else 110
    bw.u(8, profile)
    bw.u(8, 0)                      # constraint flags + reserved
    bw.u(8, 40)                     # level_idc
    bw.ue(0)                        # seq_parameter_set_id
    bw.ue(1)                        # chroma_format_idc
    bw.ue(p.bit_depth - 8)
    bw.ue(p.bit_depth - 8)
    bw.u(1, 0)                      # qpprime_y_zero_transform_bypass_flag
    bw.u(1, 0)                      # seq_scaling_matrix_present_flag
    bw.ue(p.log2_max_frame_num - 4)
    bw.ue(2)                        # pic_order_cnt_type
    bw.ue(p.num_ref_frames)
    bw.u(1, 0)                      # gaps_in_frame_num_value_allowed_flag
    bw.ue(p.mb_w - 1)
    bw.ue((p.mb_h if p.frame_mbs_only else p.mb_h // 2) - 1)
    bw.u(1, p.frame_mbs_only)
    if not p.frame_mbs_only:
        bw.u(1, p.mbaff)
    bw.u(1, 1)                      # direct_8x8_inference_flag
    bw.u(1, 0)                      # frame_cropping_flag
    bw.u(1, 0)                      # vui_parameters_present_flag
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
    bw.u(1, 0)                      # weighted_pred_flag
    bw.u(2, 0)                      # weighted_bipred_idc
    bw.se(p.init_qp - 26)
    bw.se(0)
    bw.se(p.chroma_qp_offset)
    bw.u(1, 1)                      # deblocking_filter_control_present_flag
    bw.u(1, 0)                      # constrained_intra_pred_flag
    bw.u(1, 0)                      # redundant_pic_cnt_present_flag
    bw.trailing()
    return nal(3, 8, bw.bytes())


I4_DC = 2


class Picture:
    """CAVLC context of the picture being written (frame or field): per 4x4-block coefficient counts, per-macroblock slice number and
    kind, Intra4x4 modes."""

    def __init__(self, mb_w, mb_h):
        self.mb_w, self.mb_h = mb_w, mb_h
        self.tc = [np.zeros((4 * mb_h, 4 * mb_w), np.int32), np.zeros((2 * mb_h, 2 * mb_w), np.int32), np.zeros((2 * mb_h, 2 * mb_w), np.int32)]
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
        self.stats = {"pcm": 0, "i16": 0, "i4": 0, "p16": 0, "p168": 0, "p88": 0, "skip": 0, "coded_blocks": 0, "escapes": 0}

    # ---- helpers -------------------------------------------------------------------------------------------------------------------
    def avail(self, pic, mx, my, sl):
        return 0 <= mx < pic.mb_w and 0 <= my < pic.mb_h and pic.slice_of[my, mx] == sl

    def nC(self, pic, plane, bx, by, sl):
        """9.2.1: from the left (A) and upper (B) neighbouring blocks' total_coeff; a block is available with its macroblock"""
        per = 4 if plane == 0 else 2
        tc = pic.tc[plane]
        a = b = None
        if bx > 0 and self.avail(pic, (bx - 1) // per, by // per, sl):
            a = tc[by, bx - 1]
        if by > 0 and self.avail(pic, bx // per, (by - 1) // per, sl):
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
        if cbp & 0x30:
            for _ in range(2):
                residual_block(bw, self.rand_coefs(4, 0.8), -1, 4)
        if cbp & 0x20:
            for pl in (1, 2):
                for b in range(4):
                    bx, by = 2 * mx + (b & 1), 2 * my + (b >> 1)
                    pic.tc[pl][by, bx] = residual_block(bw, self.rand_coefs(15, 0.2), self.nC(pic, pl, bx, by, sl), 15)

    def qp_delta(self, st):
        """keeps the running QP inside a band where random levels stay far from the coefficient range"""
        lo, hi = self.p.init_qp - 10, self.p.init_qp + 8
        d = int(self.rng.integers(-3, 4))
        if not lo <= st["qp"] + d <= hi:
            d = 0
        st["qp"] += d
        return d

    # ---- macroblocks -----------------------------------------------------------------------------------------------------------------
    def intra_mb(self, bw, pic, mx, my, sl, st, type_offset, allow_i4=True):
        r = self.rng
        left, top = self.avail(pic, mx - 1, my, sl), self.avail(pic, mx, my - 1, sl)
        topleft = self.avail(pic, mx - 1, my - 1, sl)
        if self.small:   # MBAFF: the neighbour derivation is the decoder's business; DC prediction is legal wherever the macroblock sits
            left = top = topleft = False
        k = r.random()
        if k < 0.08 and not self.small:
            # I_PCM: pcm_alignment_zero_bit, 256 + 2 x 64 samples of bit_depth bits
            bw.ue(type_offset + 25)
            bw.align_zero()
            for v in r.integers(0, 1 << self.p.bit_depth, 384):
                bw.u(self.p.bit_depth, int(v))
            pic.tc[0][4 * my:4 * my + 4, 4 * mx:4 * mx + 4] = 16
            for pl in (1, 2):
                pic.tc[pl][2 * my:2 * my + 2, 2 * mx:2 * mx + 2] = 16
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
        # I_NxN with Intra4x4 prediction (transform_8x8_mode_flag = 0: no transform_size_8x8_flag)
        bw.ue(type_offset + 0)
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
            for s in subs:
                bw.ue(s)
            if not ref0:
                for _ in range(4):
                    ref()
            for s in subs:
                for _ in range((1, 2, 2, 4)[s]):
                    self.mvd(bw)
            self.stats["p88"] += 1
        cbp = int(r.integers(0, 48)) if r.random() < 0.7 else 0
        bw.ue(INTER_CBP_CODE[cbp])
        if cbp:
            bw.se(self.qp_delta(st))
        pic.kind[my, mx] = 0
        self.write_residual(bw, pic, mx, my, sl, False, cbp, True, st["qp"])

    # ---- slices and pictures ---------------------------------------------------------------------------------------------------------
    def slice_nal(self, pic, desc, sl, first_mb, end_mb, idr, field):
        p, r = self.p, self.rng
        ptype = desc["type"]
        bw = BitWriter()
        mbaff = p.mbaff and not field
        bw.ue(first_mb >> mbaff)    # MBAFF: in units of macroblock pairs
        bw.ue(2 if ptype == "I" else 0)
        bw.ue(0)
        bw.u(p.log2_max_frame_num, self.frame_num)
        if not p.frame_mbs_only:
            bw.u(1, 1 if field else 0)
            if field:
                bw.u(1, int(field == "bottom"))
        if idr:
            bw.ue(self.idr_id)
        num_ref = desc.get("num_ref", 1)
        if ptype == "P":
            bw.u(1, 1)              # num_ref_idx_active_override_flag
            bw.ue(num_ref - 1)
            bw.u(1, 0)              # ref_pic_list_modification_flag_l0
        if idr:
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
            skip = ptype == "P" and r.random() < 0.22
            if skip:
                skip_run += 1
                pic.kind[my, mx] = 0
                self.stats["skip"] += 1
                prev_skipped = True
                i += 1
                continue
            if ptype == "P":
                bw.ue(skip_run)
                skip_run = 0
            if mbaff and ((addr & 1) == 0 or prev_skipped):
                # mb_field_decoding_flag (7.3.4): with the top macroblock of a pair, or with the bottom one when the top was skipped
                pair_field = int(r.random() < 0.5)
                bw.u(1, pair_field)
            prev_skipped = False
            if ptype == "I":
                self.intra_mb(bw, pic, mx, my, sl, st, 0, allow_i4=not mbaff)
            elif r.random() < 0.2:
                self.intra_mb(bw, pic, mx, my, sl, st, 5, allow_i4=not mbaff)
            else:
                self.inter_mb(bw, pic, mx, my, sl, st, num_ref * 2 if mbaff and pair_field else num_ref)
            i += 1
        if skip_run:
            bw.ue(skip_run)
        bw.trailing()
        return nal(1, 5 if idr else 1, bw.bytes())

    def picture(self, desc, idr=False):
        """one access unit (a frame, or one field) as bytes"""
        p = self.p
        field = desc.get("field")
        mb_h = p.mb_h // 2 if field else p.mb_h
        pic = Picture(p.mb_w, mb_h)
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
            # frame_num: every picture here is a reference picture; the second field of a frame shares its first field's
            if not d.get("field") or d.get("second_field"):
                self.frame_num = (self.frame_num + 1) % (1 << self.p.log2_max_frame_num)
        return aus
