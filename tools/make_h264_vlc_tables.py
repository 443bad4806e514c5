#!/usr/bin/env python3
"""tools/make_h264_vlc_tables.py — writes tests/h264_vlc_tables.py: the CAVLC code tables of ITU-T H.264 (Table 9-4: coded_block_pattern
mapping; Table 9-5: coeff_token; Tables 9-7 / 9-8 / 9-9: total_zeros; Table 9-10: run_before) as (length, bits) pairs, for the
test-side bitstream WRITER (tests/h264_bitstream.py).  The numbers are the standard's; they are read here from the arrays the reference
decoder builds its VLC readers from (libavcodec/h264_cavlc.c:35-236, h264data.c:42-58), so that writer and reader agree by construction.  Run in the
container that has /root/reference; the output is committed (the GPU box has no reference tree)."""
import os
import re
import sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
src = open(os.path.join(REF, "libavcodec", "h264_cavlc.c")).read() + open(os.path.join(REF, "libavcodec", "h264data.c")).read()


def array(name):
    m = re.search(r"(?:static )?const uint8_t %s(?:\[[^\]]*\])+\s*=\s*\{(.*?)\};" % re.escape(name), src, re.S)
    assert m, name
    body = re.sub(r"/\*.*?\*/|//[^\n]*", "", m.group(1), flags=re.S)

    def parse(s):
        s = s.strip()
        out, depth, cur, items = [], 0, "", []
        # split top-level by braces
        if "{" not in s:
            return [int(x) for x in s.replace("\n", " ").split(",") if x.strip()]
        i = 0
        while i < len(s):
            if s[i] == "{":
                depth += 1
                if depth == 1:
                    cur = ""
                else:
                    cur += s[i]
            elif s[i] == "}":
                depth -= 1
                if depth == 0:
                    items.append(parse(cur))
                else:
                    cur += s[i]
            elif depth >= 1:
                cur += s[i]
            i += 1
        return items
    return parse(body)


names = ["ff_h264_golomb_to_inter_cbp", "ff_h264_golomb_to_intra4x4_cbp", "golomb_to_inter_cbp_gray", "golomb_to_intra4x4_cbp_gray",
         "chroma_dc_coeff_token_len", "chroma_dc_coeff_token_bits", "chroma422_dc_coeff_token_len", "chroma422_dc_coeff_token_bits",
         "coeff_token_len", "coeff_token_bits", "total_zeros_len", "total_zeros_bits", "chroma_dc_total_zeros_len",
         "chroma_dc_total_zeros_bits", "chroma422_dc_total_zeros_len", "chroma422_dc_total_zeros_bits", "run_len", "run_bits"]
out = ['"""CAVLC code tables of ITU-T H.264 (Tables 9-4, 9-5, 9-7, 9-8, 9-9, 9-10) for the test-side bitstream writer.',
       "Written by tools/make_h264_vlc_tables.py; do not edit.  *_len[i] bits of code *_bits[i]; coeff_token tables are indexed",
       '[4 * total_coeff + trailing_ones]; golomb_to_*_cbp[codeNum] = coded_block_pattern."""', ""]
for n in names:
    out.append("%s = %r" % (n.replace("ff_h264_", ""), array(n)))
open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "h264_vlc_tables.py"), "w").write("\n".join(out) + "\n")
print("wrote tests/h264_vlc_tables.py")
