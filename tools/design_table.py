#!/usr/bin/env python3
"""tools/design_table.py <driver BENCH_rNN.json> <builder bench json> — DESIGN.md §5's table: one row per kernel with what bounds it, its
algorithmic bytes per unit, and two measured columns — the DRIVER's record of the previous round (everything its 8 KB of stdout tail still
holds) and this round's closing `python bench.py --steps 20 --warmup 5` on a pool box (profiles/).  Fractions are of the 8 TB/s HBM3E peak
unless a unit is given."""
import json
import re
import sys


def driver_lines(path):
    d = json.load(open(path))
    tail = d["run"]["stdout_tail"]
    out = {}
    for m in re.finditer(r'"([a-z0-9_]+)": \{([^{}]*)\}', tail):
        try:
            out[m.group(1)] = json.loads("{" + m.group(2) + "}")
        except ValueError:
            pass
    p = d.get("parsed") or {}
    out["_roofline"] = p.get("roofline", {})
    out["_value"] = p.get("value")
    return out


def fmt(e, keys=("hbm_frac",)):
    if not e:
        return "—"
    for k in keys:
        if k in e:
            v = e[k]
            return ("%.3f" % v) if isinstance(v, float) and v < 10 else str(v)
    return "—"


ROWS = [
    # (kernel, what, bound, bytes/unit, extras key, value keys)
    ("`k_sws_up2<6,1,0,0,1>` (`sws_up2.hip`) — **the bench kernel**", "nv12 / yuv420p exact-2× bicubic, H+V fused on a static schedule; round 6: horizontal bank in SGPRs, six rows in flight, non-temporal stores", "HBM writes (80 % of the bytes) with the VALU 70 % busy beside them", "1.875 B / output px (15,552,000 B per 1080p→4K frame)", "_headline", None),
    ("`k_yuv420p_rgb24_t` (`sws_yuv2rgb.hip`)", "unscaled yuv420p→rgb24 4K, 64 frames (north_star's ≥ 0.70 line); round 6: the workgroup numbering chosen per context by measurement", "HBM", "4.5 B / px", "yuv420p_rgb24_4k", ("hbm_frac",)),
    ("`k_h264_idct8_add` (`h264_idct.hip`)", "8×8 IDCT + add over 32 4K luma planes (BASELINE's second metric)", "HBM", "384 B / block", "h264_idct8_add", ("hbm_frac",)),
    ("`k_sws_down2`", "nv12 4K→1080p bicubic (exact 2:1)", "HBM / VALU", "7.5 B / output px", "sws_nv12_4k_to_1080p_bicubic", ("hbm_frac",)),
    ("`k_sws_up2_rgb` (`sws_up2rgb.hip`)", "yuv420p 1080p→rgb24 4K (exact 2× with the packed-RGB writer fused); bgra", "VALU (74 % busy)", "3.375 / 4.375 B / output px", ("sws_yuv420p_1080p_to_rgb24_4k_bicubic", "sws_yuv420p_1080p_to_bgra_4k_bicubic"), ("hbm_frac",)),
    ("`k_sws_down2_rgb`, `k_sws_eq_rgb`", "nv12 4K→rgb24 1080p; nv12→rgb24 at the source's size", "HBM / VALU", "9 B / 4.5 B per output px", ("sws_nv12_4k_to_rgb24_1080p_bicubic", "sws_nv12_1080p_to_rgb24_1080p_bicubic"), ("hbm_frac",)),
    ("`k_sws_up2<…,RC>`, `k_sws_up2` on 4:4:4", "yuvj420p→yuv420p 1080p→4K (range conversion between the passes); yuv444p 1080p→4K", "as the bench kernel", "1.875 / 3.75 B / output px", ("sws_yuvj420p_1080p_to_yuv420p_4k_bicubic", "sws_yuv444p_1080p_to_4k_bicubic"), ("hbm_frac",)),
    ("`k_sws_up2<.,.,1>`, `k_sws_down2<1>`", "exact 2× / 2:1 at 10 bits: p010 1080p→4K, yuv420p10 1080p→4K, p010 4K→1080p, p010 4K→nv12 1080p", "HBM / VALU", "3.75 B / output px (up), 15 (down)", ("sws_p010_1080p_to_4k_bicubic", "sws_yuv420p10_1080p_to_4k_bicubic", "sws_p010_4k_to_1080p_bicubic", "sws_p010_4k_to_nv12_1080p_bicubic"), ("hbm_frac",)),
    ("`k_sws_up32` (`sws_up32.hip`), `k_sws_down32h` (`sws_down32.hip`)", "round 6, exact 3:2 and 4:3 above 8 bits on static schedules: p010 720p→1080p (period 2 in, 3 out; was the walker at 0.36), yuv420p10 1080p→1440p (3 in, 4 out; was 0.40), p010 4K→1440p (3 in, 2 out; was 0.37)", "HBM / latency", "4.33 / 4.69 / 9.75 B per output px", ("sws_p010_720p_to_1080p_bicubic", "sws_yuv420p10_1080p_to_1440p_bicubic", "sws_p010_4k_to_1440p_bicubic", "sws_p010_1440p_to_1080p_bicubic"), ("hbm_frac",)),
    ("`k_sws_walk16` (`sws_walk16.hip`)", "every other ratio above 8 bits, RGB sources, deeper sources into RGB (first stage); round 6: a wave's source row segment through LDS when the picture grows", "VALU + issue", "by conversion", ("sws_p010_1080p_to_bgra_1080p_bicubic",), ("hbm_frac",)),
    ("`k_sws_rgb420`, `k_sws_rgb_in` with direct planes (`sws_rgbin.hip`)", "round 6: packed RGB into 4:2:0 / planar 4:4:4 at the source's size in one kernel: bgra 1080p→nv12 (was two stages at 0.19), bgra 1080p→yuv444p (0.17)", "HBM", "5.5 / 7 B / px", ("sws_bgra_1080p_to_nv12_1080p_bicubic", "sws_bgra_1080p_to_yuv444p_1080p"), ("hbm_frac",)),
    ("`k_sws_down32`", "nv12 1080p→720p (exact 3:2); round 6: rows straight-line, dots in hand-scheduled blocks (was 0.31)", "VALU", "4.875 B / output px", "sws_nv12_1080p_to_720p_bicubic", ("hbm_frac",)),
    ("`k_yuv444_rgb_full`, `k_sws_copy420`, yuv444p→yuv420p", "conversions at the source's size on streaming kernels", "HBM", "6 / 3 / 4.5 B / px", ("sws_yuv444p_1080p_to_rgb24_1080p", "sws_nv12_1080p_to_yuv420p_1080p", "sws_yuv444p_1080p_to_yuv420p_1080p"), ("hbm_frac",)),
    ("`sws_uops` generated kernels", "`SwsOpBackend` micro-op lists on 4K pictures: yuv444p→rgb24, rgb24→yuv444p, yuv444p10→rgb48, rgba→argb", "HBM", "by list", ("sws_ops_yuv444p_rgb24_4k", "sws_ops_rgb24_yuv444p_4k", "sws_ops_yuv444p10_rgb48_4k", "sws_ops_rgba_argb_4k"), ("hbm_frac",)),
    ("`k_h264_qpel_m` (`h264_qpel.hip`)", "luma MC, every 16×16 block of 8 / 32 4K planes, mixed positions, ±24 (BASELINE configs[2])", "the per-wave dependent chain; ceiling measured 0.29 – 0.36 (`profiles/r05_qpel_limits.txt`)", "2 B / sample", ("h264_qpel16_mixed", "h264_qpel16_mixed_32_planes"), ("hbm_frac",)),
    ("`k_h264_loop_filter`", "function-level luma edge batches, h / v", "requests", "268 B / edge", ("h264_h_loop_filter_luma", "h264_v_loop_filter_luma"), ("hbm_frac",)),
    ("`k_h264_deblock_skew`", "frame-order luma deblocking of 4K planes: ms per plane alone / in a batch of 32", "a lone wave's chain along mb_w + mb_h", "2.25 B / px", "h264_deblock_frame_4k", ("ms_per_frame_one_stream", "ms_per_frame_batch_of_32")),
    ("`k_h264_intra_frame`", "1080p I-picture through the picture layer: ms alone; pictures/s at 32 wavefronts per launch", "the dependency chain mb_w + 2 mb_h", "—", "h264_intra_picture_1080p", ("ms_per_picture", "wavefront_pictures_per_s_32_per_launch")),
    ("`FFHipH264Picture` P-pictures 1080p", "MC + weights + IDCT + deblock per picture: ms alone; pictures/s for 16 flushed together", "—", "—", "h264_picture_pipeline_1080p", ("ms_per_picture_alone", "pictures_per_s_batched_flush_of_16")),
    ("`k_hevc_idct`, `k_hevc_idct32_mfma`", "HEVC 8×8 / 16×16 / 32×32 inverse transform + add_residual", "HBM / LDS", "6 B / sample", ("hevc_idct8_add", "hevc_idct16_add", "hevc_idct32_add"), ("hbm_frac",)),
    ("`k_hevc_qpel_m` (`hevc_qpel_m.hip`), `k_vp9_mc_m` (`vp9_mc.hip`), `k_h264_chroma_mc`", "round 6, the block-MC family on the matrix cores: put_hevc_qpel_uni 16×16, mixed positions; VP9 8-tap MC 16×16, three sets mixed; H.264 chroma MC 8×8 (VALU)", "as block MC", "2 B / sample", ("hevc_qpel_uni16_mixed", "vp9_mc16_8tap_mixed", "h264_chroma_mc8_mixed"), ("hbm_frac",)),
    ("`k_txw` (`tx_wide.hip`)", "round 6: AV_TX_INT32_MDCT 1024 forward, 16,384 transforms, bit-exact fixed point", "LDS network + 64-bit products", "12,288 B per MDCT", "mdct1024_int32_fwd", ("hbm_frac",)),
    ("`k_dcst1_m` (`tx_dcst1.hip`)", "round 6: AV_TX_FLOAT_DCT_I / _DST_I of 64 reals (wmavoice), 2^20 transforms: the matrix on `v_mfma_f64_16x16x4_f64`", "FP64 matrix rate (78.6 TFLOP/s: 9.6 G transforms/s)", "512 B, 8,192 flop per transform", ("dctI_64", "dstI_64"), ("fp64_TFLOP/s", "hbm_frac")),
    ("`k_vp9_itxfm`, `k_vp9_lf_frame_wg`", "VP9 32×32 inverse transform + add; superblock-order loop filter of a 4K picture: ms alone / pictures/s at 32 per launch", "HBM / chain", "6 B / sample", ("vp9_itxfm32_add", "vp9_loopfilter_frame_4k"), ("hbm_frac", "ms_per_picture_one_stream")),
    ("`k_me_esa_g` (`me_cmp.hip`)", "exhaustive SAD search 16×16, R = 7, 8 pairs of 4K planes (BASELINE configs[4])", "`v_sad_u8` issue", "57,600 abs-diff / MB", "me_esa_sad_r7", ("MB-searches/s", "sad_issue_roof_frac")),
    ("`k_me_esa_satd_mx` (`me_satd.hip`)", "the same search with the 8×8 Hadamard cost on the matrix cores, 8 / 32 pairs", "VALU issue (one `v_sad_u32` per coefficient)", "—", ("me_esa_satd_r7", "me_esa_satd_r7_32_pairs"), ("MB-searches/s",)),
    ("`k_mdct_r`, `k_fft_r` (`tx_radix.hip`)", "MDCT-1024 forward / inverse, 65,536 transforms (BASELINE configs[3]); FFT-1024; FFT-16384", "the box's 1 : 1 read : write roof", "12,288 / 8,192 B per MDCT; 16 B / FFT point", ("mdct1024_fwd", "mdct1024_inv", "fft1024_fwd", "fft16384"), ("hbm_frac",)),
    ("`k_fft_z`, `k_mdct_pfa` (`tx_api.hip`)", "the reference-order network: FFT-1024 under `FFHIP_TX_BITEXACT`, FFT-960, IMDCT-960, MDCT-1536", "LDS latency chain", "as above", ("fft1024_fwd_bitexact", "fft960_pfa", "imdct960_pfa15", "mdct1536_pfa3_fwd"), ("hbm_frac",)),
    ("`k_dct`, `k_fdsp`, `k_aac_*`", "DCT-II / DCT-III 1024; vector_fmul_window; AAC imdct_and_windowing (long)", "LDS passes / HBM", "8 / 16 / 18,432 B", ("dct2_1024", "dct3_1024", "fdsp_vector_fmul_window_1024", "aac_imdct_and_windowing_long"), ("hbm_frac",)),
    ("`ffhip_sws_scale` on host frames", "the SwsFunc face, PCIe both ways: ms per nv12 1080p→4K frame (pageable memory)", "PCIe", "15.5 MB / frame", "sws_host_pointer_end_to_end", ("ms_per_frame",)),
]


def cell(lines, key, vkeys):
    if key == "_headline":
        r = lines.get("_roofline", {})
        if not r:
            return "—"
        s = "**%.4f** settled" % r["frac"]
        if "frac_sustained" in r:
            s += ", %.4f sustained, %.4f cold" % (r["frac_sustained"], r.get("frac_cold", float("nan")))
        s += " (%.4f ms" % r["kernel_ms"]
        if r.get("traffic"):
            s += ", traffic %.3f×" % (r["traffic"] / r["algorithmic_bytes_per_launch"])
        return s + ")"
    keys = key if isinstance(key, tuple) else (key,)
    return " / ".join(fmt(lines.get(k), vkeys) if len(vkeys) == 1 else "; ".join(fmt(lines.get(k), (vk,)) for vk in vkeys) for k in keys)


def main():
    drv = driver_lines(sys.argv[1])
    b = json.load(open(sys.argv[2]))
    cur = dict(b.get("extras", {}))
    cur["_roofline"] = b["roofline"]
    print("| kernel (file) | what the line measures | bound | algorithmic bytes | driver r05 | round 6 (builder's box) |")
    print("|---|---|---|---|---|---|")
    for k, what, bound, byt, key, vk in ROWS:
        print("| %s | %s | %s | %s | %s | %s |" % (k, what, bound, byt, cell(drv, key, vk), cell(cur, key, vk)))


if __name__ == "__main__":
    main()
