#!/usr/bin/env python3
"""tools/readme_table.py <bench json> — README.md's headline rows from a bench line (the flat scalars of roofline / cpu_baseline)."""
import json
import sys

b = json.load(open(sys.argv[1]))
r, c = b["roofline"], b.get("cpu_baseline", {})


def g(d, k, fmt="%s"):
    return fmt % d[k] if k in d else "—"


rows = [
    ("scaler nv12 1080p→4K bicubic, 256 frames (**the bench metric**, BASELINE configs[1])",
     "%.2f Tpixel/s" % (b["value"] / 1e6),
     "**%s** settled, %s sustained, %s cold; traffic %.3f× algorithmic; %s of the box's own 1 : 4 streaming probe" % (
         g(r, "frac"), g(r, "frac_sustained"), g(r, "frac_cold"), (r.get("traffic") or 0) / r["algorithmic_bytes_per_launch"], g(r, "frac_of_probe_read1_write4")),
     "%s Mpixel/s (`sws_scale_frame`, %s slice threads): %s×" % (g(c, "sws_slice_threads_Mpix"), g(c, "usable_cores"), g(r, "headline_x_cpu"))),
    ("yuv420p→rgb24 4K unscaled, 64 frames (north_star's ≥ 0.70 line)", "%.2f Tpixel/s" % (r.get("rgb24_4k_Mpix", 0) / 1e6),
     "**%s** (numbering the tuner kept: %s; %s of the box's 1 : 2 probe)" % (g(r, "rgb24_4k_frac"), g(r, "rgb24_4k_tuned_numbering"), g(r, "rgb24_4k_frac_of_probe_read1_write2")),
     "%s Mpixel/s on all cores: %s×" % (g(c, "rgb24_4k_all_cores_Mpix"), g(r, "rgb24_4k_x_cpu_all_cores"))),
    ("H.264 8×8 IDCT + add, 32 4K planes (BASELINE's second metric; north_star: ≥ 10× the CPU)", "%s G blocks/s" % g(r, "idct8_Gblocks"), "**%s**" % g(r, "idct8_frac"),
     "%s G blocks/s on all cores: **%s×**" % (g(c, "idct8_all_cores_Gblocks"), g(r, "idct8_x_cpu_all_cores"))),
    ("H.264 luma qpel 16×16, mixed positions, 8 4K planes (configs[2])", "%s Mpixel/s" % g(r, "qpel16_mixed_Mpix"), g(r, "qpel16_mixed_frac"),
     "%s Mpixel/s: %s×" % (g(c, "h264_qpel16_mixed_all_cores_Mpix"), g(r, "qpel16_mixed_x_cpu_all_cores"))),
    ("H.264 v / h luma loop filter, one edge per 16×16 tile of 8 4K planes (configs[2])", "%s / %s M edges/s" % (g(r, "h264_v_loop_filter_luma_Medges"), g(r, "h264_h_loop_filter_luma_Medges")), "—",
     "%s / %s M edges/s: %s× / %s×" % (g(c, "h264_v_loop_filter_luma_all_cores_Medges"), g(c, "h264_h_loop_filter_luma_all_cores_Medges"),
                                       g(r, "h264_v_loop_filter_luma_x_cpu_all_cores"), g(r, "h264_h_loop_filter_luma_x_cpu_all_cores"))),
    ("MDCT-1024 forward / inverse, 65,536 transforms (configs[3])", "%s / %s M transforms/s" % (g(r, "mdct1024_fwd_Mtransforms"), g(r, "mdct1024_inv_Mtransforms")),
     "%s / %s" % (g(r, "mdct1024_fwd_frac"), g(r, "mdct1024_inv_frac")),
     "%s / %s M transforms/s: %s× / %s×" % (g(c, "mdct1024_fwd_all_cores_Mtransforms"), g(c, "mdct1024_inv_all_cores_Mtransforms"), g(r, "mdct1024_fwd_x_cpu_all_cores"), g(r, "mdct1024_inv_x_cpu_all_cores"))),
    ("full search 16×16, R = 7, SAD / SATD, 8 pairs of 4K planes (configs[4])", "%.0f / %.0f M MB-searches/s" % (r.get("me_esa_sad_r7_MBsearches", 0) / 1e6, r.get("me_esa_satd_r7_MBsearches", 0) / 1e6),
     "%s / %s of the issue roof" % (g(r, "me_esa_sad_r7_issue_frac"), g(r, "me_esa_satd_r7_issue_frac")),
     "%s / %s MB-searches/s: %s× / %s×" % (g(c, "me_esa_sad_r7_all_cores_MBsearches"), g(c, "me_esa_satd_r7_all_cores_MBsearches"), g(r, "me_esa_sad_r7_x_cpu_all_cores"), g(r, "me_esa_satd_r7_x_cpu_all_cores"))),
    ("`ffhip_sws_scale` on pageable host frames, nv12 1080p→4K (PCIe both ways)", "%s ms per frame" % g(r, "sws_host_pointer_ms_per_frame"), "—", "one thread: %s Mpixel/s" % g(c, "sws_1_thread_Mpix")),
]
print("| workload | rate | of the 8 TB/s HBM peak | the reference's C on this box's %s usable cores (%s) |" % (g(c, "usable_cores"), g(c, "cpu_model")))
print("|---|---|---|---|")
for row in rows:
    print("| %s | %s | %s | %s |" % row)
