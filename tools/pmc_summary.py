#!/usr/bin/env python3
"""tools/pmc_summary.py <gpurun_out> <tag> — folds the rocprofv3 PMC passes of tools/gpu.sh into
profiles/<tag>_pmc.txt (readable) and profiles/<tag>_pmc.json (what bench.py reports as roofline.traffic).
HBM bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE (KB units; the x2 is the gfx950 correction for wide
coalesced reads given in MI355X_MICROARCH.md's HBM section; WRITE_SIZE is taken as reported)."""
import collections
import csv
import glob
import json
import os
import sys

out_dir, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(out_dir, "pmc_*", "*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if k.startswith("void at::") or "rocclr" in k:
            continue
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
lines, js = [], {}
for k, cs in acc.items():
    lines.append(k)
    avg = {c: sum(v) / len(v) for c, v in cs.items()}
    for c in sorted(avg):
        lines.append("    %-22s launches=%d avg=%.6g" % (c, len(cs[c]), avg[c]))
    if "FETCH_SIZE" in avg and "WRITE_SIZE" in avg:
        traffic = (2 * avg["FETCH_SIZE"] + avg["WRITE_SIZE"]) * 1024
        lines.append("    => HBM traffic per launch: 2*FETCH + WRITE = %.4g bytes" % traffic)
        js[k.split("(")[0].replace("void ", "")] = {"fetch_kb": avg["FETCH_SIZE"], "write_kb": avg["WRITE_SIZE"],
                                                     "traffic_bytes_per_launch": traffic}
    if "SQ_ACTIVE_INST_VALU" in avg and "GRBM_GUI_ACTIVE" in avg:
        cyc = avg["GRBM_GUI_ACTIVE"] / 8
        lines.append("    => kernel cycles (GRBM_GUI_ACTIVE / 8 XCDs) %.4g; VALU busy %.1f %% (ACTIVE_INST_VALU*4 / 1024 SIMDs / cycles)"
                     % (cyc, 100 * avg["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / cyc))
    if "SQ_WAVE_CYCLES" in avg:
        w = avg["SQ_WAVE_CYCLES"]
        lines.append("    => of wave cycles: active %.0f %%, issue-stall %.0f %%, waitcnt/barrier %.0f %%"
                     % (100 * avg.get("SQ_ACTIVE_INST_ANY", 0) / w, 100 * avg.get("SQ_WAIT_INST_ANY", 0) / w,
                        100 * avg.get("SQ_WAIT_ANY", 0) / w))
open(os.path.join(root, "profiles", tag + "_pmc.txt"), "w").write("\n".join(lines) + "\n")
open(os.path.join(out_dir, tag + "_pmc.txt"), "w").write("\n".join(lines) + "\n")  # gpurun merges only gpurun_out/ back
json.dump(js, open(os.path.join(root, "profiles", tag + "_pmc.json"), "w"), indent=1)
json.dump(js, open(os.path.join(out_dir, tag + "_pmc.json"), "w"), indent=1)
print("\n".join(lines))
