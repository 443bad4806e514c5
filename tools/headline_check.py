#!/usr/bin/env python3
"""tools/headline_check.py <kernel_stats.csv> <bench log> — the anchor recomputed from rocprofv3's own numbers.

The CSV is `rocprofv3 --kernel-trace --stats` of `bench.py --no-extras --no-cpu-baseline --no-pmc` (tools/final_run.sh): every call of
the headline kernel in it is a 256-frame launch.  algorithmic bytes per launch (frames x 15,552,000) / AverageNs of that
row is compared with roofline.achieved (settled) and roofline.frac_sustained of the JSON line the same run printed (3 % of the latter)."""
import csv
import json
import sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    line = [l for l in open(sys.argv[2]) if l.startswith("{")][-1]
    b = json.loads(line)
    frames = b["config"].get("frames_per_gpu", b["config"].get("frames", 256))
    bytes_per_launch = frames * 15552000
    up2 = [r for r in rows if "k_sws_up2" in r["Name"]]
    if not up2:
        print("no k_sws_up2 row in", sys.argv[1])
        return 1
    r = max(up2, key=lambda r: int(r["Calls"]))
    avg_ns = float(r["AverageNs"])
    achieved = bytes_per_launch / avg_ns   # B/ns == GB/s
    # round 6: the traced command runs the cold steps, the settle phase, the K timed steps and >= 1.2 s of sustained launches; the CSV's
    # average is over ALL of them, i.e. the sustained state with the settled burst inside it.  It is compared with both figures of the
    # same (traced) run's JSON line.
    rf = b["roofline"]
    want = rf["achieved"]
    dev = abs(achieved - want) / want
    print("kernel %s: %s calls, AverageNs %.0f (min %s, max %s)" % (r["Name"][:40], r["Calls"], avg_ns, r["MinNs"], r["MaxNs"]))
    print("bytes/launch %d / AverageNs -> %.1f GB/s = %.4f of 8 TB/s; bench.py roofline.achieved (settled, %d steps) %.1f GB/s (frac %.4f); deviation %.2f %%"
          % (bytes_per_launch, achieved, achieved / 8000.0, b["steps"], want, rf["frac"], 100 * dev))
    if rf.get("kernel_ms_sustained"):
        sus = bytes_per_launch / (rf["kernel_ms_sustained"] * 1e6)
        dev = abs(achieved - sus) / sus
        print("roofline.frac_sustained %.4f (%d launches back to back): %.1f GB/s; deviation of the CSV average from it %.2f %%"
              % (rf["frac_sustained"], rf["sustained_launches"], sus, 100 * dev))
    return 0 if dev <= 0.03 else 2


if __name__ == "__main__":
    sys.exit(main())
