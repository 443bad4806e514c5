#!/bin/bash
# tools/pmc_pass.sh (COUNTERS="SQ_WAVES ..." in the environment; SQ_WAVES must be among them) <kernel-name-substring> <out-file> -- <command...>
# Instruction mix of the kernels whose name contains the substring: rocprofv3 PMC pass (counters only, with the kernel trace), summed
# over the launches of the command and divided by their number.  Run on the GPU box (gpurun); the summary goes to <out-file>.
pat="$1"; out="$2"; shift 3
root="${GRAFT_REPO_ROOT:-/root/repo}"
dir="$root/gpurun_out/pmc_insts"
rm -rf "$dir"; mkdir -p "$dir"
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --pmc $COUNTERS \
    -d "$dir" -o pmc --output-format csv -- "$@" > "$dir/cmd.log" 2>&1 )
PAT="$pat" python - "$dir" <<'PY' | tee "$out"
import csv, glob, collections, os, sys
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for fn in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        k = r["Kernel_Name"]
        if os.environ["PAT"] not in k:
            continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVES":
            n[k] += 1
for k in acc:
    print(k[:90], "launches", n[k], {c: round(v / max(n[k], 1)) for c, v in sorted(acc[k].items())})
PY
