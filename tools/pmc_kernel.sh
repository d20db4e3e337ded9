#!/bin/bash
# Usage (GPU box, repo root): tools/pmc_kernel.sh <outdir-under-gpurun_out> "<counters pass 1>" ["<counters pass 2>" ...] -- <command>
# One rocprofv3 --pmc pass per counter group (kernel trace only), then per-kernel means via tools/show_pmc.py-style CSV.
set -u
out=gpurun_out/$1; shift
groups=()
while [ "$1" != "--" ]; do groups+=("$1"); shift; done
shift
export TMPDIR=/tmp
i=0
for grp in "${groups[@]}"; do
  d=$out/pass$i; mkdir -p $d
  timeout 300 rocprofv3 --pmc $grp --kernel-trace -d $d -o r --output-format csv -- "$@" > $d/cmd.log 2>&1
  rm -f $d/r_kernel_trace.csv
  i=$((i+1))
done
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pass*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(out + "/summary.txt", "w") as fh:
    for k, cs in agg.items():
        fh.write(k + "\n")
        for c, v in sorted(cs.items()):
            fh.write(f"  {c:28s} mean {sum(v)/len(v):16.1f}  n={len(v)}\n")
print(open(out + "/summary.txt").read())
PY
