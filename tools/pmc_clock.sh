#!/bin/bash
# GPU box: the shader clock a kernel actually ran at = GRBM_GUI_ACTIVE (GPU-active cycles of the dispatch) / its duration.
# Usage: tools/pmc_clock.sh <outdir-under-gpurun_out> -- <command>
set -u
out=gpurun_out/$1; shift; shift
export TMPDIR=/tmp
mkdir -p $out
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace -d $out -o r --output-format csv -- "$@" > $out/cmd.log 2>&1
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
f = glob.glob(out + "/**/*counter_collection.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
agg = collections.defaultdict(list)
for r in rows:
    if r["Counter_Name"] != "GRBM_GUI_ACTIVE": continue
    dur = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    if dur > 0: agg[r["Kernel_Name"][:70]].append((float(r["Counter_Value"]), dur))
with open(out + "/clock.txt", "w") as fh:
    for k, v in sorted(agg.items(), key=lambda kv: -sum(d for _, d in kv[1])):
        cyc = sum(c for c, _ in v); ns = sum(d for _, d in v)
        fh.write(f"{k:70s} n={len(v):4d} mean {ns / len(v) / 1e3:9.1f} us  {cyc / ns:6.3f} GHz (GRBM_GUI_ACTIVE / duration)\n")
print(open(out + "/clock.txt").read())
PY
