#!/bin/bash
# Usage (GPU box, repo root): tools/gpu_kernel_stats.sh <tag> <python script> [args]  -- rocprofv3 kernel stats of a probe script,
# compact summary into gpurun_out/<tag>_stats.txt
set -u
tag=$1; shift
export TMPDIR=/tmp
out=gpurun_out/ks_$tag
mkdir -p $out
timeout 600 rocprofv3 --kernel-trace --stats -d $out -o r --output-format csv -- python "$@" > $out/run.log 2>&1
python3 tools/show_stats.py $out/r_kernel_stats.csv 24 > gpurun_out/${tag}_stats.txt 2>&1
rm -f $out/r_kernel_trace.csv
