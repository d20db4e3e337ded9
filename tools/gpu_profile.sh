#!/bin/bash
# Usage (on the GPU box, from the repo root): tools/gpu_profile.sh <tag> <workload> [bench args...]
# Writes gpurun_out/prof_<tag>/<workload>/ (rocprofv3 kernel trace + stats).
set -u
tag=$1; w=$2; shift 2
export TMPDIR=/tmp
out=gpurun_out/prof_$tag/$w
mkdir -p $out
rocprofv3 --kernel-trace --stats -d $out -o r --output-format csv -- python bench.py --workload $w --no-cpu-baseline ${FNX_BENCH_PROFILE_ARGS:-} "$@" > $out/bench.log 2>&1
grep "^{\"metric\"" $out/bench.log | tail -1 > $out/bench.json
rm -f $out/r_kernel_trace.csv.gz
