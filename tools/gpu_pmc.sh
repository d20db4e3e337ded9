#!/bin/bash
# Usage (GPU box, repo root): tools/gpu_pmc.sh <tag> <workload> [bench args]
# Two separate PMC passes (FETCH_SIZE, WRITE_SIZE cannot share a pass: TCC has 4 slots) with kernel trace only.
set -u
tag=$1; w=$2; shift 2
export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  out=gpurun_out/pmc_$tag/$w/$c
  mkdir -p $out
  rocprofv3 --pmc $c --kernel-trace -d $out -o r --output-format csv -- python bench.py --workload $w --no-cpu-baseline --no-graph ${FNX_BENCH_PROFILE_ARGS:-} "$@" > $out/bench.log 2>&1
  rm -f $out/r_kernel_trace.csv
done
