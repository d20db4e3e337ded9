#!/bin/bash
# Usage (GPU box, repo root): tools/gpu_refresh.sh <tag>  -- the default bench line, kernel-trace stats + bench line for the
# workloads, then FETCH_SIZE / WRITE_SIZE passes for the roofline kernels.  Every rocprofv3 run is bounded by `timeout`.
set -u
tag=$1
mkdir -p gpurun_out/prof_$tag
timeout 900 python bench.py > gpurun_out/prof_$tag/bench_default.log 2>&1
grep "^{\"metric\"" gpurun_out/prof_$tag/bench_default.log | tail -1 > gpurun_out/prof_$tag/bench_default.json
for w in plume3d_slab_jacobi plume3d_256_jacobi plume2d_1024_cnn plume2d_1024_jacobi rt2d_2048_jacobi plume2d_128_jacobi plume2d_128_b32_cnn plume2d_1024_cnn_bf16x6; do
  timeout 300 tools/gpu_profile.sh $tag $w --steps 20 --warmup 3
done
for w in plume3d_256_cnn plume3d_hbm_jacobi plume3d_256_cnn_bf16x6; do
  timeout 400 tools/gpu_profile.sh $tag $w --steps 5 --warmup 2
done
for w in plume3d_slab_jacobi plume3d_256_jacobi plume2d_1024_cnn rt2d_2048_jacobi plume2d_1024_cnn_bf16x6 plume3d_256_cnn_bf16x6; do
  timeout 400 tools/gpu_pmc.sh $tag $w --steps 5 --warmup 1
  python3 tools/show_pmc.py gpurun_out/pmc_$tag/$w > gpurun_out/pmc_$tag/${w}_pmc_summary.txt 2>&1
done
