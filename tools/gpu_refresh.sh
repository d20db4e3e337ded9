#!/bin/bash
# Usage (GPU box, repo root): tools/gpu_refresh.sh <tag>  -- kernel-trace stats + bench line for the main workloads, then
# FETCH_SIZE / WRITE_SIZE passes for the roofline kernels.  Every rocprofv3 run is bounded by `timeout`.
set -u
tag=$1
for w in plume3d_slab_jacobi plume3d_256_jacobi plume2d_1024_cnn plume2d_1024_jacobi rt2d_2048_jacobi; do
  timeout 300 tools/gpu_profile.sh $tag $w --steps 20 --warmup 3
done
for w in plume3d_slab_jacobi plume3d_256_jacobi plume2d_1024_cnn; do
  timeout 400 tools/gpu_pmc.sh $tag $w --steps 5 --warmup 1
  python3 tools/show_pmc.py gpurun_out/pmc_$tag/$w > gpurun_out/pmc_$tag/${w}_pmc_summary.txt 2>&1
done
