#!/bin/bash
# Usage (GPU box, repo root): tools/gpu_refresh.sh <tag>  -- the default bench line, kernel-trace stats + bench line for the
# workloads, then FETCH_SIZE / WRITE_SIZE passes for the roofline kernels and the SQ counters of the 3D advection kernels.  Every
# rocprofv3 run is bounded by `timeout`.  The slab workload is profiled with --no-native: its dominant kernel's average must not
# be mixed with the short plane-range launches of the bench line's middle-rank-model leg.
set -u
tag=$1
mkdir -p gpurun_out/prof_$tag
timeout 1500 python bench.py --steps 20 --warmup 5 > gpurun_out/prof_$tag/bench_default.log 2>&1
grep "^{\"metric\"" gpurun_out/prof_$tag/bench_default.log | tail -1 > gpurun_out/prof_$tag/bench_default.json
cp gpurun_out/bench_detail.json gpurun_out/prof_$tag/bench_detail.json
timeout 300 tools/gpu_profile.sh $tag plume3d_slab_jacobi --steps 20 --warmup 3 --no-native
for w in plume3d_256_jacobi plume2d_1024_cnn plume2d_1024_jacobi rt2d_2048_jacobi plume2d_128_jacobi plume2d_128_b32_cnn plume2d_128_b32_jacobi plume2d_1024_cnn_bf16x6 plume2d_1024_cnn_f2; do
  timeout 300 tools/gpu_profile.sh $tag $w --steps 20 --warmup 3
done
for w in plume3d_256_cnn plume3d_hbm_jacobi plume3d_256_cnn_bf16x6 plume3d_256_cnn_f2; do
  timeout 400 tools/gpu_profile.sh $tag $w --steps 5 --warmup 2
done
for w in plume3d_slab_jacobi plume3d_256_jacobi plume2d_1024_cnn rt2d_2048_jacobi plume2d_1024_cnn_bf16x6 plume3d_256_cnn_bf16x6; do
  extra=""; [ $w = plume3d_slab_jacobi ] && extra="--no-native"
  timeout 400 tools/gpu_pmc.sh $tag $w --steps 5 --warmup 1 $extra
  python3 tools/show_pmc.py gpurun_out/pmc_$tag/$w > gpurun_out/pmc_$tag/${w}_pmc_summary.txt 2>&1
done
# SQ counters of a developed 512 x 512 x 64 plume step (the 3D advection tile kernels are VALU-issue bound)
timeout 600 tools/pmc_kernel.sh pmc_$tag/advect3d_sq "SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES" -- python tools/advect_probe.py > gpurun_out/pmc_$tag/advect3d_sq_counters.txt 2>&1
