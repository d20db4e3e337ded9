# Usage (GPU box, repo root): tools/refresh_cnn2d_r06.sh <tag>  -- after the 2D default went back to F(2x2): the GPU suite, the default line, the
# 2D CNN workloads' profiles (kernel stats + PMC passes) and the per-layer trace of both Winograd kernels, into the same gpurun_out/ places
# tools/run_r06_snapshot.sh <tag> writes (tools/collect_snapshot.py <tag> r06 a then copies the generation as a whole)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
tag=${1:-r06b}
export FNX_BENCH_PROFILE_ARGS="--no-dropin"
mkdir -p gpurun_out/prof_$tag
timeout 1800 python -m pytest tests -m gpu -q -x > gpurun_out/${tag}_pytest_gpu.log 2>&1; tail -2 gpurun_out/${tag}_pytest_gpu.log
timeout 1500 python bench.py --steps 20 --warmup 5 > gpurun_out/prof_$tag/bench_default.log 2>&1
grep "^{\"metric\"" gpurun_out/prof_$tag/bench_default.log | tail -1 > gpurun_out/prof_$tag/bench_default.json
cp gpurun_out/bench_detail.json gpurun_out/prof_$tag/bench_detail.json
rm -rf gpurun_out/prof_$tag/plume2d_1024_cnn_f2
for w in plume2d_1024_cnn plume2d_128_b32_cnn plume2d_1024_cnn_f4; do
  timeout 300 tools/gpu_profile.sh $tag $w --steps 20 --warmup 3
done
timeout 400 tools/gpu_pmc.sh $tag plume2d_1024_cnn --steps 5 --warmup 1
python3 tools/show_pmc.py gpurun_out/pmc_$tag/plume2d_1024_cnn > gpurun_out/pmc_$tag/plume2d_1024_cnn_pmc_summary.txt 2>&1
bash tools/trace_cnn_f4.sh > gpurun_out/${tag}_wino4_per_layer_trace.txt 2>&1; cat gpurun_out/${tag}_wino4_per_layer_trace.txt
wc -c gpurun_out/prof_$tag/bench_default.json
