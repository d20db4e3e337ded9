cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
tag=${1:-r05t}
mkdir -p gpurun_out/prof_$tag
timeout 900 python bench.py > gpurun_out/prof_$tag/bench_default.log 2>&1
grep "^{\"metric\"" gpurun_out/prof_$tag/bench_default.log | tail -1 > gpurun_out/prof_$tag/bench_default.json
cp gpurun_out/bench_detail.json gpurun_out/prof_$tag/bench_detail.json
for w in plume2d_1024_cnn plume2d_128_b32_cnn plume2d_1024_cnn_bf16x6; do
  timeout 300 tools/gpu_profile.sh $tag $w --steps 20 --warmup 3
done
for w in plume3d_256_cnn plume3d_256_cnn_bf16x6; do
  timeout 400 tools/gpu_profile.sh $tag $w --steps 5 --warmup 2
done
for w in plume2d_1024_cnn plume2d_1024_cnn_bf16x6 plume3d_256_cnn_bf16x6; do
  timeout 400 tools/gpu_pmc.sh $tag $w --steps 5 --warmup 1
  python3 tools/show_pmc.py gpurun_out/pmc_$tag/$w > gpurun_out/pmc_$tag/${w}_pmc_summary.txt 2>&1
done
for w in plume2d_1024_cnn_bf16x3 plume3d_256_cnn_bf16x3; do
  timeout 300 python bench.py --workload $w --steps 5 --warmup 2 2>/dev/null | grep "^{\"metric\"" | tail -1 > gpurun_out/prof_$tag/${w}_bench.json
done
ls gpurun_out/prof_$tag | head -50
wc -c gpurun_out/prof_$tag/bench_default.json
