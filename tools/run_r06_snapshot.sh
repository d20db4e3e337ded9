# Usage (GPU box, repo root): tools/run_r06_snapshot.sh <tag>  -- the round-6 snapshot: full GPU suite, the default line (N = 1 headline
# plume3d_256_jacobi, dropin rows), per-workload rocprofv3 kernel stats + PMC passes (tools/gpu_refresh.sh with --no-dropin: the
# reference-shaped legs launch other kernels and must not enter a workload's per-kernel means), the middle-rank link model, the
# peer-store probe, the 3D advection A/B (two backward marches / one) and the eight-rank rehearsal of the multi-GPU job on the one GPU.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
tag=${1:-r06a}
export FNX_BENCH_PROFILE_ARGS="--no-dropin"
timeout 1800 python -m pytest tests -m gpu -q -x > gpurun_out/${tag}_pytest_gpu.log 2>&1; tail -3 gpurun_out/${tag}_pytest_gpu.log
bash tools/gpu_refresh.sh $tag > gpurun_out/${tag}_refresh.log 2>&1
export MODEL_LINKS="0:0,9:150,9:75,9:55,9:40,20:75,25:55" MODEL_STEPS=20 MODEL_GRAPH=force
rm -f gpurun_out/${tag}_link_model.txt
MODEL_DIRECT=auto timeout 900 python tools/slab_native_model.py deep_first,deep_beside 6 6 2>&1 | grep "ms/step" >> gpurun_out/${tag}_link_model.txt
cat gpurun_out/${tag}_link_model.txt
timeout 300 python tools/peer_probe.py 2>&1 | grep peer-store > gpurun_out/${tag}_peer_probe.txt; cat gpurun_out/${tag}_peer_probe.txt
bash tools/trace_cnn_f4.sh > gpurun_out/${tag}_wino4_per_layer_trace.txt 2>&1; cat gpurun_out/${tag}_wino4_per_layer_trace.txt
timeout 300 python tools/advect_ab.py 512 64 > gpurun_out/${tag}_advect_ab.txt 2>&1; cat gpurun_out/${tag}_advect_ab.txt
timeout 600 python bench.py --gpus 8 --steps 5 --warmup 2 --rehearse-one-gpu --no-cpu-baseline 2>/dev/null | grep '^{"metric"' > gpurun_out/${tag}_bench_rehearsal_8_ranks_one_gpu.json; wc -c gpurun_out/${tag}_bench_rehearsal_8_ranks_one_gpu.json
