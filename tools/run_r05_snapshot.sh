# Usage (GPU box, repo root): the round-5 snapshot -- full GPU suite, the default line + per-workload rocprofv3 stats + PMC passes
# (tools/gpu_refresh.sh), the middle-rank link model for every schedule / direct-send setting, the peer-store probe.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
tag=${1:-r05p}
timeout 1800 python -m pytest tests -m gpu -q -x > gpurun_out/${tag}_pytest_gpu.log 2>&1; tail -3 gpurun_out/${tag}_pytest_gpu.log
bash tools/gpu_refresh.sh $tag > gpurun_out/${tag}_refresh.log 2>&1
export MODEL_LINKS="0:0,9:150,9:75,9:55,9:40,20:75,25:55" MODEL_STEPS=20 MODEL_GRAPH=force
rm -f gpurun_out/${tag}_link_model.txt
MODEL_DIRECT=auto timeout 900 python tools/slab_native_model.py deep_first,deep_beside 6 6 2>&1 | grep "ms/step" >> gpurun_out/${tag}_link_model.txt
MODEL_DIRECT=never timeout 900 python tools/slab_native_model.py deep_beside 6 6 2>&1 | grep "ms/step" >> gpurun_out/${tag}_link_model.txt
cat gpurun_out/${tag}_link_model.txt
timeout 300 python tools/peer_probe.py 2>&1 | grep peer-store > gpurun_out/${tag}_peer_probe.txt; cat gpurun_out/${tag}_peer_probe.txt
