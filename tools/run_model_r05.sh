cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05e
export MODEL_LINKS="0:0,9:150,9:75,9:55,9:40,20:75,25:55" MODEL_STEPS=20
MODEL_GRAPH=0 timeout 600 python tools/slab_native_model.py deep_first,deep_beside 6 6 > gpurun_out/r05e/model_eager.txt 2>&1
MODEL_GRAPH=force MODEL_LINKS="0:0,9:75,9:55,20:75" timeout 600 python tools/slab_native_model.py deep_first,deep_beside 6 6 > gpurun_out/r05e/model_graph.txt 2>&1
grep -v amdgpu.ids gpurun_out/r05e/model_eager.txt; grep -v amdgpu.ids gpurun_out/r05e/model_graph.txt
timeout 300 python -m pytest tests/test_slab.py -m gpu -x -q -k "link_model or model" 2>&1 | tail -3
