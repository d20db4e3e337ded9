cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05n
export MODEL_LINKS="0:0,9:150,9:75,9:55,9:40" MODEL_STEPS=20
for d in 1 0; do
  MODEL_DIRECT=$d MODEL_GRAPH=force timeout 900 python tools/slab_native_model.py deep_first,deep_beside 6 6 2>&1 | grep "ms/step" >> gpurun_out/r05n/model_direct.txt
done
cat gpurun_out/r05n/model_direct.txt
