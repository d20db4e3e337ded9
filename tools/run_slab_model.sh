cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05c
export MODEL_LINKS="0:0,15:150,20:75,25:55,30:40" MODEL_STEPS=20
for cfg in "lagged 4 8" "lagged 6 12" "lagged 3 6" "lagged 2 6" "lagged 8 16" "deep_first,deep_beside 6 6"; do
  set -- $cfg
  MODEL_GRAPH=0 timeout 600 python tools/slab_native_model.py $1 $2 $3 >> gpurun_out/r05c/model_eager.txt 2>&1
done
MODEL_LINKS="0:0,20:75,25:55" timeout 300 python tools/slab_native_model.py lagged 4 8 > gpurun_out/r05c/model_graph_lagged4.txt 2>&1
cat gpurun_out/r05c/model_eager.txt; cat gpurun_out/r05c/model_graph_lagged4.txt
