# Usage (GPU box, repo root): tools/trace_cnn_variants.sh <name>...  -- per-launch durations of the conv3_wino4_kernel launches of one
# 1024^2 CNN step for each variants/libfluidnet_hip_<name>.so (tools/ab_libs.sh build), rocprofv3 --kernel-trace
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
cp fluidnet_cxx_amd/libfluidnet_hip.so /tmp/libfluidnet_hip.keep
for v in "$@"; do
cp variants/libfluidnet_hip_$v.so fluidnet_cxx_amd/libfluidnet_hip.so
rm -rf gpurun_out/prof_cnn
timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_cnn -o t -- python bench.py --workload plume2d_1024_cnn --no-cpu-baseline --no-dropin --steps 3 --warmup 1 --no-graph > /dev/null 2>&1
python - "$v" <<'PY'
import csv, glob, sys
f = glob.glob('gpurun_out/prof_cnn/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'pack_div' in r['Kernel_Name'] or 'pack_input' in r['Kernel_Name']]
a = idx[-1]
t = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in rows[a:a + 40] if 'conv3_wino4' in r['Kernel_Name']]
print(f"{sys.argv[1]:>10}: " + " ".join(f"{x:7.1f}" for x in t) + f"   sum {sum(t):7.1f} us")
PY
done
cp /tmp/libfluidnet_hip.keep fluidnet_cxx_amd/libfluidnet_hip.so
