# Usage (GPU box, repo root): per-launch durations of the last CNN step, F(2x2) against F(4x4) for the 64- / 128-output-channel 3x3(x3) layers:
# 1024^2 and 256^3, fp32_f2 against the default.  (Eager launches behind the profiler's gaps: each kernel starts at a lower clock -- for
# the kernels as they run in a step use tools/debug/trace_step_span.sh.)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for w in plume2d_1024_cnn_f2 plume2d_1024_cnn plume3d_256_cnn_f2 plume3d_256_cnn; do
rm -rf gpurun_out/prof_cnn
timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_cnn -o t -- python bench.py --workload $w --no-cpu-baseline --no-dropin --steps 3 --warmup 1 --no-graph > /dev/null 2>&1
echo "== $w"
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/prof_cnn/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'pack_div' in r['Kernel_Name'] or 'pack_input' in r['Kernel_Name']]
a = idx[-1]
for r in rows[a:a + 40]:
    n = r['Kernel_Name'].replace('void fnx::(anonymous namespace)::', '').replace('fnx::(anonymous namespace)::', '')[:48]
    if 'conv' in n:
        print(f"{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:8.1f} us  wgs {int(r.get('Grid_Size_X', 0)) // max(int(r.get('Workgroup_Size_X', 1)), 1):6d}  {n}")
PY
done
