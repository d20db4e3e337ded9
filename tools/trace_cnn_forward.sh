cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_cnn
timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_cnn -o t -- python bench.py --workload plume2d_1024_cnn --no-cpu-baseline --steps 3 --warmup 1 --no-graph > /dev/null 2>&1
python - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/prof_cnn/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# last forward: find last 'unscale'
idx = [i for i, r in enumerate(rows) if 'pack_div' in r['Kernel_Name'] or 'pack_input' in r['Kernel_Name']]
a = idx[-1]
for r in rows[a:a + 40]:
    n = r['Kernel_Name'].replace('void fnx::(anonymous namespace)::', '').replace('fnx::(anonymous namespace)::', '')[:60]
    print(f"{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:8.1f} us  grid {r.get('Grid_Size_X','?')}x{r.get('Grid_Size_Y','?')}x{r.get('Grid_Size_Z','?')}  {n}")
PY
