cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for w in plume2d_128_jacobi plume2d_1024_jacobi; do
python bench.py --workload $w --no-cpu-baseline --steps 200 --warmup 20 | cut -c1-700 | tr "," "\n" | grep -i "ms_per_step\|\"jacobi\"\|advect\|stage"
rm -rf gpurun_out/t128
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/t128 -o t -- python bench.py --workload $w --no-cpu-baseline --steps 3 --warmup 1 --no-graph > /dev/null 2>&1
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/t128/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'advect_fwd' in r['Kernel_Name']]
a = idx[-2]; b = idx[-1]
t0 = int(rows[a]['Start_Timestamp'])
for r in rows[a:b]:
    n = r['Kernel_Name'].replace('void (anonymous namespace)::', '')[:50]
    print(f"{(int(r['Start_Timestamp'])-t0)/1e3:8.1f} {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:7.1f} us  {n}")
PY
done
