# usage: trace_schedule.sh <schedule> <latency_us:GB/s> [first_row] [rows]  -- kernel timeline of one step of the overlap model
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
S=${1:-edge_first}; L=${2:-0:0}; A=${3:-60}; N=${4:-70}
rm -rf gpurun_out/trace_$S
MODEL_LINKS="$L" MODEL_STEPS=2 timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace_$S -o t -- python tools/slab_overlap_model.py $S 6 2>&1 | grep "ms/step"
python - "$S" "$A" "$N" <<'PY'
import csv, glob, sys
S, A, N = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
f = glob.glob(f'gpurun_out/trace_{S}/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# the last step starts at the last advect_fwd launch
idx = max(i for i, r in enumerate(rows) if 'advect_fwd' in r['Kernel_Name'])
rows = rows[idx - 2:]
t0 = int(rows[0]['Start_Timestamp'])
for r in rows[A:A + N]:
    n = r['Kernel_Name'].replace('void (anonymous namespace)::', '')[:48]
    print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:7.1f} q{r.get('Queue_Id', '?')} {n}")
PY
