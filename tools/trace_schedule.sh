# usage: trace_schedule.sh <schedule> <latency_us:GB/s> [first_row] [rows] [native] [w] [halo]  -- kernel timeline of one step of the overlap model
# (5th argument "native": the C++ driver with the link-model communicator, tools/slab_native_model.py)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
S=${1:-deep_first}; L=${2:-0:0}; A=${3:-0}; N=${4:-120}; TOOL=tools/slab_overlap_model.py
if [ "${5:-}" = native ]; then TOOL=tools/slab_native_model.py; fi
rm -rf gpurun_out/trace_$S
MODEL_LINKS="$L" MODEL_STEPS=2 timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace_$S -o t -- env MODEL_GRAPH=0 python $TOOL $S ${6:-6} ${7:-} 2>&1 | grep "ms/step"
python - "$S" "$A" "$N" <<'PY'
import csv, glob, re, sys
S, A, N = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
f = glob.glob(f'gpurun_out/trace_{S}/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# the last step starts at the third-last forward advection launch (interior window + two edge windows per step)
idxs = [i for i, r in enumerate(rows) if 'advect3d_fwd_tile' in r['Kernel_Name']]
rows = rows[idxs[-3]:]
t0 = int(rows[0]['Start_Timestamp'])
prev_end = None
for r in rows[A:A + N]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    n = re.sub(r'\(.*', '', r['Kernel_Name'].replace('void (anonymous namespace)::', ''))[:56]
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    prev_end = max(prev_end or 0, e)
    print(f"{(s - t0) / 1e3:9.1f} dur {(e - s) / 1e3:7.1f} gap {gap:6.1f} q{r['Queue_Id']} grid {int(r['Grid_Size_X']) // max(int(r['Workgroup_Size_X']), 1):6d} {n}")
PY
