#!/bin/bash
# GPU box: per-launch durations of one MultiScaleNet forward (rocprofv3 kernel trace of tools/cnn_wino_probe.py).
# Usage: tools/trace_wino_probe.sh [res] [depth]   (env FNX_* switches pass through)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_wino
timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_wino -o t -- python tools/cnn_wino_probe.py "$@" > gpurun_out/prof_wino.log 2>&1
tail -1 gpurun_out/prof_wino.log
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/prof_wino/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'resize_kernel' in r['Kernel_Name']]
# last forward = the last 5 resize launches; start a little before the first of them
a = idx[-5] - 1
for r in rows[a:]:
    n = r['Kernel_Name'].replace('void fnx::(anonymous namespace)::', '').replace('fnx::(anonymous namespace)::', '')[:50]
    print(f"{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:8.1f} us  grid {r.get('Grid_Size_X','?'):>7}x{r.get('Grid_Size_Y','?'):>5}x{r.get('Grid_Size_Z','?'):>4} wg {r.get('Workgroup_Size_X','?'):>4} vgpr {r.get('VGPR_Count','?')} lds {r.get('LDS_Block_Size','?')}  {n}")
PY
rm -rf gpurun_out/prof_wino
