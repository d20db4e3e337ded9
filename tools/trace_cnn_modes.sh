# usage: trace_cnn_modes.sh [2d|3d]  -- kernel timeline (start, duration, gap to the previous kernel's end) of the last forward of each precision mode
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
C=${1:-2d}
rm -rf gpurun_out/trace_cnn_$C
timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace_cnn_$C -o t -- python tools/cnn_mode_time.py $C 2>&1 | grep "forward"
python - "$C" <<'PY'
import csv, glob, re, sys
C = sys.argv[1]
f = glob.glob(f'gpurun_out/trace_cnn_{C}/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# forwards start at the first resize_kernel of a group of launches; print the LAST TWO forwards that contain conv3_wbf and the last two that do not
starts = [i for i, r in enumerate(rows) if 'resize_kernel' in r['Kernel_Name'] and (i == 0 or 'conv_direct_kernel<1' in rows[i - 1]['Kernel_Name'] or 'resize' not in rows[i - 1]['Kernel_Name'] and 'conv' not in rows[i - 1]['Kernel_Name'])]
def show(a, b):
    t0 = int(rows[a]['Start_Timestamp']); prev = None
    for r in rows[a:b]:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        n = re.sub(r'\(.*', '', r['Kernel_Name'].replace('void fnx::(anonymous namespace)::', '').replace('fnx::(anonymous namespace)::', ''))[:44]
        gap = (s - prev) / 1e3 if prev else 0.0
        prev = max(prev or 0, e)
        print(f"{(s - t0) / 1e3:9.1f} dur {(e - s) / 1e3:8.1f} gap {gap:7.1f} grid {int(r['Grid_Size_X']) // max(int(r['Workgroup_Size_X']), 1):6d} {n}")
wbf = [i for i in range(len(starts) - 1) if any('conv3_wbf' in r['Kernel_Name'] for r in rows[starts[i]:starts[i + 1]])]
non = [i for i in range(len(starts) - 1) if i not in wbf and any('conv3_wino3' in r['Kernel_Name'] for r in rows[starts[i]:starts[i + 1]])]
for name, lst in (("fp32", non), ("bf16x6", wbf)):
    if lst:
        i = lst[-2] if len(lst) > 1 else lst[-1]
        print(f"--- {name}: forward {i}")
        show(starts[i], starts[i + 1])
PY
