# Usage (GPU box, repo root): tools/debug/trace_step_span.sh <workload>... -- rocprofv3 --kernel-trace of graph-replayed steps: for the last
# replayed step, the span from its first kernel's start to its last kernel's end, the sum of the kernel durations and the sum per kernel name
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for w in "$@"; do
rm -rf gpurun_out/prof_span
timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_span -o t -- python bench.py --workload $w --no-cpu-baseline --no-dropin --steps 30 --warmup 5 > gpurun_out/prof_span_bench.json 2>/dev/null
python - "$w" <<'PY'
import csv, glob, sys, json, collections
f = glob.glob('gpurun_out/prof_span/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# steps start with the first advection kernel; take the timed region = the 30 steps before the profiling steps (eager); find all starts
idx = [i for i, r in enumerate(rows) if any(k in r['Kernel_Name'] for k in ('advect_fwd_kernel', 'advect2d_fwd', 'advect3d_fwd'))]
ms = json.loads(open('gpurun_out/prof_span_bench.json').read().strip().split('\n')[-1])['ms_per_step']
# replayed steps: consecutive starts at a regular distance; use steps 10..25 of the sequence after the development steps (take the last 45 starts: 5 warmup + 30 timed + 10 eager profile)
idx = idx[-45:]
spans, sums = [], []
per = collections.Counter()
for a, b in zip(idx[8:30], idx[9:31]):
    seg = rows[a:b]
    spans.append((int(rows[b]['Start_Timestamp']) - int(rows[a]['Start_Timestamp'])) / 1e3)
    sums.append(sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in seg) / 1e3)
    for r in seg:
        per[r['Kernel_Name'].replace('void fnx::(anonymous namespace)::', '').replace('fnx::(anonymous namespace)::', '')[:40]] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
n = len(spans)
print(f"== {sys.argv[1]}: bench {ms:.4f} ms/step under the profiler; step-to-step span {sum(spans)/n:.1f} us, sum of kernel durations {sum(sums)/n:.1f} us ({n} replayed steps)")
for k, v in per.most_common(8):
    print(f"   {v/n:9.1f} us  {k}")
PY
done
