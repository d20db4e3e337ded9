# Usage (GPU box, repo root): [STEPS=30] tools/debug/trace_step_span.sh <workload>... -- rocprofv3 --kernel-trace of graph-REPLAYED steps (CNN
# a step is marked by its pack_div_kernel / stage3d_kernel / stage2d_div_kernel launch): mean over the timed steps of the step-to-step span, of the sum of the kernel durations
# and of the sum per kernel name.  Kernels launched one by one behind idle gaps (eager traces) start at a lower clock and read 5-10 % longer.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for w in "$@"; do
rm -rf gpurun_out/prof_span
timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_span -o t -- python bench.py --workload $w --no-cpu-baseline --no-dropin --steps ${STEPS:-30} --warmup 5 > gpurun_out/prof_span_bench.json 2>/dev/null
python - "$w" <<'PY'
import csv, glob, sys, json, collections
f = glob.glob('gpurun_out/prof_span/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# steps start with the first advection kernel; take the timed region = the 30 steps before the profiling steps (eager); find all starts
idx = []
for mark in ('pack_div_kernel', 'stage3d_kernel', 'stage2d_div_kernel'):      # one launch per step: CNN steps, 3D / 2D Jacobi steps
    idx = [i for i, r in enumerate(rows) if mark in r['Kernel_Name']]
    if idx:
        break
ms = json.loads(open('gpurun_out/prof_span_bench.json').read().strip().split('\n')[-1])['ms_per_step']
# the run's CNN steps: 5 warm-up (the graph is captured in them) + STEPS timed (replayed) + up to 10 eager ones for the HIP-event profile
import os
nst = int(os.environ.get("STEPS", "30"))
tail = idx[-(nst + 5 + min(nst, 10)):]
timed = tail[5 + 2:5 + nst - 1]                             # replayed steps, without the first two and the last
spans, sums = [], []
per = collections.Counter()
for a, b in zip(timed[:-1], timed[1:]):
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
