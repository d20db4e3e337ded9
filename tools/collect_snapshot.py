#!/usr/bin/env python3
"""Copy what tools/run_r06_snapshot.sh <tag> left under gpurun_out/ into profiles/<round>/<prefix>_* (the tracked, judged copies):
   python tools/collect_snapshot.py r06b r06 a
bench_default.json / bench_detail.json, per workload <w>_kernel_stats.csv + <w>_bench.json, pmc/<prefix>_<w>_pmc_summary.txt,
pmc/<prefix>_advect3d_sq_counters.txt, and the snapshot's text outputs (pytest log, link model, peer probe, advection A/B, per-layer
trace, eight-rank rehearsal)."""
import glob
import os
import shutil
import sys

tag, rnd, pre = sys.argv[1:4]
repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(repo, "gpurun_out"), os.path.join(repo, "profiles", rnd)
os.makedirs(os.path.join(dst, "pmc"), exist_ok=True)


def cp(a, b):
    if os.path.exists(a) and os.path.getsize(a) > 0:
        shutil.copy(a, b)
        print("  ", os.path.relpath(b, repo))
    else:
        print("   MISSING", os.path.relpath(a, repo))


p = os.path.join(src, f"prof_{tag}")
cp(os.path.join(p, "bench_default.json"), os.path.join(dst, f"{pre}_bench_default.json"))
cp(os.path.join(p, "bench_detail.json"), os.path.join(dst, f"{pre}_bench_detail.json"))
for d in sorted(glob.glob(os.path.join(p, "*", ""))):
    w = os.path.basename(os.path.dirname(d))
    cp(os.path.join(d, "r_kernel_stats.csv"), os.path.join(dst, f"{pre}_{w}_kernel_stats.csv"))
    cp(os.path.join(d, "bench.json"), os.path.join(dst, f"{pre}_{w}_bench.json"))
for f in sorted(glob.glob(os.path.join(src, f"pmc_{tag}", "*_pmc_summary.txt"))):
    cp(f, os.path.join(dst, "pmc", f"{pre}_{os.path.basename(f)}"))
cp(os.path.join(src, f"pmc_{tag}", "advect3d_sq_counters.txt"), os.path.join(dst, "pmc", f"{pre}_advect3d_sq_counters.txt"))
for name in ["pytest_gpu.log", "link_model.txt", "peer_probe.txt", "advect_ab.txt", "wino4_per_layer_trace.txt",
             "bench_rehearsal_8_ranks_one_gpu.json", "long_parity.txt", "step_span.txt"]:
    cp(os.path.join(src, f"{tag}_{name}"), os.path.join(dst, f"{pre}_{name}"))
