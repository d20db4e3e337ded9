#!/usr/bin/env python3
"""Compact view of a bench.py JSON line: tools/show_bench.py file.json"""
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
def show(name, r):
    rf = r["roofline"]
    ft = rf.get("frac_traffic"); mu = rf.get("mfma_util")
    print(f"{name:22s} {r['ms_per_step']:9.4f} ms {r['value']:9.1f} Mcells/s  step_frac={r.get('step_hbm_frac', 0):.3f} frac={rf['frac']:.3f}"
          f" frac_traffic={'-' if ft is None else f'{ft:.3f}'} mfma_util={'-' if mu is None else f'{mu:.3f}'}  {r['config']['launch']}")
    print("      ", {k: round(v, 4) for k, v in r["kernel_ms_per_step"].items()})
show(d["config"]["workload"], d)
for k, v in d.get("also", {}).items():
    print(k, v) if "error" in v else show(k, v)
print("cpu_baseline:", d.get("cpu_baseline"))
