#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc output: mean counter value per kernel.  tools/show_pmc.py <dir with FETCH_SIZE/ WRITE_SIZE/>"""
import csv, glob, os, sys, collections
root = sys.argv[1]
res = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    fs = glob.glob(os.path.join(root, c, "*counter_collection.csv"))
    if not fs:
        continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if r.get("Counter_Name") == c:
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        res[k][c] = (sum(v) / len(v), len(v))
for k, d in sorted(res.items(), key=lambda kv: -sum(x[0] for x in kv[1].values())):
    f = d.get("FETCH_SIZE", (0, 0)); w = d.get("WRITE_SIZE", (0, 0))
    print(f"{k[:70]:70s} n={f[1]:5d} FETCH_SIZE={f[0]/1024:10.2f} MiB  WRITE_SIZE={w[0]/1024:10.2f} MiB (raw counter KiB/1024)")
