#!/usr/bin/env python3
"""Print a rocprofv3 kernel_stats.csv compactly: tools/show_stats.py <csv> [n]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
tot = sum(float(r['TotalDurationNs']) for r in rows)
print(f"total kernel time {tot/1e6:.3f} ms")
for r in rows[:n]:
    print(f"{r['Name'][:84]:84s} calls={r['Calls']:>6s} avg_us={float(r['AverageNs'])/1e3:9.2f} min={float(r['MinNs'])/1e3:8.2f} max={float(r['MaxNs'])/1e3:8.2f} pct={float(r['Percentage']):5.1f}")
