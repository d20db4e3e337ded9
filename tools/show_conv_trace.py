#!/usr/bin/env python3
"""Durations of the conv kernels of the LAST forward in a rocprofv3 kernel trace: tools/show_conv_trace.py trace.csv [n=17]"""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'conv' in r['Kernel_Name'] and 'pack' not in r['Kernel_Name']]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 17
tot = 0.0
for r in rows[-n:]:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    tot += d
    k = r['Kernel_Name']; k = k[k.find('conv'):k.find('(')][:40]
    print(f"{k:42s} grid={r.get('Grid_Size_X', '?'):>8s} {d:10.1f} us")
print(f"sum {tot/1e3:.2f} ms")
