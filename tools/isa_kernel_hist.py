#!/usr/bin/env python3
"""Instruction histogram of whole kernels (label .. .Lfunc_end) in a hipcc --save-temps .s file, grouped by class:
tools/isa_kernel_hist.py file.s substring [top]"""
import collections, re, sys
s = open(sys.argv[1]).read()
pat = sys.argv[2]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 30
for m in re.finditer(r'^(_Z\S*' + re.escape(pat) + r'\S*):[^\n]*\n(.*?)^\.Lfunc_end', s, re.S | re.M):
    name, body = m.group(1), m.group(2)
    c = collections.Counter()
    for l in body.split('\n'):
        l = l.strip()
        if not l or l[0] in ';.' or l.endswith(':'): continue
        c[l.split()[0]] += 1
    tot = sum(c.values())
    cls = collections.Counter()
    for k, n in c.items():
        if k.startswith('v_'): cls['VALU'] += n
        elif k.startswith('ds_'): cls['LDS'] += n
        elif k.startswith(('buffer_', 'global_', 'scratch_', 'flat_')): cls['VMEM'] += n
        elif k.startswith('s_'): cls['SALU'] += n
        else: cls['other'] += n
    print(f"{name[:100]}: {tot} instrs " + " ".join(f"{k}={v}" for k, v in cls.most_common()))
    print("  " + ", ".join(f"{k} {n}" for k, n in c.most_common(top)))
