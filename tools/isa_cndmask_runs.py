#!/usr/bin/env python3
"""Counts back-to-back VCC-reading v_cndmask_b32_e32 pairs per kernel in a hipcc --save-temps .s file (on gfx950 the second of two
consecutive VCC-reading v_cndmask costs ~22 cycles instead of ~2.5, tools/ubench/pk_bench.hip): tools/isa_cndmask_runs.py file.s [substr]"""
import re, sys
s = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ''
for m in re.finditer(r'^(_Z\S*' + re.escape(pat) + r'\S*):[^\n]*\n(.*?)^\.Lfunc_end', s, re.S | re.M):
    name, body = m.group(1), m.group(2)
    ins = []
    for l in body.split('\n'):
        l = l.strip()
        if not l or l[0] in ';.' or l.endswith(':'): continue
        ins.append(l.split()[0])
    n32 = sum(1 for i in ins if i == 'v_cndmask_b32_e32')
    pairs = sum(1 for a, b in zip(ins, ins[1:]) if a == 'v_cndmask_b32_e32' and b == 'v_cndmask_b32_e32')
    valu = sum(1 for i in ins if i.startswith('v_'))
    if n32: print(f"{name[:90]:90s} VALU {valu:6d}  cndmask_e32 {n32:5d}  back-to-back {pairs:5d}")
