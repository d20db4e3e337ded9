#!/usr/bin/env python3
"""Instruction histogram of one kernel in a hipcc --save-temps .s file: tools/isa_stats.py file.s substring [top]."""
import collections, re, sys
s = open(sys.argv[1]).read()
pat = sys.argv[2]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 25
for m in re.finditer(r'^(\S*' + re.escape(pat) + r'\S*):.*?\n(.*?)\n\s*s_endpgm', s, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if name.startswith('.'): continue
    c = collections.Counter()
    for l in body.split('\n'):
        l = l.strip()
        if not l or l[0] in ';.' or l.endswith(':'): continue
        c[l.split()[0]] += 1
    tot = sum(c.values()); valu = sum(n for k, n in c.items() if k.startswith('v_'))
    print(f"{name}: {tot} instrs, {valu} VALU, scratch ops {sum(n for k, n in c.items() if k.startswith('scratch_'))}")
    print("  " + ", ".join(f"{k} {n}" for k, n in c.most_common(top)))
    k = re.search(r'\.amdhsa_kernel ' + re.escape(name) + r'\n(.*?)\.end_amdhsa_kernel', s, re.S)
    if k:
        for key in ("next_free_vgpr", "next_free_sgpr", "private_segment_fixed_size", "group_segment_fixed_size"):
            mm = re.search(key + r'\s+(\S+)', k.group(1))
            if mm: print(f"  {key} = {mm.group(1)}")
