#!/usr/bin/env python3
"""Instruction counts per kernel from a hipcc -S --cuda-device-only listing: VALU / SALU / LDS / scratch / readlane / IEEE divisions.
   python tools/isa_counts.py listing.s substring [substring ...]"""
import re
import sys


def main():
    s = open(sys.argv[1]).read()
    keys = sys.argv[2:]
    parts = re.split(r'\n\t\.type\t(_Z\S+),@function\n', s)
    for i in range(1, len(parts), 2):
        name, f = parts[i], parts[i + 1]
        if keys and not any(k in name for k in keys):
            continue
        body = f.split('.Lfunc_end')[0]
        lines = body.split('\n')
        cnt = lambda pat: len([l for l in lines if re.match(pat, l)])  # noqa: E731
        print(name[:90])
        print('   valu', cnt(r'\s+v_'), 'salu', cnt(r'\s+s_'), 'ds', cnt(r'\s+ds_'), 'vmem', cnt(r'\s+(buffer_|global_|flat_)'),
              'scratch ld/st', cnt(r'\s+scratch_load'), cnt(r'\s+scratch_store'), 'readlane/writelane', cnt(r'\s+v_readlane'), cnt(r'\s+v_writelane'),
              'div_fixup', cnt(r'\s+v_div_fixup'), 'rcp', cnt(r'\s+v_rcp'), 'sqrt', cnt(r'\s+v_sqrt'), 'barrier', cnt(r'\s+s_barrier'), 'waitcnt', cnt(r'\s+s_waitcnt'))


if __name__ == "__main__":
    main()
