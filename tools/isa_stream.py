#!/usr/bin/env python3
"""The instruction-class stream of a kernel's loops from a hipcc -S --cuda-device-only listing (M MFMA, v VALU, d LDS, b buffer/global,
B barrier, w s_waitcnt, n s_nop, c branch, s other SALU, | label): how the filler work sits between the MFMAs.
   python tools/isa_stream.py listing.s kernel_substring [all]"""
import re
import sys


def main():
    s = open(sys.argv[1]).read()
    parts = re.split(r'\n\t\.type\t(_Z\S+),@function\n', s)
    for i in range(1, len(parts), 2):
        if sys.argv[2] not in parts[i]:
            continue
        body = parts[i + 1].split('.Lfunc_end')[0]
        out, inloop = [], len(sys.argv) > 3
        for l in body.split('\n'):
            if 'Loop Header' in l:
                inloop = True
            if not inloop:
                continue
            m = re.match(r'\s+([a-z_0-9]+)', l)
            if not m:
                if l.startswith('.LBB'):
                    out.append('|')
                continue
            op = m.group(1)
            out.append('M' if 'mfma' in op else 'v' if op.startswith('v_') else 'd' if op.startswith('ds_') else 'B' if 'barrier' in op else
                       'w' if 'waitcnt' in op else 'b' if op.startswith(('buffer', 'global', 'flat', 'scratch')) else 'n' if op == 's_nop' else
                       'c' if 'cbranch' in op or op == 's_branch' else 's')
        print(parts[i][:80])
        print(''.join(out))


if __name__ == "__main__":
    main()
