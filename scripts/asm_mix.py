#!/usr/bin/env python3
"""Static instruction mix of a kernel's loops from its assembly (hipcc -S --cuda-device-only): VALU by kind (real arithmetic,
v_readlane / v_writelane = scalar registers spilled into vector lanes, compares, selects, moves), SALU, LDS, memory -- per loop
depth.  Usage: asm_mix.py file.s <substring of the mangled kernel name> ..."""
import collections
import re
import sys


def analyze(t, name):
    i = t.index(name + ':')
    j = t.index('.Lfunc_end', i)
    agg = collections.Counter()
    depth = 0
    for l in t[i:j].splitlines():
        s = l.strip()
        m = re.match(r'^(\.LBB\d+_\d+):\s*(;.*)?$', s)
        if m:
            d = re.search(r'Depth=(\d+)', m.group(2) or '')
            depth = int(d.group(1)) if d else 0
            continue
        if not s or s.startswith(';') or s.startswith('.'):
            continue
        op = s.split()[0]
        if op.startswith('v_readlane') or op.startswith('v_writelane'): k = 'lane'
        elif op.startswith('v_cndmask'): k = 'select'
        elif op.startswith('v_cmp'): k = 'cmp'
        elif op.startswith('v_mov') or op.startswith('v_accvgpr'): k = 'mov'
        elif op.startswith('v_'): k = 'valu'
        elif op.startswith('s_'): k = 'salu'
        elif op.startswith('ds_'): k = 'lds'
        elif op.startswith('global_') or op.startswith('scratch') or op.startswith('flat'): k = 'mem'
        else: k = 'other'
        agg[(min(depth, 3), k)] += 1
    return agg


if __name__ == '__main__':
    t = open(sys.argv[1]).read()
    names = re.findall(r'^(_Z\S+):', t, re.M)
    for pat in sys.argv[2:]:
        for n in names:
            if pat in n:
                a = analyze(t, n)
                print(n[-60:])
                for d in (1, 2, 3):
                    print('   depth %d: ' % d + '  '.join('%s %d' % (k, a[(d, k)]) for k in ('valu', 'lane', 'cmp', 'select', 'mov', 'salu', 'lds', 'mem')))
