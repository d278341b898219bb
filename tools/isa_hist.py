#!/usr/bin/env python3
"""Histogram of instruction mnemonics between the first and last v_mfma of the kernels whose names contain the given substrings."""
import sys, re
from collections import Counter
s = open(sys.argv[1]).read()
for m in re.finditer(r'^(_Z\w+):.*\n', s, re.M):
    name = m.group(1)
    if not any(p in name for p in sys.argv[2:]):
        continue
    body = s[m.end():s.index('s_endpgm', m.end())].split('\n')
    mf = [i for i, l in enumerate(body) if 'v_mfma' in l]
    c = Counter()
    for l in body[mf[0]:mf[-1] + 1]:
        l = l.strip()
        if not l or l.startswith(';') or l.startswith('.'):
            continue
        op = l.split()[0]
        if op == 's_waitcnt':
            op = ' '.join(l.split()[:2])
        c[op] += 1
    print(name)
    print('   ' + ', '.join(f'{k}:{v}' for k, v in c.most_common(40)))
