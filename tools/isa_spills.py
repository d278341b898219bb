#!/usr/bin/env python3
"""Count scratch (spill) instructions of selected kernels, split into 'inside the MFMA loop' and 'elsewhere'.
usage: isa_spills.py file.s substring [substring...]   (loop = region between the first and last v_mfma of the kernel)"""
import sys
s = open(sys.argv[1]).read()
import re
for m in re.finditer(r'^(_Z\w+):.*\n', s, re.M):
    name = m.group(1)
    if not any(p in name for p in sys.argv[2:]):
        continue
    body = s[m.end():s.index('s_endpgm', m.end())].split('\n')
    mf = [i for i, l in enumerate(body) if 'v_mfma' in l]
    if not mf:
        continue
    inside = [l.strip() for l in body[mf[0]:mf[-1]] if 'scratch_' in l]
    outside = [l for l in body[:mf[0]] + body[mf[-1]:] if 'scratch_' in l]
    ld = sum('scratch_load' in l for l in inside); st = sum('scratch_store' in l for l in inside)
    print(f"{name[:60]:60s} mfma {len(mf):4d}  in-loop scratch loads {ld:3d} stores {st:3d}   outside {len(outside)}")
