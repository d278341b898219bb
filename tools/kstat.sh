#!/bin/bash
# usage: tools/kstat.sh <file.hip> <kernel-name-regex>   -> VGPRs / spills / scratch of matching kernels (gfx950); EXTRA="-D..." adds flags
cd "$(dirname "$0")/../discrete_dppm_graphs_amd/csrc" || exit 1
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -fno-slp-vectorize -I../../include -I. ${EXTRA} -c "$1" -o "/tmp/kstat_$$.o" \
  -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import sys, re
pat = re.compile(sys.argv[1]); cur = None; rows = {}
for l in sys.stdin:
    if 'error' in l: print(l.rstrip())
    m = re.search(r'Function Name: (\S+)', l)
    if m: cur = m.group(1); rows.setdefault(cur, {}); continue
    m = re.search(r'remark:\s+(VGPRs|AGPRs|VGPRs Spill|ScratchSize \[bytes/lane\]|LDS Size \[bytes/block\]): (\d+)', l)
    if m and cur: rows[cur][m.group(1)] = int(m.group(2))
for k, v in rows.items():
    if pat.search(k): print('%-70s vgpr %3d spill %3d scratch %4d' % (k[:70], v.get('VGPRs', -1), v.get('VGPRs Spill', -1), v.get('ScratchSize [bytes/lane]', -1)))
" "$2"
