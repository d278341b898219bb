#!/bin/bash
# usage: tools/kstat.sh <file.hip> <kernel-name-substring>   -> VGPRs / spills / scratch of matching kernels (gfx950)
cd "$(dirname "$0")/../discrete_dppm_graphs_amd/csrc" || exit 1
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -I../../include -I. ${EXTRA} -c "$1" -o "${1%.hip}.o" \
  -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "error|Function Name|VGPRs:|VGPRs Spill|ScratchSize|SGPRs:" | grep -A4 "$2" | grep -v "^--"
