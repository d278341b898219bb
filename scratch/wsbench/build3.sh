#!/bin/bash
cd "$(dirname "$0")" && /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -Wno-unused-function -Wno-unused-value -I../../include -I../../discrete_dppm_graphs_amd/csrc ${1:-bench3.hip} ../../discrete_dppm_graphs_amd/csrc/dppm_common.hip -o ${2:-wsbench3} --save-temps=obj 2>&1 | grep -E "error|warning: v|spill" ; ls -la ${2:-wsbench3} | awk '{print $6,$7,$8}'
