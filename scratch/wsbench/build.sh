#!/bin/bash
cd "$(dirname "$0")" && /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -Wno-unused-function -Wno-unused-value -I../../include -I../../discrete_dppm_graphs_amd/csrc bench.hip ../../discrete_dppm_graphs_amd/csrc/dppm_common.hip -o wsbench 2>&1 | grep -E "error" ; ls -la wsbench | awk '{print $6,$7,$8}'
