#!/bin/bash
# builds the product library, then the diagnostic one (dppm_smallgraph.hip with -DSG_STAMP) next to this script
set -e
cd "$(dirname "$0")/../../discrete_dppm_graphs_amd/csrc"
make -j4 2>&1 | grep -E "error|Error" || true
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -w -fno-slp-vectorize -I../../include -I. -DSG_STAMP -c dppm_smallgraph.hip -o /tmp/sg_diag.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../../scratch/sgdiag/libdppm_hip_diag.so dppm_common.o dppm_diffusion.o dppm_ppgn.o dppm_head.o dppm_w4.o dppm_w6.o dppm_plane4.o /tmp/sg_diag.o dppm_optim.o dppm_stats.o
