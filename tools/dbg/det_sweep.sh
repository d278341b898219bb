#!/bin/bash
run() { TAG="$1" timeout -k 10 280 python tools/dbg/determinism.py 32 200 1 60 2>&1 | grep -v amdgpu.ids | tail -1; }
run default
DPPM_DISABLE_W4=1 DPPM_DISABLE_INLINFUSE=1 run noW4
DPPM_DISABLE_PLANE4=1 run plane3
DPPM_DISABLE_INLINFUSE=1 run noinl
DPPM_W3MLP=1 DPPM_DISABLE_INLINFUSE=1 run w3mlp_noinl
DPPM_DISABLE_NORMFUSE=1 run nonormfuse
