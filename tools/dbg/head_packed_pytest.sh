#!/bin/bash
# the determinism test (in its pytest neighbourhood: after the LDS-poison test) against the packed-form read-out library, N times
lib=discrete_dppm_graphs_amd/libdppm_hip.so
cp $lib /tmp/libdppm_keep.so
cp scratch/ab/libdppm_${HEADLIB:-headpacked}.so $lib
for i in $(seq 1 ${1:-4}); do
  timeout -k 10 300 python -u -m pytest tests/test_gpu_ppgn.py -q -m gpu --timeout 300 --timeout-method thread -k "${KSEL:-poison or determin}" 2>&1 | grep -E "differ|passed|failed" | cut -c1-600
done
cp /tmp/libdppm_keep.so $lib
