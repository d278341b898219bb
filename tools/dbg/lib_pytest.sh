#!/bin/bash
# usage: tools/dbg/lib_pytest.sh <runs> <tag>...: the determinism neighbourhood of tests against scratch/ab/libdppm_<tag>.so
n=$1; shift
lib=discrete_dppm_graphs_amd/libdppm_hip.so
cp $lib /tmp/libdppm_keep.so
for v in "$@"; do
  cp scratch/ab/libdppm_$v.so $lib
  for i in $(seq 1 $n); do
    echo "== $v run $i"
    timeout -k 10 300 python -u -m pytest tests/test_gpu_ppgn.py -q -m gpu --timeout 300 --timeout-method thread -k "head or determin or poison or headline or n512" 2>&1 | grep -E "differ|passed|failed" | cut -c1-300
  done
done
cp /tmp/libdppm_keep.so $lib
