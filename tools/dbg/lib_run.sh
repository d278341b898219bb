#!/bin/bash
# usage: tools/dbg/lib_run.sh <runs> "<command>" <tag>...: run a command against scratch/ab/libdppm_<tag>.so (keep = in-tree), <runs> times each
n=$1; cmd=$2; shift; shift
lib=discrete_dppm_graphs_amd/libdppm_hip.so
cp $lib /tmp/libdppm_keep.so
for v in "$@"; do
  if [ $v != keep ]; then cp scratch/ab/libdppm_$v.so $lib; else cp /tmp/libdppm_keep.so $lib; fi
  for i in $(seq 1 $n); do echo "== $v run $i"; timeout -k 10 300 bash -c "$cmd" 2>&1 | tail -6; done
done
cp /tmp/libdppm_keep.so $lib
