#!/bin/bash
# Determinism of the read-out kernel with and without the compiler's packed form of the epilogue's side-input term (-DW3_SIDE_PACKED,
# scratch/ab/libdppm_headpacked.so): tools/dbg/head_race.py B N L runs per library.  Usage: tools/dbg/head_packed_ab.sh [runs]
runs=${1:-200}
lib=discrete_dppm_graphs_amd/libdppm_hip.so
cp $lib /tmp/libdppm_keep.so
for v in keep headpacked keep; do
  if [ $v != keep ]; then cp scratch/ab/libdppm_$v.so $lib; else cp /tmp/libdppm_keep.so $lib; fi
  echo "== $v"
  timeout -k 10 400 python tools/dbg/head_race.py 32 200 1 $runs || { cp /tmp/libdppm_keep.so $lib; exit 1; }
done
cp /tmp/libdppm_keep.so $lib
