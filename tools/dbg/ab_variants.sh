#!/bin/bash
# tools/dbg/ab_variants.sh "<command>" variant [variant...]: runs the command once per library variant scratch/ab/libdppm_<variant>.so
# ("keep" = the in-tree library) inside one GPU call and restores the in-tree library.
cmd=$1; shift
mkdir -p gpurun_out
lib=discrete_dppm_graphs_amd/libdppm_hip.so
cp $lib /tmp/libdppm_keep.so
for v in "$@"; do
  if [ $v != keep ]; then cp scratch/ab/libdppm_$v.so $lib; else cp /tmp/libdppm_keep.so $lib; fi
  echo "== $v"
  timeout -k 10 300 bash -c "$cmd" 2>&1 | grep -v amdgpu.ids | tail -${TAILN:-12} || { cp /tmp/libdppm_keep.so $lib; exit 1; }
done | tee gpurun_out/ab_variants.txt
cp /tmp/libdppm_keep.so $lib
