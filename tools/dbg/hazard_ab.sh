#!/bin/bash
# One GPU call: (1) the ds_write2st64_b64 source-overwrite probe, (2) o1_diff.py (layer-0 MLP output, in_lin-fused against unfused) with
# the in-tree library, the no-select build and the no-select build with two wait states behind every image store.
mkdir -p gpurun_out
lib=discrete_dppm_graphs_amd/libdppm_hip.so
cp $lib /tmp/libdppm_keep.so
timeout -k 10 120 tools/dbg/ds_write2_hazard 20000 > gpurun_out/haz_probe.txt 2>&1 || { echo probe failed; cat gpurun_out/haz_probe.txt; exit 1; }
cat gpurun_out/haz_probe.txt
for v in keep nosel nosel_guard; do
  if [ $v != keep ]; then cp scratch/ab/libdppm_$v.so $lib; fi
  for rep in 1 2; do
    echo "== $v run $rep"
    timeout -k 10 200 python -u tools/dbg/o1_diff.py 2>&1 | tail -8 || { cp /tmp/libdppm_keep.so $lib; exit 1; }
  done
done | tee gpurun_out/haz_ab.txt
cp /tmp/libdppm_keep.so $lib
