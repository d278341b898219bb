#!/bin/bash
# tools/ab_bench.sh "<bench flags>" variant...: quick sampling bench per library variant (scratch/ab/libdppm_<v>.so; keep = in-tree), twice
flags=$1; shift
lib=discrete_dppm_graphs_amd/libdppm_hip.so
cp $lib /tmp/libdppm_keep.so
for rep in 1 2; do for v in "$@"; do
  if [ $v != keep ]; then cp scratch/ab/libdppm_$v.so $lib; else cp /tmp/libdppm_keep.so $lib; fi
  tools/bench_quick.sh $v $flags | grep -E "graphs/s|${ABGREP:-mlp2}" || { cp /tmp/libdppm_keep.so $lib; exit 1; }
done; done
cp /tmp/libdppm_keep.so $lib
