#!/bin/bash
# timing ablations of w4_mlp_kernel: one library per -DW4_ABLATE=bits (scratch/ab/libdppm_abl<bits>.so; 1 no output stores, 2 no
# second-product MFMAs, 4 no first-product MFMAs, 8 stagers request no new tiles).  Results are wrong on purpose.
lib=discrete_dppm_graphs_amd/libdppm_hip.so
cp $lib /tmp/libdppm_keep.so
tools/bench_quick.sh abl0 --levels 8 --steps 2 | grep -E "mlp2" | sed "s/^/bits 0: /"
for bits in "$@"; do
  cp scratch/ab/libdppm_abl$bits.so $lib
  tools/bench_quick.sh abl$bits --levels 8 --steps 2 | grep -E "mlp2" | sed "s/^/bits $bits: /"
done
cp /tmp/libdppm_keep.so $lib
