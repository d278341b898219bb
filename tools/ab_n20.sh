#!/bin/bash
# tools/ab_n20.sh variant...: community_small sampling bench (N = 20, 6x32, 16 levels) at 32 and 256 graphs per call, per library variant
lib=discrete_dppm_graphs_amd/libdppm_hip.so
cp $lib /tmp/libdppm_keep.so
for rep in 1 2; do for v in "$@"; do
  if [ $v != keep ]; then cp scratch/ab/libdppm_$v.so $lib; else cp /tmp/libdppm_keep.so $lib; fi
  for b in 32 256; do
    timeout -k 10 200 python bench.py --no-cpu-baseline --no-exact-f32 --no-prof --nodes 20 --hidden 32 --layers 6 --levels 16 --batch $b --steps 20 --warmup 3 --train-steps 0 > gpurun_out/abn_${v}_$b.json 2> gpurun_out/abn_$v.err || { cp /tmp/libdppm_keep.so $lib; exit 1; }
  done
  python - <<PY
import json
a=json.load(open("gpurun_out/abn_${v}_32.json")); b=json.load(open("gpurun_out/abn_${v}_256.json"))
print("$v", "B=32:", round(a["value"],0), "graphs/s, level", round(a["ms_per_step"]/16,4), "ms | B=256:", round(b["value"],0), "graphs/s, level", round(b["ms_per_step"]/16,4), "ms")
PY
done; done
cp /tmp/libdppm_keep.so $lib
