#!/bin/bash
# tools/ab_h32.sh variant...: 6x32 N = 200 sampling bench per library variant (scratch/ab/libdppm_<v>.so; keep = in-tree), twice
lib=discrete_dppm_graphs_amd/libdppm_hip.so
cp $lib /tmp/libdppm_keep.so
for rep in 1 2; do for v in "$@"; do
  if [ $v != keep ]; then cp scratch/ab/libdppm_$v.so $lib; else cp /tmp/libdppm_keep.so $lib; fi
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-exact-f32 --nodes 200 --hidden 32 --layers 6 --levels 64 --batch 32 --steps 3 --warmup 1 --train-steps 0 > gpurun_out/abh_$v.json 2> gpurun_out/abh_$v.err || { cp /tmp/libdppm_keep.so $lib; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/abh_$v.json")); print("$v", round(d["value"],1), "graphs/s, level", round(d["ms_per_step"]/64,3), "ms;", "  ".join("%s n=%d %.4f" % (k, v["launches"], v["avg_launch_ms"]) for k,v in d["kernels"].items()))
PY
done; done
cp /tmp/libdppm_keep.so $lib
