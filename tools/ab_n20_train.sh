#!/bin/bash
# tools/ab_n20_train.sh variant...: community_small (N = 20, 6x32) sampling + training bench at 32 graphs per call, per library variant
lib=discrete_dppm_graphs_amd/libdppm_hip.so
cp $lib /tmp/libdppm_keep.so
for rep in 1 2; do for v in "$@"; do
  if [ $v != keep ]; then cp scratch/ab/libdppm_$v.so $lib; else cp /tmp/libdppm_keep.so $lib; fi
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-exact-f32 --no-prof --nodes 20 --hidden 32 --layers 6 --levels 16 --batch 32 --steps 20 --warmup 3 --train-steps 200 > gpurun_out/abt_${v}.json 2> gpurun_out/abt_$v.err || { cp /tmp/libdppm_keep.so $lib; exit 1; }
  python - <<PY
import json
a=json.load(open("gpurun_out/abt_${v}.json"))
print("$v", "sampling", round(a["value"],0), "graphs/s | training", round(a["train"]["train_steps_per_s"],1), "steps/s")
PY
done; done
cp /tmp/libdppm_keep.so $lib
