#!/bin/bash
# tools/ab_n20_chain.sh: community_small sampling bench (N = 20, 6x32, 16 levels), the one-launch chain against the level-by-level loop
for rep in 1 2; do for v in chain levels; do
  for b in 32 256; do
    if [ $v = levels ]; then export DPPM_DISABLE_SGCHAIN=1; else unset DPPM_DISABLE_SGCHAIN; fi
    timeout -k 10 200 python bench.py --no-cpu-baseline --no-exact-f32 --no-prof --nodes 20 --hidden 32 --layers 6 --levels 16 --batch $b --steps 20 --warmup 3 --train-steps 0 > gpurun_out/abc_${v}_$b.json 2> gpurun_out/abc_$v.err || exit 1
  done
  python - <<PY
import json
a=json.load(open("gpurun_out/abc_${v}_32.json")); b=json.load(open("gpurun_out/abc_${v}_256.json"))
print("$v", "B=32:", round(a["value"],0), "graphs/s, level", round(a["ms_per_step"]/16,4), "ms | B=256:", round(b["value"],0), "graphs/s, level", round(b["ms_per_step"]/16,4), "ms")
PY
done; done
