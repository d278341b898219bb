#!/bin/bash
# Run on the GPU box (through gpurun) from the repo root: bench.py at the four graph sizes the north star names
# (N = 20 / 60 / 200 / 512 with the reference configs' model sizes), one JSON line each -> gpurun_out/sizes_<tag>.jsonl
# usage: bash tools/size_sweep.sh <tag>
set -u
TAG=${1:-r01}
OUT=gpurun_out/sizes_$TAG.jsonl
: > "$OUT"
run() { timeout -k 10 "$1" python3 bench.py --no-cpu-baseline --no-exact-f32 "${@:2}" >> "$OUT" 2>> gpurun_out/sizes_$TAG.err || { echo "failed: ${*:2}"; exit 1; }; echo "done: ${*:2}"; }
# community_20_small: 6 x 32, 16 levels, batch 32 (train_com_small_ddpm.yaml / sample_com_small_ddpm_16.yaml)
run 200 --no-prof --nodes 20 --hidden 32 --layers 6 --levels 16 --batch 32 --steps 20 --warmup 3 --train-steps 20
# planar_60_200: 8 x 128, batch 64 (BASELINE configs[2]), 64 levels
run 200 --no-prof --nodes 60 --hidden 128 --layers 8 --levels 64 --batch 64 --steps 4 --warmup 1 --train-steps 6
# sbm_200: the headline (BASELINE configs[3])
run 300 --nodes 200 --hidden 128 --layers 8 --levels 64 --batch 32 --steps 2 --warmup 1 --train-steps 3
# sbm_200 with the 6 x 32 network (gridsearch variant of configs[3]): the hidden-32 kernels at N > 20
run 200 --nodes 200 --hidden 32 --layers 6 --levels 64 --batch 32 --steps 2 --warmup 1 --train-steps 3
# N = 512, batch 64 (BASELINE configs[4]): sampling only (the training workspace would need 480 GB), 2 levels of the chain
run 300 --nodes 512 --hidden 128 --layers 8 --levels 2 --batch 64 --steps 2 --warmup 1 --train-steps 0
