#!/bin/bash
# Run on the GPU box (through gpurun) from the repo root: kernel-trace summary of the small-graph sampling bench
# (community_small geometry: N = 20, 6 x 32 PPGN, 16 levels, 32 graphs).   usage: bash tools/gpu_profile_small.sh <tag> [nodes levels]
set -u
TAG=${1:-r01}
NODES=${2:-20}
LEVELS=${3:-16}
REPO=$PWD
OUT=$REPO/gpurun_out/prof_small_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -o kt -- \
    python3 "$REPO/bench.py" --no-cpu-baseline --no-exact-f32 --no-prof --nodes "$NODES" --hidden 32 --layers 6 --levels "$LEVELS" --batch 32 \
    --steps 20 --warmup 3 --train-steps 0 > "$OUT/bench_under_rocprof.json" 2> "$OUT/kt.err" || { echo "kernel-trace pass failed"; tail -5 "$OUT/kt.err"; exit 1; }
cd "$REPO"
K=$(find "$OUT/kt" -name '*kernel_stats.csv' | head -n 1)
cp "$K" "$OUT/kernel_stats.csv"
rm -rf "$OUT/kt"
head -8 "$OUT/kernel_stats.csv"
