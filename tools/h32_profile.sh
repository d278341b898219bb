#!/bin/bash
# 6x32 network at N = 200: bench line + rocprofv3 kernel-trace summary -> gpurun_out/h32_<tag>.json, gpurun_out/h32_<tag>_kernel_stats.csv
TAG=${1:-r03}
R=$PWD
timeout -k 10 200 python bench.py --no-cpu-baseline --no-exact-f32 --nodes 200 --hidden 32 --layers 6 --levels 64 --batch 32 --steps 2 --warmup 1 --train-steps 3 > gpurun_out/h32_$TAG.json 2> gpurun_out/h32_$TAG.err || exit 1
export TMPDIR=/tmp
cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/h32kt -o kt -- python3 $R/bench.py --no-cpu-baseline --no-exact-f32 --no-prof --nodes 200 --hidden 32 --layers 6 --levels 64 --batch 32 --steps 1 --warmup 0 --train-steps 0 > /dev/null 2>&1 || exit 1
cd $R
cp $(find gpurun_out/h32kt -name "*kernel_stats.csv" | head -n 1) gpurun_out/h32_${TAG}_kernel_stats.csv
rm -rf gpurun_out/h32kt
