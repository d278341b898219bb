#!/bin/bash
# Run on the GPU box (through gpurun) from the repo root: kernel-trace summary of the bench command + two PMC passes.
# usage: bash tools/gpu_profile.sh <tag>
set -u
TAG=${1:-r01}
REPO=$PWD
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
timeout -k 10 420 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -o kt -- \
    python3 "$REPO/bench.py" --steps 1 --warmup 1 --train-steps 2 --no-cpu-baseline > "$OUT/bench_under_rocprof.json" 2> "$OUT/kt.err" || { echo "kernel-trace pass failed"; tail -5 "$OUT/kt.err"; exit 1; }
echo "kernel-trace pass done"
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -o fetch -- \
    python3 "$REPO/bench.py" --steps 1 --warmup 0 --levels 2 --train-steps 0 --no-cpu-baseline --no-exact-f32 > "$OUT/fetch.out" 2> "$OUT/fetch.err" || { echo "fetch pass failed"; tail -5 "$OUT/fetch.err"; exit 1; }
echo "fetch pass done"
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -o write -- \
    python3 "$REPO/bench.py" --steps 1 --warmup 0 --levels 2 --train-steps 0 --no-cpu-baseline --no-exact-f32 > "$OUT/write.out" 2> "$OUT/write.err" || { echo "write pass failed"; tail -5 "$OUT/write.err"; exit 1; }
echo "write pass done"
cd "$REPO"
F=$(find "$OUT/fetch" -name '*counter_collection.csv' | head -n 1)
W=$(find "$OUT/write" -name '*counter_collection.csv' | head -n 1)
K=$(find "$OUT/kt" -name '*kernel_stats.csv' | head -n 1)
cp "$K" "$OUT/kernel_stats.csv"
python3 tools/pmc_aggregate.py "$F" "$W" "$OUT/pmc_traffic.json" 200 128 8 32
# matrix-pipe and LDS activity per kernel (both precisions: the exact-f32 leg runs too), its own pass
bash tools/gpu_pmc.sh ${TAG}_sq "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" \
    bench.py --steps 1 --warmup 0 --levels 2 --train-steps 0 --no-cpu-baseline > /dev/null && cp gpurun_out/pmc_${TAG}_sq/summary.txt "$OUT/pmc_mfma_lds.txt" && echo "sq pass done"
# the raw per-dispatch CSVs are large; keep only the summaries
rm -rf "$OUT/fetch" "$OUT/write" "$OUT/kt"
