#!/bin/bash
# usage (on the GPU box): bash tools/gpu_pmc.sh <tag> "<counters>" <script.py> [args]   -> gpurun_out/pmc_<tag>.csv (per-kernel averages)
set -u
TAG=$1; CTRS=$2; shift 2
REPO=$PWD
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
timeout -s KILL 120 rocprofv3 --pmc $CTRS --output-format csv -d "$OUT/raw" -o pmc -- python3 "$REPO/$1" "${@:2}" > "$OUT/stdout.txt" 2> "$OUT/stderr.txt" || { echo "pmc pass failed"; tail -5 "$OUT/stderr.txt"; exit 1; }
cd "$REPO"
F=$(find "$OUT/raw" -name '*counter_collection.csv' | head -n 1)
python3 - "$F" > "$OUT/summary.txt" <<'PY'
import csv, sys
from collections import defaultdict
tot = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(lambda: defaultdict(int))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"][:70]
    tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
    if "Start_Timestamp" in r and "End_Timestamp" in r:
        tot[k]["~duration_ns"] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"]); cnt[k]["~duration_ns"] += 1
for k in tot:
    print(k)
    for c in sorted(tot[k]):
        print(f"   {c:32s} {tot[k][c] / cnt[k][c]:16.1f}  (n={cnt[k][c]})")
PY
rm -rf "$OUT/raw"
cat "$OUT/summary.txt"
