set -u
REPO=$PWD
OUT=$REPO/gpurun_out/prof_train_small
mkdir -p "$OUT"; export TMPDIR=/tmp; cd /tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -o kt -- python3 "$REPO/bench.py" --no-cpu-baseline --no-exact-f32 --no-prof --nodes 20 --hidden 32 --layers 6 --levels 16 --batch 32 --steps 1 --warmup 1 --train-steps 50 > "$OUT/line.json" 2> "$OUT/err.txt" || { tail -5 "$OUT/err.txt"; exit 1; }
cd "$REPO"
K=$(find "$OUT/kt" -name '*kernel_stats.csv' | head -n 1); cp "$K" "$OUT/kernel_stats.csv"; rm -rf "$OUT/kt"
python3 - "$OUT/kernel_stats.csv" "$OUT/line.json" <<'PY'
import csv, json, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows); calls = sum(int(r["Calls"]) for r in rows)
print("total kernel ms", tot / 1e6, "calls", calls)
for r in rows[:14]:
    print(r["Name"][:90], r["Calls"], round(float(r["TotalDurationNs"]) / 1e6, 2), "ms", round(float(r["AverageNs"]) / 1e3, 1), "us")
d = json.load(open(sys.argv[2])); print(d["train"]["ms_per_train_step"], "ms/step")
PY
