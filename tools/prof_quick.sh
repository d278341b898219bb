#!/bin/bash
# rocprofv3 kernel stats of one sampling chain: tools/prof_quick.sh <tag> [bench flags]
tag=$1; shift
GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$tag
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o p -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --train-steps 0 --no-exact-f32 --no-prof --steps 1 --warmup 1 "$@" > /tmp/prof_$tag.out 2>&1 || { tail -5 /tmp/prof_$tag.out; exit 1; }
f=$(find /tmp/prof_$tag -name "*kernel_stats.csv" | head -1)
cp "$f" $GRAFT_REPO_ROOT/gpurun_out/prof_${tag}_kernel_stats.csv
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:16]:
    print("%-60s calls %6s avg %9.1f us  %5.1f%%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
print("total ms", tot / 1e6)
PY
