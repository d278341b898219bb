#!/bin/bash
# quick sampling-only bench line: tools/bench_quick.sh <tag> [extra bench.py flags]
tag=$1; shift
timeout -k 10 600 python bench.py --no-cpu-baseline --train-steps 0 --no-exact-f32 "$@" > gpurun_out/bq_$tag.json 2> gpurun_out/bq_$tag.err || { tail -5 gpurun_out/bq_$tag.err; exit 1; }
python - <<PY
import json
d=json.load(open('gpurun_out/bq_$tag.json'))
print('$tag', 'graphs/s', round(d['value'],2), 'ms/level', round(d['ms_per_step']/d['config']['noise_levels'],3))
for k,v in d['kernels'].items(): print('  %-13s n=%5d avg %.4f ms share %.3f %s frac %.3f' % (k, v['launches'], v['avg_launch_ms'], v['share_of_step_time'], v['bound'], v['frac']))
PY
