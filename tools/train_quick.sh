#!/bin/bash
# quick training-only bench line: tools/train_quick.sh <tag> [extra bench.py flags]
tag=$1; shift
timeout -k 10 600 python bench.py --no-cpu-baseline --no-exact-f32 --steps 1 --levels 1 --warmup 0 --train-steps 4 "$@" > gpurun_out/tq_$tag.json 2> gpurun_out/tq_$tag.err || { tail -5 gpurun_out/tq_$tag.err; exit 1; }
python - <<PY
import json
d=json.load(open('gpurun_out/tq_$tag.json'))
t=d['train']
print('$tag', 'steps/s', round(t['train_steps_per_s'],3), 'ms/step', round(t['ms_per_train_step'],2), 'loss', t['loss'])
for k,v in t.get('kernels',{}).items(): print('  %-13s n=%5d avg %.4f ms share %.3f %s frac %.3f' % (k, v['launches'], v['avg_launch_ms'], v['share_of_step_time'], v['bound'], v['frac']))
PY
