#!/bin/bash
# tools/ab_full.sh variant...: headline bench (sampling + exact-f32 + training), N = 20 and 6x32 N = 200 per library variant (scratch/ab/libdppm_<v>.so; keep = in-tree)
lib=discrete_dppm_graphs_amd/libdppm_hip.so
cp $lib /tmp/libdppm_keep.so
for v in "$@"; do
  if [ $v != keep ]; then cp scratch/ab/libdppm_$v.so $lib; else cp /tmp/libdppm_keep.so $lib; fi
  timeout -k 10 500 python bench.py --no-cpu-baseline --steps 2 --warmup 1 --train-steps 10 > gpurun_out/abf_$v.json 2> gpurun_out/abf_$v.err || { tail -3 gpurun_out/abf_$v.err; cp /tmp/libdppm_keep.so $lib; exit 1; }
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-exact-f32 --no-prof --nodes 20 --hidden 32 --layers 6 --levels 16 --batch 32 --steps 20 --warmup 3 --train-steps 20 > gpurun_out/abf_${v}_n20.json 2>> gpurun_out/abf_$v.err || { cp /tmp/libdppm_keep.so $lib; exit 1; }
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-exact-f32 --nodes 200 --hidden 32 --layers 6 --levels 64 --batch 32 --steps 2 --warmup 1 --train-steps 3 > gpurun_out/abf_${v}_h32.json 2>> gpurun_out/abf_$v.err || { cp /tmp/libdppm_keep.so $lib; exit 1; }
  python - <<PY
import json
d=json.load(open('gpurun_out/abf_$v.json')); a=json.load(open('gpurun_out/abf_${v}_n20.json')); b=json.load(open('gpurun_out/abf_${v}_h32.json'))
t=lambda x:(x.get('train') or {}).get('steps_per_s')
print('$v', 'headline', round(d['value'],2), 'exact_f32', round(d['exact_f32']['value'],2), 'train', t(d), '| N=20', round(a['value'],1), 'train', t(a), '| 6x32 N=200', round(b['value'],1), 'train', t(b))
PY
done
cp /tmp/libdppm_keep.so $lib
