#!/bin/bash
# sampling quick lines with the role-split MLP (default) and the two-group MLP (DPPM_W3MLP=1) at N = 60 / 200 / 512
for cfg in "n60 --nodes 60 --batch 64 --levels 16 --steps 2" "n200 --levels 8 --steps 1" "n512 --nodes 512 --batch 64 --levels 2 --steps 1"; do
  set -- $cfg; tag=$1; shift
  tools/bench_quick.sh ${tag}_w4 "$@" | grep -E "graphs|mlp2" || exit 1
  DPPM_W3MLP=1 tools/bench_quick.sh ${tag}_w3 "$@" | grep -E "graphs|mlp2" || exit 1
done
