#!/bin/bash
# tools/ab_variant_build.sh <tag> <file.hip> "<extra flags>": library variant with one object rebuilt with extra flags -> scratch/ab/libdppm_<tag>.so
set -e
tag=$1; src=$2; extra=$3
cs=/root/repo/discrete_dppm_graphs_amd/csrc; cd $cs
obj=${src%.hip}.o
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -w -fno-slp-vectorize -I../../include -I. $extra -c $src -o /tmp/abv_$tag.o
mkdir -p ../../scratch/ab
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../../scratch/ab/libdppm_$tag.so $(for o in dppm_common.o dppm_diffusion.o dppm_ppgn.o dppm_head.o dppm_w4.o dppm_w6.o dppm_plane4.o dppm_smallgraph.o dppm_optim.o dppm_stats.o; do if [ $o = $obj ]; then echo /tmp/abv_$tag.o; else echo $o; fi; done)
ls -la ../../scratch/ab/libdppm_$tag.so
