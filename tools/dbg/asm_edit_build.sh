#!/bin/bash
# Build a library in which one source file's kernels come from EDITED device assembly.
# usage: [OBJ=dppm_head.o] [EXTRAF=-D..] asm_edit_build.sh <src.hip> <python filter: stdin asm -> stdout asm> <out tag>
# steps: device asm (hipcc -S) -> filter -> assemble -> lld -> bundle -> host object with that bundle -> scratch/ab/libdppm_<tag>.so
set -e
src=$1; filter=$2; tag=$3
cs=/root/repo/discrete_dppm_graphs_amd/csrc; L=/opt/rocm/lib/llvm/bin; W=/tmp/asmedit_$tag; mkdir -p $W
cp $src $cs/_asmedit_tmp.hip
trap "rm -f $cs/_asmedit_tmp.hip" EXIT
cd $cs
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -w -I../../include -I. $EXTRAF"
OBJ=${OBJ:-dppm_head.o}
/opt/rocm/bin/hipcc $F -S --cuda-device-only _asmedit_tmp.hip -o $W/dev.s
python3 $filter < $W/dev.s > $W/edited.s
$L/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c $W/edited.s -o $W/dev.o
$L/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared -o $W/dev.out $W/dev.o
$L/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 -input=/dev/null -input=$W/dev.out -output=$W/dev.hipfb
/opt/rocm/bin/hipcc $F --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang $W/dev.hipfb -c _asmedit_tmp.hip -o $W/host.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o /root/repo/scratch/ab/libdppm_$tag.so $(for o in dppm_common.o dppm_diffusion.o dppm_ppgn.o dppm_head.o dppm_w4.o dppm_w6.o dppm_plane4.o dppm_smallgraph.o dppm_optim.o dppm_stats.o; do if [ $o = $OBJ ]; then echo $W/host.o; else echo $o; fi; done)
ls -la /root/repo/scratch/ab/libdppm_$tag.so
