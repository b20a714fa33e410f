#!/bin/bash
# Build a variant of libsdeng.so in which ONE generated translation unit goes through an assembly-level patch
# (tools/hazard/patch_asm.py) between hipcc's code generation and the assembler: the way the packed-fp32 corruption of
# DESIGN 4a was narrowed down to an instruction pair.  The other objects are reused from csrc/obj.
#   usage: tools/hazard/asm_variant.sh <name> <tu, e.g. sim_8_2_0> "<patch spec>" [extra hipcc flags...]
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
CS=$ROOT/sde_sampler_lrds_amd/csrc
LLVM=/opt/rocm/lib/llvm/bin
NAME=$1; TU=$2; SPEC=$3; shift 3
W=/tmp/asmvar_$NAME; mkdir -p $W
FLAGS="--offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -fPIC -Wno-comment -Wno-unused-command-line-argument $*"
/opt/rocm/bin/hipcc $FLAGS --cuda-device-only -S $CS/gen/$TU.hip -o $W/dev.s
python3 $ROOT/tools/hazard/patch_asm.py $W/dev.s $W/dev_p.s "$SPEC"
$LLVM/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c $W/dev_p.s -o $W/dev.o
$LLVM/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared -o $W/dev.out $W/dev.o
$LLVM/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 \
  -input=/dev/null -input=$W/dev.out -output=$W/dev.hipfb
/opt/rocm/bin/hipcc $FLAGS --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang $W/dev.hipfb -c $CS/gen/$TU.hip -o $W/host.o
OBJS=$(ls $CS/obj/*.o | grep -v "/$TU.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/tools/var_$NAME.so.bin $OBJS $W/host.o
echo built tools/var_$NAME.so.bin
