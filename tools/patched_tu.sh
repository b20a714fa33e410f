#!/bin/bash
# Experiment harness: compile ONE generated translation unit through its assembly so that a sed script can rewrite the device ISA
# (e.g. the encoding of an instruction) before it is assembled, and link a variant of libsdeng.so with it.
#   usage: tools/patched_tu.sh <name> <tu, e.g. sim_8_2_0> '<sed script>' [extra hipcc flags...]
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CS=$ROOT/sde_sampler_lrds_amd/csrc
LLVM=/opt/rocm/lib/llvm/bin
NAME=$1; TU=$2; SED=$3; shift 3
W=/tmp/ptu_$NAME; mkdir -p $W
FL="--offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -fPIC -Wno-comment -Wno-unused-command-line-argument -Xclang -target-feature -Xclang -packed-fp32-ops $@"
/opt/rocm/bin/hipcc $FL --cuda-device-only -S $CS/gen/$TU.hip -o $W/dev.s 2>/dev/null
sed -E "$SED" $W/dev.s > $W/dev_p.s
echo "lines changed: $(diff $W/dev.s $W/dev_p.s | grep -c '^>')"
$LLVM/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c $W/dev_p.s -o $W/dev.o
$LLVM/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared -o $W/dev.out $W/dev.o
$LLVM/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 -input=/dev/null -input=$W/dev.out -output=$W/dev.hipfb
/opt/rocm/bin/hipcc $FL --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang $W/dev.hipfb -c $CS/gen/$TU.hip -o $W/tu.o
OBJS=$(ls $CS/obj/*.o | grep -v "/$TU.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/tools/var_$NAME.so.bin $OBJS $W/tu.o
echo built tools/var_$NAME.so.bin
