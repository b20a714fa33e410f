#!/bin/bash
# Build a variant of libsdeng.so in which ONE generated translation unit is recompiled with extra flags
# (the other objects are reused from csrc/obj).   usage: tools/variant_lib.sh <name> <tu, e.g. sim_8_2_0> <flags...>
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CS=$ROOT/sde_sampler_lrds_amd/csrc
NAME=$1; TU=$2; shift 2
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -fPIC -Wno-comment -Wno-unused-command-line-argument -Xclang -target-feature -Xclang -packed-fp32-ops "$@" -c $CS/gen/$TU.hip -o /tmp/var_$NAME.o
OBJS=$(ls $CS/obj/*.o | grep -v "/$TU.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/tools/var_$NAME.so.bin $OBJS /tmp/var_$NAME.o
echo built tools/var_$NAME.so.bin
