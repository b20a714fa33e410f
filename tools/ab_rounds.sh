#!/bin/bash
# A/B of the cfg-2 step-loop kernel on ONE box: the round-1 library (mkdir .ab_r01 && git archive 80d3ce0 | tar -x -C .ab_r01, then
# python -m sde_sampler_lrds_amd.build inside it) against the current one,
# alternating, kernel durations from rocprofv3 --kernel-trace.   usage: tools/ab_rounds.sh  (on the GPU box)
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/ab_rounds
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PROBE_REPS=12
for i in 1 2; do
  timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r01_$i -- python3 $R/.ab_r01/tools/bench_kernel_only.py > $OUT/r01_$i.log 2>&1 || exit 1
  PROBE_DRAW=1 timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r02draw_$i -- python3 $R/tools/bench_kernel_only.py > $OUT/r02draw_$i.log 2>&1 || exit 1
  PROBE_DRAW=0 timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r02x0_$i -- python3 $R/tools/bench_kernel_only.py > $OUT/r02x0_$i.log 2>&1 || exit 1
done
for d in r01_1 r02draw_1 r02x0_1 r01_2 r02draw_2 r02x0_2; do echo "$d: $(grep -h k_simulate $OUT/$d/*/*kernel_stats.csv | cut -d, -f1-4,6,7)"; done | tee $OUT/summary.txt
