#!/bin/bash
# In-kernel x0 draw (X0 = 1 twin) against a materialised x0 (k_sample_x0 + X0 = 0 kernel), same box, alternating.
# usage: tools/ab_draw.sh <cfg>   (on the GPU box)
set -u
R=$GRAFT_REPO_ROOT; CFG=${1:-pis_phi4}
OUT=$R/gpurun_out/ab_draw_$CFG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PROBE_REPS=8 PROBE_CFG=$CFG
for i in 1 2; do
  for dr in 1 0; do
    PROBE_DRAW=$dr timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/draw${dr}_$i -- python3 $R/tools/bench_kernel_only.py > $OUT/draw${dr}_$i.log 2>&1 || exit 1
  done
done
for d in draw1_1 draw0_1 draw1_2 draw0_2; do echo "$d: $(grep -h k_simulate $OUT/$d/*/*kernel_stats.csv | awk -F'","' '{printf "%s calls %s avg %.3f ms min %.3f\n", $1, $2, $4/1e6, $6/1e6}')"; done | tee $OUT/summary.txt
