#!/bin/bash
# Instruction-cache counters of the step-loop kernel of one workload, with the in-kernel x0 draw and with a materialised x0.
# usage: tools/pmc_icache.sh <cfg>   (on the GPU box)
set -u
R=$GRAFT_REPO_ROOT; CFG=${1:-rds_gmm}
OUT=$R/gpurun_out/icache_$CFG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PROBE_REPS=2 PROBE_CFG=$CFG
for dr in 1 0; do
  PROBE_DRAW=$dr timeout -k 10 180 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $OUT/draw$dr -- python3 $R/tools/bench_kernel_only.py > $OUT/draw$dr.log 2>&1 || exit 1
  PROBE_DRAW=$dr timeout -k 10 180 rocprofv3 --pmc SQ_IFETCH_LEVEL SQC_ICACHE_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/drawb$dr -- python3 $R/tools/bench_kernel_only.py > $OUT/drawb$dr.log 2>&1 || exit 1
done
python3 - <<PY | tee $OUT/summary.txt
import csv, glob, collections
for dr in ("1", "0"):
    acc = collections.defaultdict(list)
    for f in glob.glob("$OUT/draw%s/**/*counter_collection.csv" % dr, recursive=True) + glob.glob("$OUT/drawb%s/**/*counter_collection.csv" % dr, recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_simulate" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("draw =", dr, {k: "%.4g" % (sum(v) / len(v)) for k, v in sorted(acc.items())})
PY
