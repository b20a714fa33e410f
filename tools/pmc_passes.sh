#!/bin/bash
# rocprofv3 counter passes over the step-loop kernel of one workload (one --pmc set per run; no tracing flags), then a kernel-trace
# pass for the per-kernel average duration.   usage: tools/pmc_passes.sh <cfg: rds_gmm|pis_phi4|cmcd_logreg> <tag>
# Writes gpurun_out/pmc_<tag>/{summary.txt, counters.json, kernel_stats.csv}; copy what you want judged into profiles/.
set -u
CFG=${1:-rds_gmm}
TAG=${2:-r02_$CFG}
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PROBE_REPS=2 PROBE_CFG=$CFG
i=0
for SET in \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" \
  "SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INST_CYCLES_SALU SQ_IFETCH SQ_BUSY_CU_CYCLES" \
  "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS" \
  "SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT64" \
  "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_LEVEL_VMEM SQ_WAVES SQ_INSTS_SMEM SQ_INSTS_BRANCH" \
  "FETCH_SIZE GRBM_GUI_ACTIVE" \
  "WRITE_SIZE GRBM_GUI_ACTIVE" ; do
  i=$((i+1))
  timeout -k 10 180 rocprofv3 --pmc $SET --output-format csv -d $OUT/p$i -- python $GRAFT_REPO_ROOT/tools/bench_kernel_only.py > $OUT/p$i.log 2>&1
  echo "pass $i ($SET) rc=$?" >> $OUT/summary.txt
done
PROBE_REPS=6 timeout -k 10 180 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python $GRAFT_REPO_ROOT/tools/bench_kernel_only.py > $OUT/kt.log 2>&1
echo "kernel-trace rc=$?" >> $OUT/summary.txt
cp $(ls $OUT/kt/*/*kernel_stats.csv 2>/dev/null | head -1) $OUT/kernel_stats.csv 2>/dev/null
python3 - <<PY
import csv, glob, collections, json, sys
out, cfg = "$OUT", "$CFG"
sys.path.insert(0, "$GRAFT_REPO_ROOT")
from sde_sampler_lrds_amd import build as _build
acc = collections.defaultdict(list)
kname = None
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_simulate" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
            kname = r["Kernel_Name"]
res = {k: sum(v) / len(v) for k, v in acc.items()}
with open(out + "/summary.txt", "a") as fo:
    for k in sorted(res):
        line = f"{k:32s} mean/dispatch {res[k]:.6g}  (n={len(acc[k])})"
        print(line); fo.write(line + "\n")
# HBM traffic per launch (MI355X_MICROARCH.md, HBM section): FETCH_SIZE / WRITE_SIZE in units of 1024 B; on gfx950 FETCH_SIZE
# tallies the 128-B requests of 16-B-per-lane loads (all these kernels issue) at 64 B: doubled.  Separate --pmc passes.
traffic = None
if "FETCH_SIZE" in res and "WRITE_SIZE" in res:
    traffic = {"fetch_bytes": 2.0 * 1024.0 * res["FETCH_SIZE"], "write_bytes": 1024.0 * res["WRITE_SIZE"]}
    traffic["bytes"] = traffic["fetch_bytes"] + traffic["write_bytes"]
    traffic["correction"] = "FETCH_SIZE x2 (gfx950, 16-B-per-lane loads), WRITE_SIZE as read; separate --pmc passes"
kt = None
try:
    for r in csv.DictReader(open(out + "/kernel_stats.csv")):
        if "k_simulate" in r["Name"]:
            kt = {"name": r["Name"], "calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]), "min_ns": float(r["MinNs"]), "max_ns": float(r["MaxNs"])}
except Exception as e:
    print("no kernel stats:", e)
json.dump({"workload": cfg, "kernel": kname, "library_digest": _build._digest(), "counters_per_launch": res, "traffic": traffic, "kernel_trace": kt}, open(out + "/counters.json", "w"), indent=1)
PY
