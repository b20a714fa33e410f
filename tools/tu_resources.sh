#!/bin/bash
# Registers / scratch / occupancy of every kernel of ONE generated translation unit, compiled with the product flags.
#   usage: tools/tu_resources.sh <tu, e.g. cmcd_8> [extra flags]      (object and remarks land in /tmp/tu_<tu>.*)
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TU=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -fPIC -Wno-comment -Wno-unused-command-line-argument -Xclang -target-feature -Xclang -packed-fp32-ops \
  -Rpass-analysis=kernel-resource-usage "$@" -c $ROOT/sde_sampler_lrds_amd/csrc/gen/$TU.hip -o /tmp/tu_$TU.o 2> /tmp/tu_$TU.err || { tail -20 /tmp/tu_$TU.err; exit 1; }
python3 - /tmp/tu_$TU.err <<'PY'
import re, sys
name = None
row = {}
for l in open(sys.argv[1]):
    m = re.search(r"remark: +(Function Name|VGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]): (\S+)", l)
    if not m:
        continue
    if m.group(1) == "Function Name":
        name = m.group(2)
        row[name] = {}
    elif name:
        row[name][m.group(1).split()[0]] = m.group(2)
for k, v in row.items():
    print(k, "VGPRs", v.get("VGPRs"), "scratch", v.get("ScratchSize"), "occupancy", v.get("Occupancy"))
PY
