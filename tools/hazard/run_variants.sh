#!/bin/bash
# rerun-determinism probe (cfg 2, 32 768 x 64) for each tools/var_<name>.so.bin given; one line per variant
for n in "$@"; do
  out=$(SDENG_LIB=$PWD/tools/var_$n.so.bin PROBE_ONLY=1 timeout -k 10 120 python tools/probe_determinism.py 2>/dev/null | grep -o "x rows differing [0-9]*" | awk '{printf " %s", $4}')
  echo "$n: rows differing in 3 reruns:$out"
done
