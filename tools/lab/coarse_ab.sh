#!/bin/bash
# Development: configs 5 and 3 of BASELINE.json with the product library and with build variants, one box (tools/build_variant.sh).
# usage: bash tools/lab/coarse_ab.sh "5 3" base srt ...
cfgs=${1:-5}; shift
for c in $cfgs; do
  echo "== config $c, product"; timeout 600 python3 tools/perf_configs.py $c 2>&1 | grep -v "^\[spcg_reg\]\|^\[gmres" | tail -4
  for v in "$@"; do
    echo "== config $c, variant $v"
    FASP_HIP_LIB=lab_build/libfasp_hip_$v.so timeout 600 python3 tools/perf_configs.py $c 2>&1 | awk '/^\[spcg_reg\]|^\[gmres/ {n++; if (n % 97 == 1) print; next} {print}' | tail -14
  done
done
