#!/bin/bash
# Development: whole P7(n) solves with the product library and with build variants on ONE box (tools/build_variant.sh):
#   bash tools/lab/lib_ab.sh 256 "" head p12      ("var" as second argument: the variable-coefficient operator)
n=$1; var=$2; shift; shift
for rep in 1 2; do
  echo "product: $(timeout 600 python3 tools/lab/solve_ab.py $n estream 1 $var 2 2>&1 | grep -v "^\[" | tail -1)"
  for v in "$@"; do
    echo "$v: $(FASP_HIP_LIB=lab_build/libfasp_hip_$v.so timeout 600 python3 tools/lab/solve_ab.py $n estream 1 $var 2 2>&1 | grep -v "^\[" | tail -1)"
  done
done
