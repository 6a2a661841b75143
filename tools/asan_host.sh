#!/bin/bash
# AddressSanitizer + UBSan pass over the HOST side of libfasp_hip.so (setup, partition plans, readers;
# CPU build only -- GPU ASan is not available on the pool).  The HIP object is linked as built.
# usage: bash tools/asan_host.sh      (runs the product/oracle CPU tests that exercise the host code)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=${TMPDIR:-/tmp}/fasp_asan
mkdir -p $OUT
cd $ROOT/faspsolver_amd/csrc
make -s
for f in host_setup dist_plan comm param_input seq_sched reorder; do
  g++ -O1 -g -fPIC -std=c++17 -ffp-contract=off -fopenmp -fsanitize=address,undefined -fno-omit-frame-pointer \
      -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ -c $f.cpp -o $OUT/$f.o
done
GCCLIB=$(dirname $(gcc -print-file-name=libasan.so))
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-Bsymbolic -o $OUT/libfasp_hip.so $OUT/host_setup.o $OUT/dist_plan.o \
    $OUT/comm.o $OUT/param_input.o $OUT/seq_sched.o $OUT/reorder.o comm_ipc.o solver.o -L/opt/rocm/lib -lamdhip64 -lgomp -ldl -Wl,-rpath,/opt/rocm/lib -L$GCCLIB -lasan -lubsan
cat > $OUT/run.py <<PY
import sys
sys.path.insert(0, "$ROOT")
import faspsolver_amd as fa
fa.LIB_PATH = "$OUT/libfasp_hip.so"
import pytest
# tests that call the compiled REFERENCE are left out: it is not built with the sanitizer runtime
sys.exit(pytest.main(["tests/test_host_setup_parallel.py", "tests/test_ua_amg.py", "tests/test_bsr_amg.py", "tests/test_sa_amg.py",
                      "tests/test_param_input.py", "tests/test_golden_fixtures.py", "tests/test_matrix_coding.py", "tests/test_dist_cpu.py",
                      "tests/test_seq_schedule.py", "tests/test_host_reorder.py",
                      "-q", "-m", "not gpu", "-k", "not reference and not two_rank", "-p", "no:cacheprovider"]))
PY
cd $ROOT
LD_PRELOAD="$GCCLIB/libasan.so $GCCLIB/libubsan.so" ASAN_OPTIONS=detect_leaks=0:verify_asan_link_order=0 OMP_NUM_THREADS=4 \
    python $OUT/run.py 2>&1 | grep -v "^###\|reading file\|writing to" | grep -E "ERROR: AddressSanitizer|runtime error|SUMMARY|passed|failed" || true
