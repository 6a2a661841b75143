#!/bin/bash
# Development: a second libfasp_hip.so with extra -D flags beside the product build, for in-one-box A/B runs
# (FASP_HIP_LIB=lab_build/libfasp_hip_<tag>.so python tools/...).  usage: tools/build_variant.sh <tag> [-DFLAG ...]
set -e
cd "$(dirname "$0")/.."
tag=$1; shift
mkdir -p lab_build
make -s -C faspsolver_amd/csrc
/opt/rocm/bin/hipcc "$@" --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -fopenmp=libgomp -Wno-unused-function -Wno-unused-result \
    -c faspsolver_amd/csrc/solver.hip -o lab_build/solver_$tag.o
g++ "$@" -O3 -fPIC -std=c++17 -ffp-contract=off -fopenmp -Wall -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ -c faspsolver_amd/csrc/seq_sched.cpp -o lab_build/seq_sched_$tag.o   # (the sweep schedules share constants with the kernels)
cd faspsolver_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-Bsymbolic -o ../../lab_build/libfasp_hip_$tag.so host_setup.o dist_plan.o comm.o param_input.o ../../lab_build/seq_sched_$tag.o reorder.o comm_ipc.o ../../lab_build/solver_$tag.o \
    -L/opt/rocm/lib -lamdhip64 -lgomp -ldl -Wl,-rpath,/opt/rocm/lib
echo built lab_build/libfasp_hip_$tag.so
