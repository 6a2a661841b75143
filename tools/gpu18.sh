cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export OMP_NUM_THREADS=${OMP_NUM_THREADS:-64}
export BENCH_CPU_BUDGET_S=120
timeout 1500 python tools/check512.py 512 2>&1 | grep -v "^###\|^$" | tee gpurun_out/check512.log
