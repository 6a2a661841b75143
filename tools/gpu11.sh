cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export OMP_NUM_THREADS=${OMP_NUM_THREADS:-32}
C5_SMOOTHER=jacobi timeout 900 python tools/perf_configs.py 5 2>&1 | grep -v "^###" | tee gpurun_out/config5_jacobi.log
C5_SMOOTHER=default timeout 1500 python tools/perf_configs.py 5 123 32 2>&1 | grep -v "^###" | tee gpurun_out/config5_default.log
