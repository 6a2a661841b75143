cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export OMP_NUM_THREADS=${OMP_NUM_THREADS:-32}
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -5
timeout 900 python tools/sweep_spmv.py 256 0,1,2,6,8,9 0,2,5 2>&1 | tee gpurun_out/sweep256d.log
SWEEP_NT=0 timeout 900 python tools/sweep_spmv.py 256 0,1 0,2 2>&1 | tee gpurun_out/sweep256d_nt0.log
timeout 900 python tools/perf_levels.py 256 10 2>&1 | grep -E "^solve|setup" 
