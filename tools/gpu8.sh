cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export OMP_NUM_THREADS=${OMP_NUM_THREADS:-32}
timeout 600 python tools/perf_bsr.py 128 2>&1 | tee gpurun_out/perf_bsr.log
timeout 900 python tools/perf_gs.py 128 2>&1 | tee gpurun_out/perf_gs128.log
