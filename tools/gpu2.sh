cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export OMP_NUM_THREADS=${OMP_NUM_THREADS:-32}
timeout 300 python tools/perf_levels.py 128 20 2>&1 | tee gpurun_out/perf128.log
timeout 900 python tools/perf_levels.py 256 10 2>&1 | tee gpurun_out/perf256.log
