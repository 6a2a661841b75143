cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export OMP_NUM_THREADS=${OMP_NUM_THREADS:-32}
timeout 900 python tools/perf_vec.py 2>&1 | grep -v "^###" | tee gpurun_out/perf_vec.log
timeout 1800 python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|Error" | tail -3 | tee gpurun_out/t_all.log
timeout 900 python tools/perf_batch.py 256 2>&1 | grep -v "^###" | tail -2 | tee gpurun_out/perf_batch.log
