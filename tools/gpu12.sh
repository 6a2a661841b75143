cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export OMP_NUM_THREADS=${OMP_NUM_THREADS:-32}
timeout 600 python -m pytest tests/test_bsr_amg.py tests/test_sa_amg.py tests/test_golden_fixtures.py -x -q -m gpu 2>&1 | grep -v "^###\|^$" | tail -15 | tee gpurun_out/t_small.log
timeout 600 python tools/perf_configs.py 3 2>&1 | grep -v "^###" | tee gpurun_out/config3.log
C5_SMOOTHER=jacobi timeout 900 python tools/perf_configs.py 5 2>&1 | grep -v "^###" | tee gpurun_out/config5_jacobi.log
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | grep -v "^###\|^$" | tail -8 | tee gpurun_out/t_all.log
