cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export OMP_NUM_THREADS=${OMP_NUM_THREADS:-32}
timeout 900 python -m pytest tests/test_plugin_krylov.py tests/test_amg_solver.py tests/test_golden_fixtures.py -x -q -m gpu 2>&1 | grep -v "^###\|^$" | tail -15 | tee gpurun_out/t_new.log
