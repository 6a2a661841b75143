cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export OMP_NUM_THREADS=${OMP_NUM_THREADS:-32}
timeout 900 python -m pytest tests/test_ua_amg.py tests/test_edge_cases.py -q -m gpu 2>&1 | grep -v "^###\|^$" | tail -8 | tee gpurun_out/t_ua.log
