cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export OMP_NUM_THREADS=${OMP_NUM_THREADS:-32}
timeout 900 python -m pytest tests/test_bsr_amg.py -x -q -m gpu -s 2>&1 | grep -v "^###\|^$" | tail -15 | tee gpurun_out/t_bsr.log
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 | tee gpurun_out/t_all.log
