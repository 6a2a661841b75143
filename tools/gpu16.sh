cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export OMP_NUM_THREADS=${OMP_NUM_THREADS:-32}
timeout 1800 python -m pytest tests -x -q -m gpu 2>&1 | grep -v "^###\|^$" | tail -30 | tee gpurun_out/t_all.log
