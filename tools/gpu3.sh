cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export OMP_NUM_THREADS=${OMP_NUM_THREADS:-32}
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -15
