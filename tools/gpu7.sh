cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export OMP_NUM_THREADS=64
timeout 1500 python tools/check512.py 512 2>&1 | grep -v "^ *[0-9]* |" | tee gpurun_out/check512.log
