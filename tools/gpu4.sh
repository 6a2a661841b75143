cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export OMP_NUM_THREADS=${OMP_NUM_THREADS:-32}
timeout 1400 python tools/sweep_spmv.py 256 0,1,2,3,4,6,9 0,2,5,6,7 2>&1 | tee gpurun_out/sweep256b.log | tail -150
