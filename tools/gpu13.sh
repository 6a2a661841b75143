cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export OMP_NUM_THREADS=${OMP_NUM_THREADS:-32}
timeout 900 python tools/perf_rowpat_dbg.py 256 2>&1 | grep -v "^###" | tee gpurun_out/rowpat_dbg.log
