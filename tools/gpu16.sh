cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export OMP_NUM_THREADS=${OMP_NUM_THREADS:-32}
for s in 0 16 6; do
  echo "=== FASP_HIP_SORT_STREAM=$s"
  FASP_HIP_SORT_STREAM=$s timeout 900 python tools/perf_levels.py 256 20 2>&1 | grep "^L[0-3] \|^solve" | cut -c1-330
done | tee gpurun_out/sort_stream.log
