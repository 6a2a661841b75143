cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export OMP_NUM_THREADS=${OMP_NUM_THREADS:-32}
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
timeout 900 python bench.py --steps 5 --warmup 2 2> gpurun_out/bench_err.log | tee gpurun_out/bench_r01.json
tail -8 gpurun_out/bench_err.log
