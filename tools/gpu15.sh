cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export OMP_NUM_THREADS=${OMP_NUM_THREADS:-32}
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_compress.py tests/test_gpu_dist.py -x -q -m gpu 2>&1 | grep -v "^###\|^$" | tail -5 | tee gpurun_out/t_par.log
timeout 900 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2> gpurun_out/bench.err | tee gpurun_out/bench.json | cut -c1-400
tail -2 gpurun_out/bench.err
