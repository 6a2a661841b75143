cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export OMP_NUM_THREADS=${OMP_NUM_THREADS:-16}
timeout 900 python -m pytest tests/test_gpu_dist.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -15
BENCH_COMM=shm BENCH_N=128 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 2 --warmup 1 2>&1 | tail -8
