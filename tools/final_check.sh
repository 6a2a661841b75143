cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export OMP_NUM_THREADS=${OMP_NUM_THREADS:-32}
BENCH_COMM=shm BENCH_N=128 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 2 --warmup 1 2>&1 | grep -v "^###\|^$" | tail -6 | tee gpurun_out/dist2.log
python -c "
import __graft_entry__ as g
g.smoke()
print('smoke ok')
" 2>&1 | grep -v '^###' | tail -3 | tee gpurun_out/smoke.log
timeout 900 python bench.py 2> gpurun_out/bench.err | tee gpurun_out/bench.json | cut -c1-2500
tail -3 gpurun_out/bench.err
