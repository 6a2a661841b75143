# usage (GPU box): bash tools/pmc/run_levels_traffic.sh <n> <ops> <reps>   -- HBM-side traffic per level: FETCH_SIZE and WRITE_SIZE,
# each in its OWN rocprofv3 pass (nothing but --kernel-trace beside --pmc), summarised per run of one kernel (tools/pmc/summary_runs.py)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp OMP_NUM_THREADS=32
N=${1:-256}; OPS=${2:-0}; REPS=${3:-3}
OUT=gpurun_out/pmc_levels_traffic; rm -rf $OUT; mkdir -p $OUT
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/$c -- python3 tools/pmc/levels_workload.py $N $OPS $REPS > $OUT/$c.log 2>&1
  python3 tools/pmc/summary_runs.py $OUT/$c > $OUT/summary_$c.txt 2>&1
done
grep "^level" $OUT/FETCH_SIZE.log > $OUT/times.txt
find $OUT -name "*.csv" -delete; find $OUT -name "*.db" -delete
