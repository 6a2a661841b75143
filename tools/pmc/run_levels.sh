# (a pass that mixed FETCH_SIZE / WRITE_SIZE with SQ counters aborted rocprofv3 and then ignored SIGTERM: SQ groups only, timeout -s KILL)
# usage (GPU box): bash tools/pmc/run_levels.sh <n> <ops> <reps>   -- SQ counter groups, per level (separate passes, no TA/TCP groups)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp OMP_NUM_THREADS=32
N=${1:-256}; OPS=${2:-0}; REPS=${3:-3}
OUT=gpurun_out/pmc_levels; rm -rf $OUT; mkdir -p $OUT
CTRS=(
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
 "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INST_LEVEL_VMEM SQ_WAVES GRBM_GUI_ACTIVE"
)
g=0
for grp in "${CTRS[@]}"; do
  timeout -s KILL 240 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/g$g -- python3 tools/pmc/levels_workload.py $N $OPS $REPS > $OUT/g$g.log 2>&1
  g=$((g+1))
done
python3 tools/pmc/summary_runs.py $OUT/g0 > $OUT/summary_g0.txt 2>&1
python3 tools/pmc/summary_runs.py $OUT/g1 > $OUT/summary_g1.txt 2>&1
cp $OUT/g0.log $OUT/times.txt
find $OUT -name "*.csv" -delete; find $OUT -name "*.db" -delete
