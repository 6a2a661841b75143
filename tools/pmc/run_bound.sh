# usage (GPU box): bash tools/pmc/run_bound.sh <n> <kinds> <reps> [var]  -- TA / TCP / TCC / SQ counter groups of every level's
# operator kernel, COLD, beside a plain read of the same level (separate rocprofv3 passes, nothing but --kernel-trace beside --pmc)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp OMP_NUM_THREADS=32
N=${1:-256}; KINDS=${2:-0,8}; REPS=${3:-4}; VAR=${4:-}
OUT=gpurun_out/pmc_bound$VAR; rm -rf $OUT; mkdir -p $OUT
CTRS=(
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE"
 "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TOTAL_WAVEFRONTS_sum"
 "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum"
 "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_TCC_READ_REQ_LATENCY_sum"
 "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_TAG_STALL_sum"
 "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_LEVEL_sum TCC_BUSY_sum"
 "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_DRAM_sum TCC_CYCLE_sum TCC_READ_sum"
 "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TCP_TA_TCP_STATE_READ_sum TCP_TOTAL_READ_sum"
 "TD_TD_BUSY_sum TD_TC_STALL_sum TD_LOAD_WAVEFRONT_sum TD_SPI_STALL_sum"
)
g=0
for grp in "${CTRS[@]}"; do
  timeout -s KILL 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/g$g -- python3 tools/pmc/bound_workload.py $N $KINDS $REPS $VAR > $OUT/g$g.log 2>&1
  echo "pass $g rc $?" >> $OUT/passes.txt
  g=$((g+1))
done
python3 tools/pmc/bound_summary.py $OUT > $OUT/summary.txt 2>&1
find $OUT -name "*.csv" -size +30M -delete; find $OUT -name "*.db" -delete
tail -5 $OUT/passes.txt
