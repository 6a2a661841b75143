# usage (GPU box): bash tools/lab/run_pmc2.sh <n> <what> "<xcd filter nt>" ...   -- SQ counter groups per configuration
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp OMP_NUM_THREADS=32
N=$1; WHAT=$2; shift 2
OUT=gpurun_out/labpmc2; rm -rf $OUT; mkdir -p $OUT
CTRS=(
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
 "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INST_LEVEL_VMEM SQ_WAVES GRBM_GUI_ACTIVE"
 "SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_IFETCH"
)
for cfg in "$@"; do
  set -- $cfg
  tag=$(echo "$cfg" | tr ' <>,' '____')
  g=0
  for grp in "${CTRS[@]}"; do
    timeout -s KILL 90 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/${tag}_g$g -- tools/lab/lab $N $WHAT 3 $1 $2 $3 1 > $OUT/${tag}_g$g.log 2>&1
    g=$((g+1))
  done
done
python3 tools/lab/pmc_summary.py $OUT/*_g? > $OUT/summary.txt 2>&1
find $OUT -name "*.csv" -delete; find $OUT -name "*.db" -delete
