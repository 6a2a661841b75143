# usage (on the GPU box): bash tools/lab/run_pmc.sh <n> <what> "<xcd filter nt>" ...   -- one FETCH_SIZE and one WRITE_SIZE pass per configuration
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp OMP_NUM_THREADS=32
N=$1; WHAT=$2; shift 2
OUT=gpurun_out/labpmc; rm -rf $OUT; mkdir -p $OUT
for cfg in "$@"; do
  set -- $cfg
  tag=$(echo "$cfg" | tr ' <>,' '____')
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $OUT/${tag}_$ctr -- tools/lab/lab $N $WHAT 3 $1 $2 $3 1 > $OUT/${tag}_$ctr.log 2>&1
  done
done
python3 tools/lab/pmc_summary.py $OUT/*_FETCH_SIZE $OUT/*_WRITE_SIZE > $OUT/summary.txt 2>&1
find $OUT -name "*.csv" -delete; find $OUT -name "*.db" -delete
cat $OUT/summary.txt
