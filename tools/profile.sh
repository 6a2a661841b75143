# rocprofv3 profiles of the bench command (run on the GPU box via gpurun): kernel trace + stats, then the
# two PMC passes (FETCH_SIZE, WRITE_SIZE -- separate runs, nothing but --kernel-trace beside them).
# usage: bash tools/profile.sh [round-tag] [constant|variable]   -> gpurun_out/prof_<tag>[_var]/{kernel_stats.csv,summary.md,traffic.json}
# (constant = the headline P7(256) solve; variable = the variable-coefficient twin, every level on plain-CSR kernels)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
export OMP_NUM_THREADS=${OMP_NUM_THREADS:-32}
TAG=${1:-r06}
WL=${2:-constant}
if [ "$WL" = "variable" ]; then
  OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_${TAG}_var
  ARGS="bench.py --only-variable"
else
  OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
  ARGS="bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-variable --no-extra --no-plain --no-ceilings"
fi
rm -rf $OUT; mkdir -p $OUT
timeout -s KILL 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ARGS > $OUT/trace.log 2>&1
timeout -s KILL 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ARGS > $OUT/pmc_fetch.log 2>&1
timeout -s KILL 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ARGS > $OUT/pmc_write.log 2>&1
python3 tools/summarize_prof.py $OUT $WL "$ARGS" > $OUT/summary.md 2>&1
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv 2>/dev/null
head -60 $OUT/summary.md
# keep the merged-back payload small
find $OUT -name "*kernel_trace.csv" -size +20M -delete
find $OUT -name "*counter_collection.csv" -size +20M -delete
find $OUT -name "*.db" -delete
