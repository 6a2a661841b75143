# rocprofv3 profiles of the bench command (run on the GPU box via gpurun)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
export OMP_NUM_THREADS=${OMP_NUM_THREADS:-32}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof
rm -rf $OUT; mkdir -p $OUT
CMD="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $CMD > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $CMD > $OUT/pmc_write.log 2>&1
find $OUT -name "*.csv" | head -20
python3 tools/summarize_prof.py $OUT > $OUT/summary.md 2>&1
cat $OUT/summary.md | head -70
# keep the merged-back payload small
find $OUT -name "*kernel_trace.csv" -size +20M -delete
find $OUT -name "*counter_collection.csv" -size +20M -delete
