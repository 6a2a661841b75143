#!/bin/bash
# Development: rocprofv3 kernel statistics of configs 3 and 5 (tools/perf_configs.py), top kernels only -> gpurun_out/prof_cfg<c>.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in ${1:-5 3}; do
  rm -rf /tmp/prof_cfg$c
  timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_cfg$c -o cfg$c --output-format csv -- python3 $R/tools/perf_configs.py $c > $R/gpurun_out/prof_cfg${c}_run.log 2>&1
  f=$(find /tmp/prof_cfg$c -name "*kernel_stats.csv" | head -1)
  { grep "^config" $R/gpurun_out/prof_cfg${c}_run.log; head -14 "$f" | cut -c1-200; } > $R/gpurun_out/prof_cfg$c.txt
done
