cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 60 ./tools/micro/gridbar 2>&1 | tee gpurun_out/gridbar.log
