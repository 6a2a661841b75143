set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
nproc; free -g | head -2; rocm-smi --showproductname 2>/dev/null | head -5
python -c "import faspsolver_amd as fa; print(fa.lib().fasp_hip_version(), fa.available())"
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -30
