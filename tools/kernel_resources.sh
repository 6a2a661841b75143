# VGPR / SGPR / LDS of every kernel whose (mangled) name matches a pattern, from a device-only compile of the product's HIP unit
# (no GPU needed):   bash tools/kernel_resources.sh estream
PAT=${1:-k_csr}
OUT=${TMPDIR:-/tmp}/fasp_kernel_resources; mkdir -p $OUT
cd "$(dirname "$0")/../faspsolver_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -fopenmp=libgomp -Wno-unused-result --cuda-device-only -c solver.hip -o $OUT/solver.co || exit 1
/opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$OUT/solver.co --output=$OUT/dev.elf --unbundle || exit 1
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $OUT/dev.elf | grep -E "^\s+\.name:|\.vgpr_count|\.sgpr_count|group_segment_fixed|\.private_segment_fixed" | paste - - - - - | grep "$PAT" | sed 's/ \+/ /g'
