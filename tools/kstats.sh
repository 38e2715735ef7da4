#!/bin/bash
# register / LDS use of every kernel in a code object or a stand-alone compile of csrc/mc_kernels.hip (no GPU needed)
#   tools/kstats.sh [-D...]            compile marching-cube-for-implicit-surfaces_amd/csrc/mc_kernels.hip with extra defines
#   tools/kstats.sh file.hsaco         inspect a code object of the JIT cache
set -e
LLVM=/opt/rocm/lib/llvm/bin
ROOT=$(cd "$(dirname "$0")/.." && pwd)
if [ -f "$1" ]; then co="$1"; else
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -ffp-contract=off -fno-slp-vectorize -O3 --genco "$@" \
      "$ROOT/marching-cube-for-implicit-surfaces_amd/csrc/mc_kernels.hip" -o /tmp/kstats.bundle
  $LLVM/clang-offload-bundler --unbundle --type=o --input=/tmp/kstats.bundle --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=/tmp/kstats.co
  co=/tmp/kstats.co
fi
$LLVM/llvm-readelf --notes "$co" | grep -E "\.name:|\.vgpr_count|\.sgpr_count|group_segment_fixed_size|private_segment_fixed_size" | paste - - - - - \
  | awk '{print $4, "lds", $2, "scratch", $6, "sgpr", $8, "vgpr", $10}'
