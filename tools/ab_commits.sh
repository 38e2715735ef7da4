#!/bin/bash
# Same-box A/B of two builds of the library (box-to-box variation on the GPU pool is +-3 %, larger than most effects left).
#   1. here (no GPU needed):   tools/ab_commits.sh prepare <old-commit>
#        builds <old-commit>'s library in a scratch worktree and parks it as libmc_hip_dev.so (the slot of the developer
#        build, which MC_AMD_DEV_LIB=1 selects); the working tree's own build stays libmc_hip.so
#   2. on the GPU box:          gpurun -- 'bash tools/ab_commits.sh run [bench.py arguments ...]'
#        alternates new / old three times and prints classify / emit / step times of each run (AB_IN_FLIGHT=3: with three
#        sweeps in flight, AB_STEPS: timed steps)
#   3. here, afterwards:       python -c "import mc_amd; mc_amd.build(force=True)"   (restores the real developer build)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
PKG="$ROOT/marching-cube-for-implicit-surfaces_amd"
case "$1" in
  prepare)
    rm -rf /tmp/ab_old && git -C "$ROOT" worktree add -f /tmp/ab_old "$2" >/dev/null
    (cd /tmp/ab_old && python -c "import __graft_entry__ as g; g.build()" >/dev/null)
    python -c "import sys; sys.path.insert(0, '$ROOT'); import __graft_entry__ as g; g.build()" >/dev/null
    cp "/tmp/ab_old/marching-cube-for-implicit-surfaces_amd/libmc_hip.so" "$PKG/libmc_hip_dev.so"
    git -C "$ROOT" worktree remove --force /tmp/ab_old
    echo "old = $2 parked as libmc_hip_dev.so; new = working tree" ;;
  run)
    shift
    export MC_JIT_CACHE=${MC_JIT_CACHE:-/tmp/jc}; mkdir -p "$MC_JIT_CACHE"
    one() { python "$ROOT/bench.py" --no-cpu-baseline --in-flight ${AB_IN_FLIGHT:-1} --steps ${AB_STEPS:-40} --warmup 5 "$@" | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms']; print('classify', k['classify'], 'emit', k['emit'], 'step', d['ms_per_step'])"; }
    for rep in 1 2 3; do echo "new  $(one "$@")"; echo "old  $(MC_AMD_DEV_LIB=1 one "$@")"; done ;;
  *) echo "usage: $0 prepare <old-commit> | run [bench args]"; exit 2 ;;
esac
