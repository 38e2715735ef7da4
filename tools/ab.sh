#!/bin/bash
# A/B harness: runs bench.py several times in one GPU session with different kernel variants.
# usage: tools/ab.sh "label1|ENV1=.. ENV2=.." "label2|..." ...
mkdir -p gpurun_out
for spec in "$@"; do
  label="${spec%%|*}"; envs="${spec#*|}"
  for rep in 1 2; do
    out=$(env $envs timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1)
    echo "$label rep$rep $(echo "$out" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["kernel_ms"], d["ms_per_step"])')"
  done
done
