#!/bin/bash
# A/B harness: runs bench.py several times in one GPU session with different kernel variants.
# usage: tools/ab.sh label1 "ENVSTRING1" label2 "ENVSTRING2" ...   (ENVSTRING = value of MC_JIT_EXTRA, may be empty)
# extra bench arguments come from $BENCH_ARGS
mkdir -p gpurun_out
while [ $# -ge 2 ]; do
  label="$1"; extra="$2"; shift 2
  for rep in 1 2; do
    out=$(MC_JIT_EXTRA="$extra" timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline $BENCH_ARGS 2>/dev/null | tail -1)
    echo "$label rep$rep $(echo "$out" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["kernel_ms"], d["ms_per_step"], d["config"]["triangles"])')"
  done
done
