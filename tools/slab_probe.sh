# developer probe: per-rank step time of an N-way Z split on one GPU, for several classify tile heights
export MC_JIT_CACHE=${MC_JIT_CACHE:-/tmp/jc}
for th in 63 31 21 15 8; do
  for n in 8 4; do
    MC_TILE_H=$th python bench.py --no-cpu-baseline --slab-of $n --steps 50 --warmup 5 2>/dev/null | tail -1 > /tmp/slab.json
    python - "$th" "$n" <<'PY'
import json, sys
d = json.load(open('/tmp/slab.json'))
print("tile_h", sys.argv[1], "slab_of", sys.argv[2], d["ms_per_step"], d["kernel_ms"])
PY
  done
done
