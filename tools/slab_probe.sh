# developer probe: per-rank step time of an N-way Z split on one GPU (the middle slab, the heaviest of a sphere) --
# the fixed per-step costs that bound strong scaling on a grid this small.  Steps are replayed back to back (no host
# round trip per step), like the multi-GPU bench does.
export MC_JIT_CACHE=${MC_JIT_CACHE:-/tmp/jc}
mkdir -p $MC_JIT_CACHE
for n in 8 4 2 1; do
  python bench.py --no-cpu-baseline --slab-of $n --steps 200 --warmup 20 2>/dev/null | tail -1 > /tmp/slab.json
  python - "$n" <<'PY'
import json, sys
d = json.load(open('/tmp/slab.json'))
c = d["config"]
print("slab_of", sys.argv[1], "ms_per_step", d["ms_per_step"], f"({c['in_flight']} sweeps in flight; with one in flight: {c['ms_per_step_one_in_flight']})", d["kernel_ms"])
PY
done
