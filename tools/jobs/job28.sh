export MC_JIT_CACHE=/tmp/jc; mkdir -p /tmp/jc
for w in "" "--grid-res|256" "--workload|gyroid|--steps|5" "--grid-res|512|--equation|x^2+y^2+z^2-1/(x^2+4)"; do
  echo "== $w"
  timeout -k 10 300 python tools/ab.py --reps 1 --bench-args "$w" th63= "th42=env:MC_TILE_H=42" "th32=env:MC_TILE_H=32" "th21=env:MC_TILE_H=21" 2>&1 | sed "s/'source'.*replays right behind the timed region'}//; s/'emit_kernel': 'mc_emit[_a-z]*', //" | cut -c1-150
done
