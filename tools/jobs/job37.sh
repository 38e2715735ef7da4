f() { grep -E "classify |initial fill|alive per|share of wave time: main with"; }
for w in goursat torus; do
echo "== $w base"; python tools/classify_stamps.py --workload $w | f
echo "== $w TILE_H 32"; MC_TILE_H=32 python tools/classify_stamps.py --workload $w | f
echo "== $w TILE_H 21"; MC_TILE_H=21 python tools/classify_stamps.py --workload $w | f
echo "== $w MINW 6"; MC_JIT_EXTRA="#define MC_CLASSIFY_MINW 6" python tools/classify_stamps.py --workload $w | f
echo "== $w MINW 6 TILE_H 21"; MC_TILE_H=21 MC_JIT_EXTRA="#define MC_CLASSIFY_MINW 6" python tools/classify_stamps.py --workload $w | f
done
