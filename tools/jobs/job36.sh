f() { grep -E "classify |initial fill|alive on average"; }
echo "== base"; python tools/classify_stamps.py --workload sphere | f
echo "== WPB 2"; MC_WPB_CLASSIFY=2 python tools/classify_stamps.py --workload sphere | f
echo "== WPB 1"; MC_WPB_CLASSIFY=1 python tools/classify_stamps.py --workload sphere | f
echo "== WPB 8"; MC_WPB_CLASSIFY=8 python tools/classify_stamps.py --workload sphere | f
echo "== MINW 6"; MC_JIT_EXTRA="#define MC_CLASSIFY_MINW 6" python tools/classify_stamps.py --workload sphere | f
echo "== MINW 8"; MC_JIT_EXTRA="#define MC_CLASSIFY_MINW 8" python tools/classify_stamps.py --workload sphere | f
echo "== REC_CAP 256 (patched by sed in the source define is not overridable) skipped"
