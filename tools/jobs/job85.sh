echo "== split (default)"; python tools/emit_stamps.py --workload torus --grid-res 512
echo "== one wave per group"; MC_ES_SPLIT_MAX=0 python tools/emit_stamps.py --workload torus --grid-res 512
