python tools/classify_stamps.py --workload torus | grep -v "alive per\|started per\|last waves\|per XCC"
python tools/classify_stamps.py --workload goursat --iso -0.7 | grep -v "alive per\|started per\|last waves\|per XCC"
