python tools/classify_stamps.py --workload goursat | grep -v "per XCC\|share of"
MC_LATE=0 python tools/classify_stamps.py --workload goursat | grep "classify \|alive per\|started per"
MC_LATE=6,21 python tools/classify_stamps.py --workload goursat | grep "classify \|alive per\|started per"
python tools/classify_stamps.py --workload torus | grep "classify \|alive per\|started per\|last waves"
