for w in goursat torus; do python tools/classify_stamps.py --workload $w; done
python tools/classify_stamps.py --workload goursat --iso -0.7
python tools/classify_stamps.py --workload sphere --grid-res 512
python tools/classify_stamps.py --workload sphere --grid-res 1024
