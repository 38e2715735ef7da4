for a in "--workload goursat" "--workload torus" "--workload sphere" "--workload sphere --grid-res 512"; do
python tools/classify_stamps.py $a | grep -v "alive per\|started per\|last waves\|per XCC\|share of"
done
