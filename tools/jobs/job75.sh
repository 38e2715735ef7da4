echo "== torus 512"; bash tools/ab_commits.sh run --workload torus
echo "== torus, 3 in flight"; AB_IN_FLIGHT=3 AB_STEPS=90 bash tools/ab_commits.sh run --workload torus
echo "== rational, 3 in flight"; AB_IN_FLIGHT=3 AB_STEPS=90 bash tools/ab_commits.sh run --grid-res 512 --equation "x^2+y^2+z^2-1/(x^2+4)"
echo "== gyroid 1024 split forced"; MC_ES_SPLIT_MAX=1000000 bash tools/ab_commits.sh run --workload gyroid
echo "== gyroid 1024 split forced, 3 in flight"; AB_IN_FLIGHT=3 AB_STEPS=20 MC_ES_SPLIT_MAX=1000000 bash tools/ab_commits.sh run --workload gyroid
echo "== torus 1024 split forced"; MC_ES_SPLIT_MAX=1000000 bash tools/ab_commits.sh run --workload torus --grid-res 1024
