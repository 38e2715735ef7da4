export AB_IN_FLIGHT=3 AB_STEPS=90
echo "== torus 512 (3 in flight)"; bash tools/ab_commits.sh run --workload torus
echo "== goursat isosweep (3 in flight)"; bash tools/ab_commits.sh run --mode isosweep
echo "== sphere 1024 (3 in flight)"; bash tools/ab_commits.sh run
echo "== sphere 512 (3 in flight)"; bash tools/ab_commits.sh run --grid-res 512
