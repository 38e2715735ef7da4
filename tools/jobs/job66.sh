echo "== goursat isosweep"; bash tools/ab_commits.sh run --mode isosweep
echo "== sphere 512"; bash tools/ab_commits.sh run --grid-res 512
echo "== sphere 1024"; bash tools/ab_commits.sh run
echo "== sphere 256"; bash tools/ab_commits.sh run --grid-res 256
echo "== sphere 1024, 3 in flight"; AB_IN_FLIGHT=3 AB_STEPS=90 bash tools/ab_commits.sh run
echo "== goursat 3 in flight"; AB_IN_FLIGHT=3 AB_STEPS=90 bash tools/ab_commits.sh run --mode isosweep
