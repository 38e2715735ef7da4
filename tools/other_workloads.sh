# Secondary workloads of the round (run on the GPU box from the repo root); output -> profiles/<tag>_other_workloads.txt
export MC_JIT_CACHE=${MC_JIT_CACHE:-/tmp/jc}
EQ3='(x^2+y^2+z^2+(1/3)^2-(1/5)^2)^2-4*((1/2)*x-(2.36/6)*(1/5))^2-4*(1/3)^2*y^2'
echo "# torus-like equation_3, grid_res 512 (BASELINE config 3)"
python tools/ab.py --bench-args "--grid-res|512|--equation|$EQ3" eq3_512= | grep rep2
echo "# sphere, grid_res 256 / 512 / 1024"
python tools/ab.py --bench-args "--grid-res|256" sphere256= | grep rep2
python tools/ab.py --bench-args "--grid-res|512" sphere512= | grep rep2
python tools/ab.py sphere1024= | grep rep2
echo "# rational f (interval walk with division): x^2+y^2+z^2-1/(x^2+4), 512"
python tools/ab.py --bench-args "--grid-res|512|--equation|x^2+y^2+z^2-1/(x^2+4)" div512= | grep rep2
echo "# sampling-walk fallback (division by a variable that crosses zero): x/y+z, 512"
python tools/ab.py --bench-args "--grid-res|512|--equation|x/y+z" divzero512= | grep rep2
echo "# sphere 512 with the interval walk compiled out (MC_NO_CULL: sampling walk)"
python tools/ab.py --bench-args "--grid-res|512" "sphere512_nocull=#define MC_NO_CULL 1" | grep rep2
echo "# gyroid sin x cos y + sin y cos z + sin z cos x, 4 periods per axis, grid_res 1024 (BASELINE config 4; grammar extension E1)"
python bench.py --workload gyroid --steps 10 --warmup 2 --no-cpu-baseline | tail -1
echo "# iso sweep on the 512^3 Goursat surface, hipGraph replay per frame (BASELINE config 5)"
python bench.py --mode isosweep --steps 60 --warmup 5 | tail -1
