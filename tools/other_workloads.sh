# Secondary workloads of the round (run on the GPU box from the repo root); output -> profiles/<tag>_other_workloads.txt
export MC_JIT_CACHE=${MC_JIT_CACHE:-/tmp/jc}
mkdir -p $MC_JIT_CACHE
row() { python - "$@" <<'PY'
import json, subprocess, sys
label, args = sys.argv[1], sys.argv[2:]
o = subprocess.run([sys.executable, "bench.py", "--no-cpu-baseline", *args], capture_output=True, text=True).stdout
d = json.loads([l for l in o.splitlines() if l.startswith("{")][-1])
k = d["kernel_ms"]
print(f"{label:34s} step {d['ms_per_step']:.4f} ms  classify {k['classify']:.4f} scan {k['scan']:.4f} emit {k['emit']:.4f} ({k['emit_kernel']})  "
      f"tris {d['config']['triangles']}  one-in-flight {d.get('ms_per_step_one_in_flight')}  {d['roofline']['kernel']} frac {d['roofline']['frac']:.3f}  "
      f"{d['second_roofline']['kernel']} frac {d['second_roofline']['frac']:.3f}  pipeline frac {d['pipeline']['frac']:.3f}", flush=True)
PY
}
EQ3='(x^2+y^2+z^2+(1/3)^2-(1/5)^2)^2-4*((1/2)*x-(2.36/6)*(1/5))^2-4*(1/3)^2*y^2'
row "sphere 256"   --grid-res 256
row "sphere 512"   --grid-res 512
row "sphere 1024 (headline)"
row "sphere 1024, no normals" --no-normals
row "equation_3 512 (config 3)" --workload torus
row "equation_3 1024" --grid-res 1024 --equation "$EQ3"
row "rational x^2+y^2+z^2-1/(x^2+4) 512" --grid-res 512 --equation "x^2+y^2+z^2-1/(x^2+4)"
row "sampling walk x/y+z 512" --grid-res 512 --equation "x/y+z"
row "gyroid 1024 (config 4)" --workload gyroid --steps 10 --warmup 2
echo "# iso sweep on the 512^3 Goursat surface, one hipGraph replay per frame (config 5)"
python bench.py --mode isosweep --steps 60 --warmup 5 | tail -1
echo "# seed mode against the dense sweep (tools/seed_probe.py)"
python tools/seed_probe.py
