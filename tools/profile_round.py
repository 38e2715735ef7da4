#!/usr/bin/env python3
"""Collect the round's profile set on the GPU box (run from the repo root):

    python tools/profile_round.py [--tag r01]

  1. rocprofv3 --kernel-trace --stats of `bench.py --steps 20 --warmup 3`  -> <tag>_kernel_stats_sphere1024.csv
  2. rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes        -> <tag>_pmc_traffic_sphere1024.json
     (HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) KB: FETCH_SIZE counts 64-byte requests as 32 on
     gfx950, see MI355X_MICROARCH.md, HBM / rocprofv3 section)
  3. rocprofv3 --pmc SQ_* instruction counters                                -> <tag>_sq_counters_sphere1024.json
  4. the default bench.py run (with the CPU baseline)                         -> <tag>_bench_sphere1024.json
Everything lands in gpurun_out/profile/; copy what should be judged into profiles/."""
import glob
import json
import os
import shutil
import signal
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[sys.argv.index("--tag") + 1] if "--tag" in sys.argv else "r01"
out = os.path.join(ROOT, "gpurun_out", "profile")
os.makedirs(out, exist_ok=True)
env = dict(os.environ, TMPDIR="/tmp")


def run(cmd, timeout, cwd="/tmp"):
    print("+", " ".join(cmd), flush=True)
    p = subprocess.Popen(cmd, cwd=cwd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, start_new_session=True)
    try:
        o, _ = p.communicate(timeout=timeout)
    except subprocess.TimeoutExpired:
        os.killpg(p.pid, signal.SIGKILL)
        print("TIMEOUT", flush=True)
        return None
    return o


bench = [sys.executable, os.path.join(ROOT, "bench.py")]
# 1. kernel trace
d = os.path.join(out, "stats")
shutil.rmtree(d, ignore_errors=True)
run(["rocprofv3", "--kernel-trace", "--stats", "-d", d, "-o", "run", "--output-format", "csv", "--", *bench, "--steps", "20", "--warmup", "3",
     "--no-cpu-baseline"], 300)
for f in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True):
    shutil.copy(f, os.path.join(out, f"{tag}_kernel_stats_sphere1024.csv"))
    print(open(f).read(), flush=True)

# 2. traffic, one counter per pass
pm = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    j = os.path.join(out, f"pmc_{c}.json")
    o = run([sys.executable, os.path.join(ROOT, "tools", "pmc.py"), j, c], 200, cwd=ROOT)
    print(o, flush=True)
    if os.path.exists(j):
        for k, v in json.load(open(j)).items():
            pm.setdefault(k, {})[c] = v.get(c, 0.0)
traffic = {k: {"FETCH_SIZE_KB_avg_per_launch": v.get("FETCH_SIZE", 0.0), "WRITE_SIZE_KB_avg_per_launch": v.get("WRITE_SIZE", 0.0),
               "hbm_bytes_per_launch_fetch_x2": int((2 * v.get("FETCH_SIZE", 0.0) + v.get("WRITE_SIZE", 0.0)) * 1024),
               "hbm_bytes_per_launch_raw": int((v.get("FETCH_SIZE", 0.0) + v.get("WRITE_SIZE", 0.0)) * 1024)} for k, v in pm.items()}
json.dump(traffic, open(os.path.join(out, f"{tag}_pmc_traffic_sphere1024.json"), "w"), indent=1, sort_keys=True)
if "mc_classify" in traffic:  # bench.py quotes the committed traffic profile: refresh it before step 4
    shutil.copy(os.path.join(out, f"{tag}_pmc_traffic_sphere1024.json"), os.path.join(ROOT, "profiles", f"{tag}_pmc_traffic_sphere1024.json"))

# 3. instruction counters
j = os.path.join(out, f"{tag}_sq_counters_sphere1024.json")
print(run([sys.executable, os.path.join(ROOT, "tools", "pmc.py"), j, "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_WR",
           "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAVES"], 200, cwd=ROOT), flush=True)

# 4. the bench line
o = run([*bench], 400, cwd=ROOT)
line = [l for l in (o or "").splitlines() if l.startswith("{")]
if line:
    open(os.path.join(out, f"{tag}_bench_sphere1024.json"), "w").write(line[-1] + "\n")
    print(line[-1], flush=True)
else:
    print(o)
