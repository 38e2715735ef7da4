#!/usr/bin/env python3
"""Collect the round's profile set on the GPU box (run from the repo root):

    python tools/profile_round.py [--tag r03] [--workloads sphere1024,torus512,gyroid1024,goursat512]

Per workload (BASELINE.json configs 2/headline, 3, 4, 5):
  1. rocprofv3 --kernel-trace --stats of the bench command, with one sweep in flight (kernels alone: the durations the
     roofline uses) -> <tag>_kernel_stats_<workload>.csv, and as it runs by default (three in flight, kernels of
     consecutive sweeps overlap) -> <tag>_kernel_stats_<workload>_inflight3.csv
  2. rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes       -> <tag>_pmc_traffic_<workload>.json
     (HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) KB: FETCH_SIZE counts 64-byte requests as 32 on
     gfx950, see MI355X_MICROARCH.md, HBM / rocprofv3 section)
  3. rocprofv3 --pmc SQ_* instruction counters                              -> <tag>_sq_counters_<workload>.json
  4. the bench line itself (sphere1024: with the CPU baseline)              -> <tag>_bench_<workload>.json
Everything lands in gpurun_out/profile/ AND in profiles/ (the traffic file first: bench.py quotes it in step 4)."""
import glob
import json
import os
import shutil
import signal
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[sys.argv.index("--tag") + 1] if "--tag" in sys.argv else "r04"
wl = (sys.argv[sys.argv.index("--workloads") + 1] if "--workloads" in sys.argv else "sphere1024,torus512,gyroid1024,goursat512").split(",")
out = os.path.join(ROOT, "gpurun_out", "profile")
prof = os.path.join(ROOT, "profiles")
os.makedirs(out, exist_ok=True)
env = dict(os.environ, TMPDIR="/tmp")

BENCH_ARGS = {
    "sphere1024": [],
    "torus512": ["--workload", "torus"],
    "gyroid1024": ["--workload", "gyroid", "--steps", "5", "--warmup", "1"],
    "goursat512": ["--mode", "isosweep", "--steps", "30", "--warmup", "3"],
}


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


def keep(path, name):
    shutil.copy(path, os.path.join(out, name))
    shutil.copy(path, os.path.join(prof, name))


for w in wl:
    extra = BENCH_ARGS[w]
    bench = [sys.executable, os.path.join(ROOT, "bench.py"), *extra]
    steps = [] if "--steps" in extra else ["--steps", "20", "--warmup", "3"]
    nocpu = [] if "--mode" in extra else ["--no-cpu-baseline"]
    # 1. kernel trace
    d = os.path.join(out, f"stats_{w}")
    shutil.rmtree(d, ignore_errors=True)
    # (a) with ONE sweep in flight: the kernels run alone, their durations are what the bench line's roofline uses;
    # (b) the default command (three sweeps in flight: consecutive sweeps overlap, so single launches take longer)
    for suffix, more in (("", ["--in-flight", "1"]), ("_inflight3", [])):
        shutil.rmtree(d, ignore_errors=True)
        run(["rocprofv3", "--kernel-trace", "--stats", "-d", d, "-o", "run", "--output-format", "csv", "--", *bench, *steps, *nocpu, *more], 400)
        for f in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True):
            keep(f, f"{tag}_kernel_stats_{w}{suffix}.csv")
            print(open(f).read(), flush=True)
    # 2. traffic, one counter per pass (one sweep in flight: per-launch counts do not depend on it, attribution is cleaner)
    pm = {}
    barg = "|".join([*extra, "--in-flight", "1"])
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        j = os.path.join(out, f"pmc_{w}_{c}.json")
        cmd = [sys.executable, os.path.join(ROOT, "tools", "pmc.py"), j]
        if barg:
            cmd += ["--bench-args", barg]
        o = run([*cmd, c], 400, cwd=ROOT)
        print(o, flush=True)
        if os.path.exists(j):
            for k, v in json.load(open(j)).items():
                pm.setdefault(k, {})[c] = v.get(c, 0.0)
    traffic = {k: {"FETCH_SIZE_KB_avg_per_launch": v.get("FETCH_SIZE", 0.0), "WRITE_SIZE_KB_avg_per_launch": v.get("WRITE_SIZE", 0.0),
                   "hbm_bytes_per_launch_fetch_x2": int((2 * v.get("FETCH_SIZE", 0.0) + v.get("WRITE_SIZE", 0.0)) * 1024),
                   "hbm_bytes_per_launch_raw": int((v.get("FETCH_SIZE", 0.0) + v.get("WRITE_SIZE", 0.0)) * 1024)} for k, v in pm.items()}
    tj = os.path.join(out, f"{tag}_pmc_traffic_{w}.json")
    json.dump(traffic, open(tj, "w"), indent=1, sort_keys=True)
    if "mc_classify" in traffic:
        shutil.copy(tj, os.path.join(prof, f"{tag}_pmc_traffic_{w}.json"))
    # 3. instruction counters
    j = os.path.join(out, f"{tag}_sq_counters_{w}.json")
    cmd = [sys.executable, os.path.join(ROOT, "tools", "pmc.py"), j]
    if barg:
        cmd += ["--bench-args", barg]
    print(run([*cmd, "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_WR", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAVES",
               "SQ_ACTIVE_INST_VALU"], 400, cwd=ROOT), flush=True)
    if os.path.exists(j):
        shutil.copy(j, os.path.join(prof, f"{tag}_sq_counters_{w}.json"))
    # 4. the bench line
    o = run([*bench, *([] if w == "sphere1024" else nocpu)], 600, cwd=ROOT)
    line = [l for l in (o or "").splitlines() if l.startswith("{")]
    if line:
        for dst in (out, prof):
            open(os.path.join(dst, f"{tag}_bench_{w}.json"), "w").write(line[-1] + "\n")
        print(line[-1], flush=True)
    else:
        print(o)
