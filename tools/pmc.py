#!/usr/bin/env python3
"""Collect hardware counters for one bench.py run with rocprofv3 --pmc and print per-kernel means.

    python tools/pmc.py OUT.json [--extra "#define X 1"] [--bench-args "a|b"] COUNTER [COUNTER ...]

Run on the GPU box from the repo root.  Counters are collected in their own pass (no tracing flags).
Values are the mean over dispatches of the kernel's counter (summed over the counter's instances)."""
import csv
import glob
import json
import os
import subprocess
import sys
from collections import defaultdict

args = sys.argv[1:]
out_json = args.pop(0)
extra, bench_args = None, []
while args and args[0].startswith("--"):
    if args[0] == "--extra":
        extra = args[1]
    elif args[0] == "--bench-args":
        bench_args = args[1].split("|")
    args = args[2:]
counters = args
repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
outdir = os.path.join(repo, "gpurun_out", "pmc_" + os.path.basename(out_json).replace(".json", ""))
env = dict(os.environ, TMPDIR="/tmp")
if extra:
    env["MC_JIT_EXTRA"] = extra
if os.environ.get("PMC_SCRIPT"):   # another driver than bench.py (e.g. tools/index_probe.py)
    cmd = ["rocprofv3", "--pmc", *counters, "-d", outdir, "-o", "run", "--output-format", "csv", "--",
           sys.executable, os.path.join(repo, os.environ["PMC_SCRIPT"]), *bench_args]
else:
    cmd = ["rocprofv3", "--pmc", *counters, "-d", outdir, "-o", "run", "--output-format", "csv", "--",
           sys.executable, os.path.join(repo, "bench.py"), *([] if "--steps" in bench_args else ["--steps", "3", "--warmup", "1"]),
           *([] if "--mode" in bench_args else ["--no-cpu-baseline"]), *bench_args]
import signal
proc = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, start_new_session=True)
try:
    out, _ = proc.communicate(timeout=300)
except subprocess.TimeoutExpired:
    os.killpg(proc.pid, signal.SIGKILL)   # the profiler starts the program as a grandchild: kill the whole group
    print("rocprofv3 timed out (300 s); counters:", counters)
    sys.exit(2)
class r:  # noqa
    stdout, stderr = out, ""
files = glob.glob(os.path.join(outdir, "**", "*counter_collection.csv"), recursive=True)
if not files:
    print("no counter file", r.stdout[-500:], r.stderr[-1500:])
    sys.exit(1)
acc = defaultdict(lambda: defaultdict(lambda: defaultdict(float)))   # kernel -> counter -> dispatch -> value
for f in files:
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0]
        acc[k][row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
res = {k: {c: sum(d.values()) / len(d) for c, d in cs.items()} for k, cs in acc.items()}
res = {k: v for k, v in res.items() if k.startswith("mc_")}
json.dump(res, open(out_json, "w"), indent=1, sort_keys=True)
for k, v in sorted(res.items()):
    print(k, {c: round(x) for c, x in sorted(v.items())}, flush=True)
