#!/usr/bin/env python3
"""A/B harness: run bench.py for several kernel variants in ONE GPU session.

    python tools/ab.py [--bench-args "..."] label1="EXTRA SOURCE LINE" label2= ...

Each variant's value is passed as MC_JIT_EXTRA (a source line prepended to the JIT-compiled
kernels, e.g. "#define MC_NO_CULL 1"); an empty value is the default build."""
import json
import os
import subprocess
import sys

args = sys.argv[1:]
bench_args = []
if args and args[0] == "--bench-args":
    bench_args = args[1].split("|")
    args = args[2:]
for spec in args:
    label, _, extra = spec.partition("=")
    for rep in (1, 2):
        env = dict(os.environ)
        if extra:
            env["MC_JIT_EXTRA"] = extra
        r = subprocess.run([sys.executable, "bench.py", "--steps", "20", "--warmup", "3", "--no-cpu-baseline", *bench_args],
                           capture_output=True, text=True, env=env, timeout=300)
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if not lines:
            print(label, rep, "FAILED", r.stderr[-400:], flush=True)
            continue
        d = json.loads(lines[-1])
        print(f"{label:18s} rep{rep} {d['kernel_ms']} step {d['ms_per_step']} tris {d['config']['triangles']}", flush=True)
