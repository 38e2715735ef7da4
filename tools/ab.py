#!/usr/bin/env python3
"""A/B harness: run bench.py for several kernel variants in ONE GPU session (developer build of the library).

    python tools/ab.py [--bench-args "a|b|c"] [--reps N] label1=SPEC label2=SPEC ...

SPEC is a `;`-separated list of
    #define ...            a source line prepended to the JIT-compiled kernels (MC_JIT_EXTRA)
    patch:FILE             a file of text replacements applied to the kernel source (MC_JIT_PATCH; tools/probes/*.patch:
                           blocks  OLD / a line "====" / NEW  separated by lines "@@@@") -- the timing probes
                           ("what if this phase did nothing") live there, not in the shipped kernels
    env:NAME=VALUE         an environment variable of the developer build (MC_WPB_EMIT, MC_WPB_CLASSIFY, MC_TILE_H ...;
                           MC_JIT_OPTS=further hiprtc options, e.g. `-mllvm -amdgpu-sched-strategy=max-ilp`)
An empty SPEC is the default build.  Everything runs against libmc_hip_dev.so (MC_AMD_DEV_LIB=1); the shipped
libmc_hip.so honours none of these hooks."""
import json
import os
import subprocess
import sys

args = sys.argv[1:]
bench_args, reps = [], 2
while args and args[0].startswith("--"):
    if args[0] == "--bench-args":
        bench_args = args[1].split("|")
    elif args[0] == "--reps":
        reps = int(args[1])
    args = args[2:]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for spec in args:
    label, _, what = spec.partition("=")
    env = dict(os.environ, MC_AMD_DEV_LIB="1", MC_JIT_CACHE=os.environ.get("MC_JIT_CACHE", "/tmp/jc_ab"))
    os.makedirs(env["MC_JIT_CACHE"], exist_ok=True)
    extra = []
    for item in filter(None, (w.strip() for w in what.split(";"))):
        if item.startswith("patch:"):
            env["MC_JIT_PATCH"] = os.path.join(root, item[6:])
        elif item.startswith("env:"):
            k, _, v = item[4:].partition("=")
            env[k] = v
        else:
            extra.append(item)
    if extra:
        env["MC_JIT_EXTRA"] = "\n".join(extra)
    for rep in range(1, reps + 1):
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "20", "--warmup", "3", "--no-cpu-baseline", *bench_args],
                           capture_output=True, text=True, env=env, timeout=120)
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if not lines:
            print(label, rep, "FAILED", r.stderr[-600:], flush=True)
            continue
        d = json.loads(lines[-1])
        print(f"{label:22s} rep{rep} {d['kernel_ms']} step {d['ms_per_step']} tris {d['config'].get('triangles', d['config'].get('triangles_total'))} one-in-flight {d['config'].get('ms_per_step_one_in_flight')}", flush=True)
