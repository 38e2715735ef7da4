#!/usr/bin/env python3
"""Developer probe (GPU box, developer build): where mc_emit's waves spend their cycles, phase by phase, and how the
waves fill the chip over time.
    python tools/emit_stamps.py [--workload gyroid] [--grid-res N]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MC_AMD_DEV_LIB", "1")
os.environ.setdefault("MC_JIT_PATCH", os.path.join(os.path.dirname(os.path.abspath(__file__)), "probes", "emit_stamps.patch"))
os.environ.setdefault("MC_JIT_CACHE", "/tmp/jc_ab")
os.environ.setdefault("MC_EMIT_MODE", "shared")   # (a cheap f would take mc_emit_direct, which carries no stamps)
os.makedirs(os.environ["MC_JIT_CACHE"], exist_ok=True)
import mc_amd  # noqa: E402

n = int(sys.argv[sys.argv.index("--grid-res") + 1]) if "--grid-res" in sys.argv else 1024
eq, scale = "x^2+y^2+z^2-1", (1.0,) * 3
if "--workload" in sys.argv and sys.argv[sys.argv.index("--workload") + 1] == "gyroid":
    mc_amd.set_extensions(1)
    eq, scale = "sin(x)*cos(y)+sin(y)*cos(z)+sin(z)*cos(x)", (12.566371,) * 3
if "--workload" in sys.argv and sys.argv[sys.argv.index("--workload") + 1] == "torus":
    eq = "(x^2+y^2+z^2+(1/3)^2-(1/5)^2)^2-4*((1/2)*x-(2.36/6)*(1/5))^2-4*(1/3)^2*y^2"
ctx = mc_amd.Context(0)
step = float(np.float32(2.0) / np.float32(n))
L = mc_amd.lib()
L.mc_dev_read_symbol.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_size_t, C.c_int]
buf = np.zeros(8 * 131072, np.uint64)
ctx.march(eq, step, 0.0, scale, flags=mc_amd.FLAG_NORMALS)
assert L.mc_dev_read_symbol(ctx._h, b"mc_dbgw", buf.ctypes.data, buf.nbytes, 1) == 0
r = ctx.march(eq, step, 0.0, scale, flags=mc_amd.FLAG_NORMALS)
assert L.mc_dev_read_symbol(ctx._h, b"mc_dbgw", buf.ctypes.data, buf.nbytes, 1) == 0
d = buf.reshape(-1, 8)
d = d[d[:, 1] > 0]
start, end = d[:, 0].astype(np.int64), d[:, 1].astype(np.int64)
t0 = start.min()
dur_us = (end - start) / 100.0
nch = (d[:, 7] & 0xFFFF).astype(np.int64)
nrec = ((d[:, 7] >> 16) & 0xFFFF).astype(np.int64)
print(f"emit {r.ms_emit:.3f} ms; non-empty waves {len(d)}; span of their stamps {(end.max() - t0) / 100.0:.1f} us")
print(f"chunks/wave mean {nch.mean():.2f} max {nch.max()}; records/wave mean {nrec.mean():.1f} max {nrec.max()}; records/chunk {nrec.sum() / nch.sum():.1f}")
print(f"wave lifetime us: mean {dur_us.mean():.2f} p50 {np.percentile(dur_us, 50):.2f} p90 {np.percentile(dur_us, 90):.2f} max {dur_us.max():.2f}")
tot = d[:, 2:7].astype(np.float64)
print("cycles per wave (mean): prologue %.0f | per chunk: wait-for-records %.0f  B %.0f  C %.0f  D %.0f" %
      (tot[:, 0].mean(), tot[:, 1].sum() / nch.sum(), tot[:, 2].sum() / nch.sum(), tot[:, 3].sum() / nch.sum(), tot[:, 4].sum() / nch.sum()))
# concurrency over time: non-empty waves alive per 10 us bucket
edges = np.arange(0, (end.max() - t0) / 100.0 + 10, 10)
alive = [(int(((start - t0) / 100.0 < b + 10).sum() - ((end - t0) / 100.0 < b).sum())) for b in edges[:-1]]
print("non-empty waves alive per 10 us:", alive)
