"""Probe: throughput of repeated sweeps with 1 context vs 2 contexts driven from 2 host threads
(kernels of different sweeps overlap on the GPU through their own streams)."""
import sys, threading, time
sys.path.insert(0, '.')
import numpy as np
import mc_amd

eq, step = "x^2+y^2+z^2-1", float(np.float32(2.0) / np.float32(1024))
K = 30

FLAGS = mc_amd.FLAG_NO_EMIT if "--classify-only" in sys.argv else mc_amd.FLAG_NORMALS


def worker(ctx, n):
    for _ in range(n):
        ctx.march(eq, step, 0.0, flags=FLAGS)

for nctx in (1, 2, 3):
    ctxs = [mc_amd.Context(0) for _ in range(nctx)]
    for c in ctxs:
        worker(c, 2)
    t0 = time.perf_counter()
    th = [threading.Thread(target=worker, args=(c, K)) for c in ctxs]
    [t.start() for t in th]
    [t.join() for t in th]
    dt = time.perf_counter() - t0
    print(f"{nctx} context(s): {dt / (K * nctx) * 1e3:.4f} ms per sweep", flush=True)
    for c in ctxs:
        c.close()
