"""Probe: one sweep of the 1025^3 grid done as S Z-slabs by S contexts on S host threads (their kernels overlap on
the GPU through separate streams) against the same sweep by one context -- what staged pipelining inside one sweep
could gain at best."""
import sys, threading, time
sys.path.insert(0, '.')
import numpy as np
import mc_amd

eq, step = "x^2+y^2+z^2-1", float(np.float32(2.0) / np.float32(1024))
K = 40
n1 = mc_amd.cells_per_axis(step)

def worker(ctx, zb, ze, n, bar):
    for _ in range(n):
        if bar is not None:
            bar.wait()
        ctx.march(eq, step, 0.0, flags=mc_amd.FLAG_NORMALS, z_begin=zb, z_end=ze)

for S in (1, 2, 3, 4):
    ctxs = [mc_amd.Context(0) for _ in range(S)]
    slabs = [mc_amd.shard_layers(n1, S, r) for r in range(S)]
    for c, (zb, ze) in zip(ctxs, slabs):
        worker(c, zb, ze, 2, None)
    bar = threading.Barrier(S)
    t0 = time.perf_counter()
    th = [threading.Thread(target=worker, args=(c, zb, ze, K, bar)) for c, (zb, ze) in zip(ctxs, slabs)]
    [t.start() for t in th]
    [t.join() for t in th]
    dt = time.perf_counter() - t0
    print(f"{S} slab(s) in parallel: {dt / K * 1e3:.4f} ms per full sweep", flush=True)
    for c in ctxs:
        c.close()
