"""Developer probe: do sweeps of Z slabs on SEPARATE contexts (streams) overlap usefully on one GPU?
K replays of the whole grid on one context, against K replays of each of N slabs on N contexts running concurrently."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import mc_amd as mc
eq = "x^2+y^2+z^2-1"
n = 1024
K = 40
step = float(np.float32(2.0) / np.float32(n))
n1 = mc.cells_per_axis(step)
flags = mc.FLAG_NORMALS | mc.FLAG_NO_TIMING


def run(nslab):
    cs = [mc.Context(0) for _ in range(nslab)]
    for r, c in enumerate(cs):
        zb, ze = mc.shard_layers(n1, nslab, r)
        c.graph_build(eq, step, 0.0, flags=flags, z_begin=zb, z_end=ze)
        c.graph_replay(0.0)
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(K):
            for c in cs:
                c.graph_replay_async(0.0)
        tris = sum(c.graph_wait().n_tris for c in cs)
        best = min(best, (time.perf_counter() - t0) / K * 1e3)
    for c in cs:
        c.close()
    return best, tris


for nslab in (1, 2, 3, 4, 8):
    ms, tris = run(nslab)
    print(f"{nslab} concurrent slab contexts: {ms:.4f} ms per whole-grid step, {tris} triangles", flush=True)
