"""Developer probe: K independent sweeps of the SAME grid (or slab), round-robin over C contexts (= C streams, C sets of
buffers): do consecutive sweeps overlap usefully?  Time per sweep."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import mc_amd as mc
eq = "x^2+y^2+z^2-1"
K = 48
flags = mc.FLAG_NORMALS | mc.FLAG_NO_TIMING


def run(n, nctx, zb=0, ze=-1):
    step = float(np.float32(2.0) / np.float32(n))
    cs = [mc.Context(0) for _ in range(nctx)]
    for c in cs:
        c.graph_build(eq, step, 0.0, flags=flags, z_begin=zb, z_end=ze)
        c.graph_replay(0.0)
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        for k in range(K):
            cs[k % nctx].graph_replay_async(0.0)
        tris = [c.graph_wait().n_tris for c in cs]
        best = min(best, (time.perf_counter() - t0) / K * 1e3)
    for c in cs:
        c.close()
    return best, tris[0]


for label, n, zb, ze in (("1025^3 whole", 1024, 0, -1), ("1025^3 middle 1/8 slab", 1024, 448, 577), ("1025^3 middle 1/4 slab", 1024, 384, 641),
                         ("513^3 whole", 512, 0, -1)):
    for nctx in (1, 2, 3, 4):
        ms, tris = run(n, nctx, zb, ze)
        print(f"{label:24s} {nctx} contexts: {ms:.4f} ms per sweep, {tris} triangles", flush=True)
