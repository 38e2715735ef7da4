"""Developer probe: what ONE sweep would cost if it ran as N sub-slabs of layers on N streams (classify / scan / emit of the
sub-slabs overlapping freely), emulated with N contexts that each sweep 1/N of the slab: per step all N are launched, then
all are waited for (a host round trip per step, as with one mc_march per step)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mc_amd as mc
K = 60
flags = mc.FLAG_NORMALS | mc.FLAG_NO_TIMING
EQ3 = "(x^2+y^2+z^2+(1/3)^2-(1/5)^2)^2-4*((1/2)*x-(2.36/6)*(1/5))^2-4*(1/3)^2*y^2"
GOURSAT = "(x^2)^2+(y^2)^2+(z^2)^2-(x^2+y^2+z^2)"


def run(eq, n, iso, nsub, zb=0, ze=-1):
    step = float(np.float32(2.0) / np.float32(n))
    n1 = mc.cells_per_axis(step)
    ze = n1 if ze < 0 else ze
    cs = [mc.Context(0) for _ in range(nsub)]
    for r, c in enumerate(cs):
        a, b = mc.shard_layers(ze - zb, nsub, r)
        c.graph_build(eq, step, iso, flags=flags, z_begin=zb + a, z_end=zb + b)
        c.graph_replay(iso)
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(K):
            for c in cs:
                c.graph_replay_async(iso)
            tris = sum(c.graph_wait().n_tris for c in cs)
        best = min(best, (time.perf_counter() - t0) / K * 1e3)
    for c in cs:
        c.close()
    return best, tris


for label, eq, n, iso, zb, ze in (("sphere 513^3", "x^2+y^2+z^2-1", 512, 0.0, 0, -1), ("equation_3 513^3", EQ3, 512, 0.0, 0, -1),
                                  ("goursat 513^3 iso -0.4", GOURSAT, 512, -0.4, 0, -1), ("sphere 1025^3", "x^2+y^2+z^2-1", 1024, 0.0, 0, -1),
                                  ("sphere 1025^3 middle 1/8", "x^2+y^2+z^2-1", 1024, 0.0, 448, 577), ("sphere 257^3", "x^2+y^2+z^2-1", 256, 0.0, 0, -1)):
    print(label, " ".join(f"{nsub}: {run(eq, n, iso, nsub, zb, ze)[0]:.4f}" for nsub in (1, 2, 3, 4, 6)), "ms per step", flush=True)
