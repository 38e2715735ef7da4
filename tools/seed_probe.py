"""Developer probe: wall time of a seed-mode sweep against the dense sweep."""
import sys, time
import numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mc_amd as mc
c = mc.Context(0)
for n in (32, 256, 1024):
    step = float(np.float32(2.0) / np.float32(n))
    for seed in (None, (1.0, 0.0, 0.0)):
        c.seed_mode(seed is not None)
        if seed:
            c.set_seed(*seed)
        c.march("x^2+y^2+z^2-1", step, flags=mc.FLAG_NORMALS)
        t0 = time.perf_counter()
        for _ in range(3):
            r = c.march("x^2+y^2+z^2-1", step, flags=mc.FLAG_NORMALS)
        print(n, "seed" if seed else "dense", f"{(time.perf_counter() - t0) / 3 * 1e3:.3f} ms", r.n_tris, flush=True)
