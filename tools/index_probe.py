import sys, time, numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mc_amd as mc
c = mc.Context(0)
for n in (256, 512, 1024):
    step = float(np.float32(2.0) / np.float32(n))
    for _ in range(2):
        r = c.march("x^2+y^2+z^2-1", step, flags=mc.FLAG_INDEXED | mc.FLAG_NO_EMIT)
    t0 = time.perf_counter(); v, t, nn = r.indexed(); dt = time.perf_counter() - t0
    print(n, "verts", r.n_verts, "tris", r.n_tris, "ms_index", round(r.ms_index, 3), "classify", round(r.ms_classify, 3), "copy_ms", round(dt * 1e3, 1), flush=True)
mc.set_extensions(1)
r = c.march("sin(x)*cos(y)+sin(y)*cos(z)+sin(z)*cos(x)", float(np.float32(2.0) / np.float32(1024)), 0.0, (12.566371,) * 3, flags=mc.FLAG_INDEXED | mc.FLAG_NO_EMIT)
print("gyroid 1024 verts", r.n_verts, "tris", r.n_tris, "ms_index", round(r.ms_index, 3))
