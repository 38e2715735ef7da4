import sys
sys.path.insert(0, '.')
import mc_amd
print("devices", mc_amd.device_count(), flush=True)
c = mc_amd.Context(0)
print("ctx ok", flush=True)
import numpy as np
print(c.eval_points("x+y", np.array([[1, 2, 3]], np.float32)), flush=True)
r = c.march("x+y", 0.25)
print("march ok", r.n_tris, r.n_cells, flush=True)
print(r.codes()[:16], r.vertices()[:2], flush=True)
