"""Kernel trace of a seed-mode sweep at 1025^3 (run under rocprofv3 --kernel-trace --stats)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mc_amd as mc

c = mc.Context(0)
step = float(np.float32(2.0) / np.float32(int(sys.argv[1]) if len(sys.argv) > 1 else 1024))
c.set_seed(1.0, 0.0, 0.0)
c.seed_mode(True)
for _ in range(10):
    r = c.march("x^2+y^2+z^2-1", step, flags=mc.FLAG_NORMALS | mc.FLAG_NO_INTERP)
print("tris", r.n_tris, "ms_total", round(r.ms_total, 3))
