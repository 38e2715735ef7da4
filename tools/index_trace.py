"""Kernel trace of the indexed-mesh path at 1025^3 (run under rocprofv3 --kernel-trace --stats): ten sweeps with MC_FLAG_INDEXED."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mc_amd as mc

c = mc.Context(0)
step = float(np.float32(2.0) / np.float32(int(sys.argv[1]) if len(sys.argv) > 1 else 1024))
for _ in range(12):
    r = c.march("x^2+y^2+z^2-1", step, flags=mc.FLAG_INDEXED | mc.FLAG_NO_EMIT | mc.FLAG_NO_INTERP)
print("verts", r.n_verts, "tris", r.n_tris, "ms_index", round(r.ms_index, 3), "sweep", round(r.ms_total, 3))
