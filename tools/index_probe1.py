"""one indexed sweep of the 1025^3 sphere, three times (for counter collection: tools/pmc.py with PMC_SCRIPT)"""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mc_amd as mc
c = mc.Context(0)
step = float(np.float32(2.0) / np.float32(1024))
for _ in range(3):
    r = c.march("x^2+y^2+z^2-1", step, flags=mc.FLAG_INDEXED | mc.FLAG_NO_EMIT)
print(r.n_verts, r.ms_index)
