"""Developer aid: compare cube codes with the oracle for one equation / grid and list the rows that differ."""
import sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "oracle")
import mc_amd as mc
import pyoracle as orc
eq = sys.argv[1] if len(sys.argv) > 1 else "(x-0.1)*(y-0.07)-0.001"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4
step = float(np.float32(2.0) / np.float32(n))
c = mc.Context(0)
r = c.march(eq, step)
o = orc.march(eq, step, pow_mode=orc.POW_EXACT, want=3)
n1 = o.n1
a, b = r.codes().reshape(n1, n1, n1), o.codes.reshape(n1, n1, n1)
bad = np.argwhere((a != b).any(axis=2))
print("n1", n1, "tris", r.n_tris, o.n_tris, "rows differing:", len(bad), "of", n1 * n1)
for z, y in bad[:12]:
    print("z", z, "y", y, "got", a[z, y][:24], "want", b[z, y][:24], "first x", np.argwhere(a[z, y] != b[z, y])[:4].ravel())
