"""Developer probe: which triangles of the dense sweep a seed-mode sweep leaves out."""
import sys
import numpy as np
sys.path.insert(0, ".")
import mc_amd as mc
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
c = mc.Context(0)
step = float(np.float32(2.0) / np.float32(n))
d = c.march("x^2+y^2+z^2-1", step, flags=0).vertices()[:, :, :3].copy()
c.set_seed(1.0, 0.0, 0.0)
c.seed_mode(True)
for it in range(3):
    r = c.march("x^2+y^2+z^2-1", step, flags=0)
    s = r.vertices()[:, :, :3]
    dk = d.reshape(len(d), 9).view(np.uint32)
    sk = s.reshape(len(s), 9).view(np.uint32)
    j = 0
    missing = []
    for i in range(len(dk)):
        if j < len(sk) and np.array_equal(dk[i], sk[j]):
            j += 1
        else:
            missing.append(i)
        if len(missing) > 40:
            break
    print(it, r.n_tris, len(d), "missing:", [(i, d[i].mean(axis=0).round(4).tolist()) for i in missing], flush=True)
