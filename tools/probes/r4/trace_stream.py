import os, sys
sys.path.insert(0, "/root/repo")
os.environ["MC_TRACE"] = "1"
import mc_amd as mc
c = mc.Context(0)
r = c.march("x^2+y^2+z^2-1", 2.0/1024, 0.0, flags=mc.FLAG_NORMALS | mc.FLAG_NO_INTERP)
print(r.n_tris)
