import sys, zlib
import numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mc_amd as mc
mc.set_extensions(mc.EXT_TRIG)
cases = [("sin(x)*cos(y)+sin(y)*cos(z)+sin(z)*cos(x)", 300, 0.1, (12.566371, 9.0, 12.566371), [("x+y", "<=", 0.4)]),
         ("(x^2)^2+(y^2)^2+(z^2)^2-(x^2+y^2+z^2)", 512, -0.4, (1.0, 1.0, 1.0), []),
         ("x^2+y^2+z^2-1", 1024, 0.0, (1.0, 1.0, 1.0), []),
         ("(x^2+y^2+z^2+(1/3)^2-(1/5)^2)^2-4*((1/2)*x-(2.36/6)*(1/5))^2-4*(1/3)^2*y^2", 400, 0.0, (1.1, 1.1, 1.1), [("z", ">", -0.2), ("x*y", "<", 0.3)])]
for eq, n, iso, scale, cons in cases:
    step = float(np.float32(2.0) / np.float32(n))
    c = mc.Context(0)
    sh = mc.Sharded([0] * 8)
    for ctx in [c] + sh.ctxs:
        for i, (l, o, r) in enumerate(cons):
            ctx.set_constraint(i, l, o, r)
    for flags in (mc.FLAG_NORMALS | mc.FLAG_KEEP_CODES, mc.FLAG_INDEXED | mc.FLAG_NO_EMIT):
        w = c.march(eq, step, iso, scale, flags=flags)
        s = sh.march(eq, step, iso, scale, flags=flags)
        assert (w.n_tris, w.n_active) == (s.n_tris, s.n_active), (eq, w.n_tris, s.n_tris)
        if flags & mc.FLAG_INDEXED:
            a, b = w.indexed(), s.indexed()
            assert w.n_verts == s.n_verts
            for x, y in zip(a, b):
                assert zlib.crc32(x.tobytes()) == zlib.crc32(y.tobytes()) or np.array_equal(x.view(np.uint32), y.view(np.uint32)), eq
        else:
            assert zlib.crc32(w.vertices().tobytes()) == zlib.crc32(s.vertices().tobytes()), eq
            assert zlib.crc32(w.codes().tobytes()) == zlib.crc32(s.codes().tobytes()), eq
    print("ok", eq[:30], n, w.n_tris, w.n_verts, flush=True)
    sh.close(); c.close()
print("STRESS_OK")
