#!/usr/bin/env python3
"""One sweep in flight, in one piece and as parts side by side (MC_FLAG_NO_PARTS off / on): same bytes? how long?
   python tools/parts_probe.py            (developer build: MC_AMD_DEV_LIB=1 MC_PIPE_PARTS=2|3|4)"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mc_amd as mc

EQ3 = "(x^2+y^2+z^2+(1/3)^2-(1/5)^2)^2-4*((1/2)*x-(2.36/6)*(1/5))^2-4*(1/3)^2*y^2"
CASES = [("sphere1024", "x^2+y^2+z^2-1", 1024, 0.0), ("torus512", EQ3, 512, 0.0),
         ("goursat512", "(x^2)^2+(y^2)^2+(z^2)^2-(x^2+y^2+z^2)", 512, -0.3), ("sphere512", "x^2+y^2+z^2-1", 512, 0.0)]
c = mc.Context(0)
for name, eq, n, iso in CASES:
    step = 2.0 / n
    out = {}
    for label, extra in (("one piece", mc.FLAG_NO_PARTS), ("parts", 0)):
        flags = mc.FLAG_NORMALS | mc.FLAG_NO_INTERP | extra
        for _ in range(4):
            r = c.march(eq, step, iso, flags=flags)
        ms, wall = [], []
        for _ in range(12):
            t0 = time.perf_counter()
            r = c.march(eq, step, iso, flags=flags)
            wall.append((time.perf_counter() - t0) * 1e3)
            ms.append(r.ms_total)
        v = r.vertices() if n <= 512 else None
        codes = r.codes() if n <= 512 else None
        out[label] = (r.n_tris, r.n_active, v, codes)
        print(f"{name:12s} {label:10s} parts={r.parts} tris={r.n_tris} ms_total min {min(ms):.4f} med {sorted(ms)[6]:.4f}  wall min {min(wall):.3f}", flush=True)
    a, b = out["one piece"], out["parts"]
    same = a[0] == b[0] and a[1] == b[1]
    if a[2] is not None:
        same = same and np.array_equal(a[2].view(np.uint32), b[2].view(np.uint32)) and np.array_equal(a[3], b[3])
    print(f"{name:12s} same output: {same}", flush=True)
c.close()

# the same through the captured graph (what a frame loop replays): one sweep in flight at a time
print("--- captured graph, one replay at a time", flush=True)
for name, eq, n, iso in CASES:
    step = 2.0 / n
    for label, extra in (("one piece", mc.FLAG_NO_PARTS), ("parts", 0)):
        c = mc.Context(0)
        flags = mc.FLAG_NORMALS | mc.FLAG_NO_INTERP | mc.FLAG_NO_TIMING | extra
        c.graph_build(eq, step, iso=iso, flags=flags)
        for _ in range(5):
            r = c.graph_replay(iso)
        best = 1e9
        for rep in range(5):
            t0 = time.perf_counter()
            for _ in range(20):
                c.graph_replay_async(iso)
            r = c.graph_wait()
            best = min(best, (time.perf_counter() - t0) / 20 * 1e3)
        v = r.vertices() if n <= 512 else None
        print(f"{name:12s} {label:10s} parts={r.parts} tris={r.n_tris} ms per replay (20 back to back) {best:.4f}", flush=True)
        if label == "one piece":
            ref = (r.n_tris, v)
        else:
            ok = ref[0] == r.n_tris and (v is None or np.array_equal(ref[1].view(np.uint32), v.view(np.uint32)))
            print(f"{name:12s} same output: {ok}", flush=True)
        c.close()
