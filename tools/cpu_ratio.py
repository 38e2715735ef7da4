#!/usr/bin/env python3
"""Port-vs-reference CPU speed, single thread, on the machine BASELINE.md's reference timings were taken on (this build
container: Intel Xeon @ 2.1 GHz, 8 vCPU).  The reference itself cannot be built here without a stand-in windows.h (DESIGN.md
section 5), so its side of the ratio is BASELINE.md section 2's table (recorded by the survey from the unmodified sources);
the port's side -- oracle/mc_oracle.c, libm powf, ONE thread, the same equations and grid sizes -- is measured by this
script.  Writes profiles/r03_cpu_port_vs_reference.json; bench.py carries the ratio in cpu_baseline.ratio_to_reference so
that the GPU box's port timing can be related to the true reference.

    python tools/cpu_ratio.py            # ~1 minute, CPU only (run it on an otherwise idle container)
"""
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "oracle"))
import pyoracle as orc  # noqa: E402

EQ3 = "(x^2+y^2+z^2+(1/3)^2-(1/5)^2)^2-4*((1/2)*x-(2.36/6)*(1/5))^2-4*(1/3)^2*y^2"
# BASELINE.md section 2 (reference proper: g++ 11.4 -O2, glibc 2.35, one thread): equation, N, cells, wall seconds
REFERENCE = [("x^2+y^2+z^2-1", 64, 274625, 0.704), ("x^2+y^2+z^2-1", 128, 2146689, 5.14), ("x^2+y^2+z^2-1", 256, 16974593, 42.5),
             (EQ3, 64, 274625, 3.29), (EQ3, 128, 2146689, 27.0), ("x+y", 32, 35937, 0.013)]

rows = []
for eq, n, cells, ref_s in REFERENCE:
    step = float(np.float32(2.0) / np.float32(n))
    best = None
    for _ in range(3 if cells < 5e6 else 1):
        t0 = time.perf_counter()
        m = orc.march(eq, step, 0.0, pow_mode=orc.POW_LIBM, want=orc.WANT_CODES | orc.WANT_SOUP, nthreads=1)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    assert m.n_cells == cells
    rows.append({"equation": eq, "grid_res": n, "cells": cells, "reference_s": ref_s, "port_s": round(best, 4),
                 "reference_us_per_voxel": round(ref_s / cells * 1e6, 4), "port_us_per_voxel": round(best / cells * 1e6, 4),
                 "port_over_reference_speed": round(ref_s / best, 2)})
    print(rows[-1], flush=True)
sph = [r for r in rows if r["equation"] == "x^2+y^2+z^2-1"]
ratio = float(np.exp(np.mean([np.log(r["port_over_reference_speed"]) for r in sph])))
out = {"script": "tools/cpu_ratio.py", "machine": "build container: Intel Xeon @ 2.1 GHz (8 vCPU), gcc 11.4 -O2, glibc 2.35, one thread",
       "reference_timings": "BASELINE.md section 2 (survey, unmodified reference sources)",
       "port": "oracle/mc_oracle.c, ORC_POW_LIBM, codes + soup, one thread",
       "sphere_port_over_reference_speed": round(ratio, 2), "rows": rows}
(ROOT / "profiles" / "r03_cpu_port_vs_reference.json").write_text(json.dumps(out, indent=1) + "\n")
print("sphere: the port is", round(ratio, 2), "x the reference's single-thread speed")
