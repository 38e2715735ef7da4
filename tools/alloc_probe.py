#!/usr/bin/env python3
"""Developer probe (GPU box): does mc_classify's time depend on WHERE a context's buffers landed?  Several contexts in one
process sweep the same grid one after the other; per context the classify / emit times of repeated sweeps.
    python tools/alloc_probe.py [--contexts 6] [--grid-res 1024]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mc_amd  # noqa: E402


def arg(name, default, conv=str):
    return conv(sys.argv[sys.argv.index(name) + 1]) if name in sys.argv else default


n = arg("--grid-res", 1024, int)
k = arg("--contexts", 6, int)
step = float(np.float32(2.0) / np.float32(n))
eq = "x^2+y^2+z^2-1"
ctxs = []
for i in range(k):
    c = mc_amd.Context(0)
    c.march(eq, step, 0.0, flags=mc_amd.FLAG_NORMALS)
    ctxs.append(c)
for rnd in range(3):
    for i, c in enumerate(ctxs):
        cl, em = [], []
        for _ in range(8):
            r = c.march(eq, step, 0.0, flags=mc_amd.FLAG_NORMALS)
            cl.append(r.ms_classify)
            em.append(r.ms_emit)
        print(f"round {rnd} context {i}: classify min {min(cl) * 1e3:6.1f} median {np.median(cl) * 1e3:6.1f} us   emit min {min(em) * 1e3:6.1f} median {np.median(em) * 1e3:6.1f} us")
