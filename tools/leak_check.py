#!/usr/bin/env python3
"""Repeated context life cycles: device memory and host RSS must not creep.
   python tools/leak_check.py [rounds]"""
import os, sys, resource
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MC_COLD_START", "jit")
import ctypes as C
import mc_amd as mc

hip = C.CDLL("libamdhip64.so")
def free_bytes():
    f, t = C.c_size_t(), C.c_size_t()
    assert hip.hipMemGetInfo(C.byref(f), C.byref(t)) == 0
    return f.value
def rss_mb():
    with open("/proc/self/statm") as fh:
        return int(fh.read().split()[1]) * os.sysconf("SC_PAGE_SIZE") / 2**20

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 60
EQS = ["x^2+y^2+z^2-1", "(x^2)^2+(y^2)^2+(z^2)^2-(x^2+y^2+z^2)", "x+y"]
c0 = mc.Context(0); c0.march(EQS[0], 2.0 / 32); c0.close()   # runtime warm-up (module caches, pools)
base_free, base_rss = None, None
for i in range(rounds):
    c = mc.Context(0)
    eq = EQS[i % 3]
    iso = -0.3 if i % 3 == 1 else 0.0
    c.march(eq, 2.0 / 64, iso, flags=mc.FLAG_NORMALS)
    c.march(eq, 2.0 / 96, iso, flags=mc.FLAG_NORMALS | mc.FLAG_INDEXED)
    c.set_seed(0.0, 0.0, 1.0 if i % 3 == 0 else 0.0)
    if i % 3 == 0:
        c.seed_mode(True)
        c.march(eq, 2.0 / 64, iso)
        c.seed_mode(False)
    c.graph_build(eq, 2.0 / 64, iso=iso, flags=mc.FLAG_NORMALS | (mc.FLAG_INDEXED if i % 2 else 0))
    for k in range(3):
        c.graph_replay(iso + 0.01 * k)
    if i % 5 == 0:
        s = mc.Sharded([0, 0, 0])
        s.march(eq, 2.0 / 64, iso, flags=mc.FLAG_NORMALS)
        s.close()
    c.close()
    if i == 9:
        base_free, base_rss = free_bytes(), rss_mb()
    if i % 10 == 9:
        print(f"round {i + 1}: device free {free_bytes() / 2**20:.1f} MiB, host RSS {rss_mb():.1f} MiB", flush=True)
d_dev = (base_free - free_bytes()) / 2**20
d_rss = rss_mb() - base_rss
print(f"after {rounds} rounds: device memory grew by {d_dev:.1f} MiB, host RSS by {d_rss:.1f} MiB (since round 10)")
print("LEAK_OK" if d_dev < 64 and d_rss < 200 else "LEAK_SUSPECT")
