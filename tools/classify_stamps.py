#!/usr/bin/env python3
"""Developer probe (GPU box, developer build): where mc_classify's waves spend their time, tile by tile -- wall stamps and
phase cycle counts patched in by tools/probes/classify_stamps.patch -- and how the waves fill the chip over time.
    python tools/classify_stamps.py [--workload sphere|torus|goursat|gyroid] [--grid-res N] [--iso V]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MC_AMD_DEV_LIB", "1")
os.environ.setdefault("MC_JIT_PATCH", os.path.join(os.path.dirname(os.path.abspath(__file__)), "probes", "classify_stamps.patch"))
os.environ.setdefault("MC_JIT_CACHE", "/tmp/jc_ab")
os.makedirs(os.environ["MC_JIT_CACHE"], exist_ok=True)
import mc_amd  # noqa: E402


def arg(name, default, conv=str):
    return conv(sys.argv[sys.argv.index(name) + 1]) if name in sys.argv else default


wl = arg("--workload", "sphere")
n = arg("--grid-res", 1024 if wl in ("sphere", "gyroid") else 512, int)
iso = arg("--iso", -0.4 if wl == "goursat" else 0.0, float)
eq, scale = "x^2+y^2+z^2-1", (1.0,) * 3
if wl == "torus":
    eq = "(x^2+y^2+z^2+(1/3)^2-(1/5)^2)^2-4*((1/2)*x-(2.36/6)*(1/5))^2-4*(1/3)^2*y^2"
elif wl == "goursat":
    eq = "(x^2)^2+(y^2)^2+(z^2)^2-(x^2+y^2+z^2)"
elif wl == "gyroid":
    mc_amd.set_extensions(1)
    eq, scale = "sin(x)*cos(y)+sin(y)*cos(z)+sin(z)*cos(x)", (12.566371,) * 3
ctx = mc_amd.Context(0)
step = float(np.float32(2.0) / np.float32(n))
L = mc_amd.lib()
L.mc_dev_read_symbol.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_size_t, C.c_int]
buf = np.zeros(8 * 131072, np.uint64)
for _ in range(3):
    r = ctx.march(eq, step, iso, scale, flags=mc_amd.FLAG_NORMALS)
    assert L.mc_dev_read_symbol(ctx._h, b"mc_dbgw", buf.ctypes.data, buf.nbytes, 1) == 0
d = buf.reshape(-1, 8)
tiles = np.nonzero(d[:, 1] > 0)[0]
d = d[tiles]
start, end = d[:, 0].astype(np.int64), d[:, 1].astype(np.int64)
t0 = start.min()
dur = (end - start) / 100.0
tail = (d[:, 6] >> 31) & 1 == 1
npend = (d[:, 6] & 0xFF).astype(np.int64)
entries = ((d[:, 6] >> 8) & 0x7FFFFF).astype(np.int64)
entries[tail] = npend[tail]
span = (end.max() - t0) / 100.0
print(f"{wl} {n}: classify {r.ms_classify * 1e3:.1f} us (with the probe); tiles {len(d)} ({int(tail.sum())} tail); span of the stamps {span:.1f} us; "
      f"sum of wave lifetimes / span = {dur.sum() / span:.0f} waves alive on average")
ph = d[:, 2:6].astype(np.float64)


def stats(name, m):
    if not m.any():
        return
    print(f"  {name:34s} n={int(m.sum()):6d}  life us mean {dur[m].mean():6.2f} p50 {np.percentile(dur[m], 50):6.2f} p90 {np.percentile(dur[m], 90):6.2f} "
          f"max {dur[m].max():6.2f} | cycles: prologue {ph[m, 0].mean():6.0f} walk {ph[m, 1].mean():6.0f} back-end {ph[m, 2].mean():6.0f} "
          f"end {ph[m, 3].mean():6.0f} | pending rows {npend[m].mean():5.1f} entries {entries[m].mean():6.1f}")


main = ~tail
stats("tail tiles", tail)
stats("main tiles, no pending row", main & (npend == 0))
stats("main tiles, pending rows, 0 entries", main & (npend > 0) & (entries == 0))
stats("main tiles, 1..64 entries", main & (entries > 0) & (entries <= 64))
stats("main tiles, 65..256 entries", main & (entries > 64) & (entries <= 256))
stats("main tiles, > 256 entries", main & (entries > 256))
tot = dur.sum()
for name, m in (("tail", tail), ("main, no pending", main & (npend == 0)), ("main, pending, 0 entries", main & (npend > 0) & (entries == 0)),
                ("main with entries", main & (entries > 0))):
    print(f"  share of wave time: {name:26s} {100 * dur[m].sum() / tot:5.1f} %")
mm = main & (npend > 0)
if mm.any():
    print(f"  walk cycles per pending row (main tiles with pending rows): {ph[mm, 1].sum() / npend[mm].sum():.0f}; "
          f"back-end cycles per entry: {ph[main & (entries > 0), 2].sum() / max(entries[main].sum(), 1):.1f}; "
          f"entries per pending row: {entries[mm].sum() / npend[mm].sum():.2f}")
w = 5.0
edges = np.arange(0, span + w, w)
alive = [int((((start - t0) / 100.0) < b + w).sum() - (((end - t0) / 100.0) < b).sum()) for b in edges[:-1]]
print(f"  waves alive per {w:.0f} us:", alive)
started = np.histogram((start - t0) / 100.0, bins=edges)[0]
print(f"  waves started per {w:.0f} us:", started.tolist())
print(f"  waves started in the first 2 us (the initial fill of the chip): {int(((start - t0) < 200).sum())}")
late = np.argsort(end)[-8:]
print("  last waves to end: " + ", ".join(f"tile {int(tiles[i])} ({'tail' if tail[i] else 'main'}, {int(entries[i])} entries, start {(start[i] - t0) / 100.0:.1f} "
                                        f"life {dur[i]:.1f})" for i in late))
fl_cyc, fl_n = (d[:, 7] & 0xFFFFFF).astype(np.float64), ((d[:, 7] >> 24) & 0xFF).astype(np.float64)
st_cyc, nchunk = ((d[:, 7] >> 32) & 0xFFFFFF).astype(np.float64), ((d[:, 7] >> 56) & 0xFF).astype(np.float64)
for name, m in (("main tiles with entries", main & (entries > 0)), ("main tiles, > 256 entries", main & (entries > 256))):
    if m.any():
        be = ph[m, 2].sum()
        print(f"  back-end of {name}: {nchunk[m].mean():.1f} chunks, {fl_n[m].mean():.2f} flushes per tile; of its cycles {100 * fl_cyc[m].sum() / be:.0f} % in "
              f"flushes ({fl_cyc[m].sum() / max(fl_n[m].sum(), 1):.0f} cycles each), {100 * st_cyc[m].sum() / be:.0f} % in the row / dword stores, "
              f"{(be - fl_cyc[m].sum() - st_cyc[m].sum()) / max(nchunk[m].sum(), 1):.0f} cycles per chunk in the rest")
