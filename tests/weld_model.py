"""Executable specification of the DEVICE welding rule (MC_FLAG_INDEXED, csrc/mc_kernels.hip: mc_resolve), in Python.

The reference welds the sweep's vertices through a std::set with a 1e-6 tolerance comparator (marching.cpp:599-654,
marching.h:32-55): the first point inserted wins, later points within the tolerance get its index.  A GPU cannot replay
a red-black tree, so the device uses a closed form that gives the same mesh whenever the tolerance classes are what one
expects them to be -- points meet only on a shared lattice edge or at a lattice corner:

  key(vertex)   = the lattice CORNER it sits on when its +axis-direction intersection point is closer than 1e-6 to an
                  end of its lattice edge, else the lattice EDGE itself;
  owner(key)    = the first cell of the sweep (z, y, x) that produces a vertex with this key, and inside that cell the
                  lowest-numbered such edge: the vertex keeps the position THAT cell computes (first inserted wins);
  index(vertex) = number of keys owned by earlier cells + rank of the key among the owner's own edges.

`weld_by_keys` is the rule as a dictionary walk (obviously "first seen wins"); `resolve` is the closed form the kernel
uses (no dictionary: the owner follows from lattice indices, and around a snapped corner from the 7 samples of f there);
tests/test_weld_model.py checks one against the other and both against the oracle's std::set replay.

Test infrastructure: imports the oracle."""
import ctypes as C
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "oracle"))
import pyoracle as orc  # noqa: E402

f32 = np.float32
EDGE_CORNER = [(0, 1), (1, 2), (2, 3), (3, 0), (4, 5), (5, 6), (6, 7), (7, 4), (0, 4), (1, 5), (2, 6), (3, 7)]  # marching_lookup.h:10-23
CORNER_OFF = [(0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 1, 0), (0, 0, 1), (1, 0, 1), (1, 1, 1), (0, 1, 1)]        # marching.cpp:471-472
# edge number of a lattice edge inside a cell: EDGE_OF[axis][(offset of its lower end on the two other axes)]
EDGE_OF = {0: {(0, 0): 0, (1, 0): 2, (0, 1): 4, (1, 1): 6},       # x edges, key (dy, dz)
           1: {(1, 0): 1, (0, 0): 3, (1, 1): 5, (0, 1): 7},       # y edges, key (dx, dz)
           2: {(0, 0): 8, (1, 0): 9, (1, 1): 10, (0, 1): 11}}     # z edges, key (dx, dy)


class _Step(C.Structure):
    _fields_ = [("skipped", C.c_int), ("code", C.c_int), ("row", C.c_int), ("val", C.c_float * 8), ("n_points", C.c_int),
                ("edge", C.c_int * 12), ("point", (C.c_float * 3) * 12), ("n_tris", C.c_int), ("tri_vlist", C.c_int * 15)]


def interp(iso, xs, xe, vs, ve):
    """marching.cpp:437-446 in float32."""
    with np.errstate(all="ignore"):
        v = f32(f32(f32(iso - vs) / f32(ve - vs)) * f32(xe - xs))
        if np.isinf(v) or np.isnan(v):
            return f32(np.float64(xs) + 0.5 * np.float64(f32(xe - xs)))
        return f32(xs + v)


class Sweep:
    """Per-cell results of the oracle's calculate_step for a whole grid (or Z slab)."""

    def __init__(self, eq, step, iso=0.0, scale=(1.0, 1.0, 1.0), constraints=(), z_begin=0, z_end=-1, pow_mode=orc.POW_EXACT):
        L = orc.lib()
        L.orc_step_begin.restype = C.c_void_p
        L.orc_step_begin.argtypes = [C.c_char_p, C.c_float, C.c_float, C.POINTER(C.c_float), C.c_int, C.POINTER(orc._Constraint), C.c_int]
        L.orc_step_cell.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(_Step)]
        L.orc_step_n1.argtypes = [C.c_void_p]
        L.orc_step_end.argtypes = [C.c_void_p]
        cons = (orc._Constraint * max(len(constraints), 1))()
        for i, (lhs, op, rhs) in enumerate(constraints):
            cons[i] = orc._Constraint(lhs.encode(), orc.CMP[op], float(rhs))
        h = L.orc_step_begin(eq.encode(), C.c_float(step), C.c_float(iso), (C.c_float * 3)(*scale), pow_mode, cons, len(constraints))
        assert h, eq
        self.n1 = n1 = L.orc_step_n1(h)
        self.iso = f32(iso)
        self.ax = orc.axis_coords(step)
        self.zb = max(z_begin, 0)
        self.ze = n1 if z_end < 0 or z_end > n1 else z_end
        self.cells = {}      # (ix,iy,iz) -> dict(code,row,val,edges{e: pos},tris[(e,e,e)])   active cells only
        self.skipped = set()
        self.vals = {}       # lattice sample -> f (from the corner values of unskipped cells)
        s = _Step()
        for iz in range(self.zb, self.ze):
            for iy in range(n1):
                for ix in range(n1):
                    assert L.orc_step_cell(h, ix, iy, iz, C.byref(s)) == 0
                    if s.skipped:
                        self.skipped.add((ix, iy, iz))
                        continue
                    for v in range(8):
                        o = CORNER_OFF[v]
                        self.vals[(ix + o[0], iy + o[1], iz + o[2])] = f32(s.val[v])
                    if s.code in (0, 255):
                        continue
                    edges = {s.edge[i]: tuple(f32(s.point[i][k]) for k in range(3)) for i in range(s.n_points)}
                    tris = [tuple(s.edge[s.tri_vlist[3 * t + k]] for k in range(3)) for t in range(s.n_tris)]
                    self.cells[(ix, iy, iz)] = dict(code=s.code, row=s.row, edges=edges, tris=tris,
                                                    val=[f32(s.val[v]) for v in range(8)])
        L.orc_step_end(h)

    # ---- geometry shared by both formulations -------------------------------------------------------------------
    def lattice_edge(self, cell, e):
        """(axis, lower end P) of edge e of `cell`."""
        v1, v2 = EDGE_CORNER[e]
        o1, o2 = CORNER_OFF[v1], CORNER_OFF[v2]
        axis = [a for a in range(3) if o1[a] != o2[a]][0]
        lo = o1 if o1[axis] == 0 else o2
        return axis, (cell[0] + lo[0], cell[1] + lo[1], cell[2] + lo[2])

    def snap(self, axis, P, vlo, vhi):
        """None, or the lattice corner the +axis intersection point of the edge (P, P + e_axis) sits on."""
        clo, chi = self.ax[P[axis]], self.ax[P[axis] + 1]
        p = interp(self.iso, clo, chi, vlo, vhi)
        if abs(np.float64(f32(p - clo))) < 0.000001:   # marching.h:38-40 close_enough: float difference, double compare
            return P
        if abs(np.float64(f32(p - chi))) < 0.000001:
            Q = list(P)
            Q[axis] += 1
            return tuple(Q)
        return None

    def key(self, cell, e):
        axis, P = self.lattice_edge(cell, e)
        v1, v2 = EDGE_CORNER[e]
        val = self.cells[cell]["val"]
        lo_is_v1 = CORNER_OFF[v1][axis] == 0
        vlo, vhi = (val[v1], val[v2]) if lo_is_v1 else (val[v2], val[v1])
        c = self.snap(axis, P, vlo, vhi)
        return ("c", c) if c is not None else ("e", axis, P)

    def in_grid(self, q):
        return 0 <= q[0] < self.n1 and 0 <= q[1] < self.n1 and self.zb <= q[2] < self.ze

    def active(self, q):
        return self.in_grid(q) and q not in self.skipped

    # ---- formulation 1: dictionary walk -------------------------------------------------------------------------
    def weld_by_keys(self):
        first = {}          # key -> (index, owner cell, owner edge)
        verts, tris = [], []
        for cell in sorted(self.cells, key=lambda c: (c[2], c[1], c[0])):
            d = self.cells[cell]
            vi = {}
            for e in sorted(d["edges"]):
                pos = d["edges"][e]
                if np.isnan(pos[0]):        # marching.cpp:611-613
                    vi[e] = 0xFFFFFFFF
                    continue
                k = self.key(cell, e)
                if k not in first:
                    first[k] = (len(verts), cell, e)
                    verts.append(pos)
                vi[e] = first[k][0]
            for t in d["tris"]:
                tris.append(tuple(vi[e] for e in t))
        return np.array(verts, f32).reshape(-1, 3), np.array(tris, np.uint32).reshape(-1, 3), first

    # ---- formulation 2: closed form (what the kernel does) ------------------------------------------------------
    def f_at(self, S):
        return self.vals.get(S)

    def resolve(self, cell, e):
        """(owner cell, owner edge) of the vertex on edge e of `cell`, from lattice indices and local samples only."""
        k = self.key(cell, e)
        if k[0] == "e":
            _, axis, P = k
            others = [a for a in range(3) if a != axis]
            cands = []
            for d1 in (0, 1):
                for d0 in (0, 1):
                    q = list(P)
                    q[others[0]] -= d0
                    q[others[1]] -= d1
                    cands.append((tuple(q), (d0, d1)))
            cands.sort(key=lambda t: (t[0][2], t[0][1], t[0][0]))
            for q, dd in cands:
                if self.active(q):
                    return q, EDGE_OF[axis][dd]
            raise AssertionError("no owner")
        Cn = k[1]
        fc = self.f_at(Cn)
        S = {}   # (axis, d_a) -> the incident edge inside cells with offset d_a on that axis is crossed and snaps to Cn
        for a in range(3):
            for da in (0, 1):           # da = 0: the edge Cn -> Cn + e_a; da = 1: the edge Cn - e_a -> Cn
                N = list(Cn)
                N[a] += 1 if da == 0 else -1
                fn = self.f_at(tuple(N))
                ok = False
                if fn is not None and (fc > self.iso) != (fn > self.iso):
                    P = Cn if da == 0 else tuple(N)
                    vlo, vhi = (fc, fn) if da == 0 else (fn, fc)
                    ok = self.snap(a, P, vlo, vhi) == Cn
                S[(a, da)] = ok
        cands = []
        for dz in (1, 0):
            for dy in (1, 0):
                for dx in (1, 0):
                    cands.append(((Cn[0] - dx, Cn[1] - dy, Cn[2] - dz), (dx, dy, dz)))   # already in sweep order
        for q, d in cands:
            if not self.active(q) or q not in self.cells:
                continue
            mine = []
            for a in range(3):
                if S[(a, d[a])]:
                    others = [b for b in range(3) if b != a]
                    # lower end of that lattice edge, as an offset inside cell q
                    lo = list(d)
                    lo[a] = 0          # the edge spans the cell on axis a
                    mine.append(EDGE_OF[a][(lo[others[0]], lo[others[1]])])
            if mine:
                return q, min(mine)
        raise AssertionError("no owner for corner key")

    def weld_closed_form(self):
        order = sorted(self.cells, key=lambda c: (c[2], c[1], c[0]))
        own, base = {}, {}
        n = 0
        verts = []
        for cell in order:
            d = self.cells[cell]
            m = 0
            for e in sorted(d["edges"]):
                if not np.isnan(d["edges"][e][0]) and self.resolve(cell, e) == (cell, e):
                    m |= 1 << e
                    verts.append(d["edges"][e])
            own[cell], base[cell] = m, n
            n += bin(m).count("1")
        tris = []
        for cell in order:
            d = self.cells[cell]
            for t in d["tris"]:
                idx = []
                for e in t:
                    if np.isnan(d["edges"][e][0]):
                        idx.append(0xFFFFFFFF)
                        continue
                    q, eo = self.resolve(cell, e)
                    idx.append(base[q] + bin(own[q] & ((1 << eo) - 1)).count("1"))
                tris.append(tuple(idx))
        return np.array(verts, f32).reshape(-1, 3), np.array(tris, np.uint32).reshape(-1, 3)
