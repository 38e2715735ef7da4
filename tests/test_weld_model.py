"""CPU: the closed-form welding rule the GPU uses (tests/weld_model.py, mirrored by mc_resolve in csrc/mc_kernels.hip)
against the oracle's replay of the reference's std::set welding (oracle/mc_oracle_weld.cpp; marching.cpp:599-654,
marching.h:32-55) -- vertex_list bit for bit, tri_list index for index -- and the oracle's welded counts against the
numbers SURVEY.md section 4 recorded from the unmodified reference."""
import numpy as np
import pytest

import weld_model as wm
from conftest import EQ

f32 = np.float32


def step_of(n):
    return float(f32(2.0) / f32(n))


# indexed vertex / triangle counts of the unmodified reference
REFERENCE_COUNTS = [
    ("x+y", step_of(32), 0.0, 1.0, 1122, 4290),
    (EQ["sphere"], step_of(32), 0.0, 1.0, 4758, 9548),
    (EQ["sphere"], step_of(64), 0.0, 1.0, 19230, 38492),
    (EQ["sphere"], step_of(128), 0.0, 1.0, 77094, 154220),
    (EQ["eq2"], step_of(32), 0.0, 1.0, 1188, 4612),
    (EQ["eq3"], step_of(32), 0.0, 1.0, 1237, 2436),
    (EQ["eq3"], step_of(64), 0.0, 1.0, 4835, 9596),
    (EQ["eq8"], step_of(32), 0.0, 1.0, 2820, 5632),
    (EQ["goursat"], step_of(32), -0.4, 1.0, 8628, 16912),
    (EQ["ui_default"], 0.2, 0.0, 1.1, 648, 1312),
    ("(x-0.1)*(y-0.07)-0.001", step_of(4), 0.0, 1.0, 72, 100),
    ("(x-0.1)*(y+0.07)-0.001", step_of(4), 0.0, 1.0, 72, 100),
    ("(x-0.1)*(y-0.07)*(z-0.13)-0.0001", step_of(4), 0.0, 1.0, 108, 148),
    ("(x-0.1)*(y+0.07)*(z-0.13)-0.0005", step_of(4), 0.0, 1.0, 108, 148),
]


@pytest.mark.parametrize("eq,step,iso,scale,verts,tris", REFERENCE_COUNTS)
def test_oracle_welding_reproduces_the_reference_counts(orc, eq, step, iso, scale, verts, tris):
    m = orc.march_indexed(eq, step, iso, (scale,) * 3, pow_mode=orc.POW_LIBM)
    assert (m.n_verts, m.n_tris) == (verts, tris)
    assert m.tris.max() < m.n_verts
    e = orc.march_indexed(eq, step, iso, (scale,) * 3, pow_mode=orc.POW_EXACT)   # what the GPU is compared with
    assert (e.n_verts, e.n_tris) == (verts, tris)


CASES = [
    ("x+y", step_of(32), 0.0, 1.0, (), (0, -1)),                  # every vertex sits on a lattice corner
    (EQ["sphere"], step_of(32), 0.0, 1.0, (), (0, -1)),           # six exact corner hits
    (EQ["sphere"], 0.1, 0.1, 1.0, (), (0, -1)),                   # drifting lattice, 144 corner keys
    (EQ["eq3"], step_of(32), 0.0, 1.0, (), (0, -1)),
    (EQ["goursat"], step_of(24), -0.4, 1.0, (), (0, -1)),
    ("x*y*z", step_of(16), 0.0, 1.0, (), (0, -1)),                # ambiguous cells on lattice planes
    ("(x-0.1)*(y-0.07)*(z-0.13)-0.0001", step_of(7), 0.0, 1.0, (), (0, -1)),
    (EQ["ui_default"], 0.2, 0.0, 1.1, (), (0, -1)),
    (EQ["sphere"], step_of(32), 0.0, 1.0, (("x", ">", -0.5),), (0, -1)),           # constraints skip cells: owners move
    (EQ["eq3"], step_of(32), 0.0, 1.0, (("x", ">", -0.5), ("y+z", "<=", 0.25), ("x*y", ">=", -0.1)), (0, -1)),
    ("x+y", step_of(16), 0.0, 1.0, (("z", "<", 0.3), ("x", ">=", -0.5)), (0, -1)),  # corner keys under constraints
    (EQ["sphere"], step_of(32), 0.0, 1.0, (), (11, 19)),           # a Z slab is welded on its own
    ("x+y", step_of(16), 0.0, 1.0, (), (3, 9)),
]


@pytest.mark.parametrize("eq,step,iso,scale,cons,z", CASES)
def test_closed_form_welding_equals_the_std_set(orc, eq, step, iso, scale, cons, z):
    ref = orc.march_indexed(eq, step, iso, (scale,) * 3, pow_mode=orc.POW_EXACT, constraints=cons, z_begin=z[0], z_end=z[1])
    sw = wm.Sweep(eq, step, iso, (scale,) * 3, constraints=cons, z_begin=z[0], z_end=z[1])
    v1, t1, _ = sw.weld_by_keys()
    assert v1.shape == ref.vertices.shape and np.array_equal(v1.view(np.uint32), ref.vertices.view(np.uint32))
    assert np.array_equal(t1, ref.tris)
    v2, t2 = sw.weld_closed_form()
    assert v2.shape == v1.shape and np.array_equal(v2.view(np.uint32), v1.view(np.uint32)) and np.array_equal(t2, t1)


def test_oracle_calculate_normal(orc):
    """normal.h:3-41 on the sphere: area-weighted, unit length, outward like the winding."""
    m = orc.march_indexed(EQ["sphere"], step_of(24))
    assert np.abs(np.linalg.norm(m.normals, axis=1) - 1).max() < 1e-6
    assert (np.sum(m.normals * m.vertices, axis=1) > 0.9).all()


def seam_slab_by_keys(eq, step, iso, scale, cons, zb, ze, n1):
    """What MC_FLAG_INDEXED | MC_FLAG_SEAM specifies for the slab [zb, ze) (include/mc_hip.h: mc_index_rebase), stated with
    the dictionary walk: weld the slab together with one ghost layer above and TWO below (the second one only so that the
    first ghost layer's own ownership comes out as in the whole grid; the device gets that from lattice indices instead),
    then hand out the slab's own triangles and the vertices its cells own, tri_list relative to the first own vertex
    (a vertex of the slab below: negative)."""
    lo, hi = max(zb - 2, 0), min(ze + 1, n1)
    sw = wm.Sweep(eq, step, iso, (scale,) * 3, constraints=cons, z_begin=lo, z_end=hi)
    v, t, first = sw.weld_by_keys()
    owner_layer = np.empty(len(v), np.int64)
    for _, (idx, cell, _e) in first.items():
        owner_layer[idx] = cell[2]
    # triangles in sweep order with their cell's layer
    tri_layer = []
    for cell in sorted(sw.cells, key=lambda c: (c[2], c[1], c[0])):
        tri_layer += [cell[2]] * len(sw.cells[cell]["tris"])
    tri_layer = np.array(tri_layer, np.int64)
    keep_v = (owner_layer >= zb) & (owner_layer < ze)
    keep_t = (tri_layer >= zb) & (tri_layer < ze)
    assert not len(v) or (np.diff(np.flatnonzero(keep_v)) == 1).all()      # contiguous: everything is in sweep order
    v0 = int(np.flatnonzero(keep_v)[0]) if keep_v.any() else int((owner_layer < zb).sum())
    return v[keep_v], t[keep_t].astype(np.int64) - v0


SEAM_SPEC_CASES = [
    (EQ["sphere"], 16, 0.0, 1.0, (), [0, 5, 6, 11, 17]),
    ("x+y", 12, 0.0, 1.0, (), [0, 4, 9, 13]),                                   # corner keys on every seam
    ("x*y*z", 10, 0.0, 1.0, (), [0, 5, 6, 11]),
    ("z", 6, 0.0, 1.0, (), [0, 3, 4, 7]),                                        # the surface is a seam plane
    (EQ["sphere"], 16, 0.0, 1.0, (("z", ">", 0.02), ("x", ">", -0.5)), [0, 8, 9, 17]),
    (EQ["goursat"], 16, -0.4, 1.0, (), [0, 6, 12, 17]),
]


@pytest.mark.parametrize("eq,n,iso,scale,cons,bounds", SEAM_SPEC_CASES)
def test_seam_rule_slabs_concatenate_to_the_std_set_of_the_whole_grid(orc, eq, n, iso, scale, cons, bounds):
    """The rule behind MC_FLAG_SEAM, on the CPU: slabs welded with ghost layers, cut back and re-based by the running vertex
    count concatenate to the reference's std::set welding of the WHOLE grid (oracle replay), bit for bit."""
    step = step_of(n)
    ref = orc.march_indexed(eq, step, iso, (scale,) * 3, pow_mode=orc.POW_EXACT, constraints=cons)
    n1 = bounds[-1]
    vs, ts, off = [], [], 0
    for zb, ze in zip(bounds[:-1], bounds[1:]):
        v, t = seam_slab_by_keys(eq, step, iso, scale, cons, zb, ze, n1)
        vs.append(v)
        ts.append(t + off)                      # mc_index_rebase(offset)
        off += len(v)
    V, T = np.concatenate(vs), np.concatenate(ts)
    assert V.shape == ref.vertices.shape and np.array_equal(V.view(np.uint32), ref.vertices.view(np.uint32))
    assert np.array_equal(T.astype(np.uint32), ref.tris)
