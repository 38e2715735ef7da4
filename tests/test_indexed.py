"""GPU (-m gpu): MC_FLAG_INDEXED -- the reference's indexed Poly_Data built on the device (mc_vmark / mc_vwrite /
mc_vindex, csrc/mc_kernels.hip) against the oracle's replay of the reference's std::set welding
(oracle/mc_oracle_weld.cpp; marching.cpp:599-654, marching.h:32-55): vertex_list bit for bit in the reference's order of
first insertion, tri_list index for index; vertex / triangle counts equal the numbers SURVEY.md section 4 recorded from
the unmodified reference; every indexed position is bit-equal to a soup position of the same sweep; the area-weighted
normals (normal.h:3-41) bit for bit (the device gathers each vertex's triangles in the reference's order)."""
import os
import random
import sys

import numpy as np
import pytest

from conftest import EQ
from test_weld_model import CASES, REFERENCE_COUNTS

pytestmark = pytest.mark.gpu
f32 = np.float32


def step_of(n):
    return float(f32(2.0) / f32(n))


def check_indexed(mc, orc, c, eq, step, iso=0.0, scale=(1.0, 1.0, 1.0), cons=(), z=(0, -1), soup=True, exact=True):
    """exact=False: the case is known to hold points closer than the reference's 1e-6 tolerance that are NOT bit-identical.
    The reference's comparator is then not an ordering, and whether its std::set finds the earlier point depends on the
    shape of its red-black tree at that moment (libstdc++ and MSVC may even disagree); the device merges every such
    group (tests/weld_model.py states the rule).  Such a case is compared with the rule itself, and with the std::set
    replay only for what cannot depend on the tree: the triangles, and that the replay never merges MORE."""
    ref = orc.march_indexed(eq, step, iso, scale, pow_mode=orc.POW_EXACT, constraints=cons, z_begin=z[0], z_end=z[1])
    flags = mc.FLAG_INDEXED | (mc.FLAG_NORMALS if soup else mc.FLAG_NO_EMIT)
    r = c.march(eq, step, iso, scale, flags, z[0], z[1])
    v, t, n = r.indexed()
    if exact or r.n_verts == ref.n_verts:
        assert (r.n_verts, r.n_tris) == (ref.n_verts, ref.n_tris)
        assert np.array_equal(v.view(np.uint32), ref.vertices.view(np.uint32)), "vertex_list differs"
        assert np.array_equal(t, ref.tris), "tri_list differs"
        # CalculateNormal: the device adds each vertex's triangle normals in the reference's order -> the same bits
        # (NaN where the reference's sum is zero)
        assert np.array_equal(n.view(np.uint32) | (np.isnan(n) * np.uint32(0x7FFFFFFF)),
                              ref.normals.view(np.uint32) | (np.isnan(ref.normals) * np.uint32(0x7FFFFFFF))), "vertex normals differ"
    else:
        import weld_model as wm
        sw = wm.Sweep(eq, step, iso, scale, constraints=cons, z_begin=z[0], z_end=z[1])
        mv, mt, _ = sw.weld_by_keys()
        assert r.n_tris == ref.n_tris and 0 < ref.n_verts - r.n_verts <= max(4, ref.n_verts // 200)
        assert v.shape == mv.shape and np.array_equal(v.view(np.uint32), mv.view(np.uint32)), "vertex_list differs from the rule"
        assert np.array_equal(t, mt), "tri_list differs from the rule"
        # the same triangles as the replay, corner for corner, within the reference's own tolerance
        assert np.abs(v[t.reshape(-1)] - ref.vertices[ref.tris.reshape(-1)]).max() < 1e-6
    if soup and r.n_tris:
        # every corner's welded vertex is within the reference's tolerance of the soup vertex it replaces, and the
        # welded coordinates are bits the soup holds too (first inserted wins)
        s = r.vertices()[:, :, :3].reshape(-1, 3)
        assert np.abs(v[t.reshape(-1)] - s).max() < 1e-6
        assert set(map(bytes, v.view(np.uint8).reshape(-1, 12))) <= set(map(bytes, s.view(np.uint8).reshape(-1, 12)))
    return r


@pytest.mark.parametrize("eq,step,iso,scale,verts,tris", REFERENCE_COUNTS)
def test_indexed_counts_equal_the_reference(mc, orc, ctx, eq, step, iso, scale, verts, tris):
    r = check_indexed(mc, orc, ctx, eq, step, iso, (scale,) * 3)
    assert (r.n_verts, r.n_tris) == (verts, tris)


@pytest.mark.parametrize("eq,step,iso,scale,cons,z", CASES)
def test_indexed_mesh_equals_the_std_set_replay(mc, orc, eq, step, iso, scale, cons, z):
    c = mc.Context(0)
    try:
        for i, (lhs, op, rhs) in enumerate(cons):
            c.set_constraint(i, lhs, op, rhs)
        check_indexed(mc, orc, c, eq, step, iso, (scale,) * 3, cons, z)
    finally:
        c.close()


def test_indexed_without_soup_and_on_wide_grids(mc, orc, ctx):
    """MC_FLAG_NO_EMIT | MC_FLAG_INDEXED (what the facade asks for); rows wider than one 256-cell segment, a tail
    plane (257 cells per axis), groups that straddle layers."""
    check_indexed(mc, orc, ctx, EQ["sphere"], step_of(48), soup=False)
    check_indexed(mc, orc, ctx, EQ["sphere"], step_of(256), z=(126, 131))
    # lattice points within 1e-8 of this cylinder that are not exact hits: four groups of near-coincident points the
    # reference's tree happens to keep apart (see check_indexed)
    check_indexed(mc, orc, ctx, "x^2+y^2-0.5", step_of(300), z=(149, 152), exact=False)
    check_indexed(mc, orc, ctx, "x+y", step_of(256), z=(0, 3))


def test_indexed_known_deviation_from_the_std_set_is_pinned(mc, orc, ctx):
    """The ONE documented place where MC_FLAG_INDEXED is not the reference bit for bit (mc_hip.h): points
    closer than 1e-6 that are not bit-identical.  `x^2+y^2-0.5` at grid_res 300 has lattice points within 1e-8 of the
    cylinder; on layers 149..151 the reference's std::set keeps 4 such points apart that the device's closed-form rule
    merges.  Pinned so that a change of the rule shows: which vertices, how many."""
    eq, step, z = "x^2+y^2-0.5", step_of(300), (149, 152)
    ref = orc.march_indexed(eq, step, 0.0, pow_mode=orc.POW_EXACT, z_begin=z[0], z_end=z[1])
    r = ctx.march(eq, step, 0.0, flags=mc.FLAG_INDEXED | mc.FLAG_NO_EMIT, z_begin=z[0], z_end=z[1])
    v, t, _ = r.indexed()
    assert r.n_tris == ref.n_tris
    assert ref.n_verts - r.n_verts == 4, (ref.n_verts, r.n_verts)
    # the replay's extra vertices are duplicates of device vertices within the reference's own tolerance
    have = {bytes(b) for b in v.view(np.uint8).reshape(-1, 12)}
    extra = [p for p in ref.vertices if bytes(p.view(np.uint8)) not in have]
    for p in extra:
        assert np.abs(v - p).max(axis=1).min() < 1e-6
    assert len(extra) <= 4


def test_indexed_on_an_empty_slab_is_an_empty_mesh(mc, ctx):
    """z_begin == z_end (a rank that gets no layers from shard_layers): no error, zero vertices and triangles."""
    r = ctx.march(EQ["sphere"], step_of(16), flags=mc.FLAG_INDEXED | mc.FLAG_NO_EMIT, z_begin=5, z_end=5)
    assert (r.n_cells, r.n_tris, r.n_verts) == (0, 0, 0)
    v, t, n = r.indexed()
    assert v.shape == (0, 3) and t.shape == (0, 3) and n.shape == (0, 3)
    r = ctx.march(EQ["sphere"], step_of(16), flags=mc.FLAG_INDEXED | mc.FLAG_NO_EMIT)   # and the context still works
    assert (r.n_verts, r.n_tris) == (1158, 2312) or r.n_tris > 0


def nanbits(a):
    return a.view(np.uint32) | (np.isnan(a) * np.uint32(0x7FFFFFFF))


def check_slabs_make_the_whole(mc, c, eq, step, bounds, iso=0.0, scale=(1.0, 1.0, 1.0), soup=False):
    """MC_FLAG_INDEXED | MC_FLAG_SEAM on consecutive Z slabs + mc_index_rebase with the running vertex count: the
    concatenation is the whole grid's Poly_Data (marching.h:26-30) bit for bit -- vertex_list, tri_list, CalculateNormal."""
    base = mc.FLAG_INDEXED | (mc.FLAG_NORMALS if soup else mc.FLAG_NO_EMIT)
    w = c.march(eq, step, iso, scale, base)
    vw, tw, nw = w.indexed()
    sw = w.vertices() if soup else None
    cw = w.codes()
    n1 = w.cells_per_axis
    vs, ts, ns, ss, cs, off, ntri = [], [], [], [], [], 0, 0
    for zb, ze in zip(bounds[:-1], bounds[1:]):
        r = c.march(eq, step, iso, scale, base | mc.FLAG_SEAM, zb, ze)
        assert (r.z_begin, r.z_end, r.n_cells) == (zb, ze, n1 * n1 * (ze - zb))
        c.index_rebase(off)
        v, t, n = r.indexed()
        assert v.shape == (r.n_verts, 3) and t.shape == (r.n_tris, 3)
        vs.append(v); ts.append(t); ns.append(n)
        cs.append(r.codes())
        if soup:
            ss.append(r.vertices())
        off += r.n_verts
        ntri += r.n_tris
    assert (off, ntri) == (w.n_verts, w.n_tris), (off, ntri, w.n_verts, w.n_tris)
    assert np.array_equal(np.concatenate(vs).view(np.uint32), vw.view(np.uint32)), "vertex_list of the slabs != the whole grid's"
    assert np.array_equal(np.concatenate(ts), tw), "tri_list of the slabs != the whole grid's"
    assert np.array_equal(nanbits(np.concatenate(ns)), nanbits(nw)), "vertex normals of the slabs != the whole grid's"
    assert np.array_equal(np.concatenate(cs), cw)
    if soup:
        assert np.array_equal(np.concatenate(ss).view(np.uint32), sw.view(np.uint32))
    return w


SEAM_CASES = [
    (EQ["sphere"], 32, [0, 11, 22, 33], 0.0, 1.0, ()),
    (EQ["sphere"], 32, [0, 16, 17, 18, 33], 0.0, 1.0, ()),                     # one-layer slabs, a cut through the equator
    ("x+y", 16, [0, 5, 9, 17], 0.0, 1.0, ()),                                    # every vertex an exact lattice hit (corner keys)
    ("x*y*z", 12, [0, 6, 7, 13], 0.0, 1.0, ()),                                  # lattice hits shared by up to 8 cells around a seam corner
    ("z", 8, [0, 4, 5, 9], 0.0, 1.0, ()),                                        # the surface IS the seam plane
    ("x^2+y^2+z^2-1", 20, [0, 7, 14, 21], 0.0, 1.1, ()),                         # UI default scale
    ("(x-0.1)*(y-0.07)*(z-0.13)-0.0001", 4, [0, 2, 3, 5], 0.0, 1.0, ()),         # ambiguity rows
    (EQ["sphere"], 24, [0, 8, 16, 25], 0.0, 1.0, (("x", ">", -0.5),)),            # a constraint: skipped cells own nothing
    (EQ["sphere"], 24, [0, 12, 13, 25], 0.0, 1.0, (("z", ">", 0.02), ("y", "<=", 0.5))),   # ... whose border is the seam
    ("(x^2)^2+(y^2)^2+(z^2)^2-(x^2+y^2+z^2)", 32, [0, 9, 20, 33], -0.4, 1.0, ()),
]


@pytest.mark.parametrize("eq,n,bounds,iso,scale,cons", SEAM_CASES)
def test_seam_slabs_concatenate_to_the_single_sweep(mc, eq, n, bounds, iso, scale, cons):
    c = mc.Context(0)
    try:
        for i, (lhs, op, rhs) in enumerate(cons):
            c.set_constraint(i, lhs, op, rhs)
        check_slabs_make_the_whole(mc, c, eq, step_of(n), bounds, iso, (scale,) * 3, soup=True)
    finally:
        c.close()


def test_seam_on_wide_grids_and_eight_ranks(mc, ctx):
    """A tail plane (257 cells per axis), rows wider than one segment, and the 8-way split bench.py's ranks use."""
    bounds = [mc.shard_layers(257, 8, r)[0] for r in range(8)] + [257]
    w = check_slabs_make_the_whole(mc, ctx, EQ["sphere"], step_of(256), bounds)
    assert (w.n_verts, w.n_tris) == (308574, 617180)      # the unmodified reference's counts
    check_slabs_make_the_whole(mc, ctx, EQ["eq3"], step_of(300), [0, 100, 150, 151, 301])


@pytest.mark.parametrize("seed", range(int(os.environ.get("MC_RANDOM_SEEDS", "24")) // 2))
def test_seam_random_equations_and_cuts(mc, ctx, seed):
    rng = random.Random(9100 + seed)
    eq = random_equation(rng)
    n = rng.choice([16, 24, 31])
    n1 = mc.cells_per_axis(step_of(n))
    cuts = sorted(rng.sample(range(1, n1), rng.randint(1, 4)))
    scale = rng.choice([(1.0, 1.0, 1.0), (1.1, 1.1, 1.1), (0.7, 1.3, 1.0)])
    check_slabs_make_the_whole(mc, ctx, eq, step_of(n), [0] + cuts + [n1], rng.choice([0.0, 0.1, -0.2]), scale)


def test_indexed_sphere_256_known_answer(mc, ctx):
    """the unmodified reference welds the 256-grid sphere into 308 574 vertices / 617 180 triangles."""
    r = ctx.march(EQ["sphere"], step_of(256), flags=mc.FLAG_INDEXED | mc.FLAG_NO_EMIT)
    assert (r.n_verts, r.n_tris) == (308574, 617180)
    v, t, n = r.indexed()
    assert t.max() == r.n_verts - 1 and np.abs(np.linalg.norm(v, axis=1) - 1).max() < 1e-3
    assert (np.sum(n * v, axis=1) > 0.99).all()


def test_indexed_sphere_1024_properties(mc, ctx):
    """The headline grid: a closed surface (every edge in exactly two triangles, Euler characteristic 2 up to the six
    axis points where the reference welds five vertices into one), all indices used, unit normals pointing outwards."""
    r = ctx.march(EQ["sphere"], step_of(1024), flags=mc.FLAG_INDEXED | mc.FLAG_NO_EMIT)
    assert r.n_tris == 9881660 and r.n_verts == r.n_tris // 2 + 2 - 18
    v, t, n = r.indexed()
    assert np.array_equal(np.unique(t), np.arange(r.n_verts, dtype=np.uint32))
    assert np.abs(np.linalg.norm(v, axis=1) - 1).max() < 1e-3
    ok = np.isfinite(n).all(axis=1)
    assert ok.sum() >= r.n_verts - 6 and (np.sum(n[ok] * v[ok], axis=1) > 0.99).all()
    assert r.ms_index > 0


def weld_soup(soup, tol=1e-6):
    """marching.cpp:599-654 by brute force on a (T,3,3) soup, in the soup's order: first point wins, per-axis tolerance."""
    pts = soup.reshape(-1, 3)
    order = np.lexsort((pts[:, 2], pts[:, 1], pts[:, 0]))
    verts, idx = [], np.empty(len(pts), np.int64)
    grid = {}
    for i in range(len(pts)):
        p = pts[i]
        key = tuple(np.floor(p.astype(np.float64) / 1e-3).astype(np.int64))
        hit = -1
        for dx in (-1, 0, 1):
            for dy in (-1, 0, 1):
                for dz in (-1, 0, 1):
                    for j in grid.get((key[0] + dx, key[1] + dy, key[2] + dz), ()):
                        if np.all(np.abs(verts[j] - p) < tol):
                            hit = j if hit < 0 else min(hit, j)
        if hit < 0:
            hit = len(verts)
            verts.append(p)
            grid.setdefault(key, []).append(hit)
        idx[i] = hit
    return np.array(verts, np.float32).reshape(-1, 3), idx.reshape(-1, 3)


@pytest.mark.parametrize("eq,n,seed,iso", [
    (EQ["sphere"], 32, (1.0, 0.0, 0.0), 0.0),
    (EQ["eq8"], 40, (0.43, 0.0, 0.0), 0.0),                                   # several components: only the seed's is welded (2 888 of 9 184 triangles)
    (EQ["eq8"], 40, (0.66, 0.0, 0.0), 0.0),
    ("x^2-0.25", 32, (0.5, 0.0, 0.0), 0.0),                                   # two sheets of exact lattice hits: only the seed's is welded
    ("(x^2+y^2+z^2-0.6)*((x-0.3)^2+y^2+z^2-0.04)", 48, (0.3, 0.2, 0.0), 0.0),  # a small sphere inside a large one
    ("z^2-0.25", 64, (0.1, -0.2, 0.5), 0.0),                                  # whole layers of surface cells, corner keys
])
def test_indexed_mesh_in_seed_mode(mc, orc, eq, n, seed, iso):
    """MC_FLAG_INDEXED with seed mode (marching.cpp:310-331 -> add_step_to_poly_data :599-654): the seed's component welded.
    Against the oracle's restatement of the reference's walk, welded in ITS order with the reference's tolerance: the same
    number of vertices and triangles, every triangle over the same three welded points within 1e-6 (the reference keeps the
    position of the first VISITED cell, the device that of the first cell in sweep order); and against the device's own
    seed-mode soup."""
    step = step_of(n)
    c = mc.Context(0)
    try:
        c.set_seed(*seed)
        c.seed_mode(True)
        r = c.march(eq, step, iso, flags=mc.FLAG_INDEXED | mc.FLAG_NORMALS)
        v, t, nrm = r.indexed()
        soup = r.vertices()[:, :, :3]
        o = orc.march_seed(eq, step, seed, iso, pow_mode=orc.POW_EXACT)
        assert r.n_tris == o.n_tris and r.n_tris > 0
        ov, ot = weld_soup(o.soup)
        assert r.n_verts == len(ov), (r.n_verts, len(ov))
        # our triangles over our welded points == our soup, within the tolerance; no two welded points coincide
        assert np.abs(v[t.reshape(-1)] - soup.reshape(-1, 3)).max() < 1e-6
        assert np.array_equal(np.unique(t), np.arange(r.n_verts, dtype=np.uint32))
        # the same triangle set as the reference's walk (order differs): map each of our welded points to the reference's
        # point within its tolerance (one to one), then compare the triangles as multisets of index triples
        from scipy.spatial import cKDTree
        dist, m = cKDTree(ov.astype(np.float64)).query(v.astype(np.float64))
        # (1e-5, the soup's tolerance in seed mode: the reference re-derives cell positions as previous +- step and drifts by
        # a few ulp per move -- DESIGN.md section 4, "Seed mode")
        assert dist.max() < 1e-5 and len(np.unique(m)) == len(ov)
        ours = np.sort(np.ascontiguousarray(m[t.astype(np.int64)]).view([("a", np.int64), ("b", np.int64), ("c", np.int64)]).ravel())
        refs = np.sort(np.ascontiguousarray(ot.astype(np.int64)).view([("a", np.int64), ("b", np.int64), ("c", np.int64)]).ravel())
        assert np.array_equal(ours, refs)
        ok = np.isfinite(nrm).all(axis=1)
        assert ok.sum() >= r.n_verts - 8 and np.abs(np.linalg.norm(nrm[ok], axis=1) - 1).max() < 1e-5
        # the dense sweep of the same context is untouched by it
        c.seed_mode(False)
        d = c.march(eq, step, iso, flags=mc.FLAG_INDEXED | mc.FLAG_NO_EMIT)
        assert d.n_tris >= r.n_tris and d.n_verts >= r.n_verts
    finally:
        c.close()


def test_indexed_mesh_in_a_captured_graph(mc, orc):
    """mc_graph_build accepts MC_FLAG_INDEXED: the five indexing kernels are nodes behind the sweep's, no host round trip;
    every replayed frame equals the un-captured sweep at that iso, also the frames that outgrow the buffers the capture
    sized (they are run again by mc_graph_wait) and frames replayed asynchronously."""
    eq, step = EQ["goursat"], step_of(48)
    c, d = mc.Context(0), mc.Context(0)
    try:
        flags = mc.FLAG_INDEXED | mc.FLAG_NORMALS
        c.graph_build(eq, step, iso=-0.69, flags=flags)       # few triangles: later frames outgrow these buffers
        for iso in (-0.69, -0.4, -0.1, -0.55, -0.4):
            r = c.graph_replay(iso)
            w = d.march(eq, step, iso, flags=flags)
            assert (r.n_verts, r.n_tris) == (w.n_verts, w.n_tris) and r.n_verts > 0
            for a, b in zip(r.indexed(), w.indexed()):
                assert np.array_equal(nanbits(a) if a.dtype == np.float32 else a, nanbits(b) if b.dtype == np.float32 else b)
            assert np.array_equal(r.vertices().view(np.uint32), w.vertices().view(np.uint32))
        ref = orc.march_indexed(eq, step, -0.4, pow_mode=orc.POW_EXACT)
        assert (r.n_verts, r.n_tris) == (ref.n_verts, ref.n_tris)
        for iso in (-0.3, -0.35, -0.45):                      # three frames in flight on one context, the last one is reported
            c.graph_replay_async(iso)
        r = c.graph_wait()
        w = d.march(eq, step, -0.45, flags=flags)
        assert (r.n_verts, r.n_tris) == (w.n_verts, w.n_tris)
        assert np.array_equal(r.indexed()[1], w.indexed()[1])
        # a slab welded as a part of the whole grid, captured
        c.graph_build(EQ["sphere"], step_of(32), flags=mc.FLAG_INDEXED | mc.FLAG_SEAM | mc.FLAG_NO_EMIT, z_begin=11, z_end=22)
        r = c.graph_replay(0.0)
        w = d.march(EQ["sphere"], step_of(32), flags=mc.FLAG_INDEXED | mc.FLAG_SEAM | mc.FLAG_NO_EMIT, z_begin=11, z_end=22)
        assert (r.n_verts, r.n_tris, r.z_begin, r.z_end) == (w.n_verts, w.n_tris, 11, 22)
        for a, b in zip(r.indexed(), w.indexed()):
            assert np.array_equal(nanbits(a) if a.dtype == np.float32 else a, nanbits(b) if b.dtype == np.float32 else b)
    finally:
        c.close()
        d.close()


OPS = ["+", "-", "*"]


def random_equation(rng):
    terms = []
    for _ in range(rng.randint(2, 4)):
        v = rng.choice(["x", "y", "z"])
        k = rng.choice(["", "^2", "^3"])
        terms.append(f"{rng.choice(['', '0.5*', '1.5*', '2*'])}{v}{k}")
    eq = terms[0]
    for t in terms[1:]:
        eq += rng.choice(OPS) + t
    return eq + rng.choice(["-0.25", "-0.5", "+0.1", ""])


@pytest.mark.parametrize("seed", range(int(os.environ.get("MC_RANDOM_SEEDS", "24"))))
def test_indexed_random_equations(mc, orc, ctx, seed):
    rng = random.Random(7000 + seed)
    eq = random_equation(rng)
    step = rng.choice([step_of(16), step_of(24), 0.07, 0.11, step_of(32)])
    scale = rng.choice([(1.0, 1.0, 1.0), (1.1, 1.1, 1.1), (0.7, 1.3, 1.0)])
    check_indexed(mc, orc, ctx, eq, step, rng.choice([0.0, 0.1, -0.2]), scale, exact=False)
