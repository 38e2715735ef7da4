"""GPU (-m gpu): the HIP path, called through the C ABI, against the oracle and the golden vectors.

Bar: cube codes bit-exact; triangle soup bit-exact against the oracle in exact-power mode
(which is far inside the north star's 1e-5) and within 1e-5 of the libm-powf (reference)
vectors; normals within 1e-6 of the oracle's N1 definition.
"""
import os

import zlib

import numpy as np
import pytest

from conftest import EQ, GOLDEN

pytestmark = pytest.mark.gpu
f32 = np.float32
TOL_POS = 1e-5   # north star: vertex positions within 1e-5 of the reference
TOL_NRM = 1e-6


def step_of(n):
    return float(f32(2.0) / f32(n))


def u32(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def assert_same_floats(a, b, what):
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    eq = (u32(a) == u32(b)) | (np.isnan(a) & np.isnan(b))
    assert eq.all(), f"{what}: {np.count_nonzero(~eq)} of {eq.size} floats differ, max |d| = {np.nanmax(np.abs(a - b))}"


def check_against_oracle(mc, orc, ctx, eq, step, iso=0.0, scale=(1.0, 1.0, 1.0), flags=None, z=(0, -1)):
    flags = (mc.FLAG_NORMALS | mc.FLAG_KEEP_CODES) if flags is None else flags
    r = ctx.march(eq, step, iso, scale, flags, z[0], z[1])
    o = orc.march(eq, step, iso, scale, pow_mode=orc.POW_EXACT, want=7, z_begin=z[0], z_end=z[1])
    assert r.cells_per_axis == o.n1 and r.n_cells == o.n_cells
    assert np.array_equal(r.codes(), o.codes), "cube codes differ"
    assert (r.n_tris, r.n_active) == (o.n_tris, o.n_active)
    v = r.vertices()
    assert_same_floats(v[:, :, :3], o.soup, "positions")
    assert_same_floats(r.soup(), o.soup, "soup copy")
    assert np.array_equal(u32(r.soup_normals()), u32(v[:, :, 3:])), "mc_copy_soup_normals is not the normal half of mc_copy_vertices"
    if flags & mc.FLAG_NORMALS:
        d = np.abs(v[:, :, 3:] - o.normals)
        assert not d.size or np.nanmax(d) <= TOL_NRM, f"normals differ by {np.nanmax(d)}"
    # both emit kernels (mc_emit_direct, the default for cheap f; mc_emit, which shares vertices inside a chunk, the default
    # for expensive f) must write the same bytes
    # ... and so must 63-row classify tiles (small grids get shorter ones by default)
    # ... and mc_emit with its waves one per group (FLAG_BATCH) or four per group (an expensive f on a small grid)
    # (the one-wave-per-group launch is a second module per equation: compiled for BASELINE's equation_3 and for every other
    # equation besides, by a hash of its text)
    forces = [mc.FLAG_EMIT_DIRECT, mc.FLAG_EMIT_SHARED, mc.FLAG_TILE63]
    if eq == EQ["eq3"] or zlib.crc32(eq.encode()) % 2 == 0:
        forces.insert(2, mc.FLAG_EMIT_SHARED | mc.FLAG_BATCH)
    for force in forces:
        r2 = ctx.march(eq, step, iso, scale, flags | force, z[0], z[1])
        if force == mc.FLAG_TILE63:
            assert np.array_equal(r2.codes(), o.codes), "cube codes differ with 63-row tiles"
        v2 = r2.vertices()
        assert np.array_equal(u32(v2[:, :, :3]), u32(v[:, :, :3])), f"positions differ with emit flag {force}"
        if flags & mc.FLAG_NORMALS:
            ok = ~(np.isnan(v2[:, :, 3:]) & np.isnan(v[:, :, 3:]))
            assert not ok.any() or np.abs(v2[:, :, 3:] - v[:, :, 3:])[ok].max() <= 1e-6, f"normals differ with emit flag {force}"
    return r, o


# ---------------------------------------------------------------- golden vectors
@pytest.mark.parametrize("path", sorted(p for p in GOLDEN.glob("*.npz") if p.stem.split("_")[0] in ("constraint", "constraint3", "seed")),
                         ids=lambda p: p.stem)
def test_golden_constraints_and_seed(mc, path):
    """Committed vectors for the 'next' rows (oracle-generated; no reference fingerprint exists for them)."""
    g = np.load(path)
    c = mc.Context(0)
    try:
        if str(g["kind"]) == "constraint":
            for i, row in enumerate(g["constraints"]):
                lhs, op, rhs = str(row).split("|")
                c.set_constraint(i, lhs, op, float(rhs))
            r = c.march(str(g["equation"]), float(g["step"]))
            assert np.array_equal(r.codes(), g["codes"]) and r.n_tris == int(g["n_tris"])
            assert_same_floats(r.vertices()[:, :, :3], g["soup"], "positions")
        else:
            c.set_seed(*[float(v) for v in g["seed"]])
            c.seed_mode(True)
            r = c.march(str(g["equation"]), float(g["step"]))
            assert r.n_tris == int(g["n_tris"])
            _match_triangle_sets(r.vertices()[:, :, :3], g["soup"], TOL_POS)   # the reference's order is breadth-first
    finally:
        c.close()


@pytest.mark.parametrize("path", sorted(p for p in GOLDEN.glob("*.npz") if p.stem.split("_")[0] not in ("constraint", "constraint3", "seed")),
                         ids=lambda p: p.stem)
def test_golden(mc, ctx, path):
    g = np.load(path)
    r = ctx.march(str(g["equation"]), float(g["step"]), float(g["iso"]), tuple(float(s) for s in g["scale"]))
    assert r.cells_per_axis == int(g["n1"])
    assert np.array_equal(r.codes(), g["codes"])            # bit-exact cube indices (reference semantics)
    assert r.n_tris == int(g["n_tris"]) and r.n_active == int(g["n_active"])
    v = r.vertices()
    assert np.nanmax(np.abs(v[:, :, :3] - g["soup"]), initial=0) <= TOL_POS   # vs libm-powf vectors
    assert_same_floats(v[:, :, :3], g["soup_exact"], "positions vs exact-power vectors")
    assert np.nanmax(np.abs(v[:, :, 3:] - g["normals"]), initial=0) <= TOL_NRM


# ---------------------------------------------------------------- f itself
@pytest.mark.parametrize("name", sorted(EQ))
def test_eval_points_bit_exact(mc, orc, ctx, name):
    rng = np.random.default_rng(11)
    pts = rng.uniform(-1.3, 1.3, size=(4096, 3)).astype(np.float32)
    pts[:8] = [[0, 0, 0], [1, 0, 0], [-1, 1, -1], [0.5, 0.5, 0.5], [1e-20, 1, 1], [1, 1e-30, -1], [-0.0, 0.0, 1], [1, 1, 1]]
    got = ctx.eval_points(EQ[name], pts)
    want = orc.evaluate_many(EQ[name], pts, pow_mode=orc.POW_EXACT)
    assert_same_floats(got, want, "f(x,y,z)")


@pytest.mark.parametrize("eq", ["x/y/z", "x-y-z", "-x^2", "x/y*z", "1/x", "x/(y*y-z)", "x^3", "x^-2", "x^4-y^5+z^7",
                                "(x/3)/(y/7)", "2.36/6*x", "x^0+y^1", "-(x+ -(y)* -.021)", "xy/z^2", "x*1000000/y"])
def test_eval_points_operators(mc, orc, ctx, eq):
    rng = np.random.default_rng(5)
    pts = rng.uniform(-2, 2, size=(8192, 3)).astype(np.float32)
    pts[0] = [1, 0, 0]      # division by zero -> inf / nan must propagate like IEEE
    pts[1] = [0, 0, 0]
    got = ctx.eval_points(eq, pts)
    want = orc.evaluate_many(eq, pts, pow_mode=orc.POW_EXACT)
    assert_same_floats(got, want, eq)


def test_eval_points_general_pow_tolerance(mc, orc, ctx):
    """Non-literal-integer exponents: tolerance only -- parity unpinned at bit level."""
    rng = np.random.default_rng(5)
    pts = rng.uniform(0.05, 2, size=(4096, 3)).astype(np.float32)
    for eq in ("x^y", "x^.5+z", "xy/z^-.22"):
        got = ctx.eval_points(eq, pts)
        want = orc.evaluate_many(eq, pts, pow_mode=orc.POW_LIBM)
        assert np.allclose(got, want, rtol=2e-6, atol=1e-30), eq


# ---------------------------------------------------------------- sweeps vs the oracle
@pytest.mark.parametrize("name,n", [("sphere", 32), ("sphere", 64), ("eq1", 32), ("eq2", 32), ("eq3", 32), ("eq4", 24),
                                    ("eq5", 24), ("eq6", 32), ("eq7", 32), ("eq8", 32), ("goursat", 32), ("eq3", 64)])
def test_sweep_power_of_two(mc, orc, ctx, name, n):
    check_against_oracle(mc, orc, ctx, EQ[name], step_of(n))


@pytest.mark.parametrize("step", [0.5, 0.3, 0.25, 0.2, 0.1, 0.07, 0.0371])
def test_sweep_odd_steps(mc, orc, ctx, step):
    """Non power-of-two steps: the lattice is built by drifting float adds (marching.cpp:375-377)."""
    check_against_oracle(mc, orc, ctx, EQ["sphere"], float(f32(step)), iso=0.05)


@pytest.mark.parametrize("iso", [-0.7, -0.5, -0.4, -0.3, -0.1])
def test_goursat_iso_sweep(mc, orc, ctx, iso):
    r, _ = check_against_oracle(mc, orc, ctx, EQ["goursat"], step_of(32), iso=iso)
    assert r.n_tris == {-0.7: 1984, -0.5: 13024, -0.4: 16912, -0.3: 16144, -0.1: 5728}[iso]  # SURVEY 8d cfg 5


@pytest.mark.parametrize("scale", [(1.1, 1.1, 1.1), (0.5, 2.0, 1.0), (1.0, 1.3, 0.8)])
def test_scaling(mc, orc, ctx, scale):  # marching.cpp:209-224
    check_against_oracle(mc, orc, ctx, EQ["ui_default"], 0.2, scale=scale)
    check_against_oracle(mc, orc, ctx, EQ["sphere"], step_of(24), iso=0.2, scale=scale)


@pytest.mark.parametrize("eq,amb,flip", [("(x-0.1)*(y-0.07)-0.001", 5, 5), ("(x-0.1)*(y+0.07)-0.001", 5, 0),
                                         ("(x-0.1)*(y-0.07)*(z-0.13)-0.0001", 13, 7),
                                         ("(x-0.1)*(y+0.07)*(z-0.13)-0.0005", 13, 8)])
@pytest.mark.parametrize("n", [4, 7, 12])
def test_ambiguity_branch(mc, orc, ctx, eq, amb, flip, n):
    """marching.cpp:519-549: face-centre sample decides between row c and row 255-c."""
    _, o = check_against_oracle(mc, orc, ctx, eq, step_of(n))
    if n == 4:
        assert (o.n_amb, o.n_flipped) == (amb, flip)     # SURVEY.md section 4 known answers
    assert o.n_amb > 0


@pytest.mark.parametrize("eq,n", [("x*y*z", 50), ("(x-0.03)*(y-0.04)*(z-0.05)", 36), ("x*y*z", 300)])
def test_ambiguity_branch_larger(mc, orc, ctx, eq, n):
    """Saddle surfaces: >100 ambiguous cells, about half of them flipped, also across 256-cell segments."""
    z = (0, -1) if n < 100 else (148, 153)
    _, o = check_against_oracle(mc, orc, ctx, eq, step_of(n), z=z)
    assert o.n_amb >= 5 and 0 < o.n_flipped < o.n_amb


def test_wide_rows_multi_chunk(mc, orc, ctx):
    """n1 > 256 and n1 % 256 != 0: several 256-cell segments per row plus a ragged tail."""
    check_against_oracle(mc, orc, ctx, "x^2+y^2-0.5", step_of(300), z=(149, 153))      # 301 cells per row
    check_against_oracle(mc, orc, ctx, EQ["sphere"], step_of(513), iso=0.3, z=(200, 202))  # 514 = 2*256+2
    check_against_oracle(mc, orc, ctx, EQ["sphere"], step_of(255), z=(100, 102))       # exactly 256


def test_dense_surface_fills_emit_staging(mc, orc, ctx):
    """A surface that crosses nearly every cell of a row overflows the per-wave LDS list (flush path)."""
    check_against_oracle(mc, orc, ctx, "(x*37-y*29+z*31)*(x*37-y*29+z*31)-0.3", step_of(300), z=(10, 12))
    check_against_oracle(mc, orc, ctx, "y*y*900-0.5+x*0", step_of(300), z=(3, 5))


def test_tile_height_does_not_matter(mc, ctx):
    a = ctx.march(EQ["eq8"], step_of(80))
    b = ctx.march(EQ["eq8"], step_of(80), flags=mc.FLAG_NORMALS | mc.FLAG_KEEP_CODES | mc.FLAG_TILE1)
    ca, va = a.codes(), a.vertices()
    assert np.array_equal(ca, b.codes()) and np.array_equal(u32(va), u32(b.vertices()))


def test_empty_and_full_volumes(mc, orc, ctx):
    for eq, iso in (("x^2+y^2+z^2+5", 0.0), ("x^2+y^2+z^2-50", 0.0), ("x", 10.0)):
        r, o = check_against_oracle(mc, orc, ctx, eq, step_of(16), iso=iso)
        assert r.n_tris == 0 and o.n_tris == 0


def test_nan_and_inf_fields(mc, orc, ctx):
    """NaN compares false (bit clear, marching.cpp:498); inf/nan edge values take interp's midpoint branch."""
    check_against_oracle(mc, orc, ctx, "1/x+y", step_of(8))
    check_against_oracle(mc, orc, ctx, "x/(y*y)+z", step_of(8))
    check_against_oracle(mc, orc, ctx, "(x^2+y^2-.25)^.5-z", step_of(12), flags=2)  # NaN inside the cylinder


def test_z_slabs_equal_whole(mc, ctx):
    """Multi-GPU sharding invariant, on one GPU: concatenated Z slabs == the whole sweep, byte for byte."""
    eq, step = EQ["eq3"], step_of(96)
    whole = ctx.march(eq, step)
    wc, wv = whole.codes(), whole.vertices()
    n1 = whole.cells_per_axis
    codes, verts = [], []
    for rank in range(5):
        b, e = mc.shard_layers(n1, 5, rank)
        r = ctx.march(eq, step, z_begin=b, z_end=e)
        codes.append(r.codes())
        verts.append(r.vertices())
    assert np.array_equal(np.concatenate(codes), wc)
    assert np.array_equal(u32(np.concatenate(verts)), u32(wv))


def test_graph_replay_equals_march(mc, ctx):
    eq, step = EQ["goursat"], step_of(64)
    ctx.graph_build(eq, step, iso=-0.4)
    for iso in (-0.6, -0.45, -0.2, -0.4):
        g = ctx.graph_replay(iso)
        # the per-kernel HIP events are nodes of the captured graph: the replay reports real kernel times
        assert 0 < g.ms_classify < g.ms_total and 0 < g.ms_emit < g.ms_total and g.ms_total < 50
        gv = g.vertices()
        m = ctx.march(eq, step, iso, flags=mc.FLAG_NORMALS)
        assert g.n_tris == m.n_tris and np.array_equal(u32(gv), u32(m.vertices()))
        ctx.graph_build(eq, step, iso=-0.4)


def test_graph_survives_other_sweeps_on_its_context(mc, orc):
    """A captured graph holds kernel handles, launch sizes and raw device pointers.  Sweeps of OTHER equations, grids and
    scales on the same context (which re-target the axis tables and re-allocate buffers), and constraint changes, must
    not make a later replay run one equation's kernels over another's buffers: the replay re-captures its own sweep."""
    c = mc.Context(0)
    try:
        eq, step = EQ["goursat"], step_of(48)
        want = {iso: orc.march(eq, step, iso, pow_mode=orc.POW_EXACT, want=2) for iso in (-0.4, -0.2)}
        c.graph_build(eq, step, iso=-0.4)
        c.march(EQ["sphere"], step_of(200))                       # larger grid: every buffer is re-allocated
        g = c.graph_replay(-0.2)
        assert_same_floats(g.vertices()[:, :, :3], want[-0.2].soup, "after a larger sweep")
        c.march(EQ["eq3"], step_of(48), scale=(1.1, 1.1, 1.1))    # same size, other equation and axis tables
        g = c.graph_replay(-0.4)
        assert_same_floats(g.vertices()[:, :, :3], want[-0.4].soup, "after another equation")
        c.set_constraint(0, "x", ">", -0.25)                      # constraints are part of the compiled kernels
        g = c.graph_replay(-0.4)
        oc = orc.march(eq, step, -0.4, pow_mode=orc.POW_EXACT, want=2, constraints=[("x", ">", -0.25)])
        assert g.n_tris == oc.n_tris and g.n_tris < want[-0.4].n_tris
        assert_same_floats(g.vertices()[:, :, :3], oc.soup, "after a constraint change")
    finally:
        c.close()


def test_graph_async_replays(mc, ctx):
    """mc_graph_replay_async enqueues without a host round trip; mc_graph_wait reports the last replay.  Replays with
    different iso values in flight stay ordered (the parameter block is read when the upload node RUNS)."""
    eq, step = EQ["goursat"], step_of(64)
    ctx.graph_build(eq, step, iso=-0.4, flags=mc.FLAG_NORMALS | mc.FLAG_NO_TIMING)
    for _ in range(8):
        ctx.graph_replay_async(-0.4)
    ctx.graph_replay_async(-0.25)
    g = ctx.graph_wait()
    assert g.ms_total == 0 and g.ms_classify == 0      # FLAG_NO_TIMING: no event nodes in the graph
    m = ctx.march(eq, step, -0.25, flags=mc.FLAG_NORMALS)
    assert g.n_tris == m.n_tris and np.array_equal(u32(g.vertices()), u32(m.vertices()))


def test_errors(mc, ctx):
    for eq, code in (("sin(x)", mc.MC_ERR_PARSE), ("x+", mc.MC_ERR_EVAL)):
        with pytest.raises(mc.McError) as e:
            ctx.march(eq, 0.25)
        assert e.value.code == code
    with pytest.raises(mc.McError) as e:
        ctx.march("x+y", 0.6)          # marching.cpp:226: step outside [0.001, 0.5]
    assert e.value.code == mc.MC_ERR_STEP


# ---------------------------------------------------------------- larger sizes: counts + properties
def test_sphere_256_known_answer(mc, orc, ctx):
    """BASELINE config 2.  SURVEY.md section 4: the reference emits 617 180 triangles."""
    r = ctx.march(EQ["sphere"], step_of(256))
    assert (r.cells_per_axis, r.n_tris) == (257, 617180)
    o = orc.march(EQ["sphere"], step_of(256), pow_mode=orc.POW_EXACT, want=orc.WANT_CODES | orc.WANT_SOUP)
    assert np.array_equal(r.codes(), o.codes)
    assert_same_floats(r.soup(), o.soup, "soup @256")
    # ... and against the reference's own arithmetic (libm powf for ^): the same codes, positions within the north star's 1e-5
    ref = orc.march(EQ["sphere"], step_of(256), pow_mode=orc.POW_LIBM, want=orc.WANT_CODES | orc.WANT_SOUP)
    assert np.array_equal(r.codes(), ref.codes) and ref.n_tris == r.n_tris
    assert np.abs(r.soup() - ref.soup).max() <= TOL_POS


def test_torus_128_against_reference_semantics(mc, orc, ctx):
    """equation_3 (BASELINE config 3's surface) at a size the CPU oracle sweeps in seconds, against libm-powf semantics
    (what the reference executes): identical cube codes, positions within 1e-5; SURVEY section 4 count 38 388."""
    r = ctx.march(EQ["eq3"], step_of(128))
    ref = orc.march(EQ["eq3"], step_of(128), pow_mode=orc.POW_LIBM, want=orc.WANT_CODES | orc.WANT_SOUP)
    assert r.n_tris == ref.n_tris == 38388
    assert np.array_equal(r.codes(), ref.codes)
    assert np.abs(r.soup() - ref.soup).max() <= TOL_POS


def test_sphere_128_reference_count(mc, ctx):
    assert ctx.march(EQ["sphere"], step_of(128), flags=0).n_tris == 154220


def test_torus_512_properties(mc, ctx):
    """BASELINE config 3 at full size: size-independent properties (the oracle would need ~minutes)."""
    eq, step = EQ["eq3"], step_of(512)
    r = ctx.march(eq, step, flags=mc.FLAG_NORMALS)
    # 38 388 triangles @128, x4 per doubling
    assert abs(r.n_tris / (38388 * 16) - 1) < 0.02
    v = r.vertices()
    p, n = v[:, :, :3], v[:, :, 3:]
    assert np.isfinite(v).all() and p.min() >= -1 - 1e-6 and p.max() <= 1 + step + 1e-6
    assert np.abs(np.linalg.norm(n, axis=2) - 1).max() < 1e-5
    # every vertex lies on a lattice edge: two coordinates are lattice values
    lat = -1 + np.arange(514) * step
    on = np.isclose(p[..., None], lat, atol=1e-7).any(-1).sum(-1)
    assert (on >= 2).all()
    # orientation: cross(B-A, C-A) points toward +grad f
    cr = np.cross(p[:, 1] - p[:, 0], p[:, 2] - p[:, 0])
    good = np.linalg.norm(cr, axis=1) > 1e-12
    assert ((cr[good] * n[good, 0]).sum(1) > 0).mean() > 0.999
    # slabs == whole
    half = r.cells_per_axis // 2
    a = ctx.march(eq, step, flags=mc.FLAG_NORMALS, z_begin=0, z_end=half)
    na, va = a.n_tris, a.vertices()
    b = ctx.march(eq, step, flags=mc.FLAG_NORMALS, z_begin=half, z_end=-1)
    assert na + b.n_tris == r.n_tris
    assert np.array_equal(u32(np.concatenate([va, b.vertices()])), u32(v))


@pytest.mark.parametrize("name,n,iso,z", [
    ("sphere", 1024, 0.0, (511, 515)),     # the headline grid: the equator layers (longest surface rows) ...
    ("sphere", 1024, 0.0, (1, 4)),         # ... the pole, where the surface grazes whole layers ...
    ("sphere", 1024, 0.0, (1022, 1025)),   # ... and the last layers (tail plane, lattice end)
    ("eq3", 512, 0.0, (255, 259)),         # BASELINE config 3
    ("eq3", 512, 0.0, (300, 303)),
    ("goursat", 512, -0.4, (100, 103)),    # BASELINE config 5's surface
    ("goursat", 512, -0.7, (436, 439)),
    # ragged last chunks (44 / 127 / 129 cells, and a 2-cell tail plane) under the block level of the interval walk, which
    # these two staged polynomials take: boxes beyond the end of the grid, partly filled boxes
    ("eq3", 299, 0.0, (148, 151)),
    ("goursat", 300, -0.4, (60, 63)),
    ("eq3", 382, 0.0, (190, 193)),
    ("goursat", 640, -0.4, (100, 102)),
    ("goursat", 257, -0.5, (128, 131)),
])
def test_full_size_grids_thin_slabs_against_the_oracle(mc, orc, ctx, name, n, iso, z):
    """BASELINE's full-size grids, a few layers at a time (what the CPU oracle sweeps in seconds): cube codes and the
    soup bit for bit against the oracle in exact-power mode, and against the REFERENCE's semantics (libm powf): the
    same cube codes, positions within 1e-5 (north star)."""
    eq, step = EQ[name], step_of(n)
    r, o = check_against_oracle(mc, orc, ctx, eq, step, iso=iso, z=z)
    assert o.n_tris > 0
    ref = orc.march(eq, step, iso, pow_mode=orc.POW_LIBM, want=orc.WANT_CODES | orc.WANT_SOUP, z_begin=z[0], z_end=z[1])
    assert np.array_equal(r.codes(), ref.codes) and r.n_tris == ref.n_tris
    assert np.abs(r.soup() - ref.soup).max() <= TOL_POS


def test_sphere_1024_properties(mc, ctx):
    """The headline workload: 1025^3 cells.  Count scaling, closed-surface and on-sphere properties."""
    r = ctx.march(EQ["sphere"], step_of(1024), flags=mc.FLAG_NORMALS)
    assert r.cells_per_axis == 1025 and r.n_cells == 1025 ** 3
    assert abs(r.n_tris / (617180 * 16) - 1) < 0.01           # x4 per doubling from 256
    v = r.vertices()
    p, n = v[:, :, :3].reshape(-1, 3), v[:, :, 3:].reshape(-1, 3)
    rad = np.linalg.norm(p.astype(np.float64), axis=1)
    assert abs(rad - 1).max() < 2e-6                           # linear interpolation error of x^2 at h = 1/512
    assert np.abs((n * p).sum(1) / rad - 1).max() < 1e-4       # gradient normal is radial
    # every edge vertex is shared by the triangles around it: positions repeat (closed surface)
    sub = p[: 3 * 200000]
    _, counts = np.unique(sub.view([("", np.float32)] * 3), return_counts=True)
    assert counts.mean() > 3


RATIONAL = {
    "div_pos": "x^2+y^2+z^2-1/(x*x+4)",          # divisor bounded away from 0: interval walk
    "div_lin": "x/(y+3)+z*z-0.1",
    "negpow": "(x+3)^-2+y^2+z^2-0.2",
    "negpow_odd": "(y-2.5)^-3+x*x+z",
    "div_var": "x/(y*y+0.01)+z",                 # large but finite
    "div_zero": "x/y+z",                         # divisor crosses 0: sampling fallback, inf/nan samples
    "div_y": "x^2+z^2-1/(y*y+1.5)",              # expensive sub-expression of y alone: the staged enclosure (mc_f_iv_y)
    "pow_y": "x*y^5+z^2-0.3",                    # same, a power
    "pow_yz": "x+y^3*z^3*4-0.1",                 # a sub-expression of y AND z (no x): staged as a whole (round 4)
}


@pytest.mark.parametrize("name", sorted(RATIONAL))
@pytest.mark.parametrize("n", [24, 300])
def test_rational_equations(mc, orc, ctx, name, n):
    """Division and negative powers: bit-exact vs the oracle whichever walk (interval / sampling) is used."""
    z = (0, -1) if n < 100 else (147, 152)
    check_against_oracle(mc, orc, ctx, RATIONAL[name], step_of(n), z=z)


EQ_ALL = dict(EQ, **RATIONAL)


@pytest.mark.parametrize("name,n,iso", [("sphere", 96, 0.0), ("eq3", 96, 0.0), ("eq8", 80, 0.0), ("eq2", 64, 0.0),
                                        ("eq6", 64, 0.0), ("goursat", 96, -0.4), ("ui_default", 40, 0.0),
                                        ("div_pos", 64, 0.0), ("div_lin", 64, 0.0), ("negpow", 64, 0.0),
                                        ("negpow_odd", 64, 0.0), ("div_var", 64, 0.0), ("div_y", 64, 0.0), ("pow_y", 96, 0.0),
                                        ("div_y", 300, 0.0), ("pow_yz", 96, 0.0), ("pow_yz", 300, 0.0)])
def test_interval_row_culling_is_exact(mc, ctx, name, n, iso):
    """K1 proves rows / lanes uniform with interval arithmetic (mc_f_iv) and never samples them; the kernels
    compiled with the interval walk left out (MC_FLAG_NO_CULL: the sampling walk) must give byte-identical codes
    and vertices."""
    EQ = EQ_ALL
    a = ctx.march(EQ[name], step_of(n), iso)
    ca, va = a.codes(), a.vertices()
    b = ctx.march(EQ[name], step_of(n), iso, flags=mc.FLAG_NORMALS | mc.FLAG_KEEP_CODES | mc.FLAG_NO_CULL)
    assert np.array_equal(ca, b.codes())
    assert np.array_equal(u32(va), u32(b.vertices()))


# ---------------------------------------------------------------- constraints (marching.cpp:173-207, :255-280, :476)
CONSTRAINT_SETS = {
    "drawer_x_gt": [("x", ">", -0.5)],                       # the developer viewer's hotkey (marching_test_drawer.h)
    "slab": [("z", ">=", -0.25), ("z", "<=", 0.25)],
    "three": [("x", ">", -0.5), ("y+z", "<=", 0.25), ("x*y", ">=", -0.1)],
    "curved": [("x^2+y^2", "<", 0.49)],
    "nothing_left": [("x", ">", 5)],
    "everything": [("x", ">", -5)],
    "rational": [("1/x", "<", 3)],                           # not boundable: the sampling walk takes it
}


@pytest.mark.parametrize("cname", sorted(CONSTRAINT_SETS))
@pytest.mark.parametrize("name,n", [("sphere", 40), ("eq3", 64), ("eq8", 300)])
def test_constraints(mc, orc, name, n, cname):
    cons = CONSTRAINT_SETS[cname]
    eq, step = EQ[name], step_of(n)
    c = mc.Context(0)
    try:
        for i, (lhs, op, rhs) in enumerate(cons):
            c.set_constraint(i, lhs, op, rhs)
        r = c.march(eq, step)
        o = orc.march(eq, step, pow_mode=orc.POW_EXACT, want=7, constraints=cons)
        assert np.array_equal(r.codes(), o.codes), "cube codes differ"
        assert (r.n_tris, r.n_active) == (o.n_tris, o.n_active)
        assert_same_floats(r.vertices()[:, :, :3], o.soup, "positions")
        for force in (mc.FLAG_EMIT_DIRECT, mc.FLAG_EMIT_SHARED):     # both emit kernels (skipped cells break up the chunks)
            rf = c.march(eq, step, flags=mc.FLAG_NORMALS | force)
            assert_same_floats(rf.vertices()[:, :, :3], o.soup, f"positions, emit flag {force}")
        # switched off again: the unconstrained surface (use_constraint, marching.cpp:202-207)
        for i in range(len(cons)):
            c.use_constraint(i, False)
        r2 = c.march(eq, step)
        o2 = orc.march(eq, step, pow_mode=orc.POW_EXACT, want=3)
        assert r2.n_tris == o2.n_tris and np.array_equal(r2.codes(), o2.codes)
    finally:
        c.close()


def test_constraints_with_scaling_and_inside_volume(mc, orc):
    # a solid region (codes 255) cut by a constraint: skipped cells read 0, the rest stay 255
    c = mc.Context(0)
    try:
        c.set_constraint(0, "x+y", ">=", 0.1)
        eq, step, scale = "1-x^2-y^2-z^2", step_of(48), (1.3, 0.9, 1.1)
        r = c.march(eq, step, 0.0, scale)
        o = orc.march(eq, step, 0.0, scale, pow_mode=orc.POW_EXACT, want=7, constraints=[("x+y", ">=", 0.1)])
        assert np.array_equal(r.codes(), o.codes) and r.n_tris == o.n_tris
        assert (o.codes == 255).any() and (o.codes == 0).any()
        assert_same_floats(r.vertices()[:, :, :3], o.soup, "positions")
    finally:
        c.close()


def test_constraint_errors(mc):
    c = mc.Context(0)
    try:
        for args in [(3, "x", ">", 0.0), (-1, "x", ">", 0.0), (0, "x", "==", 0.0), (0, "x", "=>", 0.0), (0, "x+a", ">", 0.0)]:
            with pytest.raises(mc.McError):
                c.set_constraint(*args)
    finally:
        c.close()


def test_record_buffer_grows_on_demand(mc, orc):
    """The dense record array starts from a guess (16 records per row and layer); a surface with more active cells --
    here ten planes per axis, 28 % of all cells -- overflows it, the sweep's own count tells the host, which grows the
    buffer and sweeps again (the same path a replayed graph takes)."""
    c = mc.Context(0)
    try:
        roots = [-0.9, -0.7, -0.5, -0.3, -0.1, 0.1, 0.3, 0.5, 0.7, 0.9]
        eq = "*".join(f"({v}{'+' if r < 0 else '-'}{abs(r)})" for v in "xyz" for r in roots)
        step = step_of(96)
        r = c.march(eq, step)
        o = orc.march(eq, step, pow_mode=orc.POW_EXACT, want=3)
        assert r.n_active > 16 * r.cells_per_axis ** 2 + 4096, "the case no longer overflows the first guess"
        assert (r.n_tris, r.n_active) == (o.n_tris, o.n_active)
        assert np.array_equal(r.codes(), o.codes)
        assert_same_floats(r.vertices()[:, :, :3], o.soup, "positions")
        c.graph_build(eq, step, iso=0.5)            # fewer cells at iso 0.5 ...
        g = c.graph_replay(0.0)                     # ... than at 0: the replay outgrows nothing here (already grown)
        assert g.n_tris == o.n_tris
    finally:
        c.close()


def test_largest_grid_step_0001(mc):
    """The smallest step set_grid_step_size accepts (0.001, marching.cpp:226) = 2001 cells per axis: one 128-layer
    slab of it (the whole grid is 8e9 cells; the slab keeps the test short) against closed-form expectations."""
    c = mc.Context(0)
    try:
        step = 0.001
        n1 = mc.cells_per_axis(step)
        assert n1 == 2001
        zb, ze = 936, 1064                                   # around the equator
        r = c.march(EQ["sphere"], step, 0.0, flags=mc.FLAG_NORMALS, z_begin=zb, z_end=ze)
        assert r.n_cells == n1 * n1 * (ze - zb)
        # a band of a unit sphere of height h has area 2*pi*h; the whole sphere gives 3.0 triangles per step^2 of area
        # at every size (256..1024), a band's density depends on its orientation mix: between 2 and 4
        area = 2 * np.pi * (ze - zb) * step
        assert 2.0 < r.n_tris / (area / step ** 2) < 4.0
        v = r.vertices()[:: 1009].reshape(-1, 6)
        assert np.abs(np.linalg.norm(v[:, :3], axis=1) - 1).max() < 2e-6
        assert np.abs(np.sum(v[:, :3] * v[:, 3:], axis=1) - 1).max() < 1e-4   # normals point outwards (towards f > iso)
    finally:
        c.close()


# ---------------------------------------------------------------- seed mode (marching.cpp:42-137, :310-331)
def _match_triangle_sets(a, b, tol):
    """Every triangle of a has a partner in b within tol (max-norm over its 9 coordinates) and vice versa."""
    from scipy.spatial import cKDTree
    a, b = a.reshape(-1, 9).astype(np.float64), b.reshape(-1, 9).astype(np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    if len(a) == 0:
        return
    d, idx = cKDTree(b).query(a, k=1, p=np.inf)
    assert d.max() <= tol, f"max distance {d.max()}"
    d2, _ = cKDTree(a).query(b, k=1, p=np.inf)
    assert d2.max() <= tol, f"max distance {d2.max()} (the other way round)"
    # one-to-one, where triangles are distinct: exact lattice hits produce degenerate triangles that coincide (several copies of
    # one point), and a nearest-neighbour query cannot tell those apart -- the equal counts above and the two-sided match stand
    # for them
    distinct = np.unique(np.round(b, 5), axis=0).shape[0]
    assert len(np.unique(idx)) >= distinct - (len(b) - distinct), "not a one-to-one match"


@pytest.mark.parametrize("eq,n,seed,scale,empty", [
    (EQ["sphere"], 32, (1.0, 0.0, 0.0), (1.0, 1.0, 1.0), False),  # the whole sphere except the cells the bound check excludes
    (EQ["sphere"], 32, (0.0, 0.0, 0.0), (1.0, 1.0, 1.0), True),   # a seed cell without a crossing: nothing (meant to be empty)
    ("x^2-0.25", 32, (0.5, 0.0, 0.0), (1.0, 1.0, 1.0), False),    # two sheets: only the one the seed touches
    ("x^2-0.25", 32, (-0.52, 0.3, 0.2), (1.0, 1.0, 1.0), False),
    ("x^2-0.25", 32, (-0.5, 0.3, 0.2), (1.0, 1.0, 1.0), True),    # the crossing is in the cell next door: nothing (meant to be empty)
    # seeds that sit ON the surface (centroids of triangles of the dense sweep): round 3's seeds for these equations lay in
    # cells without a crossing and compared 0 triangles with 0
    (EQ["eq3"], 48, (-0.585, -0.126, -0.542), (1.0, 1.0, 1.0), False),   # 5 028 of the dense sweep's 5 316 triangles
    (EQ["eq8"], 40, (0.008, 0.102, 0.65), (1.0, 1.0, 1.0), False),       # two components: 6 296 ...
    (EQ["eq8"], 40, (0.181, -0.363, -0.183), (1.0, 1.0, 1.0), False),    # ... and 2 888 of 9 184
    # seed / scale picks the cell (marching.cpp:106-108): on the stretched sphere's x pole, then on its z side -- 4 784 of the
    # dense sweep's 4 788 triangles (the cell at the +y pole lies beyond the walk's bound, marching.cpp:84-86); a sphere that
    # the domain cuts open (24 576 of 24 912).  Power-of-two grids: on others the reference's walk re-derives cell positions
    # an ulp off the lattice and classifies exact lattice hits differently from its own dense sweep (seed mode)
    (EQ["sphere"], 32, (1.0, 0.0, 0.0), (2.0, 1.0, 1.5), False),
    (EQ["sphere"], 32, (0.0, 0.0, 1.0), (2.0, 1.0, 1.5), False),
    (EQ["sphere"], 64, (0.0, 0.0, 1.0), (1.25, 0.8, 1.6), False),
    (EQ["sphere"], 32, (0.5, 0.0, 0.0), (2.0, 1.0, 1.5), True),          # inside the stretched sphere: nothing (meant to be empty)
    # whole layers / rows of surface cells: a 64-segment group holds thousands of records (several LDS windows of the
    # component labelling), and the sheets are separate components
    ("z^2-0.25", 64, (0.1, -0.2, 0.5), (1.0, 1.0, 1.0), False),
    ("y^2-0.25", 64, (0.1, -0.51, 0.3), (1.0, 1.0, 1.0), False),
    ("y^2-0.25", 64, (0.1, 0.5, 0.3), (1.0, 1.0, 1.0), False),
    ("z^2-0.25", 150, (0.1, -0.2, -0.5), (1.0, 1.0, 1.0), False),
    ("(x^2+y^2+z^2-0.6)*((x-0.3)^2+y^2+z^2-0.04)", 48, (0.3, 0.2, 0.0), (1.0, 1.0, 1.0), False),   # a small sphere inside a large one
    ("(x^2+y^2+z^2-0.6)*((x-0.3)^2+y^2+z^2-0.04)", 48, (0.0, 0.0, 0.7746), (1.0, 1.0, 1.0), False),
])
def test_seed_mode(mc, orc, eq, n, seed, scale, empty):
    step = step_of(n)
    c = mc.Context(0)
    try:
        c.set_seed(*seed)
        c.seed_mode(True)
        r = c.march(eq, step, 0.0, scale)
        o = orc.march_seed(eq, step, seed, 0.0, scale, pow_mode=orc.POW_EXACT)
        assert r.n_tris == o.n_tris
        assert (r.n_tris == 0) == empty        # a case that compares nothing with nothing must say so
        # same triangles, the reference's in breadth-first order and at cell positions -1 + k*step (ulps off the lattice)
        _match_triangle_sets(r.vertices()[:, :, :3], o.soup, TOL_POS)
        # the code volume is the dense sweep's: seed mode only selects triangles
        c.seed_mode(False)
        full = c.march(eq, step, 0.0, scale)
        assert np.array_equal(r.codes(), full.codes()) and r.n_tris <= full.n_tris
    finally:
        c.close()


def _subsequence_gaps(full, part, limit=64):
    """Indices of `full`'s rows (u32 words) missing from `part`, which must be `full` with a few rows taken out."""
    gaps, shift, i = [], 0, 0
    while i - shift < len(part):
        n = min(len(full) - i, len(part) - (i - shift))
        ne = (full[i:i + n] != part[i - shift:i - shift + n]).any(axis=1)
        if not ne.any():
            i += n
            break
        k = int(np.argmax(ne))
        gaps.append(i + k)
        assert len(gaps) <= limit, "too many rows missing"
        i += k + 1
        shift += 1
    gaps.extend(range(i, len(full)))
    assert len(full) - len(gaps) == len(part), "not a sub-sequence"
    return gaps


def test_seed_mode_headline_grid_is_the_same_component_every_time(mc):
    """1025^3 sphere, 4.9 M records in one component, labelled by ~80 k waves racing on one union-find: the result must be
    the dense list minus the triangles of the outermost cells (index N, marching.cpp:84-86) -- every time."""
    step = step_of(1024)
    c = mc.Context(0)
    try:
        fv = u32(c.march(EQ["sphere"], step, flags=0).vertices()[:, :, :3]).reshape(-1, 9).copy()
        c.set_seed(1.0, 0.0, 0.0)
        c.seed_mode(True)
        for _ in range(3):
            r = c.march(EQ["sphere"], step, flags=0)
            assert r.n_tris == 9881649
            rv = u32(r.vertices()[:, :, :3]).reshape(-1, 9)
            gaps = _subsequence_gaps(fv, rv)
            assert len(gaps) == 11
            pos = fv[gaps].view(np.float32).reshape(-1, 3, 3)
            assert (pos.max(axis=(1, 2)) >= 1.0 - 1e-6).all()     # all at the +1 faces of the domain
    finally:
        c.close()


def test_seed_mode_errors(mc):
    c = mc.Context(0)
    try:
        with pytest.raises(mc.McError):
            c.set_seed(1.5, 0, 0)                       # marching.cpp:128
        c.set_seed(1, 0, 0)
        c.seed_mode(True)
        with pytest.raises(mc.McError):
            c.march(EQ["sphere"], step_of(32), z_begin=0, z_end=10)     # whole grid only
    finally:
        c.close()


def test_seed_mode_in_a_captured_graph(mc, orc):
    """Seed mode is capturable (round 4): the component-labelling kernels are nodes of the graph.  An iso sweep in seed mode
    -- two sheets, the seed on one of them -- replayed from one capture equals the un-captured seed-mode sweep frame by
    frame (bytes), soup and welded; moving the seed to the other sheet, or turning seed mode off, makes the next replay
    re-capture (the seed's cell is baked into the captured kernels' arguments)."""
    eq, step = "x^2-0.25", step_of(48)
    c = mc.Context(0)
    try:
        # the seed's cell is x in [0.5, 0.5417): the sheet x = sqrt(0.25 + iso) stays inside it for the first four iso values
        # and leaves it for the fifth (a seed cell without a crossing: an empty mesh, marching.cpp:310-331)
        c.set_seed(0.52, 0.0, 0.0)
        c.seed_mode(True)
        for flags in (mc.FLAG_NORMALS, mc.FLAG_INDEXED | mc.FLAG_NO_EMIT):
            c.graph_build(eq, step, iso=0.02, flags=flags)
            for iso in (0.02, 0.03, 0.011, 0.02, 0.2, 0.03):
                g = c.graph_replay(iso)
                got = g.indexed() if flags & mc.FLAG_INDEXED else (g.vertices(),)
                m = c.march(eq, step, iso, flags=flags)
                want = m.indexed() if flags & mc.FLAG_INDEXED else (m.vertices(),)
                assert g.n_tris == m.n_tris and (g.n_tris > 0) == (iso != 0.2)
                for x, y in zip(got, want):
                    assert np.array_equal(u32(x) if x.dtype == np.float32 else x, u32(y) if y.dtype == np.float32 else y)
                dense_tris = orc.march(eq, step, iso, pow_mode=orc.POW_EXACT, want=0).n_tris
                assert g.n_tris < dense_tris        # one sheet of the two
        c.graph_build(eq, step, iso=0.02)
        a = c.graph_replay(0.02).vertices()
        c.set_seed(-0.52, 0.0, 0.0)                   # the other sheet
        b = c.graph_replay(0.02).vertices()
        assert a.shape == b.shape and len(a) > 0 and a[:, :, 0].min() > 0 and b[:, :, 0].max() < 0
        c.seed_mode(False)
        d = c.graph_replay(0.02)
        assert d.n_tris == orc.march(eq, step, 0.02, pow_mode=orc.POW_EXACT, want=0).n_tris > len(a)    # (the dense sweep also has the tail plane of cells)
    finally:
        c.close()


@pytest.mark.parametrize("n,seed", [(300, (0.0, 1.0, 0.0)), (256, (0.0, 0.0, -1.0))])   # 257 cells per axis: tail plane
def test_seed_mode_large_grids_are_exact_components(mc, orc, n, seed):
    """Beyond ~200 cells per axis the REFERENCE's walk drifts: it re-derives cell positions by adding +-step per move in
    float, the error passes its own 1e-6 set tolerance and cells are visited twice (the oracle, which restates it with
    the same containers, returns 848 145 triangles at grid_res 300 where the whole dense surface has 848 100).  The
    product walks lattice indices, so here it is checked against what the walk MEANS: the dense surface minus the cells
    the bound check excludes (index N on any axis, marching.cpp:84-86) -- the sphere is one component."""
    step = step_of(n)
    c = mc.Context(0)
    try:
        full = c.march(EQ["sphere"], step)
        fv = full.vertices()[:, :, :3]
        c.set_seed(*seed)
        c.seed_mode(True)
        r = c.march(EQ["sphere"], step)
        rv = r.vertices()[:, :, :3]
        ax = np.empty(r.cells_per_axis + 1, f32)
        v = f32(-1.0)
        for i in range(len(ax)):
            ax[i] = v
            v = f32(v + f32(step))
        last = ax[r.cells_per_axis - 1]                         # lower corner of the excluded outermost cells
        excluded = (fv.min(axis=1) >= last).any(axis=1)         # a triangle of such a cell has all vertices beyond it
        # the seed result is the dense list with a few triangles removed: same order, same bits
        fk = [t.tobytes() for t in u32(fv).reshape(len(fv), -1)]
        rk = [t.tobytes() for t in u32(rv).reshape(len(rv), -1)]
        kept = np.zeros(len(fk), bool)
        j = 0
        for i, k in enumerate(fk):
            if j < len(rk) and k == rk[j]:
                kept[i] = True
                j += 1
        assert j == len(rk), "seed-mode triangles are not a sub-sequence of the dense sweep's"
        assert 0 < (~kept).sum() <= 16 and excluded[~kept].all()   # only triangles of the excluded outermost cells went
        o = orc.march_seed(EQ["sphere"], step, seed, pow_mode=orc.POW_EXACT)
        assert o.n_tris >= r.n_tris                             # the drifting walk only ever adds revisits
    finally:
        c.close()


# ---------------------------------------------------------------- randomized differential test
def _random_expr(rng, depth=0):
    """A random string of the reference's grammar (no non-integer powers: those are tolerance-only)."""
    r = rng.random()
    if depth >= 3 or r < 0.25:
        k = rng.random()
        if k < 0.55:
            return "xyz"[rng.integers(3)]
        if k < 0.8:
            return f"{rng.uniform(0.05, 2.5):.3g}"
        return f"{rng.integers(1, 4)}" + "xyz"[rng.integers(3)]           # implicit multiplication, evaluator.cpp:215-223
    if r < 0.45:
        return _random_expr(rng, depth + 1) + "+-"[rng.integers(2)] + _random_expr(rng, depth + 1)
    if r < 0.62:
        return _random_expr(rng, depth + 1) + "*" + _random_expr(rng, depth + 1)
    if r < 0.72:
        return "(" + _random_expr(rng, depth + 1) + ")/(" + _random_expr(rng, depth + 1) + ")"
    if r < 0.86:
        return "(" + _random_expr(rng, depth + 1) + ")^" + str([2, 2, 3, 4, -1, -2][rng.integers(6)])
    if r < 0.93:
        return "-(" + _random_expr(rng, depth + 1) + ")"
    return "(" + _random_expr(rng, depth + 1) + ")"


@pytest.mark.parametrize("seed", range(int(os.environ.get("MC_RANDOM_SEEDS", "48"))))
def test_random_expressions_match_the_oracle(mc, orc, seed):
    """Differential test over random equations, steps, iso values and scales: interval codegen, sampling fallback,
    NaN / inf fields, the precedence quirks -- cube codes and vertices bit for bit against the oracle."""
    rng = np.random.default_rng(1000 + seed)
    eq = _random_expr(rng) + "-" + f"{rng.uniform(0.05, 1.0):.3g}"
    if mc.expr_validate(eq) != 0:
        pytest.skip("generated string is refused (evaluation underflow)")
    n = int(rng.integers(5, 44))
    step = step_of(n)
    iso = float(f32(rng.uniform(-0.4, 0.4)))
    scale = tuple(float(f32(v)) for v in rng.uniform(0.6, 1.8, 3)) if rng.random() < 0.5 else (1.0, 1.0, 1.0)
    c = mc.Context(0)
    try:
        r = c.march(eq, step, iso, scale)
        o = orc.march(eq, step, iso, scale, pow_mode=orc.POW_EXACT, want=3)
        assert r.interpreted == 0                                  # (the specialised kernels: tests/conftest.py)
        assert np.array_equal(r.codes(), o.codes), eq
        assert (r.n_tris, r.n_active) == (o.n_tris, o.n_active), eq
        rv = r.vertices()
        assert_same_floats(rv[:, :, :3], o.soup, eq)
        # ... and the interpreter build of the same kernels (what an unseen equation's first sweeps run on): the same bytes
        try:
            ri = c.march(eq, step, iso, scale, flags=mc.FLAG_NORMALS | mc.FLAG_KEEP_CODES | mc.FLAG_INTERP)
        except mc.McError as e:
            assert e.code == mc.MC_ERR_ARG and "too long" in str(e), eq   # (beyond the interpreter's tables: it says so)
        else:
            assert ri.interpreted == 1 and np.array_equal(ri.codes(), o.codes), eq
            assert_same_floats(ri.vertices()[:, :, :3], rv[:, :, :3], "interpreter build: " + eq)
    finally:
        c.close()


@pytest.mark.parametrize("seed", range(int(os.environ.get("MC_RANDOM_SEEDS", "16"))))
def test_random_constraints_match_the_oracle(mc, orc, seed):
    rng = np.random.default_rng(5000 + seed)
    eq = ["x^2+y^2+z^2-0.8", EQ["eq3"], "x*y*z-0.02", "x^2-y*z-0.1"][rng.integers(4)]
    cons = []
    for _ in range(int(rng.integers(1, 4))):
        lhs = _random_expr(rng, 1)
        if mc.expr_validate(lhs) != 0:
            continue
        cons.append((lhs, [">=", "<=", ">", "<"][rng.integers(4)], float(f32(rng.uniform(-0.5, 0.5)))))
    if not cons:
        pytest.skip("no usable constraint generated")
    step = step_of(int(rng.integers(6, 40)))
    c = mc.Context(0)
    try:
        for i, (lhs, op, rhs) in enumerate(cons):
            c.set_constraint(i, lhs, op, rhs)
        r = c.march(eq, step)
        o = orc.march(eq, step, pow_mode=orc.POW_EXACT, want=3, constraints=cons)
        assert np.array_equal(r.codes(), o.codes), (eq, cons)
        assert r.n_tris == o.n_tris, (eq, cons)
        assert_same_floats(r.vertices()[:, :, :3], o.soup, str((eq, cons)))
    finally:
        c.close()


@pytest.mark.parametrize("seed", range(int(os.environ.get("MC_RANDOM_LARGE", "24"))))
def test_random_equations_on_multi_chunk_grids(mc, orc, seed):
    """Random equations on grids around the 256-cell chunk boundary (ragged last chunk, tail plane for 1..4 cells,
    several chunks) and random Z slabs: bit for bit against the oracle -- and the interval walk (row, block and lane level)
    against the sampling walk (MC_FLAG_NO_CULL) on the same slab, byte for byte."""
    rng = np.random.default_rng(9000 + seed)
    eq = _random_expr(rng) + "-" + f"{rng.uniform(0.05, 1.0):.3g}"
    if mc.expr_validate(eq) != 0:
        pytest.skip("generated string is refused (evaluation underflow)")
    n = int([255, 256, 257, 258, 259, 260, 300, 383, 511, 512][rng.integers(10)])
    step = step_of(n)
    n1 = mc.cells_per_axis(step)
    zb = int(rng.integers(0, n1 - 8))
    ze = int(min(n1, zb + rng.integers(4, 40)))
    iso = float(f32(rng.uniform(-0.3, 0.3)))
    c = mc.Context(0)
    try:
        r = c.march(eq, step, iso, z_begin=zb, z_end=ze)
        o = orc.march(eq, step, iso, pow_mode=orc.POW_EXACT, want=3, z_begin=zb, z_end=ze)
        assert np.array_equal(r.codes(), o.codes), (eq, n, zb, ze)
        assert (r.n_tris, r.n_active) == (o.n_tris, o.n_active), (eq, n, zb, ze)
        rv = r.vertices()
        assert_same_floats(rv[:, :, :3], o.soup, str((eq, n, zb, ze)))
        nc = c.march(eq, step, iso, flags=mc.FLAG_NORMALS | mc.FLAG_KEEP_CODES | mc.FLAG_NO_CULL, z_begin=zb, z_end=ze)
        assert np.array_equal(nc.codes(), o.codes), ("NO_CULL", eq, n, zb, ze)
        assert_same_floats(nc.vertices(), rv, "NO_CULL " + str((eq, n, zb, ze)))
    finally:
        c.close()


def test_layer_launch_order_does_not_change_the_output(mc, ctx):
    """mc_classify launches a slab's layers in z order or from the middle outwards, whichever it measured faster on the
    equation's first sweeps (equation_3 at 513^3 is where middle-out wins).  Forced either way (MC_FLAG_ORDER_Z /
    MC_FLAG_ORDER_MIDDLE_OUT) the whole grid's codes and vertices are the same bytes, and the default's."""
    eq, step = EQ["eq3"], step_of(512)
    base = mc.FLAG_NORMALS | mc.FLAG_KEEP_CODES
    crcs = []
    for f in (mc.FLAG_ORDER_Z, mc.FLAG_ORDER_MIDDLE_OUT, 0, mc.FLAG_ORDER_MIDDLE_OUT | mc.FLAG_TILE63):
        r = ctx.march(eq, step, flags=base | f)
        assert r.n_cells == 513 ** 3 and abs(r.n_tris / (38388 * 16) - 1) < 0.01      # 38 388 at 128, x4 per doubling
        crcs.append((r.n_tris, r.n_active, zlib.crc32(r.codes().tobytes()), zlib.crc32(r.vertices().tobytes())))
    assert len(set(crcs)) == 1, crcs


GOURSAT_32 = {-0.7: 1984, -0.5: 13024, -0.4: 16912, -0.3: 16144, -0.1: 5728}   # SURVEY 8d [probe]: the reference at grid_res 32


def test_goursat_512_iso_sweep_through_the_captured_graph(mc, orc):
    """BASELINE config 5 as stated: the 513^3 Goursat sweep replayed from ONE captured hipGraph with five iso values.  Every
    frame replayed is the frame swept un-captured (same count; for two frames the same bytes); the counts scale from the
    reference's own at grid_res 32 (x4 per doubling) where that coarse grid resolves the surface; and a thin slab of two
    frames -- cut out of the whole-grid result by the triangle offset of the layers below it -- equals the oracle's sweep of
    those layers bit for bit (positions) / 1e-6 (normals)."""
    eq, step = EQ["goursat"], step_of(512)
    c = mc.Context(0)
    try:
        c.graph_build(eq, step, iso=-0.4, flags=mc.FLAG_NORMALS)
        for iso, n32 in GOURSAT_32.items():
            g = c.graph_replay(iso)
            assert g.n_cells == 513 ** 3 and g.interpreted == 0
            # SURVEY 8d's counts at grid_res 32 scale by x4 per doubling where the 33-cell grid resolves the surface (within
            # 15 % at iso -0.7 ... -0.3; at -0.1 the surface's thin parts are below that grid's resolution: half of x256)
            assert abs(g.n_tris / (n32 * 256) - 1) < (0.15 if iso < -0.2 else 0.6), (iso, g.n_tris, n32 * 256)
            gv = g.vertices() if iso in (-0.4, -0.7) else None
            m = c.march(eq, step, iso, flags=mc.FLAG_NORMALS | mc.FLAG_NO_INTERP)      # (re-targets the context's buffers: gv is on the host by now)
            assert (m.n_tris, m.n_active) == (g.n_tris, g.n_active), iso
            if gv is not None:
                assert zlib.crc32(m.vertices().tobytes()) == zlib.crc32(gv.tobytes())
                zb, ze = (100, 103) if iso == -0.4 else (436, 439)
                below = c.march(eq, step, iso, flags=mc.FLAG_NO_EMIT, z_begin=0, z_end=zb).n_tris
                o = orc.march(eq, step, iso, pow_mode=orc.POW_EXACT, want=7, z_begin=zb, z_end=ze)
                assert o.n_tris > 0
                part = gv[below:below + o.n_tris]
                assert_same_floats(part[:, :, :3], o.soup, f"graph frame iso {iso}, layers {zb}..{ze}")
                assert np.nanmax(np.abs(part[:, :, 3:] - o.normals)) <= TOL_NRM
            # (the next replay finds its buffers re-targeted by the sweeps above and re-captures its own sweep first)
    finally:
        c.close()
