"""Grammar extension E1: sin(...) / cos(...) (include/mc_trig.h).  NOT a reference feature -- the reference's
tokenizer rejects these letters (evaluator.cpp:224) -- so parity here is "unpinned by the reference": the tests
pin (a) default-off behaviour = the reference's, (b) the shared arithmetic against libm to 1 ulp, (c) the
product (host DAG, device code, interval culling) against the oracle's stack walk on the same grammar."""
import numpy as np
import pytest

GYROID = "sin(x)*cos(y)+sin(y)*cos(z)+sin(z)*cos(x)"
f32 = np.float32


@pytest.fixture()
def ext(mc, orc):
    a, b = mc.set_extensions(mc.EXT_TRIG), orc.set_extensions(1)
    yield
    mc.set_extensions(a)
    orc.set_extensions(b)


def test_default_grammar_is_the_references(mc, orc):
    for eq in ["sin(x)", "cos(y)", "x+sin(z)", "s", "cos"]:
        assert not mc.expr_check(eq)
        with pytest.raises(ValueError):
            orc.march(eq, 0.25)


ACCEPT = ["sin(x)", "cos(x)", "2sin(3x)cos(y)", "-sin(x)^2", GYROID, "sin(cos(x+1)*2)", "xsin(y)", "SIN(x)+Cos(y)", "sin(1)",
          "(x)sin(y)", "sin(-x)", "sin(x)(y)", "x^sin(0)", "sin(x)+"]
REJECT = ["sin x", "sin()", "sin", "sin(", "si(x)", "sinx", "tan(x)", "sin(x", "sin)x(", "sin(+x)", "s(x)"]


def test_tokenizer_with_extension(mc, orc, ext):
    for eq in ACCEPT:
        assert mc.expr_check(eq), eq
    for eq in REJECT:
        assert not mc.expr_check(eq), eq
    assert mc.expr_validate("sin(x)+") != 0      # trailing operator: tokenizer accepts, evaluation would underflow
    # still the reference's quirks: unary minus binds tighter than ^, right-to-left chains
    assert mc.expr_debug_eval_host("-sin(x)^2", 0.5, 0, 0) > 0
    # the oracle's tokenizer agrees on every string
    for eq in ACCEPT + REJECT:
        try:
            orc.march(eq, 0.5, want=0)
            ok = True
        except ValueError:
            ok = False
        assert ok == (mc.expr_validate(eq) == 0), eq


def test_host_dag_equals_oracle_walk(mc, orc, ext):
    rng = np.random.default_rng(5)
    pts = rng.uniform(-7, 7, (300, 3)).astype(f32)
    for eq in [e for e in ACCEPT if e != "sin(x)+"]:
        for x, y, z in pts[:60]:
            a = f32(mc.expr_debug_eval_host(eq, x, y, z))
            b = f32(orc.evaluate(eq, x, y, z, orc.POW_EXACT))
            assert a.view(np.uint32) == b.view(np.uint32) or (np.isnan(a) and np.isnan(b)), (eq, x, y, z, a, b)


def test_shared_trig_accuracy_against_libm(mc, ext):
    rng = np.random.default_rng(9)
    x = np.concatenate([rng.uniform(-4, 4, 4000), rng.uniform(-100, 100, 4000), rng.uniform(-8000, 8000, 4000),
                        [0.0, np.pi / 2, np.pi, -np.pi, 1.5707964, 8191.5, -8191.5]]).astype(f32)
    for fn, ref in (("sin", np.sin), ("cos", np.cos)):
        got = np.array([mc.expr_debug_eval_host(f"{fn}(x)", v, 0, 0) for v in x], f32)
        want = ref(x.astype(np.float64))
        err = np.abs(got.astype(np.float64) - want)
        assert err.max() <= 1.2e-7                                  # the bound the interval code relies on (mc_trig_iv)
        big = np.abs(want) > 1e-3
        ulp = np.spacing(np.abs(want[big]).astype(f32)).astype(np.float64)
        assert np.max(err[big] / ulp) <= 2.0
        assert np.all(np.abs(got) <= 1.0)
    # outside the defined domain: NaN
    for v in (8192.0, -3e6, np.inf, np.nan):
        assert np.isnan(mc.expr_debug_eval_host("sin(x)", v, 0, 0)) and np.isnan(mc.expr_debug_eval_host("cos(x)", v, 0, 0))


# ------------------------------------------------------------------------------------------------ GPU
def _u32(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


@pytest.mark.gpu
@pytest.mark.parametrize("eq", [GYROID, "sin(3x)*cos(2y)-z", "sin(x*y)+cos(z/(x^2+1))", "-sin(x)^2+0.3", "sin(x)/cos(y)-z"])
def test_eval_points_bit_exact(mc, orc, ext, eq):
    rng = np.random.default_rng(3)
    pts = rng.uniform(-9, 9, (20000, 3)).astype(f32)
    c = mc.Context(0)
    try:
        got = c.eval_points(eq, pts)
    finally:
        c.close()
    want = np.array([orc.evaluate(eq, *p, orc.POW_EXACT) for p in pts[:3000]], f32)
    same = (_u32(got[:3000]) == _u32(want)) | (np.isnan(got[:3000]) & np.isnan(want))
    assert same.all()


@pytest.mark.gpu
@pytest.mark.parametrize("eq,n,iso,scale", [
    (GYROID, 48, 0.0, (6.2831853, 6.2831853, 6.2831853)),
    (GYROID, 200, 0.3, (12.566371, 12.566371, 12.566371)),
    (GYROID, 33, 0.0, (1.0, 1.0, 1.0)),
    ("cos(x)+cos(y)+cos(z)", 64, 0.0, (9.424778, 9.424778, 9.424778)),        # Schwarz P
    ("sin(x)*sin(y)*sin(z)+sin(x)*cos(y)*cos(z)+cos(x)*sin(y)*cos(z)+cos(x)*cos(y)*sin(z)", 40, 0.0, (6.2831853,) * 3),  # diamond
    ("sin(x)/cos(y)-z", 40, 0.0, (3.0, 3.0, 1.0)),                             # not boundable: sampling walk, inf / NaN cells
    ("x^2+y^2+z^2-sin(4x)^2", 40, 0.5, (1.0, 1.0, 1.0)),
    # the same functions on all three axes (MC_TAB_SYM: mc_emit's one pass over all edge directions) with a scale per axis,
    # one function next to plain variables, and a product
    (GYROID, 72, 0.1, (6.2831853, 9.424778, 12.566371)),
    ("sin(2x)*y+sin(2y)*z+sin(2z)*x", 56, 0.05, (2.0, 2.5, 3.0)),
    ("cos(x)*cos(y)*cos(z)-0.1", 64, 0.0, (6.2831853,) * 3),
])
def test_sweep_matches_oracle(mc, orc, ext, eq, n, iso, scale):
    step = float(f32(2.0) / f32(n))
    c = mc.Context(0)
    try:
        r = c.march(eq, step, iso, scale)
        o = orc.march(eq, step, iso, scale, pow_mode=orc.POW_EXACT, want=7)
        assert np.array_equal(r.codes(), o.codes)
        assert (r.n_tris, r.n_active) == (o.n_tris, o.n_active) and o.n_tris > 0
        v = r.vertices()
        same = (_u32(v[:, :, :3]) == _u32(o.soup)) | (np.isnan(v[:, :, :3]) & np.isnan(o.soup))
        assert same.all()
        d = np.abs(v[:, :, 3:] - o.normals)
        assert np.nanmax(d) <= 1e-6
    finally:
        c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("eq,n,scale", [(GYROID, 96, 12.566371), ("cos(x)+cos(y)+cos(z)", 80, 9.424778), (GYROID, 64, 1.0)])
def test_interval_culling_of_trig_is_exact(mc, ext, eq, n, scale):
    """mc_sin_iv / mc_cos_iv enclosures cull rows without sampling; compiled out (MC_FLAG_NO_CULL) the result is identical."""
    step = float(f32(2.0) / f32(n))
    c = mc.Context(0)
    try:
        a = c.march(eq, step, 0.0, (scale,) * 3)
        ca, va = a.codes(), a.vertices()
        b = c.march(eq, step, 0.0, (scale,) * 3, flags=mc.FLAG_NORMALS | mc.FLAG_KEEP_CODES | mc.FLAG_NO_CULL)
        assert np.array_equal(ca, b.codes()) and np.array_equal(_u32(va), _u32(b.vertices()))
    finally:
        c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("eq,n,scale,cons", [
    (GYROID, 64, (6.2831853,) * 3, [("x", ">", -0.5)]),
    (GYROID, 80, (12.566371, 6.2831853, 9.0), [("x+y", "<=", 0.4), ("z^2", "<", 0.6)]),
    ("sin(3y)*x+z*z-0.2", 48, (1.0, 1.0, 1.0), [("sin(2x)", ">=", -0.3)]),       # a tabulated f under a tabulated constraint
    ("x*y^5+z^2-0.3", 64, (1.0, 1.3, 1.0), [("1/(x*x+0.5)", "<", 1.7)]),
])
def test_tabulated_equations_with_constraints_and_scales(mc, orc, ext, eq, n, scale, cons):
    """Equations whose one-variable sub-expressions come from tables (MC_TAB), with constraints (marching.cpp:476: their
    left-hand sides are evaluated the ordinary way) and anisotropic scales: codes and soup bit for bit, either emit kernel."""
    step = float(f32(2.0) / f32(n))
    c = mc.Context(0)
    try:
        for i, (lhs, op, rhs) in enumerate(cons):
            c.set_constraint(i, lhs, op, rhs)
        o = orc.march(eq, step, 0.1, scale, pow_mode=orc.POW_EXACT, want=7, constraints=cons)
        assert o.n_tris > 0
        for force in (0, mc.FLAG_EMIT_DIRECT, mc.FLAG_EMIT_SHARED):
            r = c.march(eq, step, 0.1, scale, mc.FLAG_NORMALS | mc.FLAG_KEEP_CODES | force)
            assert np.array_equal(r.codes(), o.codes) and (r.n_tris, r.n_active) == (o.n_tris, o.n_active)
            v = r.vertices()
            same = (_u32(v[:, :, :3]) == _u32(o.soup)) | (np.isnan(v[:, :, :3]) & np.isnan(o.soup))
            assert same.all()
            assert np.nanmax(np.abs(v[:, :, 3:] - o.normals)) <= 1e-6
    finally:
        c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("z", [(700, 703), (0, 2), (1023, 1025)])
def test_gyroid_1024_thin_slabs_against_the_oracle(mc, orc, ext, z):
    """BASELINE config 4 at its full size (1025^3 cells, scale 4 pi), three layers at a time: codes, counts and positions
    bit for bit against the oracle, normals within 1e-6; both emit kernels."""
    n, s = 1024, 12.566371
    step = float(f32(2.0) / f32(n))
    c = mc.Context(0)
    try:
        o = orc.march(GYROID, step, 0.0, (s,) * 3, pow_mode=orc.POW_EXACT, want=7, z_begin=z[0], z_end=z[1])
        assert o.n_tris > 50000
        for force in (0, mc.FLAG_EMIT_DIRECT, mc.FLAG_EMIT_SHARED):
            r = c.march(GYROID, step, 0.0, (s,) * 3, mc.FLAG_NORMALS | mc.FLAG_KEEP_CODES | force, z[0], z[1])
            assert np.array_equal(r.codes(), o.codes)
            assert (r.n_tris, r.n_active) == (o.n_tris, o.n_active)
            v = r.vertices()
            assert np.array_equal(_u32(v[:, :, :3]), _u32(o.soup))
            assert np.nanmax(np.abs(v[:, :, 3:] - o.normals)) <= 1e-6
    finally:
        c.close()


@pytest.mark.gpu
def test_gyroid_512_slabs_and_properties(mc, ext):
    """BASELINE config 4's surface at a size the box finishes quickly: z-slabs concatenate to the whole sweep, every
    vertex lies on a cell edge and on the surface (|f| small)."""
    n, s = 512, 12.566371
    step = float(f32(2.0) / f32(n))
    c = mc.Context(0)
    try:
        whole = c.march(GYROID, step, 0.0, (s,) * 3, mc.FLAG_NORMALS)
        nt = whole.n_tris
        assert nt > 5_000_000
        parts = 0
        n1 = whole.cells_per_axis
        for rank in range(4):
            zb, ze = mc.shard_layers(n1, 4, rank)
            parts += c.march(GYROID, step, 0.0, (s,) * 3, mc.FLAG_NORMALS, zb, ze).n_tris
        assert parts == nt
        r = c.march(GYROID, step, 0.0, (s,) * 3, mc.FLAG_NORMALS)
        v = r.vertices()[::97].reshape(-1, 6)
        f = c.eval_points(GYROID, (v[:, :3] * f32(s)).astype(f32))
        assert np.abs(f).max() < 2e-3          # linear interpolation error on a cell of size 2/512 * 4pi
        assert np.abs(np.linalg.norm(v[:, 3:], axis=1) - 1).max() < 1e-5
    finally:
        c.close()


@pytest.mark.gpu
def test_gyroid_1024_whole_grid_properties(mc, orc, ext):
    """BASELINE config 4 whole: 1025^3 cells, 40.8 M triangles (2.9 GB of vertices).  Size-independent properties of the one
    sweep: eight Z slabs add up to it (counts) and concatenate to it (a checksum of checksums over the vertex bytes); three
    layers cut out of it by the triangle offset of the layers below equal the oracle's sweep of those layers bit for bit;
    on a sample, every vertex lies on a lattice edge (two lattice coordinates), on the surface (|f| small) and carries a
    unit normal that points towards f > iso."""
    import zlib
    n, s = 1024, 12.566371
    step = float(f32(2.0) / f32(n))
    c = mc.Context(0)
    try:
        whole = c.march(GYROID, step, 0.0, (s,) * 3, mc.FLAG_NORMALS)
        n1, nt = whole.cells_per_axis, whole.n_tris
        assert n1 == 1025 and whole.n_cells == 1025 ** 3 and 40_000_000 < nt < 42_000_000
        v = whole.vertices()
        # a thin slab of the whole against the oracle
        zb, ze = 700, 703
        below = c.march(GYROID, step, 0.0, (s,) * 3, mc.FLAG_NO_EMIT, 0, zb).n_tris
        o = orc.march(GYROID, step, 0.0, (s,) * 3, pow_mode=orc.POW_EXACT, want=7, z_begin=zb, z_end=ze)
        part = v[below:below + o.n_tris]
        assert np.array_equal(_u32(part[:, :, :3]), _u32(o.soup))
        assert np.nanmax(np.abs(part[:, :, 3:] - o.normals)) <= 1e-6
        # sample properties
        smp = v[::4099].reshape(-1, 6)
        ax = np.empty(n1 + 1, f32)
        a = f32(-1.0)
        for i in range(n1 + 1):
            ax[i] = a
            a = f32(a + f32(step))
        on_lattice = np.isin(smp[:, :3], ax).sum(axis=1)
        assert (on_lattice >= 2).all()
        fval = c.eval_points(GYROID, (smp[:, :3] * f32(s)).astype(f32))
        assert np.abs(fval).max() < 1e-3
        assert np.abs(np.linalg.norm(smp[:, 3:], axis=1) - 1).max() < 1e-5
        h = f32(1e-3)
        fplus = c.eval_points(GYROID, ((smp[:, :3] + h * smp[:, 3:]) * f32(s)).astype(f32))
        assert (fplus > fval).mean() > 0.999
        # slabs == whole
        at, tris, active = 0, 0, 0
        for rank in range(8):
            zb, ze = mc.shard_layers(n1, 8, rank)
            r = c.march(GYROID, step, 0.0, (s,) * 3, mc.FLAG_NORMALS, zb, ze)
            pv = r.vertices()
            assert zlib.crc32(pv.tobytes()) == zlib.crc32(v[at:at + r.n_tris].tobytes()), rank
            at += r.n_tris
            tris += r.n_tris
            active += r.n_active
        assert (tris, active) == (nt, whole.n_active)
    finally:
        c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(12))
def test_random_trig_expressions_match_the_oracle(mc, orc, ext, seed):
    rng = np.random.default_rng(7000 + seed)
    terms = []
    for _ in range(int(rng.integers(1, 4))):
        fn = ["sin", "cos"][rng.integers(2)]
        arg = f"{rng.integers(1, 7)}" + "xyz"[rng.integers(3)] + ["", "+0.3", "*y", "-z"][rng.integers(4)]
        terms.append(f"{fn}({arg})" + ["", "*x", "^2", "*cos(2z)"][rng.integers(4)])
    eq = "+".join(terms) + "-" + f"{rng.uniform(0.05, 0.6):.3g}" + ["", "+z", "-y*y"][rng.integers(3)]
    assert mc.expr_validate(eq) == 0, eq
    step = float(f32(2.0) / f32(int(rng.integers(8, 48))))
    scale = (float(f32(rng.uniform(0.7, 3.0))),) * 3
    c = mc.Context(0)
    try:
        r = c.march(eq, step, 0.0, scale)
        o = orc.march(eq, step, 0.0, scale, pow_mode=orc.POW_EXACT, want=3)
        assert np.array_equal(r.codes(), o.codes), eq
        assert r.n_tris == o.n_tris, eq
        v = r.vertices()[:, :, :3]
        assert ((_u32(v) == _u32(o.soup)) | (np.isnan(v) & np.isnan(o.soup))).all(), eq
    finally:
        c.close()


@pytest.mark.gpu
def test_extensions_pinned_per_context(mc, orc):
    """mc_context_set_extensions: one context accepts sin / cos whatever the process-wide word says, its neighbour on the
    same device keeps the reference's grammar; ext < 0 returns a context to following the process-wide setting."""
    assert mc.set_extensions(0) == 0                   # process-wide: the reference's grammar
    a, b = mc.Context(0), mc.Context(0)
    try:
        a.set_extensions(mc.EXT_TRIG)
        eq, step = "sin(x)+y", 0.25
        r = a.march(eq, step)
        old = orc.set_extensions(1)
        try:
            o = orc.march(eq, step, pow_mode=orc.POW_EXACT, want=3)
        finally:
            orc.set_extensions(old)
        assert r.n_tris == o.n_tris > 0 and np.array_equal(r.codes(), o.codes)
        with pytest.raises(mc.McError) as e:
            b.march(eq, step)                           # the neighbour: letters other than x, y, z are rejected (evaluator.cpp:224)
        assert e.value.code == mc.MC_ERR_PARSE
        a.set_extensions(-1)
        with pytest.raises(mc.McError):
            a.march("cos(x)+y", step)
        assert a.march("x+y", step).n_tris > 0          # both still sweep the reference's language
    finally:
        a.close()
        b.close()
        mc.set_extensions(0)
