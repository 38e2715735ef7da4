"""GPU (-m gpu): the cold start.  Evaluator::set_equation is instant in the reference (evaluator.cpp:15-17); specialising the
kernels for an equation takes hiprtc about a second.  The library therefore carries the same kernels compiled ahead of time
with f as an INTERPRETER of the equation's DAG (mc_kernels.hip, MC_INTERP) and runs an equation's first sweeps on it while a
host thread compiles.  Bar: the interpreter build writes the same bytes as the specialised kernels -- codes, soup, normals,
the indexed mesh -- on every kind of equation, and the first mesh of an unseen equation is there in milliseconds."""
import os
import time

import numpy as np
import pytest

from conftest import EQ
from test_gpu_parity import RATIONAL, CONSTRAINT_SETS, step_of, u32

pytestmark = pytest.mark.gpu
f32 = np.float32

CASES = [(EQ[k], 40, 0.0, (1.0, 1.0, 1.0)) for k in ("eq1", "eq2", "eq3", "eq6", "eq8", "sphere", "ui_default")] + [
    (EQ["goursat"], 48, -0.4, (1.0, 1.0, 1.0)),
    (EQ["eq3"], 36, 0.0, (1.1, 1.1, 1.1)),                       # the UI's default scale (drawer.cpp:43)
    (EQ["sphere"], 300, 0.0, (1.0, 1.0, 1.0)),                   # two chunks per row, ragged last chunk
    (EQ["sphere"], 256, 0.0, (1.0, 1.0, 1.0)),                   # tail plane (257 = 256 + 1 cells per row)
    ("(x-0.1)*(y-0.07)*(z-0.13)-0.0001", 4, 0.0, (1.0, 1.0, 1.0)),   # 13 ambiguous cells, 7 flipped
    ("x*y*z", 16, 0.0, (1.0, 1.0, 1.0)),                         # lattice hits everywhere
    ("1/x+y", 24, 0.0, (1.0, 1.0, 1.0)),                         # inf / NaN samples
    ("(x^2+y^2-0.3)^.5-z", 24, 0.0, (1.0, 1.0, 1.0)),            # general pow, NaN
] + [(RATIONAL[k], 40, 0.0, (1.0, 1.3, 0.9)) for k in sorted(RATIONAL)]


@pytest.mark.parametrize("eq,n,iso,scale", CASES)
def test_interpreter_build_writes_the_same_bytes(mc, orc, ctx, eq, n, iso, scale):
    """MC_FLAG_INTERP (the interpreter build, whatever is cached) against MC_FLAG_NO_INTERP (the kernels hiprtc specialised
    for the equation): the same cube codes, the same vertex bytes (positions AND gradient normals), either emit kernel, and
    the same welded Poly_Data; the codes and positions are the oracle's."""
    step = step_of(n)
    z = (0, -1) if n < 100 else (n // 2 - 3, n // 2 + 2)
    base = mc.FLAG_NORMALS | mc.FLAG_KEEP_CODES
    a = ctx.march(eq, step, iso, scale, base | mc.FLAG_NO_INTERP, *z)
    ca, va = a.codes(), a.vertices()
    assert a.interpreted == 0
    for force in (0, mc.FLAG_EMIT_DIRECT, mc.FLAG_EMIT_SHARED):
        b = ctx.march(eq, step, iso, scale, base | mc.FLAG_INTERP | force, *z)
        assert b.interpreted == 1 and (b.n_tris, b.n_active) == (a.n_tris, a.n_active)
        assert np.array_equal(b.codes(), ca), "cube codes differ"
        vb = b.vertices()
        assert np.array_equal(u32(vb[:, :, :3]), u32(va[:, :, :3])), "positions differ"
        same = (u32(vb[:, :, 3:]) == u32(va[:, :, 3:])) | (np.isnan(vb[:, :, 3:]) & np.isnan(va[:, :, 3:]))
        assert same.all() or np.nanmax(np.abs(vb[:, :, 3:] - va[:, :, 3:])) <= 1e-6, "normals differ"
    o = orc.march(eq, step, iso, scale, pow_mode=orc.POW_EXACT, want=3, z_begin=z[0], z_end=z[1])
    if "^.5" not in eq:   # (non-integer exponents: device pow vs glibc pow, tolerance-only -- DESIGN.md P1)
        assert np.array_equal(ca, o.codes)
        assert ((u32(va[:, :, :3]) == u32(o.soup)) | (np.isnan(va[:, :, :3]) & np.isnan(o.soup))).all()
    ai = ctx.march(eq, step, iso, scale, mc.FLAG_INDEXED | mc.FLAG_NO_EMIT | mc.FLAG_NO_INTERP, *z)
    bi = ctx.march(eq, step, iso, scale, mc.FLAG_INDEXED | mc.FLAG_NO_EMIT | mc.FLAG_INTERP, *z)
    assert (ai.n_verts, ai.n_tris) == (bi.n_verts, bi.n_tris) and bi.interpreted == 1
    for x, y in zip(ai.indexed(), bi.indexed()):
        assert np.array_equal(x.view(np.uint32), y.view(np.uint32)) or ((x == y) | (np.isnan(x) & np.isnan(y))).all()


@pytest.mark.parametrize("name", sorted(CONSTRAINT_SETS)[:8])
def test_interpreter_build_with_constraints(mc, orc, name):
    """The constraints' left-hand sides are interpreted too (marching.cpp:255-280: one Evaluator per constraint)."""
    cons = CONSTRAINT_SETS[name]
    eq, step = EQ["sphere"], step_of(36)
    c = mc.Context(0)
    try:
        for i, (lhs, op, rhs) in enumerate(cons):
            c.set_constraint(i, lhs, op, rhs)
        a = c.march(eq, step, flags=mc.FLAG_NORMALS | mc.FLAG_KEEP_CODES | mc.FLAG_NO_INTERP)
        b = c.march(eq, step, flags=mc.FLAG_NORMALS | mc.FLAG_KEEP_CODES | mc.FLAG_INTERP)
        o = orc.march(eq, step, pow_mode=orc.POW_EXACT, want=3, constraints=cons)
        assert b.interpreted == 1 and a.interpreted == 0
        assert np.array_equal(a.codes(), o.codes) and np.array_equal(b.codes(), o.codes)
        assert a.n_tris == b.n_tris == o.n_tris
        assert np.array_equal(u32(a.vertices()), u32(b.vertices()))
    finally:
        c.close()


def test_interpreter_build_with_sin_and_cos(mc, orc):
    """Grammar extension E1 under the interpreter: the same mc_sinf / mc_cosf (include/mc_trig.h)."""
    old, oldo = mc.set_extensions(mc.EXT_TRIG), orc.set_extensions(1)
    c = mc.Context(0)
    try:
        eq, step, s = "sin(x)*cos(y)+sin(y)*cos(z)+sin(z)*cos(x)", step_of(48), (6.2831853,) * 3
        a = c.march(eq, step, 0.0, s, mc.FLAG_NORMALS | mc.FLAG_KEEP_CODES | mc.FLAG_NO_INTERP)
        b = c.march(eq, step, 0.0, s, mc.FLAG_NORMALS | mc.FLAG_KEEP_CODES | mc.FLAG_INTERP)
        o = orc.march(eq, step, 0.0, s, pow_mode=orc.POW_EXACT, want=3)
        assert np.array_equal(b.codes(), o.codes) and np.array_equal(a.codes(), o.codes) and b.n_tris == o.n_tris > 0
        va, vb = a.vertices(), b.vertices()
        assert np.array_equal(u32(vb[:, :, :3]), u32(o.soup)) and np.array_equal(u32(va[:, :, :3]), u32(vb[:, :, :3]))
        assert np.nanmax(np.abs(va[:, :, 3:] - vb[:, :, 3:])) <= 1e-6
    finally:
        c.close()
        mc.set_extensions(old)
        orc.set_extensions(oldo)


def test_first_mesh_of_an_unseen_equation(mc, orc, tmp_path):
    """A fresh context, an empty code-object cache, an equation nobody has compiled: the first march() returns the oracle's
    mesh without waiting for hiprtc (interpreted = 1), a host thread compiles meanwhile, and some sweeps later the
    specialised kernels take over (interpreted = 0) with the same bytes.  The same for Evaluator::evaluate (mc_eval_points)
    and for the facade's default, the indexed mesh.  Timings are reported by bench.py; here only a loose bound."""
    old = os.environ.get("MC_JIT_CACHE")
    old_cold = os.environ.pop("MC_COLD_START", None)     # (tests/conftest.py makes the rest of the suite wait for hiprtc)
    os.environ["MC_JIT_CACHE"] = str(tmp_path)
    c = mc.Context(0)
    try:
        tag = int(time.time() * 1e3) % 100000
        eq = f"x^2+y^2+z^2-0.9{tag:05d}"                   # unseen by construction
        step = step_of(32)
        c.march("x+y", step, flags=mc.FLAG_INTERP)           # (module load and the GPU's first launch are not the equation's cost)
        t0 = time.perf_counter()
        r = c.march(eq, step, flags=mc.FLAG_NORMALS | mc.FLAG_KEEP_CODES)
        t_first = time.perf_counter() - t0
        assert r.interpreted == 1
        o = orc.march(eq, step, pow_mode=orc.POW_EXACT, want=7)
        v = r.vertices()
        assert np.array_equal(r.codes(), o.codes) and np.array_equal(u32(v[:, :, :3]), u32(o.soup))
        assert np.abs(v[:, :, 3:] - o.normals).max() <= 1e-6
        assert t_first < 0.25, t_first                      # (bench.py reports the figure; VERDICT's bar is 50 ms)
        # Evaluator::evaluate right after set_equation: one point, no hiprtc in the way
        eq2 = f"x*y-z+0.3{tag:05d}"
        t0 = time.perf_counter()
        val = c.eval_points(eq2, np.array([[0.5, 0.25, 0.125]], np.float32))
        assert time.perf_counter() - t0 < 0.25
        assert u32(val)[0] == u32(np.array([mc.expr_debug_eval_host(eq2, 0.5, 0.25, 0.125)], np.float32))[0]
        # the facade's default: the welded mesh of an unseen equation
        eq3 = f"x^2+y^2+z^2-0.8{tag:05d}"
        ri = c.march(eq3, step, flags=mc.FLAG_INDEXED | mc.FLAG_NO_EMIT)
        oi = orc.march_indexed(eq3, step, pow_mode=orc.POW_EXACT)
        assert ri.interpreted == 1 and (ri.n_verts, ri.n_tris) == (oi.n_verts, oi.n_tris)
        vl, tl, _ = ri.indexed()
        assert np.array_equal(vl.view(np.uint32), oi.vertices.view(np.uint32)) and np.array_equal(tl, oi.tris)
        # the specialised kernels arrive by themselves
        deadline, took_over = time.perf_counter() + 60.0, None
        while time.perf_counter() < deadline:
            r2 = c.march(eq, step, flags=mc.FLAG_NORMALS | mc.FLAG_KEEP_CODES)
            if r2.interpreted == 0:
                took_over = r2
                break
            time.sleep(0.05)
        assert took_over is not None, "the compile thread never delivered"
        assert np.array_equal(took_over.codes(), o.codes) and np.array_equal(u32(took_over.vertices()), u32(v))
        # a captured graph is always the specialised kernels'
        eq4 = f"x^2+y^2+z^2-0.7{tag:05d}"
        c.graph_build(eq4, step)
        assert c.graph_replay(0.0).interpreted == 0
    finally:
        c.close()
        if old_cold is not None:
            os.environ["MC_COLD_START"] = old_cold
        if old is None:
            os.environ.pop("MC_JIT_CACHE", None)
        else:
            os.environ["MC_JIT_CACHE"] = old


def test_equation_too_long_for_the_interpreter_waits_for_hiprtc(mc, orc):
    deep = "*(".join(f"(x^{k % 7 + 2}*y-{k}.5)" for k in range(20)) + "*(z" + ")" * 20 + "-0.001"
    c = mc.Context(0)
    try:
        r = c.march(deep, step_of(12))
        assert r.interpreted == 0
        o = orc.march(deep, step_of(12), pow_mode=orc.POW_EXACT, want=1)
        assert np.array_equal(r.codes(), o.codes)
        with pytest.raises(mc.McError) as e:
            c.march(deep, step_of(12), flags=mc.FLAG_INTERP)
        assert e.value.code == mc.MC_ERR_ARG
    finally:
        c.close()
