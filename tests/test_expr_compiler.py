"""CPU: the product's host-side expression compiler (csrc/mc_expr.cpp) against the oracle.

Two independent implementations of the reference's evaluation semantics are compared:
the oracle executes the two-stack walk per sample (evaluator.cpp:53-107); the product
executes it once symbolically and emits a DAG.  mc_expr_debug_eval_host interprets that
DAG with the same float ops the generated device code uses.
"""
import random

import numpy as np
import pytest

from conftest import EQ
from test_oracle_pins import PROBE_TOKENIZER_CASES, REF_TOKENIZER_CASES


def bits(x):
    return np.float32(x).view(np.uint32)


def same(a, b):
    return bits(a) == bits(b) or (np.isnan(a) and np.isnan(b))


@pytest.mark.parametrize("eq,expect", REF_TOKENIZER_CASES + PROBE_TOKENIZER_CASES)
def test_check_matches_reference_cases(mc, eq, expect):
    assert mc.expr_check(eq) == bool(expect)


def _random_string(rng, n):
    alphabet = "xyzXYZ0123456789.+-*/^() "
    return "".join(rng.choice(alphabet) for _ in range(n))


def test_check_matches_oracle_fuzz(mc, orc):
    rng = random.Random(1234)
    accepted = 0
    for _ in range(20000):
        s = _random_string(rng, rng.randint(1, 10))
        a, b = mc.expr_check(s), orc.tokenize(s)
        assert a == b, s
        accepted += a
    assert accepted > 500


def _random_expr(rng, depth=0):
    r = rng.random()
    if depth > 3 or r < 0.25:
        return rng.choice(["x", "y", "z", "x", "y", "z", "2", "0.5", ".25", "3", "1.5", "7"])
    if r < 0.35:
        return "-" + _random_expr(rng, depth + 1) if rng.random() < 0.5 else "(" + _random_expr(rng, depth + 1) + ")"
    if r < 0.45:
        return _random_expr(rng, depth + 1) + "^" + rng.choice(["2", "3", "2", "-2", "4", "0", "1", "-1", "2"])
    if r < 0.5:  # implicit multiplication
        return rng.choice(["2", "x", "(y+1)"]) + "(" + _random_expr(rng, depth + 1) + ")"
    op = rng.choice(["+", "-", "*", "/", "+", "-", "*"])
    return _random_expr(rng, depth + 1) + op + _random_expr(rng, depth + 1)


def test_dag_matches_oracle_random_expressions(mc, orc):
    rng = random.Random(99)
    pts = [(rng.uniform(-1.5, 1.5), rng.uniform(-1.5, 1.5), rng.uniform(-1.5, 1.5)) for _ in range(6)] + [(0.0, 1.0, -1.0)]
    n_ok = 0
    for _ in range(1500):
        eq = _random_expr(rng)
        ok = orc.tokenize(eq)
        assert mc.expr_check(eq) == ok, eq
        if not ok:
            continue
        rc = mc.expr_validate(eq)
        for p in pts:
            p = tuple(float(np.float32(c)) for c in p)
            want = orc.evaluate(eq, *p, pow_mode=orc.POW_EXACT)
            if want is None:
                assert rc == mc.MC_ERR_EVAL, eq
                break
            assert rc == mc.MC_OK, eq
            got = mc.expr_debug_eval_host(eq, *p)
            assert same(got, want), (eq, p, got, want)
        n_ok += 1
    assert n_ok > 1000


@pytest.mark.parametrize("name", sorted(EQ))
def test_example_equations_bit_exact(mc, orc, name):
    rng = np.random.default_rng(7)
    pts = rng.uniform(-1.2, 1.2, size=(200, 3)).astype(np.float32)
    want = orc.evaluate_many(EQ[name], pts, pow_mode=orc.POW_EXACT)
    got = np.array([mc.expr_debug_eval_host(EQ[name], *map(float, p)) for p in pts], np.float32)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_libm_vs_exact_power_is_tiny(orc):
    """glibc powf is not correctly rounded, the product's x*x is; they differ by
    <= 1 ulp of the power on a small fraction of inputs."""
    rng = np.random.default_rng(3)
    pts = rng.uniform(-1.2, 1.2, size=(4000, 3)).astype(np.float32)
    a = orc.evaluate_many(EQ["eq8"], pts, pow_mode=orc.POW_LIBM)
    b = orc.evaluate_many(EQ["eq8"], pts, pow_mode=orc.POW_EXACT)
    assert np.allclose(a, b, rtol=3e-6, atol=1e-6)
    assert (a.view(np.uint32) != b.view(np.uint32)).mean() < 0.25


@pytest.mark.parametrize("eq,code", [("x+", 2), ("x*-", 2), ("   ", 2), ("", 1), ("sin(x)", 1), ("(x", 1), ("x)", 1),
                                     ("x^2+y^2+z^2-1", 0), ("xyz", 0), ("2x(y)", 0)])
def test_validate_status(mc, eq, code):
    assert mc.expr_validate(eq) == code


def test_dump_shows_reference_order(mc):
    src = mc.expr_dump("x-y-z")
    # x-(y-z): the inner subtraction comes first and x is the minuend of the outer one
    lines = [l.strip() for l in src.splitlines() if "=" in l and l.strip().startswith("const float")]
    assert lines[0].endswith("= y - z;") and lines[1].startswith("const float") and "= x - t" in lines[1]
    assert "x * x" in mc.expr_dump("x^2") and "mc_pow_int<3>" in mc.expr_dump("x^3")
    assert "mc_pow_general" in mc.expr_dump("x^y")


@pytest.mark.parametrize("step", [2 / 32, 2 / 256, 2 / 1024, 0.2, 0.3, 0.1, 0.001, 0.5, 0.37, 0.0123])
def test_cells_per_axis_matches_oracle(mc, orc, step):
    s = float(np.float32(step))
    assert mc.cells_per_axis(s) == orc.cells_per_axis(s)


@pytest.mark.parametrize("step", [0.0005, 0.6, 0.0, -1.0, float("nan")])
def test_step_out_of_range(mc, step):  # marching.cpp:226-238
    assert mc.cells_per_axis(step) == 0


def test_integer_power_of_a_subexpression_generates_valid_code(mc):
    """(expr)^n with n not in {0, 1, 2}: the generated call must name its operand (a shared format buffer once
    produced `t2t2)`); the kernels must compile (hiprtc runs without a GPU)."""
    for eq in ["(x+y)^3-1", "((z)^3)^3-0.56", "(((y)/(z))^-1)^4-0.749", "0.712-1x+(2x)^3*(0.913*y)-0.885", "z+(x*y)^-2"]:
        src = mc.expr_dump(eq)
        assert "mc_pow_int<" in src and "t2t2" not in src
        import re
        assert not re.search(r"= t\d+t\d+\)", src), src
        assert mc.jit_precompile(eq) > 0


# ---- the interpreter's program (cold start: mc_hip.h MC_FLAG_INTERP) -------------------------------------------------
def test_interpreter_program_matches_the_dag(mc, orc):
    """The first sweeps of an equation run on an interpreter of its DAG (mc_kernels.hip MC_INTERP): one word per operation,
    16 registers allocated by liveness, a constant table.  Walked on the host exactly as the device walks it
    (mc_expr_debug_interp_host) it gives the DAG's value -- and the oracle's stack walk's -- bit for bit: register reuse,
    shared sub-expressions, constants folded or not, roots that are a bare variable or literal."""
    rng = random.Random(4242)
    pts = [tuple(float(np.float32(rng.uniform(-1.5, 1.5))) for _ in range(3)) for _ in range(5)] + [(0.0, 1.0, -1.0)]
    n_ok = 0
    for k in range(1500):
        eq = _random_expr(rng) if k % 3 else "+".join(_random_expr(rng) for _ in range(rng.randint(2, 6)))   # (long sums: many live values)
        if not mc.expr_check(eq) or mc.expr_validate(eq) != mc.MC_OK:
            continue
        for p in pts:
            want = mc.expr_debug_eval_host(eq, *p)
            try:
                got = mc.expr_debug_interp_host(eq, *p)
            except mc.McError as e:        # does not fit the tables: allowed, but must say so
                assert e.code == mc.MC_ERR_ARG and "too long" in str(e), eq
                break
            assert same(got, want), (eq, p, got, want)
            ref = orc.evaluate(eq, *p, pow_mode=orc.POW_EXACT)
            assert ref is None or same(got, ref), (eq, p, got, ref)
        else:
            n_ok += 1
    assert n_ok > 900
    for eq in ("x", "5", "(y)", "-z", "2x", "x^0", "x^1"):
        assert same(mc.expr_debug_interp_host(eq, 0.5, -0.25, 2.0), mc.expr_debug_eval_host(eq, 0.5, -0.25, 2.0)), eq


@pytest.mark.parametrize("name", sorted(EQ))
def test_interpreter_program_example_equations(mc, name):
    rng = np.random.default_rng(11)
    for p in rng.uniform(-1.2, 1.2, size=(100, 3)).astype(np.float32):
        p = tuple(map(float, p))
        assert same(mc.expr_debug_interp_host(EQ[name], *p), mc.expr_debug_eval_host(EQ[name], *p)), (name, p)


def test_interpreter_refuses_what_does_not_fit(mc):
    """More than 16 values alive at once, or more than 120 operations: the equation then always waits for hiprtc
    (mc_runtime.hip get_compiled falls back), and the diagnostic entry says why."""
    wide = "(" + ")*(".join(f"x^{k}+y^{k + 1}" for k in range(2, 14)) + ")"          # fits
    assert np.isfinite(mc.expr_debug_interp_host(wide, 0.5, 0.25, 0.125))
    # a right-nested chain keeps every left operand alive until the innermost bracket is done
    deep = "*(".join(f"(x^{k % 7 + 2}*y-{k}.5)" for k in range(20)) + "*(z" + ")" * 20
    assert mc.expr_validate(deep) == mc.MC_OK
    with pytest.raises(mc.McError) as e:
        mc.expr_debug_interp_host(deep, 0.5, 0.25, 0.125)
    assert e.value.code == mc.MC_ERR_ARG
    long = "+".join(f"x^{k % 5 + 2}*{k}.25" for k in range(70))
    with pytest.raises(mc.McError):
        mc.expr_debug_interp_host(long, 0.5, 0.25, 0.125)
