#!/usr/bin/env python3
"""How often does the product's power rule (`^n` with a literal integer n is an IEEE product) change a CUBE
CODE against the reference's glibc powf (evaluator.cpp:133)?  CPU only; drives the oracle (test infrastructure) in both
of its power modes over random polynomial equations whose squares / cubes are taken of NON-lattice sub-expressions (where
powf(a, 2) and a*a do differ in the last bit on ~0.07 % of inputs, tests/test_oracle_pins.py) and counts the cells whose
8-bit cube code differs.  A cube code only changes when the two values of f straddle iso, i.e. |f - iso| below one ulp of
a sum of O(1) terms at that lattice point -- this script puts a number on it.

    python tests/golden/power_rule_risk.py [--equations 128] [--grid-res 96] [--seed 2026] [--out tests/golden/power_rule_risk.json]

Default: 128 equations x 97^3 cells = 1.17e8 cells (about 3 minutes on 8 cores).  The committed JSON is the record of one
run of exactly this script (seed in the file); tests/test_oracle_pins.py re-runs a slice of it and compares."""
import argparse
import json
import random
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "oracle"))
import pyoracle as orc  # noqa: E402


def coef(rng, lo=0.05, hi=1.5):
    return f"{rng.uniform(lo, hi):.4f}"


def linear(rng):
    """a*x+b*y+c*z+d with non-lattice coefficients: its value at a lattice point is not a 'nice' float"""
    vs = rng.sample(["x", "y", "z"], rng.randint(1, 3))
    s = "+".join(f"{coef(rng)}*{v}" for v in vs)
    return f"({s}+{coef(rng, 0.01, 0.9)})" if rng.random() < 0.7 else f"({s})"


def random_equation(rng):
    """sum of powers of linear forms minus a constant.  Only `+` between terms and ONE trailing `-`: the reference reduces
    equal-precedence chains right to left (evaluator.cpp:22-48), a-b-c would be a-(b-c)."""
    terms = []
    for _ in range(rng.randint(2, 4)):
        n = rng.choice([2, 2, 2, 3, 4])
        t = f"{linear(rng)}^{n}"
        if rng.random() < 0.4:
            t = f"{coef(rng)}*{t}"
        terms.append(t)
    return "+".join(terms) + "-" + coef(rng, 0.2, 1.2)


def sweep(eqs, grid_res, threads=None):
    step = float(np.float32(2.0) / np.float32(grid_res))
    rows, cells, diff_cells, active, diff_eqs = [], 0, 0, 0, 0
    for eq in eqs:
        a = orc.march(eq, step, 0.0, pow_mode=orc.POW_LIBM, want=orc.WANT_CODES, nthreads=threads)
        b = orc.march(eq, step, 0.0, pow_mode=orc.POW_EXACT, want=orc.WANT_CODES, nthreads=threads)
        d = int(np.count_nonzero(a.codes != b.codes))
        rows.append({"equation": eq, "cells": int(a.n_cells), "active_libm": int(a.n_active), "code_mismatches": d,
                     "tris_libm": int(a.n_tris), "tris_exact": int(b.n_tris)})
        cells += int(a.n_cells)
        active += int(a.n_active)
        diff_cells += d
        diff_eqs += 1 if d else 0
    return rows, cells, active, diff_cells, diff_eqs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--equations", type=int, default=128)
    ap.add_argument("--grid-res", type=int, default=96)
    ap.add_argument("--seed", type=int, default=2026)
    ap.add_argument("--out", default=str(Path(__file__).with_name("power_rule_risk.json")))
    args = ap.parse_args()
    rng = random.Random(args.seed)
    eqs = [random_equation(rng) for _ in range(args.equations)]
    t0 = time.time()
    rows, cells, active, diff_cells, diff_eqs = sweep(eqs, args.grid_res)
    out = {"script": "tests/golden/power_rule_risk.py", "seed": args.seed, "grid_res": args.grid_res, "equations": len(eqs),
           "cells_swept_per_mode": cells, "active_cells_libm": active, "cube_code_mismatches": diff_cells,
           "equations_with_a_mismatch": diff_eqs,
           "mismatch_rate_per_cell": diff_cells / cells, "mismatch_rate_per_active_cell": diff_cells / max(active, 1),
           # rule of three: with 0 events in N trials the 95 % upper bound of the rate is 3 / N
           "upper_bound_95_per_cell": (3.0 / cells) if diff_cells == 0 else None,
           "upper_bound_95_per_active_cell": (3.0 / max(active, 1)) if diff_cells == 0 else None,
           "glibc": " ".join(__import__("platform").libc_ver()), "seconds": round(time.time() - t0, 1), "per_equation": rows}
    rows_txt = ",\n".join("  " + json.dumps(r) for r in out.pop("per_equation"))   # one line per equation
    Path(args.out).write_text(json.dumps(out, indent=1)[:-2] + ',\n "per_equation": [\n' + rows_txt + "\n ]\n}\n")
    out["per_equation"] = rows
    print(json.dumps({k: v for k, v in out.items() if k != "per_equation"}))


if __name__ == "__main__":
    main()
