#!/usr/bin/env python3
"""Generate tests/golden/*.npz with the CPU oracle (oracle/mc_oracle.c).

The reference itself cannot be built in this image (its hot-path TUs include
<windows.h>), so the vectors come from the oracle, which tests/test_oracle_pins.py
pins bit-for-bit to the fingerprints SURVEY.md section 4 recorded from the unmodified
reference.  Rows that also appear in that SURVEY table carry the reference fingerprint
in `ref_fnv_codes` / `ref_fnv_soup`, and this script refuses to write a row whose
oracle output does not hash to it.

Each .npz holds: equation (str), step, iso, scale, n1, codes u8[n_cells], soup f32[T,3,3]
(libm powf semantics == the reference's), normals f32[T,3,3] (exact-power
semantics), counts.  Run from the repo root:  python tests/golden/make_golden.py
"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "oracle"))
import pyoracle as orc  # noqa: E402

EQ3 = "(x^2+y^2+z^2+(1/3)^2-(1/5)^2)^2-4*((1/2)*x-(2.36/6)*(1/5))^2-4*(1/3)^2*y^2"   # example_files/equation_3.txt
EQ8 = "(x^2+y^2-(1/16))^2+(y^2+z^2-(1/16))^2+(z^2+x^2-(1/16))^2-8*(x^2+y^2+z^2-(1/4))^2"  # equation_8.txt
EQ2 = "x^2*y^2+x^2*z^2+z^2*y^2+x*y*z"  # equation_2.txt
UI_DEFAULT = "(x^2+y^2-1)^2 + (x^2+z^2-1)^2 + (z^2+y^2-1)^2 - 0.5"  # drawer.cpp:39

f32 = np.float32


def step_of(n):
    return float(f32(2.0) / f32(n))


# name, equation, step, iso, scale, (ref_fnv_codes, ref_fnv_soup) or None
ROWS = [
    ("amb2d_a_n4", "(x-0.1)*(y-0.07)-0.001", step_of(4), 0.0, (1, 1, 1), ("88343fbaa746bfc6", "f0c853cc7f615c4e")),
    ("amb3d_a_n4", "(x-0.1)*(y-0.07)*(z-0.13)-0.0001", step_of(4), 0.0, (1, 1, 1), ("3fe2175923bab521", "4b4bd108c366fe8f")),
    ("amb3d_b_n4", "(x-0.1)*(y+0.07)*(z-0.13)-0.0005", step_of(4), 0.0, (1, 1, 1), ("38924c48d8f5cae0", "ab531bb809938f33")),
    ("amb2d_b_n4", "(x-0.1)*(y+0.07)-0.001", step_of(4), 0.0, (1, 1, 1), None),
    ("plane_n32", "x+y", step_of(32), 0.0, (1, 1, 1), ("1950087ac319be97", "3081dde6768e362b")),
    ("sphere_n32", "x^2+y^2+z^2-1", step_of(32), 0.0, (1, 1, 1), ("3a293a4d31ff065b", "4598d5da3647dd7b")),
    ("eq3_n32", EQ3, step_of(32), 0.0, (1, 1, 1), ("af8da92a3fd9a13d", "5121f2bf60815b59")),
    ("eq8_n32", EQ8, step_of(32), 0.0, (1, 1, 1), ("2e75c3f8432b75b1", "f20239c35f7c33c7")),
    ("eq2_n32", EQ2, step_of(32), 0.0, (1, 1, 1), ("7fe35be4d2b04829", "aaab3ea02c309a33")),
    ("goursat_n32", "(x^2)^2+(y^2)^2+(z^2)^2-(x^2+y^2+z^2)", step_of(32), -0.4, (1, 1, 1),
     ("fd486f89f66bbc5c", "61136007ac533813")),
    ("ui_default_s11", UI_DEFAULT, 0.2, 0.0, (1.1, 1.1, 1.1), None),      # drawer.cpp:39-44 defaults
    ("sphere_step03", "x^2+y^2+z^2-1", 0.3, 0.0, (1, 1, 1), None),         # non power-of-two step: drifting float adds
    ("sphere_step01_aniso", "x^2+y^2+z^2-1", 0.1, 0.1, (1.0, 1.3, 0.8), None),
]


# "next" rows of the path: no reference fingerprint exists for these, the oracle's restatement is the source
# (parity unpinned by reference outputs).  name, equation, step, constraints, seed
EXTRA = [
    ("constraint_sphere_n32", "x^2+y^2+z^2-1", step_of(32), [("x", ">", -0.5)], None),          # the developer viewer's hotkey
    ("constraint3_eq3_n32", EQ3, step_of(32), [("x", ">", -0.5), ("y+z", "<=", 0.25), ("x*y", ">=", -0.1)], None),
    ("seed_planes_n32", "x^2-0.25", step_of(32), [], (0.5, 0.0, 0.0)),
    ("seed_sphere_n32", "x^2+y^2+z^2-1", step_of(32), [], (1.0, 0.0, 0.0)),
]


def extra(out_dir):
    for name, eq, step, cons, seed in EXTRA:
        if seed is None:
            m = orc.march(eq, step, pow_mode=orc.POW_EXACT, want=orc.WANT_CODES | orc.WANT_SOUP, constraints=cons)
            codes = m.codes
        else:
            m = orc.march_seed(eq, step, seed, pow_mode=orc.POW_EXACT)
            codes = np.zeros(0, np.uint8)
        np.savez_compressed(out_dir / f"{name}.npz", equation=np.array(eq), step=f32(step), kind=np.array("seed" if seed else "constraint"),
                            constraints=np.array([f"{l}|{o}|{r!r}" for l, o, r in cons]), seed=np.array(seed if seed else (0, 0, 0), f32),
                            codes=codes, soup=m.soup, n_tris=np.int64(m.n_tris))
        print(f"{name:22s} tris={m.n_tris:6d}")


def main():
    out_dir = Path(__file__).resolve().parent
    extra(out_dir)
    for name, eq, step, iso, scale, ref in ROWS:
        m = orc.march(eq, step, iso, scale, pow_mode=orc.POW_LIBM, want=orc.WANT_CODES | orc.WANT_SOUP)
        e = orc.march(eq, step, iso, scale, pow_mode=orc.POW_EXACT,
                      want=orc.WANT_CODES | orc.WANT_SOUP | orc.WANT_NORMALS)
        if ref is not None:
            assert f"{m.fnv_codes:016x}" == ref[0] and f"{m.fnv_soup:016x}" == ref[1], name
        np.savez_compressed(
            out_dir / f"{name}.npz", equation=np.array(eq), step=f32(step), iso=f32(iso), scale=np.array(scale, f32),
            n1=np.int32(m.n1), codes=m.codes, soup=m.soup, normals=e.normals, soup_exact=e.soup, codes_exact=e.codes,
            n_active=np.int64(m.n_active), n_tris=np.int64(m.n_tris), n_amb=np.int64(m.n_amb),
            n_flipped=np.int64(m.n_flipped), fnv_codes=np.uint64(m.fnv_codes), fnv_soup=np.uint64(m.fnv_soup),
            ref_pinned=np.bool_(ref is not None))
        same = np.array_equal(m.codes, e.codes) and np.array_equal(m.soup.view(np.uint32), e.soup.view(np.uint32))
        print(f"{name:22s} n1={m.n1:3d} tris={m.n_tris:6d} amb={m.n_amb}/{m.n_flipped} libm==exact:{same} pinned:{ref is not None}")


if __name__ == "__main__":
    main()
