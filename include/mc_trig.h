/*
 * mc_trig.h -- sin / cos of the grammar EXTENSION E1 (MC_EXT_TRIG).
 *
 * The reference's grammar has no functions (Source/evaluator.cpp:139-237 rejects any letter other
 * than x, y, z), so nothing here mirrors reference code: `sin(...)` / `cos(...)` are an opt-in
 * extension that makes BASELINE.json's gyroid workload expressible.  Because no reference result
 * exists for it, the extension defines its own arithmetic, and defines it ONCE: this header is
 * compiled into the device code (hiprtc), into the host-side expression compiler (constant
 * folding) and into the CPU oracle, so all three produce the same bits by construction.
 *
 * Definition: the argument is widened to double, reduced by a two-term Cody-Waite step against
 * pi/2 (exact product for |k| <= 2^20), the fdlibm kernel polynomials (Sun, 1993; public
 * constants) are evaluated in double in the fixed order written below, and the result is rounded
 * once to float.  |x| >= 2^20, inf and NaN give NaN.  The result is within 1 ulp (float) of the
 * true value (tests/test_trig.py measures it against a long-double libm).
 *
 * Every translation unit that includes this must be compiled with -ffp-contract=off.
 */
#ifndef MC_TRIG_H
#define MC_TRIG_H

#ifndef MC_TRIG_FN
#define MC_TRIG_FN static inline
#endif

/* which = 0: sin, 1: cos */
MC_TRIG_FN float mc_trig_eval(float xf, int which) {
    const double x = (double)xf;
    if (!(__builtin_fabs(x) < 1048576.0)) return __builtin_nanf("");
    const double k = __builtin_rint(x * 6.36619772367581382433e-01); /* x * 2/pi, ties to even */
    /* pi/2 = P1 + P1T: P1 holds the first 33 bits, so k * P1 is exact for |k| <= 2^20 */
    const double r0 = x - k * 1.57079632673412561417e+00;
    const double r = r0 - k * 6.07710050650619224932e-11;
    const double z = r * r;
    /* sin(r), |r| <= pi/4 */
    const double ps = 8.33333333332248946124e-03 +
                      z * (-1.98412698298579493134e-04 +
                           z * (2.75573137070700676789e-06 + z * (-2.50507602534068634195e-08 + z * 1.58969099521155010221e-10)));
    const double s = r + (r * z) * (-1.66666666666666324348e-01 + z * ps);
    /* cos(r) */
    const double pc = 4.16666666666666019037e-02 +
                      z * (-1.38888888888741095749e-03 +
                           z * (2.48015872894767294178e-05 +
                                z * (-2.75573143513906633035e-07 + z * (2.08757232129817482790e-09 + z * -1.13596475577881948265e-11))));
    const double c = 1.0 - (0.5 * z - (z * z) * pc);
    const int q = ((int)k + which) & 3;
    const double v = (q & 1) ? c : s;
    return (float)((q & 2) ? -v : v);
}

MC_TRIG_FN float mc_sinf(float x) { return mc_trig_eval(x, 0); }
MC_TRIG_FN float mc_cosf(float x) { return mc_trig_eval(x, 1); }

#endif
