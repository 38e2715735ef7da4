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
 * Definition (all in float, fixed operation order, no FMA): k = rint(x * 2/pi); three-term Cody-Waite reduction
 * r = ((x - k*P1) - k*P2) - k*P3 with P1 + P2 + P3 = pi/2 (the products are exact for |k| < 2^13); the cephes
 * single-precision kernel polynomials for sin r and cos r on |r| <= pi/4; quadrant select; clamp to [-1, 1].
 * |x| >= 8192, inf and NaN give NaN.  Measured against libm on 3e7 arguments: absolute error <= 9.3e-8,
 * <= 1.6 ulp (tests/test_trig.py keeps both bounds).  A double-precision version was correctly rounded almost
 * everywhere but made the gyroid sweep 3x slower (double ops run at half rate and sin/cos dominate it).
 *
 * Every translation unit that includes this must be compiled with -ffp-contract=off.
 */
#ifndef MC_TRIG_H
#define MC_TRIG_H

#ifndef MC_TRIG_FN
#define MC_TRIG_FN static inline
#endif

/* which = 0: sin, 1: cos.  Branch-free on purpose: sin(a) and cos(a) of the same a (the gyroid has three such pairs)
 * then share everything but the quadrant select once both calls are inlined -- an early return for the out-of-range
 * case put the two reductions into different basic blocks and the compiler kept both. */
MC_TRIG_FN float mc_trig_eval(float x, int which) {
    const int ok = __builtin_fabsf(x) < 8192.0f; /* false for inf and NaN too */
    const float xs = ok ? x : 0.0f;
    const float k = __builtin_rintf(xs * 0.636619772367581343f); /* x * 2/pi, ties to even */
    float r = xs - k * 1.5703125f;
    r = r - k * 4.837512969970703125e-4f;
    r = r - k * 7.54978995489188216e-8f;
    const float z = r * r;
    const float s = r + (r * z) * (-1.6666654611e-1f + z * (8.3321608736e-3f + z * -1.9515295891e-4f));
    const float c = (1.0f - 0.5f * z) + (z * z) * (4.166664568298827e-2f + z * (-1.388731625493765e-3f + z * 2.443315711809948e-5f));
    const int q = ((int)k + which) & 3;
    float v = (q & 1) ? c : s;
    v = (q & 2) ? -v : v;
    v = __builtin_fminf(1.0f, __builtin_fmaxf(-1.0f, v));
    return ok ? v : __builtin_nanf("");
}

MC_TRIG_FN float mc_sinf(float x) { return mc_trig_eval(x, 0); }
MC_TRIG_FN float mc_cosf(float x) { return mc_trig_eval(x, 1); }

#endif
