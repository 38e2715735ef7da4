/*
 * mc_oracle.c -- CPU ORACLE (test infrastructure, NOT product code; see mc_oracle.h).
 *
 * Plain-C restatement of the reference hot path.  Citations are file:line into
 * the reference repository (Source/...).  Build with -ffp-contract=off and
 * without -ffast-math: the reference is x86-64 SSE code with one rounding per
 * float operation.
 *
 * PINNING.  By reference-held material: the case tables (the reference's marching_lookup.h compiled in place,
 * oracle/check_tables_ref.c: 0 mismatches) and the tokenizer (the reference's own 9 self-test cases,
 * evaluator.h:67-77).  NUMERIC RESULTS: PARITY UNPINNED -- the reference holds no numeric fixture, and its two
 * hot-path translation units cannot be built here (both include <windows.h>; stand-in headers are not allowed).  The
 * numeric pins this oracle passes (tests/test_oracle_pins.py: codes / soup fingerprints and counts of SURVEY.md
 * section 4) were recorded by the survey session from a build of the unmodified sources that did use a stand-in
 * windows.h; they are evidence, not a pin the tier rules accept.  DESIGN.md section 5.
 */
#include "../include/mc_trig.h"
#include "mc_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "../include/mc_tables_data.h"

/* ------------------------------------------------------------------ tables */
/* marching_lookup.h:64-320 / :329-587 / :25-32 / :10-23, packed (tools/gen_tables.py) */
static const uint64_t k_tri_row[256] = MC_TRI_ROW_INIT;
static const uint8_t k_tri_count[256] = MC_TRI_COUNT_INIT;
static const uint8_t k_amb_face[256] = MC_AMB_FACE_INIT;
static const uint16_t k_face_corner[6] = MC_FACE_CORNER_INIT;
static const uint8_t k_edge_corner[12] = MC_EDGE_CORNER_INIT;

const uint64_t *orc_tri_rows(void) { return k_tri_row; }
const uint8_t *orc_tri_counts(void) { return k_tri_count; }
const uint8_t *orc_amb_faces(void) { return k_amb_face; }
const uint16_t *orc_face_corners(void) { return k_face_corner; }
const uint8_t *orc_edge_corners(void) { return k_edge_corner; }

uint64_t orc_fnv1a(const void *p, size_t n, uint64_t h) {
    const unsigned char *b = (const unsigned char *)p;
    for (size_t i = 0; i < n; i++) {
        h ^= b[i];
        h *= 1099511628211ull;
    }
    return h;
}
#define FNV_OFFSET 1469598103934665603ull

/* --------------------------------------------------------------- tokenizer */
/* evaluator.h:37-41 */
static int is_operator(char c) { return c == '+' || c == '-' || c == '*' || c == '/' || c == '^'; }
static int is_number(char c) { return (c >= '0' && c <= '9') || c == '.'; }
static int is_variable(char c) { return (c >= 'x' && c <= 'z') || (c >= 'X' && c <= 'Z'); }

static int push_tok(orc_expr *e, int type, char ch, float num) {
    if (e->n == e->cap) {
        int nc = e->cap ? e->cap * 2 : 64;
        orc_token *t = (orc_token *)realloc(e->tok, (size_t)nc * sizeof(orc_token));
        if (!t) return -1;
        e->tok = t;
        e->cap = nc;
    }
    e->tok[e->n].type = type;
    e->tok[e->n].ch = ch;
    e->tok[e->n].num = num;
    e->n++;
    return 0;
}

void orc_expr_free(orc_expr *e) {
    if (e) {
        free(e->tok);
        e->tok = NULL;
        e->n = e->cap = 0;
    }
}

/* Grammar extension E1 (NOT in the reference, whose tokenizer rejects these letters): `sin(` / `cos(`.
 * Off unless a test switches it on; the arithmetic is include/mc_trig.h, shared with the product so that the
 * extension has one definition (there is no reference result for it; tests/test_trig.py checks the accuracy). */
static unsigned g_ext = 0;
unsigned orc_set_extensions(unsigned ext) {
    unsigned old = g_ext;
    g_ext = ext & 1u;
    return old;
}
static int is_func_at(const char *s, size_t i, size_t n) {
    if (!(g_ext & 1u) || i + 3 >= n || s[i + 3] != '(') return 0;
    char a = s[i] | 0x20, b = s[i + 1] | 0x20, c = s[i + 2] | 0x20;
    if (a == 's' && b == 'i' && c == 'n') return 's';
    if (a == 'c' && b == 'o' && c == 's') return 'c';
    return 0;
}

/* evaluator.cpp:139-237 */
int orc_tokenize(const char *eq, orc_expr *out) {
    out->n = 0;
    out->cap = 0;
    out->tok = NULL;
    size_t len0 = strlen(eq);
    if (len0 == 0) return 0; /* :141 */

    /* :147 remove ' ' (only the space character) */
    char *s = (char *)malloc(len0 + 1);
    if (!s) return 0;
    size_t n = 0;
    for (size_t i = 0; i < len0; i++)
        if (eq[i] != ' ') s[n++] = eq[i];
    s[n] = 0;

    int neg = 0, brac = 0, ok = 1;
    int last = -1; /* NONE */
    for (size_t i = 0; i < n && ok; i++) {
        char ch = s[i];
        /* :162 negative sign: first char, or previous CHARACTER is '(' or an operator */
        if (ch == '-' && (i == 0 || s[i - 1] == '(' || is_operator(s[i - 1]))) {
            if (neg) { ok = 0; break; } /* :163 */
            neg = 1;
            push_tok(out, ORC_NEG, 'N', 0.0f);
            continue; /* :167 last_tok unchanged */
        } else if (is_func_at(s, i, n)) { /* E1: placed and counted like a '(' */
            if (last == ORC_VAR || last == ORC_NUM || last == ORC_BRAC_C) push_tok(out, ORC_OP, '*', 0.0f);
            push_tok(out, ORC_FUNC, (char)is_func_at(s, i, n), 0.0f);
            push_tok(out, ORC_BRAC_O, '(', 0.0f);
            brac++;
            last = ORC_BRAC_O;
            i += 3;
        } else if (ch == '(') { /* :170-178 */
            if (last == ORC_VAR || last == ORC_NUM || last == ORC_BRAC_C) push_tok(out, ORC_OP, '*', 0.0f);
            push_tok(out, ORC_BRAC_O, ch, 0.0f);
            brac++;
            last = ORC_BRAC_O;
        } else if (ch == ')') { /* :179-185 */
            if (neg || last == ORC_BRAC_O || last == ORC_OP) { ok = 0; break; }
            if (brac == 0) { ok = 0; break; }
            push_tok(out, ORC_BRAC_C, ch, 0.0f);
            brac--;
            last = ORC_BRAC_C;
        } else if (is_operator(ch)) { /* :186-191 */
            if (neg || last == ORC_BRAC_O || last == ORC_OP || last == -1) { ok = 0; break; }
            push_tok(out, ORC_OP, ch, 0.0f);
            last = ORC_OP;
        } else if (is_number(ch)) { /* :192-214 */
            if (last == ORC_VAR || last == ORC_BRAC_C) push_tok(out, ORC_OP, '*', 0.0f);
            int dot = (ch == '.');
            size_t b = i;
            while (i + 1 < n && is_number(s[i + 1])) {
                if (s[i + 1] == '.') {
                    if (dot) { ok = 0; break; }
                    dot = 1;
                }
                i++;
            }
            if (!ok) break;
            if (i == b && ch == '.') { ok = 0; break; } /* :210 just "." */
            char save = s[i + 1];
            s[i + 1] = 0;
            float v = strtof(s + b, NULL); /* :82 stof(token) at evaluation time */
            s[i + 1] = save;
            push_tok(out, ORC_NUM, '0', v);
            last = ORC_NUM;
        } else if (is_variable(ch)) { /* :215-223 */
            if (last == ORC_VAR || last == ORC_NUM || last == ORC_BRAC_C) push_tok(out, ORC_OP, '*', 0.0f);
            push_tok(out, ORC_VAR, ch, 0.0f);
            last = ORC_VAR;
        } else {
            ok = 0; /* :224 */
            break;
        }
        neg = 0; /* :227 */
    }
    if (ok && brac != 0) ok = 0; /* :231 */
    free(s);
    if (!ok) {
        orc_expr_free(out);
        return 0;
    }
    return 1;
}

/* --------------------------------------------------------------- evaluator */
typedef struct {
    float v;
    int is_const; /* only used by ORC_POW_EXACT to recognise literal integer exponents */
} orc_val;

typedef struct {
    char *ops;
    orc_val *vals;
    int opn, vn;
    int pow_mode;
} orc_stk;

/* evaluator.cpp:111-124 */
static int precedence(char c) {
    switch (c) {
    case 'N': return 4;
    case '^': return 3;
    case '/': return 2;
    case '*': return 2;
    case '+': return 1;
    case '-': return 1;
    default: return 0; /* '(' ')' */
    }
}

/* DESIGN.md "P1": the device-matching power.  Exponent literal-integer |n|<=16:
 * left-to-right product chain in double, one final rounding to float (n==2 is
 * then exactly the float product).  Otherwise (float)pow(double,double). */
static float pow_exact(float a, float b, int b_const) {
    if (b_const && b == floorf(b) && fabsf(b) <= 16.0f) {
        int n = (int)b;
        if (n == 0) return 1.0f;
        int m = n < 0 ? -n : n;
        double p = (double)a, r = p;
        for (int i = 2; i <= m; i++) r = r * p;
        if (n < 0) r = 1.0 / r;
        return (float)r;
    }
    return (float)pow((double)a, (double)b);
}

/* evaluator.cpp:127-136 */
static orc_val evaluate_operation(char op, orc_val a, orc_val b, int pow_mode) {
    orc_val r;
    r.is_const = a.is_const && b.is_const;
    switch (op) {
    case '+': r.v = a.v + b.v; break;
    case '-': r.v = a.v - b.v; break;
    case '*': r.v = a.v * b.v; break;
    case '/': r.v = a.v / b.v; break;
    default: /* '^' :133 pow(float,float) -> powf */
        r.v = (pow_mode == ORC_POW_LIBM) ? powf(a.v, b.v) : pow_exact(a.v, b.v, b.is_const);
        break;
    }
    return r;
}

/* evaluator.cpp:22-48 */
static int evaluate_op(orc_stk *s) {
    if (s->opn == 0) return -1;
    char op = s->ops[--s->opn];
    if (is_operator(op)) {
        if (s->vn == 0) return -1;
        orc_val val1 = s->vals[--s->vn];
        if (s->opn > 0) { /* :32-37 recurse ONCE if the next operator binds tighter */
            char op2 = s->ops[s->opn - 1];
            if (precedence(op2) > precedence(op))
                if (evaluate_op(s)) return -1;
        }
        if (s->vn == 0) return -1;
        orc_val val2 = s->vals[--s->vn];
        s->vals[s->vn++] = evaluate_operation(op, val2, val1, s->pow_mode); /* :39 val2 op val1 */
    } else if (op == 'N') { /* :42-46 */
        if (s->vn == 0) return -1;
        s->vals[s->vn - 1].v = -s->vals[s->vn - 1].v;
    } else {
        return -1; /* :47 throw */
    }
    return 0;
}

static int eval_with(const orc_expr *e, orc_stk *s, float x, float y, float z, float *out) {
    s->opn = s->vn = 0;
    for (int i = 0; i < e->n; i++) { /* evaluator.cpp:62-98 */
        const orc_token *t = &e->tok[i];
        switch (t->type) {
        case ORC_NEG: s->ops[s->opn++] = 'N'; break;
        case ORC_VAR: {
            orc_val v;
            v.is_const = 0;
            char c = t->ch;
            v.v = (c == 'x' || c == 'X') ? x : (c == 'y' || c == 'Y') ? y : z;
            s->vals[s->vn++] = v;
            break;
        }
        case ORC_NUM: {
            orc_val v;
            v.is_const = 1;
            v.v = t->num;
            s->vals[s->vn++] = v;
            break;
        }
        case ORC_BRAC_O: s->ops[s->opn++] = '('; break;
        case ORC_BRAC_C:
            for (;;) {
                if (s->opn == 0) return -1;
                if (s->ops[s->opn - 1] == '(') break;
                if (evaluate_op(s)) return -1;
            }
            s->opn--;
            if (s->opn > 0 && (s->ops[s->opn - 1] == 'S' || s->ops[s->opn - 1] == 'C')) { /* E1: apply the function */
                if (s->vn == 0) return -1;
                const int which = s->ops[--s->opn] == 'C';
                s->vals[s->vn - 1].v = mc_trig_eval(s->vals[s->vn - 1].v, which);
            }
            break;
        case ORC_FUNC: s->ops[s->opn++] = t->ch == 's' ? 'S' : 'C'; break;
        default: s->ops[s->opn++] = t->ch; break;
        }
    }
    while (s->opn > 0) /* :100-102 */
        if (evaluate_op(s)) return -1;
    if (s->vn == 0) return -1;
    *out = s->vals[s->vn - 1].v; /* :105 */
    return 0;
}

static int stk_init(orc_stk *s, const orc_expr *e, int pow_mode) {
    s->ops = (char *)malloc((size_t)e->n + 4);
    s->vals = (orc_val *)malloc(((size_t)e->n + 4) * sizeof(orc_val));
    s->pow_mode = pow_mode;
    s->opn = s->vn = 0;
    return (s->ops && s->vals) ? 0 : -1;
}
static void stk_free(orc_stk *s) {
    free(s->ops);
    free(s->vals);
}

int orc_evaluate(const orc_expr *e, float x, float y, float z, int pow_mode, float *out) {
    orc_stk s;
    if (stk_init(&s, e, pow_mode)) return -1;
    int r = eval_with(e, &s, x, y, z, out);
    stk_free(&s);
    return r;
}

/* ------------------------------------------------------------------- sweep */
/* marching.cpp:372-377: for (v = -1.0f; v <= (float)(1.0 + 0.5*step); v += step) */
int orc_cells_per_axis(float step) {
    if (!(step > 0.0f)) return 0;
    float upper = (float)(1.0 + 0.5 * (double)step);
    int n = 0;
    for (float v = -1.0f; v <= upper; v += step) {
        n++;
        if (n > (1 << 22)) return 0;
    }
    return n;
}

int orc_axis_coords(float step, float *c, int cap) {
    int n1 = orc_cells_per_axis(step);
    if (n1 <= 0 || cap < n1 + 1) return -1;
    float v = -1.0f;
    for (int i = 0; i <= n1; i++) { /* c[i+1] = c[i] + step: marching.cpp:377 and :458-460 are the same float add */
        c[i] = v;
        v += step;
    }
    return n1;
}

typedef struct {
    const orc_expr *e;
    orc_stk stk;
    float iso, step, sx, sy, sz;
    int ncons; /* enabled constraints, marching.h:58-69 */
    const orc_expr *ce[3];
    orc_stk cstk[3];
    int cop[3];
    float crhs[3];
} orc_ctx;

/* marching.cpp:209-224 Marching::evaluate: f(scale_x*x, scale_y*y, scale_z*z) */
static int F(orc_ctx *c, float x, float y, float z, float *out) {
    return eval_with(c->e, &c->stk, c->sx * x, c->sy * y, c->sz * z, out);
}

/* marching.cpp:255-280 check_constraints: every enabled constraint lhs(scaled point) op rhs must hold
 * (a NaN lhs fails every comparison).  Returns 1 inside, 0 outside, -1 on evaluation underflow. */
static int within_constraints(orc_ctx *c, float x, float y, float z) {
    int ok = 1;
    for (int i = 0; i < c->ncons; i++) {
        float lhs;
        if (eval_with(c->ce[i], &c->cstk[i], c->sx * x, c->sy * y, c->sz * z, &lhs)) return -1;
        const float rhs = c->crhs[i];
        switch (c->cop[i]) {
        case ORC_CMP_GE: ok &= lhs >= rhs; break;
        case ORC_CMP_LE: ok &= lhs <= rhs; break;
        case ORC_CMP_GT: ok &= lhs > rhs; break;
        default: ok &= lhs < rhs; break;
        }
    }
    return ok;
}

/* marching.cpp:437-446 */
static float interp(float iso, float x_s, float x_e, float v_s, float v_e) {
    float v = ((iso - v_s) / (v_e - v_s)) * (x_e - x_s);
    if (isinf(v)) return (float)((double)x_s + 0.5 * (double)(x_e - x_s));
    if (isnan(v)) return (float)((double)x_s + 0.5 * (double)(x_e - x_s));
    return x_s + v;
}

typedef struct {
    uint8_t *codes; /* n1*n1 per layer when wanted */
    float *soup;
    float *nrm;
    size_t ntri, cap;
    uint64_t n_active, n_amb, n_flip;
    int err;
} orc_layer;

static int layer_reserve(orc_layer *L, size_t more, int want) {
    if (L->ntri + more <= L->cap) return 0;
    size_t nc = L->cap ? L->cap * 2 : 256;
    while (nc < L->ntri + more) nc *= 2;
    if (want & ORC_WANT_SOUP) {
        float *p = (float *)realloc(L->soup, nc * 9 * sizeof(float));
        if (!p) return -1;
        L->soup = p;
    }
    if (want & ORC_WANT_NORMALS) {
        float *p = (float *)realloc(L->nrm, nc * 9 * sizeof(float));
        if (!p) return -1;
        L->nrm = p;
    }
    L->cap = nc;
    return 0;
}

/* DESIGN.md "N1": gradient normal at an emitted vertex.  Not a reference
 * feature (normal.h computes area-weighted face normals on the UI side); only
 * the orientation is pinned: it points to the f > iso side like the reference's
 * cross(B-A, C-A). */
static int grad_normal(orc_ctx *c, const float p[3], float h, float n[3], int *okflag) {
    float a, b, g[3];
    if (F(c, p[0] + h, p[1], p[2], &a) || F(c, p[0] - h, p[1], p[2], &b)) return -1;
    g[0] = a - b;
    if (F(c, p[0], p[1] + h, p[2], &a) || F(c, p[0], p[1] - h, p[2], &b)) return -1;
    g[1] = a - b;
    if (F(c, p[0], p[1], p[2] + h, &a) || F(c, p[0], p[1], p[2] - h, &b)) return -1;
    g[2] = a - b;
    float len = sqrtf((g[0] * g[0] + g[1] * g[1]) + g[2] * g[2]);
    if (len > 0.0f && !isinf(len)) {
        const float inv = 1.0f / len;
        n[0] = g[0] * inv;
        n[1] = g[1] * inv;
        n[2] = g[2] * inv;
        *okflag = 1;
    } else {
        *okflag = 0;
    }
    return 0;
}

/* marching.cpp:456-595 calculate_step for one cell, without the output side: cube code, the table row used after
 * the ambiguity test (:519-549) and the intersection point of every crossed edge in edge order (:557-583).
 * Returns 0, or -1 on evaluation underflow.  skipped = 1: a corner is outside a constraint (:476), nothing computed. */
typedef struct {
    int skipped, code, row, amb, flipped;
    float val[8];
    float ex[12], ey[12], ez[12]; /* NaN where the edge carries no intersection */
} orc_cellcore;

static int cell_core(orc_ctx *c, float x0, float x1, float y0, float y1, float z0, float z1, orc_cellcore *o) {
    /* :471-472 */
    const float cx[8] = {x0, x1, x1, x0, x0, x1, x1, x0};
    const float cy[8] = {y0, y0, y1, y1, y0, y0, y1, y1};
    const float cz[8] = {z0, z0, z0, z0, z1, z1, z1, z1};
    float *val = o->val;
    o->skipped = o->amb = o->flipped = 0;
    o->code = o->row = 0;
    for (int i = 0; i < 8; i++) { /* :475-479: the first corner outside a constraint abandons the cell */
        if (c->ncons) {
            int w = within_constraints(c, cx[i], cy[i], cz[i]);
            if (w < 0) return -1;
            if (!w) { o->skipped = 1; return 0; } /* the reference writes nothing for such a cell; its code reads 0 here */
        }
        if (F(c, cx[i], cy[i], cz[i], &val[i])) return -1;
    }
    const float iso = c->iso;
    int code = 0; /* :497-505, strict > */
    for (int i = 0; i < 8; i++)
        if (val[i] > iso) code |= 1 << i;
    o->code = o->row = code;
    if (code == 0 || code == 255) return 0; /* :508-510 */

    int face = k_amb_face[code]; /* :523-549 */
    if (face != 0xFF) {
        o->amb = 1;
        float mx = 0, my = 0, mz = 0;
        for (int i = 0; i < 4; i++) {
            int vi = (k_face_corner[face] >> (4 * i)) & 0xF;
            mx += cx[vi];
            my += cy[vi];
            mz += cz[vi];
        }
        mx = (float)((double)mx / 4.0);
        my = (float)((double)my / 4.0);
        mz = (float)((double)mz / 4.0);
        float mid;
        if (F(c, mx, my, mz, &mid)) return -1;
        if (mid > iso) {
            o->row = 255 - code;
            o->flipped = 1;
        }
    }

    for (int e = 0; e < 12; e++) { /* :557-583 */
        int v1 = k_edge_corner[e] & 0xF, v2 = k_edge_corner[e] >> 4;
        if (((code >> v1) & 1) != ((code >> v2) & 1)) {
            o->ex[e] = interp(iso, cx[v1], cx[v2], val[v1], val[v2]);
            o->ey[e] = interp(iso, cy[v1], cy[v2], val[v1], val[v2]);
            o->ez[e] = interp(iso, cz[v1], cz[v2], val[v1], val[v2]);
        } else {
            o->ex[e] = o->ey[e] = o->ez[e] = NAN;
        }
    }
    return 0;
}

/* calculate_step for the cell with lower corner (x0,y0,z0) and upper corner (x1,y1,z1), appended to the layer: the
 * dense sweep passes lattice coordinates, seed mode passes x0 and x0 + step (marching.cpp:458-460) */
static int cell_at(orc_ctx *c, orc_layer *L, int want, float x0, float x1, float y0, float y1, float z0, float z1,
                   uint8_t *code_out) {
    orc_cellcore o;
    if (cell_core(c, x0, x1, y0, y1, z0, z1, &o)) return -1;
    *code_out = (uint8_t)o.code;
    if (o.skipped || o.code == 0 || o.code == 255) return 0;
    L->n_active++;
    L->n_amb += (uint64_t)o.amb;
    L->n_flip += (uint64_t)o.flipped;
    const int code = o.code, row = o.row;
    const float iso = c->iso;

    int nt = k_tri_count[row]; /* :586-594 */
    if (!(want & (ORC_WANT_SOUP | ORC_WANT_NORMALS))) {
        L->ntri += (size_t)nt;
        return 0;
    }
    if (layer_reserve(L, (size_t)nt, want)) return -4;
    /* ONE normal per crossed lattice edge, the gradient of F at the edge's intersection point computed in
     * the edge's +axis direction (from its lower to its upper end).  For the edges the table walks upwards (0, 1, 4, 5,
     * 8..11) that is the emitted point itself; edges 2, 3, 6, 7 run downwards and their emitted point can differ from it
     * in the last bit.  Every cell that shares the lattice edge therefore gets the same normal bits. */
    float en[12][3];
    int eok[12];
    if (want & ORC_WANT_NORMALS) {
        const float cx[8] = {x0, x1, x1, x0, x0, x1, x1, x0};
        const float cy[8] = {y0, y0, y1, y1, y0, y0, y1, y1};
        const float cz[8] = {z0, z0, z0, z0, z1, z1, z1, z1};
        const float h = 0.5f * c->step;
        for (int e = 0; e < 12; e++) {
            int v1 = k_edge_corner[e] & 0xF, v2 = k_edge_corner[e] >> 4;
            eok[e] = 0;
            if (((code >> v1) & 1) == ((code >> v2) & 1)) continue;
            float p[3] = {o.ex[e], o.ey[e], o.ez[e]};
            if (e == 2 || e == 3 || e == 6 || e == 7) { /* downward edge: the point seen from the lower end */
                p[0] = interp(iso, cx[v2], cx[v1], o.val[v2], o.val[v1]);
                p[1] = interp(iso, cy[v2], cy[v1], o.val[v2], o.val[v1]);
                p[2] = interp(iso, cz[v2], cz[v1], o.val[v2], o.val[v1]);
            }
            if (grad_normal(c, p, h, en[e], &eok[e])) return -1;
        }
    }
    for (int t = 0; t < nt; t++) {
        float P[3][3];
        int E[3];
        for (int k = 0; k < 3; k++) {
            int e = (int)((k_tri_row[row] >> (4 * (3 * t + k))) & 0xF);
            E[k] = e;
            P[k][0] = o.ex[e];
            P[k][1] = o.ey[e];
            P[k][2] = o.ez[e];
        }
        if (want & ORC_WANT_SOUP) memcpy(L->soup + 9 * L->ntri, P, sizeof(P));
        if (want & ORC_WANT_NORMALS) {
            float fn[3] = {0, 0, 0};
            int have_fn = 0;
            for (int k = 0; k < 3; k++) {
                float n[3] = {en[E[k]][0], en[E[k]][1], en[E[k]][2]};
                if (!eok[E[k]]) { /* fall back to the triangle's own normal cross(B-A, C-A) */
                    if (!have_fn) {
                        float e1[3] = {P[1][0] - P[0][0], P[1][1] - P[0][1], P[1][2] - P[0][2]};
                        float e2[3] = {P[2][0] - P[0][0], P[2][1] - P[0][1], P[2][2] - P[0][2]};
                        float cr[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2],
                                       e1[0] * e2[1] - e1[1] * e2[0]};
                        float l = sqrtf((cr[0] * cr[0] + cr[1] * cr[1]) + cr[2] * cr[2]);
                        if (l > 0.0f && !isinf(l)) {
                            const float inv = 1.0f / l;
                            fn[0] = cr[0] * inv;
                            fn[1] = cr[1] * inv;
                            fn[2] = cr[2] * inv;
                        }
                        have_fn = 1;
                    }
                    n[0] = fn[0];
                    n[1] = fn[1];
                    n[2] = fn[2];
                }
                memcpy(L->nrm + 9 * L->ntri + 3 * k, n, sizeof(n));
            }
        }
        L->ntri++;
    }
    return 0;
}

static int cell(orc_ctx *c, orc_layer *L, int want, const float *ax, int ix, int iy, int iz, uint8_t *code_out) {
    return cell_at(c, L, want, ax[ix], ax[ix + 1], ax[iy], ax[iy + 1], ax[iz], ax[iz + 1], code_out);
}

void orc_mesh_free(orc_mesh *m) {
    if (!m) return;
    free(m->codes);
    free(m->soup);
    free(m->normals);
    m->codes = NULL;
    m->soup = m->normals = NULL;
}

int orc_march(const char *eq, float step, float iso, const float scale[3], int pow_mode, int want, int z_begin,
              int z_end, int nthreads, orc_mesh *out) {
    return orc_march_constrained(eq, step, iso, scale, pow_mode, want, z_begin, z_end, nthreads, NULL, 0, out);
}

int orc_march_constrained(const char *eq, float step, float iso, const float scale[3], int pow_mode, int want,
                          int z_begin, int z_end, int nthreads, const orc_constraint *cons, int ncons, orc_mesh *out) {
    memset(out, 0, sizeof(*out));
    /* marching.cpp:226-238 set_grid_step_size accepts [0.001, 0.5] (compared in double) */
    if (!((double)step >= 0.001 && (double)step <= .5)) return -3;
    if (ncons < 0 || ncons > 3) return -1; /* marching.cpp:174 */
    orc_expr e;
    orc_expr cexp[3];
    memset(cexp, 0, sizeof(cexp));
    for (int i = 0; i < ncons; i++) /* marching.cpp:192-193 */
        if (cons[i].op < ORC_CMP_GE || cons[i].op > ORC_CMP_LT || !orc_tokenize(cons[i].lhs, &cexp[i])) {
            for (int k = 0; k < i; k++) orc_expr_free(&cexp[k]);
            return -1;
        }
    if (!orc_tokenize(eq, &e)) {
        for (int k = 0; k < ncons; k++) orc_expr_free(&cexp[k]);
        return -1;
    }
    int n1 = orc_cells_per_axis(step);
    float *ax = (float *)malloc(((size_t)n1 + 1) * sizeof(float));
    if (!ax) { orc_expr_free(&e); for (int k = 0; k < ncons; k++) orc_expr_free(&cexp[k]); return -4; }
    orc_axis_coords(step, ax, n1 + 1);
    if (z_end < 0 || z_end > n1) z_end = n1;
    if (z_begin < 0) z_begin = 0;
    if (z_begin > z_end) z_begin = z_end;
    const int nz = z_end - z_begin;
    const size_t plane = (size_t)n1 * (size_t)n1;

    out->n1 = n1;
    out->n_cells = (uint64_t)plane * (uint64_t)nz;
    if (want & ORC_WANT_CODES) {
        out->codes = (uint8_t *)malloc(out->n_cells ? out->n_cells : 1);
        if (!out->codes) { free(ax); orc_expr_free(&e); for (int k = 0; k < ncons; k++) orc_expr_free(&cexp[k]); return -4; }
    }
    orc_layer *layers = (orc_layer *)calloc((size_t)(nz > 0 ? nz : 1), sizeof(orc_layer));
    if (!layers) { free(ax); orc_expr_free(&e); for (int k = 0; k < ncons; k++) orc_expr_free(&cexp[k]); orc_mesh_free(out); return -4; }
    if (nthreads < 1) nthreads = 1;

    /* z layers are independent (marching.cpp:375 outer loop); concatenating the
     * per-layer results in z order reproduces the reference's emission order. */
#pragma omp parallel num_threads(nthreads)
    {
        orc_ctx c;
        c.e = &e;
        c.iso = iso;
        c.step = step;
        c.sx = scale[0];
        c.sy = scale[1];
        c.sz = scale[2];
        int bad = stk_init(&c.stk, &e, pow_mode);
        c.ncons = ncons;
        for (int i = 0; i < ncons; i++) {
            c.ce[i] = &cexp[i];
            c.cop[i] = cons[i].op;
            c.crhs[i] = cons[i].rhs;
            bad |= stk_init(&c.cstk[i], &cexp[i], pow_mode);
        }
        uint8_t *tmp = NULL;
        if (!(want & ORC_WANT_CODES)) tmp = (uint8_t *)malloc(plane ? plane : 1);
#pragma omp for schedule(dynamic, 1)
        for (int lz = 0; lz < nz; lz++) {
            orc_layer *L = &layers[lz];
            if (bad) { L->err = -4; continue; }
            uint8_t *cd = (want & ORC_WANT_CODES) ? out->codes + (size_t)lz * plane : tmp;
            const int iz = z_begin + lz;
            for (int iy = 0; iy < n1 && !L->err; iy++)
                for (int ix = 0; ix < n1; ix++) {
                    int r = cell(&c, L, want, ax, ix, iy, iz, &cd[(size_t)iy * n1 + ix]);
                    if (r) { L->err = r == -4 ? -4 : -2; break; }
                }
            L->codes = NULL;
            /* fingerprint of this layer's codes is folded in order below, so keep a copy if not wanted */
            if (!(want & ORC_WANT_CODES)) {
                L->codes = (uint8_t *)malloc(plane ? plane : 1);
                if (L->codes) memcpy(L->codes, cd, plane);
                else L->err = -4;
            }
        }
        free(tmp);
        stk_free(&c.stk);
        for (int i = 0; i < ncons; i++) stk_free(&c.cstk[i]);
    }

    int err = 0;
    size_t ntri = 0;
    for (int lz = 0; lz < nz; lz++) {
        if (layers[lz].err && !err) err = layers[lz].err;
        ntri += layers[lz].ntri;
    }
    if (!err) {
        if (want & ORC_WANT_SOUP) out->soup = (float *)malloc((ntri ? ntri : 1) * 9 * sizeof(float));
        if (want & ORC_WANT_NORMALS) out->normals = (float *)malloc((ntri ? ntri : 1) * 9 * sizeof(float));
        if (((want & ORC_WANT_SOUP) && !out->soup) || ((want & ORC_WANT_NORMALS) && !out->normals)) err = -4;
    }
    uint64_t hc = FNV_OFFSET, hs = FNV_OFFSET;
    size_t at = 0;
    for (int lz = 0; lz < nz; lz++) {
        orc_layer *L = &layers[lz];
        if (!err) {
            const uint8_t *cd = (want & ORC_WANT_CODES) ? out->codes + (size_t)lz * plane : L->codes;
            hc = orc_fnv1a(cd, plane, hc);
            if (want & ORC_WANT_SOUP) {
                memcpy(out->soup + 9 * at, L->soup, L->ntri * 9 * sizeof(float));
                hs = orc_fnv1a(L->soup, L->ntri * 9 * sizeof(float), hs);
            }
            if (want & ORC_WANT_NORMALS) memcpy(out->normals + 9 * at, L->nrm, L->ntri * 9 * sizeof(float));
            at += L->ntri;
            out->n_active += L->n_active;
            out->n_amb += L->n_amb;
            out->n_flipped += L->n_flip;
        }
        free(L->codes);
        free(L->soup);
        free(L->nrm);
    }
    free(layers);
    free(ax);
    orc_expr_free(&e);
    for (int k = 0; k < ncons; k++) orc_expr_free(&cexp[k]);
    if (err) {
        orc_mesh_free(out);
        return err;
    }
    out->n_tris = ntri;
    out->fnv_codes = hc;
    out->fnv_soup = (want & ORC_WANT_SOUP) ? hs : 0;
    return 0;
}


/* ------------------------------------------------------------------ seed mode (marching.cpp:42-137, :310-331)
 * The breadth-first walk itself (deque + std::set<xyz> with the tolerance comparator of marching.h:32-55) is
 * restated in mc_oracle_seed.cpp with the same containers; this is its per-cell back end. */
typedef struct {
    orc_expr e;
    orc_ctx c;
    orc_layer L;
    int want;
} orc_seed_state;

void *orc_seed_begin(const char *eq, float step, float iso, const float scale[3], int pow_mode, int want) {
    if (!((double)step >= 0.001 && (double)step <= .5)) return NULL;
    orc_seed_state *s = (orc_seed_state *)calloc(1, sizeof(*s));
    if (!s) return NULL;
    if (!orc_tokenize(eq, &s->e)) { free(s); return NULL; }
    s->c.e = &s->e;
    s->c.iso = iso;
    s->c.step = step;
    s->c.sx = scale[0];
    s->c.sy = scale[1];
    s->c.sz = scale[2];
    s->c.ncons = 0;
    s->want = want;
    if (stk_init(&s->c.stk, &s->e, pow_mode)) { orc_expr_free(&s->e); free(s); return NULL; }
    return s;
}

/* calculate_step(x0, y0, z0) + add_step_to_poly_data for one cell; *code_out = its cube code */
int orc_seed_cell(void *h, float x0, float y0, float z0, uint8_t *code_out) {
    orc_seed_state *s = (orc_seed_state *)h;
    const float st = s->c.step;
    return cell_at(&s->c, &s->L, s->want, x0, x0 + st, y0, y0 + st, z0, z0 + st, code_out); /* :458-460 */
}

void orc_seed_finish(void *h, orc_mesh *out, uint64_t n_cells) {
    orc_seed_state *s = (orc_seed_state *)h;
    memset(out, 0, sizeof(*out));
    out->n1 = orc_cells_per_axis(s->c.step);
    out->n_cells = n_cells;
    out->n_active = s->L.n_active;
    out->n_amb = s->L.n_amb;
    out->n_flipped = s->L.n_flip;
    out->n_tris = s->L.ntri;
    out->soup = s->L.soup;
    out->normals = s->L.nrm;
    if ((s->want & ORC_WANT_SOUP) && !out->soup) out->soup = (float *)malloc(36);
    if ((s->want & ORC_WANT_NORMALS) && !out->normals) out->normals = (float *)malloc(36);
    out->fnv_soup = out->soup ? orc_fnv1a(out->soup, out->n_tris * 9 * sizeof(float), FNV_OFFSET) : FNV_OFFSET;
    out->fnv_codes = FNV_OFFSET;
    stk_free(&s->c.stk);
    orc_expr_free(&s->e);
    free(s);
}

/* ------------------------------------------------------------------ one cell at a time (Step_Data, marching.h:15-23)
 * Per-cell back end of the indexed-mesh oracle (mc_oracle_weld.cpp): calculate_step for the cell with lattice indices
 * (ix, iy, iz), handing back what add_step_to_poly_data (marching.cpp:599-625) consumes. */
typedef struct {
    orc_expr e;
    orc_expr cexp[3];
    orc_ctx c;
    float *ax;
    int n1;
} orc_step_state;

void *orc_step_begin(const char *eq, float step, float iso, const float scale[3], int pow_mode, const orc_constraint *cons,
                     int ncons) {
    if (!((double)step >= 0.001 && (double)step <= .5)) return NULL;
    if (ncons < 0 || ncons > 3) return NULL;
    orc_step_state *s = (orc_step_state *)calloc(1, sizeof(*s));
    if (!s) return NULL;
    int ok = orc_tokenize(eq, &s->e);
    int nc = 0;
    for (; ok && nc < ncons; nc++)
        if (cons[nc].op < ORC_CMP_GE || cons[nc].op > ORC_CMP_LT || !orc_tokenize(cons[nc].lhs, &s->cexp[nc])) break;
    if (!ok || nc < ncons) {
        if (ok) orc_expr_free(&s->e);
        for (int k = 0; k < nc; k++) orc_expr_free(&s->cexp[k]);
        free(s);
        return NULL;
    }
    s->n1 = orc_cells_per_axis(step);
    s->ax = (float *)malloc(((size_t)s->n1 + 1) * sizeof(float));
    orc_axis_coords(step, s->ax, s->n1 + 1);
    s->c.e = &s->e;
    s->c.iso = iso;
    s->c.step = step;
    s->c.sx = scale[0];
    s->c.sy = scale[1];
    s->c.sz = scale[2];
    s->c.ncons = ncons;
    int bad = stk_init(&s->c.stk, &s->e, pow_mode);
    for (int i = 0; i < ncons; i++) {
        s->c.ce[i] = &s->cexp[i];
        s->c.cop[i] = cons[i].op;
        s->c.crhs[i] = cons[i].rhs;
        bad |= stk_init(&s->c.cstk[i], &s->cexp[i], pow_mode);
    }
    if (bad) { orc_step_end(s); return NULL; }
    return s;
}

int orc_step_cell(void *h, int ix, int iy, int iz, orc_step *out) {
    orc_step_state *s = (orc_step_state *)h;
    if (ix < 0 || iy < 0 || iz < 0 || ix >= s->n1 || iy >= s->n1 || iz >= s->n1) return -1;
    const float *ax = s->ax;
    orc_cellcore o;
    if (cell_core(&s->c, ax[ix], ax[ix + 1], ax[iy], ax[iy + 1], ax[iz], ax[iz + 1], &o)) return -2;
    memset(out, 0, sizeof(*out));
    out->skipped = o.skipped;
    out->code = o.code;
    out->row = o.row;
    if (o.skipped) return 0;
    memcpy(out->val, o.val, sizeof(o.val));
    if (o.code == 0 || o.code == 255) return 0;
    int mapper[12]; /* :557-583: intersect_coord holds the crossed edges' points in edge order */
    for (int e = 0; e < 12; e++) {
        int v1 = k_edge_corner[e] & 0xF, v2 = k_edge_corner[e] >> 4;
        mapper[e] = -1;
        if (((o.code >> v1) & 1) != ((o.code >> v2) & 1)) {
            mapper[e] = out->n_points;
            out->edge[out->n_points] = e;
            out->point[out->n_points][0] = o.ex[e];
            out->point[out->n_points][1] = o.ey[e];
            out->point[out->n_points][2] = o.ez[e];
            out->n_points++;
        }
    }
    out->n_tris = k_tri_count[o.row]; /* :586-594 */
    for (int k = 0; k < 3 * out->n_tris; k++) out->tri_vlist[k] = mapper[(k_tri_row[o.row] >> (4 * k)) & 0xF];
    return 0;
}

int orc_step_n1(void *h) { return ((orc_step_state *)h)->n1; }

void orc_step_end(void *h) {
    orc_step_state *s = (orc_step_state *)h;
    if (!s) return;
    stk_free(&s->c.stk);
    for (int i = 0; i < s->c.ncons; i++) stk_free(&s->c.cstk[i]);
    orc_expr_free(&s->e);
    for (int i = 0; i < 3; i++) orc_expr_free(&s->cexp[i]);
    free(s->ax);
    free(s);
}
