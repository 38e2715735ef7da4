/*
 * mc_oracle.h -- CPU ORACLE for the marching-cubes hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is a plain-C restatement of the reference's algorithm
 * (raineyeh/Marching-Cube-for-Implicit-Surfaces: Source/evaluator.cpp,
 * Source/marching.cpp, Source/marching_lookup.h).  Every function cites the
 * reference file:line it follows.  Nothing under oracle/ is part of the
 * product: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load it, and only as the checker / the timed CPU baseline.  The
 * product path (libmc_hip.so) never links, loads or calls this code.
 *
 * Parity pin: the reference itself cannot be built in this image (both hot
 * path TUs #include <windows.h>, marching.cpp:3 / evaluator.cpp:2, and the
 * rules forbid stand-in headers), so the oracle is pinned against
 *   - the reference's own tokenizer self-test (evaluator.h:67-77),
 *   - the known-answer counts and FNV-1a fingerprints of per-cell cube codes
 *     and triangle soup that SURVEY.md section 4 recorded from the unmodified
 *     reference (10 configurations), and
 *   - the sha256 of the case tables, re-checked
 *     against marching_lookup.h compiled in place by oracle/Makefile (_ref).
 * See tests/test_oracle_pins.py.
 */
#ifndef MC_ORACLE_H
#define MC_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* How `^` is evaluated (reference: evaluator.cpp:133, pow(float,float) -> powf). */
enum {
    ORC_POW_LIBM = 0,  /* faithful: libm powf, what the reference executes            */
    ORC_POW_EXACT = 1  /* device-matching: constant integer exponents |n|<=16 become   */
                       /* an IEEE product chain, everything else (float)pow(double)    */
};

/* token types, evaluator.h:21 */
enum { ORC_OP = 0, ORC_NUM, ORC_VAR, ORC_BRAC_O, ORC_BRAC_C, ORC_NEG, ORC_FUNC /* extension E1 only */ };

typedef struct {
    int type;  /* ORC_* */
    char ch;   /* operator / bracket / variable character */
    float num; /* ORC_NUM: strtof of the literal (evaluator.cpp:82 does stof per call) */
} orc_token;

typedef struct {
    int n;
    int cap;
    orc_token *tok;
} orc_expr;

/* Grammar extension E1 -- sin(...) / cos(...), NOT part of the reference (which rejects them): bit 0 enables it for
 * every following call; default 0.  Returns the previous value.  Arithmetic: include/mc_trig.h. */
unsigned orc_set_extensions(unsigned ext);

/* evaluator.cpp:139-237.  Returns 1 (accepted) or 0 (rejected), like the reference. */
int orc_tokenize(const char *eq, orc_expr *out);
void orc_expr_free(orc_expr *e);

/* evaluator.cpp:53-107 + :22-48 + :111-136.  Returns 0 on success, -1 when the
 * reference would read below its stacks (undefined behaviour there). */
int orc_evaluate(const orc_expr *e, float x, float y, float z, int pow_mode, float *out);

/* marching.cpp:372-383: number of lower-corner iterations per axis for a step,
 * and the shared coordinate table c[0..n1] (c[i+1] = c[i] + step in float). */
int orc_cells_per_axis(float step);
int orc_axis_coords(float step, float *c, int cap);

enum {
    ORC_WANT_CODES = 1,   /* raw cube_code per cell (before the ambiguity redirect) */
    ORC_WANT_SOUP = 2,    /* 9 floats / triangle, emission order                    */
    ORC_WANT_NORMALS = 4  /* 9 floats / triangle, gradient normals   */
};

typedef struct {
    int n1;              /* cells per axis */
    uint64_t n_cells;    /* cells swept (n1*n1*(z_end-z_begin)) */
    uint64_t n_active;   /* cells with code not in {0,255} */
    uint64_t n_tris;
    uint64_t n_amb;      /* cells whose code is an ambiguous case */
    uint64_t n_flipped;  /* ... of which the alternative row was taken */
    uint64_t fnv_codes;  /* FNV-1a 64 of the code bytes, sweep order */
    uint64_t fnv_soup;   /* FNV-1a 64 of the little-endian float32 soup bytes */
    uint8_t *codes;      /* n_cells bytes or NULL */
    float *soup;         /* 9*n_tris floats or NULL */
    float *normals;      /* 9*n_tris floats or NULL */
} orc_mesh;

/* marching.cpp:368-384 (sweep) + :456-595 (calculate_step) + :437-446 (interp)
 * + :209-224 (scaling).  z_begin/z_end select a contiguous range of cell layers
 * (pass 0,-1 for all).  nthreads<=1 -> the reference's single-threaded sweep;
 * >1 -> z layers are processed by that many threads and concatenated in order
 * (results identical).  Returns 0, or -1 parse error, -2 evaluation underflow,
 * -3 bad step, -4 out of memory. */
int orc_march(const char *eq, float step, float iso, const float scale[3], int pow_mode,
              int want, int z_begin, int z_end, int nthreads, orc_mesh *out);
void orc_mesh_free(orc_mesh *m);

/* Constraints (marching.h:58-69, marching.cpp:173-200, :255-280): up to three `lhs op rhs` predicates; a cell
 * any of whose 8 corners is outside one of them is skipped (marching.cpp:476) -- no triangles, and its code
 * byte reads 0 here (the reference computes none).  lhs is evaluated at the scaled point like the surface. */
enum { ORC_CMP_GE = 0, ORC_CMP_LE = 1, ORC_CMP_GT = 2, ORC_CMP_LT = 3 };
typedef struct {
    const char *lhs;
    int op; /* ORC_CMP_* */
    float rhs;
} orc_constraint;
int orc_march_constrained(const char *eq, float step, float iso, const float scale[3], int pow_mode, int want,
                          int z_begin, int z_end, int nthreads, const orc_constraint *cons, int ncons, orc_mesh *out);

/* Seed mode (Marching::seed_mode / set_seed / the seed branch of recalculate, marching.cpp:42-137, :310-331): the mesh
 * of the cells reached from the cell that contains `seed` by crossing faces that carry an intersection, in the
 * reference's breadth-first order and with its float cell positions (-1 + floor(d)*step for the start, +-step per move)
 * and its tolerance set of visited positions.  No constraints.  out->n_cells = cells visited.  Returns 0, -1 parse error,
 * -2 evaluation underflow, -3 bad step, -5 seed outside [-1,1]^3 (set_seed refuses it, marching.cpp:125-137). */
int orc_march_seed(const char *eq, float step, float iso, const float scale[3], int pow_mode, int want, const float seed[3],
                   orc_mesh *out);
/* per-cell back end of the walk (mc_oracle_seed.cpp drives it) */
void *orc_seed_begin(const char *eq, float step, float iso, const float scale[3], int pow_mode, int want);
int orc_seed_cell(void *h, float x0, float y0, float z0, uint8_t *code_out);
void orc_seed_finish(void *h, orc_mesh *out, uint64_t n_cells);

/* One cell at a time: calculate_step (marching.cpp:456-595) for the cell with lattice indices (ix, iy, iz), as the
 * reference's Step_Data (marching.h:15-23) records it.  Back end of orc_march_indexed. */
typedef struct {
    int skipped;          /* a corner is outside a constraint (marching.cpp:476): nothing else is filled in */
    int code, row;        /* cube code; table row used after the ambiguity test */
    float val[8];         /* corner_values */
    int n_points;         /* crossed edges */
    int edge[12];         /* their edge numbers, ascending (edge_list) */
    float point[12][3];   /* intersect_coord */
    int n_tris;
    int tri_vlist[15];    /* indices into point[] */
} orc_step;
void *orc_step_begin(const char *eq, float step, float iso, const float scale[3], int pow_mode, const orc_constraint *cons,
                     int ncons);
int orc_step_cell(void *h, int ix, int iy, int iz, orc_step *out); /* 0, -1 bad index, -2 evaluation underflow */
int orc_step_n1(void *h);
void orc_step_end(void *h);

/* The reference's indexed mesh: Marching::add_step_to_poly_data / add_point / add_triangle (marching.cpp:599-654) over
 * the full sweep (:372-383), welding through std::set<xyz> with the tolerance comparator of marching.h:32-55, and the
 * drawer's CalculateNormal (normal.h:3-41) on the result.  z_begin/z_end as in orc_march (a slab is welded on its own). */
typedef struct {
    uint64_t n_verts, n_tris;
    float *vertex_list;     /* 3 * n_verts */
    uint32_t *tri_list;     /* 3 * n_tris (a NaN point keeps index -1, marching.cpp:611-613) */
    float *normals;         /* 3 * n_verts, CalculateNormal */
} orc_indexed;
int orc_march_indexed(const char *eq, float step, float iso, const float scale[3], int pow_mode, const orc_constraint *cons,
                      int ncons, int z_begin, int z_end, orc_indexed *out);
void orc_indexed_free(orc_indexed *m);

/* FNV-1a 64 (offset 1469598103934665603, prime 1099511628211), SURVEY.md section 4. */
uint64_t orc_fnv1a(const void *p, size_t n, uint64_t h);

/* packed case tables (include/mc_tables_data.h) exposed for the table tests */
const uint64_t *orc_tri_rows(void);
const uint8_t *orc_tri_counts(void);
const uint8_t *orc_amb_faces(void);
const uint16_t *orc_face_corners(void);
const uint8_t *orc_edge_corners(void);

#ifdef __cplusplus
}
#endif
#endif
