/*
 * mc_hip.h -- C ABI of libmc_hip.so, the MI355X (gfx950) marching-cubes hot path.
 *
 * Drop-in boundary for ONE path of raineyeh/Marching-Cube-for-Implicit-Surfaces:
 * the full-sweep branch of Marching::recalculate() and everything it calls
 * (Source/marching.cpp:368-384 -> calculate_step :456-595 -> Evaluator::evaluate
 * Source/evaluator.cpp:53-107, tables Source/marching_lookup.h).  The reference
 * has no FFI; its boundary is the public surface of `class Marching`
 * (Source/marching.h:72-157) and `class Evaluator` (Source/evaluator.h:24-86).
 * Each entry point below names the reference member it replaces.  A C++ facade
 * with the reference's class and method names is include/mc_marching.hpp; the
 * binding a maintainer would add is shown in INTEGRATION.md.
 *
 * Plain C types only: pointers, sizes, POD structs.  All compute runs on the
 * GPU (hand-written HIP kernels, specialised per equation with hiprtc); there
 * is no CPU fallback -- every call fails with MC_ERR_HIP when no gfx950 device
 * or no HIP runtime is present.
 */
#ifndef MC_HIP_H
#define MC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MC_ABI_VERSION 4

/* Status codes (the reference returns bool / throws std::exception without text:
 * evaluator.cpp:10-13, marching.cpp:226-238). */
enum {
    MC_OK = 0,
    MC_ERR_PARSE = 1,     /* Evaluator::set_equation would return false (evaluator.cpp:15-17)      */
    MC_ERR_EVAL = 2,      /* tokenizer accepts but Evaluator::evaluate reads below its stacks --   */
                          /* undefined behaviour in the reference (e.g. "x+"), refused here        */
    MC_ERR_STEP = 3,      /* Marching::set_grid_step_size would return false (marching.cpp:226)    */
    MC_ERR_ARG = 4,       /* null pointer / bad range / bad flag                                    */
    MC_ERR_HIP = 5,       /* HIP runtime, hiprtc or device error; text in mc_last_error()          */
    MC_ERR_NOMEM = 6,     /* device or host allocation failed                                       */
    MC_ERR_OVERFLOW = 7   /* more than 2^32-1 triangles in one slab: split with z_begin/z_end      */
};

/* Flags for mc_params.flags */
enum {
    MC_FLAG_NORMALS = 1u,      /* fill the normal half of each vertex (gradient of f);  */
                               /* without it normals are written as 0                                 */
    MC_FLAG_KEEP_CODES = 2u,   /* keep the per-cell cube codes readable through mc_copy_codes()       */
    MC_FLAG_NO_EMIT = 4u,      /* no triangle soup (cube codes, counts and -- if asked for -- the     */
                               /* indexed mesh only)                                                  */
    MC_FLAG_TILE1 = 8u,        /* diagnostic: classify with row tiles of height 1 (no sample reuse)   */
    MC_FLAG_INDEXED = 32u,     /* also build the reference's indexed Poly_Data on the GPU: welded      */
                               /* vertex_list / tri_list (marching.cpp:599-654, marching.h:32-55) and  */
                               /* the drawer's area-weighted vertex normals (normal.h:3-41).           */
                               /* DEVIATION: a closed form of the reference's std::set welding -- bit  */
                               /* for bit its mesh unless two points are closer than its 1e-6 tolerance */
                               /* WITHOUT being bit-identical; such groups are always merged here,     */
                               /* while the reference's (non-transitive) comparator sometimes keeps    */
                               /* them apart (4 of 3 392 vertices on x^2+y^2-0.5 at grid_res 300, see  */
                               /* the note under mc_copy_indexed and DESIGN.md section 4)              */
    MC_FLAG_NO_CULL = 64u,     /* diagnostic: classify every row by sampling (no interval culling);   */
                               /* the output must not change                                          */
    MC_FLAG_NO_TIMING = 128u,  /* no per-kernel hipEvents (ms_* stay 0): four fewer nodes per sweep   */
    MC_FLAG_EMIT_DIRECT = 256u,/* diagnostic: force the emit kernel that computes every output vertex  */
                               /* on its own (the default for cheap f) ...                             */
    MC_FLAG_EMIT_SHARED = 512u,/* ... or the one that computes each lattice-edge vertex once per chunk */
                               /* of cells (the default for expensive f); the output must not change   */
    MC_FLAG_TILE63 = 2048u,    /* diagnostic: classify with 63-row tiles whatever the grid size (small  */
                               /* grids get shorter tiles by default); the output must not change       */
    MC_FLAG_SEAM = 1024u,      /* with MC_FLAG_INDEXED on a Z slab: weld the slab as a PART OF THE WHOLE     */
                               /* GRID, so that the slabs' vertex_list / tri_list, concatenated in slab order, */
                               /* are the single sweep's Poly_Data bit for bit (see mc_index_rebase)          */
    MC_FLAG_BATCH = 4096u,     /* the caller keeps SEVERAL sweeps in flight (one context each): choose for     */
                               /* throughput where that differs from the fastest single sweep -- on small grids */
                               /* the emit kernel of an expensive f otherwise puts 4 waves on each group of     */
                               /* cells, which shortens one sweep by 8-20 % and costs a batch 9 %.  The output  */
                               /* does not change                                                              */
    MC_FLAG_ORDER_Z = 8192u,   /* diagnostic: classify launches the slab's layers in z order ...                */
    MC_FLAG_ORDER_MIDDLE_OUT = 16384u,/* ... or from the slab's middle outwards (by default the library times  */
                               /* both orders on an equation's first sweeps and keeps the faster); the output  */
                               /* must not change                                                              */
    /* Cold start.  Evaluator::set_equation is instant in the reference (evaluator.cpp:15-17); specialising the kernels   */
    /* for an equation takes hiprtc about a second.  By default the FIRST sweeps of an equation that neither this process */
    /* nor $MC_JIT_CACHE has compiled run on an ahead-of-time build of the same kernels in which f is an interpreter of    */
    /* the equation's DAG -- the same float operations in the same order, the same output bits -- while a host thread      */
    /* compiles; later sweeps switch to the specialised kernels by themselves.  (Environment: MC_COLD_START=jit restores   */
    /* "wait for hiprtc" process-wide.)                                                                                    */
    MC_FLAG_INTERP = 32768u,   /* diagnostic: sweep with the interpreter build whatever is cached; the output must not  */
                               /* change                                                                                 */
    MC_FLAG_NO_INTERP = 65536u /* this sweep waits for the specialised kernels (what mc_graph_build always does)         */
};
#define MC_FLAG_LAYER_ORDER_MASK (MC_FLAG_ORDER_Z | MC_FLAG_ORDER_MIDDLE_OUT)

typedef struct mc_context mc_context; /* one per GPU: stream, buffers, compiled-equation cache */

/* Parameters of one sweep.  Field <- reference setter it mirrors. */
typedef struct mc_params {
    const char *equation; /* Evaluator::set_equation(string)            evaluator.cpp:15            */
    float step;           /* Marching::set_grid_step_size(float)        marching.cpp:226, [0.001,.5] */
    float iso;            /* Marching::set_surface_constant(float)      marching.cpp:149            */
    float scale[3];       /* Marching::set_scaling_{x,y,z}(float)       marching.cpp:240-251        */
    uint32_t flags;       /* MC_FLAG_*                                                              */
    int32_t z_begin;      /* first cell layer of this slab (0-based); Z-sharding across GPUs        */
    int32_t z_end;        /* one past the last layer; <0 means "to the end" (whole grid = 0,-1)     */
} mc_params;

/* Result of one sweep.  Device buffers are owned by the context and stay valid
 * until the next mc_march()/mc_context_destroy() on it (the reference's
 * Poly_Data pointer has the same lifetime rule: marching.cpp:293-305, 656-658). */
typedef struct mc_result {
    int32_t cells_per_axis; /* n1 = loop trip count of marching.cpp:375-377 (grid_res N -> N+1)     */
    int32_t z_begin, z_end; /* slab actually swept                                                   */
    uint64_t n_cells;       /* n1*n1*(z_end-z_begin)                                                */
    uint64_t n_active;      /* cells whose cube code is neither 0 nor 255                            */
    uint64_t n_tris;        /* triangles emitted, reference emission order (z, y, x, table order)    */
    const float *d_vertices;/* device: n_tris*3 vertices * 6 floats {x,y,z,nx,ny,nz}; 72 B/triangle  */
    const uint8_t *d_codes; /* device: raw cube codes of cells x < code_main_cells, row pitch         */
                            /* code_pitch, (z-z_begin, y) rows; mc_copy_codes() gives all n1 per row */
    uint64_t code_pitch;    /* bytes between consecutive (z,y) rows of d_codes                       */
    float ms_classify;      /* GPU time of the classify kernel (hipEvent, ms)                        */
    float ms_scan;          /* ... of the triangle-count scan                                        */
    float ms_emit;          /* ... of the emit kernel                                                */
    float ms_total;         /* first kernel start -> last kernel end                                 */
    int32_t code_main_cells;/* cells per row held by d_codes (n1, or n1 - (1..4) on 2^k+1 grids)     */
    const uint32_t *d_codes_tail; /* device, or NULL: one dword per (z,y) row with the codes of cells */
                            /* code_main_cells..n1-1 (byte k = cell code_main_cells + k)              */
    /* MC_FLAG_INDEXED: Poly_Data (marching.h:26-30) in device memory, the slab welded on its own      */
    uint64_t n_verts;       /* vertex_list.size() / 3                                                 */
    const float *d_vertex_list;    /* n_verts * 3 floats, the reference's order of first insertion    */
    const uint32_t *d_tri_list;    /* n_tris * 3 indices into it, reference emission order            */
    const float *d_vertex_normals; /* n_verts * 3 floats: CalculateNormal (normal.h:3-41)             */
    float ms_index;         /* GPU time of the indexed-mesh kernels (hipEvent, ms)                   */
    const uint64_t *d_totals; /* device: {n_tris, n_active} of the sweep, valid in stream order -- lets */
                            /* a multi-GPU host exchange the counts (RCCL) without a host round trip  */
    int32_t emit_shared;    /* which emit kernel ran: 1 = mc_emit (vertices shared inside a chunk of   */
                            /* cells, the choice for expensive f), 0 = mc_emit_direct                  */
    int32_t interpreted;    /* 1: this sweep ran on the interpreter build of the kernels (cold start,   */
                            /* MC_FLAG_INTERP); 0: on the kernels specialised for the equation          */
} mc_result;

/* -- library ------------------------------------------------------------- */
int mc_abi_version(void);
const char *mc_last_error(void);  /* thread-local text of the last failing call ("" if none) */
int mc_device_count(void);        /* gfx950 devices visible to HIP; 0 when none / no driver   */

/* -- expression layer (Evaluator) ---------------------------------------- */
/* Grammar extensions, process-wide, default 0 = exactly the reference's grammar (evaluator.cpp:139-237 rejects every
 * letter but x, y, z).  MC_EXT_TRIG adds `sin(...)` and `cos(...)` (include/mc_trig.h defines their arithmetic; there is
 * no reference result to match) so that BASELINE.json's gyroid workload can be written.  Affects every
 * entry point that takes an equation or a constraint.  Returns the previous setting. */
#define MC_EXT_TRIG 1u
unsigned mc_set_extensions(unsigned ext);
/* Thread-safety contract: mc_set_extensions is ONE atomic word for the whole process, read by every later call that parses
 * an equation; it is meant to be set once at start-up.  A process whose contexts need different grammars pins each one
 * with mc_context_set_extensions(ctx, ext) -- ext >= 0 applies to every equation / constraint this context compiles from
 * then on, whatever the process-wide word says; ext < 0 returns the context to following it.  (The context-free helpers
 * mc_expr_check / mc_expr_validate / mc_expr_dump / mc_jit_precompile always use the process-wide word.)  A context itself
 * is not re-entrant, like the reference's Marching object: one thread at a time per context. */

/* Evaluator::set_equation / tokenize accept-reject only (evaluator.cpp:139-237): 1 accept, 0 reject. */
int mc_expr_check(const char *equation);
/* Compile to the evaluation DAG the reference's two-stack walk performs (evaluator.cpp:22-107).
 * Returns MC_OK / MC_ERR_PARSE / MC_ERR_EVAL.  Pure host work, no GPU needed. */
int mc_expr_validate(const char *equation);
/* Writes the generated device code (HIP source text of mc_f, then -- for equations that are finite on the unit-scale
 * domain -- its interval enclosure mc_f_iv and, when f has expensive sub-expressions of y alone, the staged form
 * mc_f_iv_y / mc_f_iv_rest the classify walk uses) into buf; returns its full length.
 * Lets a maintainer audit the evaluation order (e.g. x-y-z becomes x-(y-z)). */
size_t mc_expr_dump(const char *equation, char *buf, size_t cap);
/* Diagnostic: interpret the compiled DAG on the host for ONE point (same op order and float ops
 * as the device code, P1 power rule).  Used by the CPU-side tests of the compiler; the product
 * never calls it -- Evaluator::evaluate maps to mc_eval_points (GPU). */
int mc_expr_debug_eval_host(const char *equation, float x, float y, float z, float *out);
/* The same point through the PROGRAM the interpreter build of the kernels would run for this equation (MC_FLAG_INTERP and
 * the cold-start note there): the DAG as one word per operation with liveness-allocated registers, walked on the host
 * exactly as the device walks it.  MC_ERR_ARG when the equation does not fit the interpreter's tables (it then always
 * waits for hiprtc).  CPU-side tests compare it with mc_expr_debug_eval_host bit for bit. */
int mc_expr_debug_interp_host(const char *equation, float x, float y, float z, float *out);

/* Specialise the kernels for `equation` with hiprtc (gfx950 code object) WITHOUT touching a GPU:
 * pre-populates the on-disk cache named by $MC_JIT_CACHE and lets a build machine check that the
 * generated code compiles.  code_size (optional) receives the code-object size in bytes. */
int mc_jit_precompile(const char *equation, size_t *code_size);

/* -- context -------------------------------------------------------------- */
int mc_context_create(int device, mc_context **out);
void mc_context_destroy(mc_context *ctx);
int mc_context_set_extensions(mc_context *ctx, int ext);

/* Evaluator::evaluate(x,y,z) for n points (evaluator.cpp:53): xyz = n*3 host floats, out = n host floats. */
int mc_eval_points(mc_context *ctx, const char *equation, const float *xyz, size_t n, float *out);

/* -- the sweep: Marching::recalculate() full-sweep branch (marching.cpp:368-384) ---------- */
int mc_march(mc_context *ctx, const mc_params *p, mc_result *res);
/* north-star convenience: march(equation, grid_res, iso) with step = 2.0f/grid_res, scale 1. */
int mc_march_simple(mc_context *ctx, const char *equation, int grid_res, float iso, uint32_t flags, mc_result *res);

/* Copy results of the last mc_march() to host memory.
 * mc_copy_vertices: n_tris*18 floats (positions+normals interleaved).
 * mc_copy_soup:     n_tris*9 floats, positions only -- the reference's pre-dedup triangle soup
 *                   (Step_Data::intersect_coord[tri_vlist[k]], marching.cpp:586-594).
 * mc_copy_codes:    n_cells bytes, compact, sweep order x-fastest. */
int mc_copy_vertices(mc_context *ctx, float *host, uint64_t max_tris);
int mc_copy_soup(mc_context *ctx, float *host, uint64_t max_tris);
/* the other half of mc_copy_vertices: n_tris*9 floats, the unit normals (MC_FLAG_NORMALS) in the same order */
int mc_copy_soup_normals(mc_context *ctx, float *host, uint64_t max_tris);
int mc_copy_codes(mc_context *ctx, uint8_t *host, uint64_t max_bytes);
/* mc_copy_indexed: the indexed mesh of the last MC_FLAG_INDEXED sweep: vertex_list (n_verts*3 floats), tri_list
 * (n_tris*3 uint32) and the area-weighted vertex normals (n_verts*3 floats); any of the three may be NULL.
 * The welding rule (checked against a replay of the reference's std::set, tests/test_indexed.py, tests/weld_model.py): a
 * vertex's KEY is the lattice corner it sits on when its lattice edge's intersection point -- computed from the edge's
 * lower end, whatever direction the owning cell's table walks it in -- lies within 1e-6 of an end of that edge, else the
 * lattice edge itself; the first cell of the sweep that produces a key owns the vertex and its position is the one THAT
 * cell computes.  (Lower-end and upper-end interpolation differ by an ulp at most, far below the 1e-6 threshold; a point
 * exactly at the threshold could be keyed differently from the reference's comparison of the stored points -- one of the
 * near-tie cases of the DEVIATION noted at MC_FLAG_INDEXED, pinned by
 * tests/test_indexed.py::test_indexed_known_deviation_from_the_std_set_is_pinned.) */
int mc_copy_indexed(mc_context *ctx, float *vertex_list, uint32_t *tri_list, float *normals, uint64_t max_verts,
                    uint64_t max_tris);
/* One Poly_Data across Z slabs (marching.h:26-30 for a grid swept in parts, one slab per GPU).  Without MC_FLAG_SEAM a slab
 * is welded on its own: the vertices on the plane it shares with the slab below exist in both.  With MC_FLAG_INDEXED |
 * MC_FLAG_SEAM the slab is welded as the single sweep would weld it: a vertex belongs to the FIRST cell of the whole grid's
 * sweep that produces it (marching.cpp:627-643), so the vertices on a slab's lower plane belong to the slab below and are
 * not in this slab's vertex_list; n_verts counts the slab's own vertices, and the normals of the vertices on its upper
 * plane include the triangles of the slab above (normal.h:3-41).  No data is exchanged for this: the slab sweeps one ghost
 * layer below and one above its own layers (f is analytic) and hands out its own part.  tri_list comes back relative to the
 * slab's first own vertex -- index i < n_verts is the slab's i-th vertex, a vertex of the slab below reads as i - 2^32 < 0
 * (the j-th vertex from the END of that slab's list is -j) -- and
 *     mc_index_rebase(ctx, offset)     offset = number of vertices owned by all slabs below (the one figure the ranks
 *                                      exchange: an all-gather of n_verts, like the triangle counts)
 * makes every entry index the concatenation of the slabs' vertex lists.  The offset is a STATE of the sweep's tri_list,
 * not an increment: calling it again with the same offset changes nothing (a retry after a failed exchange is safe), with
 * another offset the list is re-targeted; a new sweep starts from offset 0.  Before the call (offset 0) an entry that
 * refers to a vertex of the slab below reads as a wrapped negative number -- mc_copy_indexed hands it out as it is.
 * The whole grid in one slab: MC_FLAG_SEAM changes nothing. */
int mc_index_rebase(mc_context *ctx, uint64_t vertex_offset);

/* Constraints: Marching::set_constraint0..2(lhs, op, rhs) (Source/marching.h:105-108, marching.cpp:173-200) and
 * use_constraint0..2(bool) (marching.h:110-113, marching.cpp:202-207).  i in 0..2; op is one of ">=", "<=", ">", "<";
 * lhs is checked like an equation.  A constraint takes part in the following sweeps of this context once it is both
 * set and in use (marching.cpp:258): a cell any of whose 8 corners has `lhs(scale*corner) op rhs` false (NaN counts
 * as false) is skipped (marching.cpp:476) -- it emits no triangles and its code byte reads 0.  The reference's
 * set_constraint falls off its end without a return value on success (marching.cpp:199-200); here it is MC_OK. */
int mc_set_constraint(mc_context *ctx, int i, const char *lhs, const char *op, float rhs);
/* Seed mode: Marching::set_seed(x,y,z) (marching.cpp:125-137: refused outside [-1,1]^3) and seed_mode(bool) (:115-118).
 * While it is on, a sweep keeps only the triangles of the cells the reference's walk visits (marching.cpp:42-101,
 * :310-331): those reached from the cell containing the seed across faces that carry an intersection, never past the
 * last cell whose centre lies inside [-1,1] (:84-86).  Differences, DESIGN.md: cells are the dense sweep's lattice cells
 * (the reference re-derives their positions as -1 + k*step, a few ulp off), and the triangles come in sweep order, not in
 * breadth-first order.  (That set is a connected component of the surface cells; it is labelled on the device by a
 * union-find over the sweep's records, not walked.)  Whole-grid sweeps only (z_begin 0, z_end -1); capturable by
 * mc_graph_build.  With MC_FLAG_INDEXED the component is welded like the dense sweep (only visited cells insert
 * vertices, marching.cpp:310-331): the reference numbers vertices and triangles in VISITATION order and keeps the position
 * computed by the first visited cell of each vertex; here both follow the sweep order -- the same triangles over the same
 * welded points within the reference's own 1e-6 tolerance, in another order. */
int mc_set_seed(mc_context *ctx, float x, float y, float z);
int mc_seed_mode(mc_context *ctx, int on);
int mc_use_constraint(mc_context *ctx, int i, int use);

/* marching.cpp:372-377: trip count of `for (v=-1.0f; v <= (float)(1.0+0.5*step); v += step)`. 0 if step rejected. */
int mc_cells_per_axis(float step);

/* -- steady-state replay (animated iso sweep): capture classify->scan->emit once as a hipGraph
 *    and replay it with a new iso value per frame (Marching::set_surface_constant + recalculate).
 *    The captured graph belongs to the sweep mc_graph_build was given (equation, step, scale, flags, slab and the
 *    constraints in force then).  Any later call that changes what it depends on -- an mc_march with other parameters
 *    that re-targets or re-allocates a buffer, mc_set_constraint / mc_use_constraint -- makes the next replay re-capture
 *    the ORIGINAL sweep first (never another equation's kernels on this one's buffers).  MC_FLAG_INDEXED (with or without
 *    MC_FLAG_SEAM) is captured too: its five kernels follow the sweep's in the graph, the index buffers get head room at
 *    capture time, and a frame that outgrows them is run again by mc_graph_wait / mc_graph_replay.  Seed mode is captured
 *    too (the component-labelling kernels become nodes of the graph; the seed is part of the capture: mc_set_seed /
 *    mc_seed_mode make the next replay re-capture). */
int mc_graph_build(mc_context *ctx, const mc_params *p);
int mc_graph_replay(mc_context *ctx, float iso, mc_result *res);
/* The same without the host round trip: enqueue one replay and return; mc_graph_wait blocks until everything enqueued
 * has run and reports the LAST replay (a replay that outgrew the vertex buffer writes nothing past it; mc_graph_wait then
 * re-runs that last frame with a larger buffer, like mc_graph_replay does). */
int mc_graph_replay_async(mc_context *ctx, float iso);
/* (Streams of independent sweeps -- animation frames, parameter studies -- run 10-90 % faster with two or three sweeps in
 * flight: one context per in-flight sweep, frame k on context k % D, mc_graph_wait on a context just before it is used
 * again.  The kernels of one sweep depend on each other; a neighbouring sweep fills their ramps and tails.  INTEGRATION.md.) */
int mc_graph_wait(mc_context *ctx, mc_result *res);
/* The HIP stream (hipStream_t) every kernel of this context runs on, for callers that order their own work after a sweep. */
void *mc_stream(mc_context *ctx);

/* -- multi-device sweep: Marching::recalculate() (marching.cpp:368-384) with the cell layers cut into one contiguous Z slab
 *    per device (SURVEY 8b "device list", 8e).  The sweep is z-major (marching.cpp:375), so the slabs' triangle lists
 *    concatenated in slab order ARE the single sweep's list; f is analytic, every device evaluates its own top sample plane
 *    and no halo exists.  The only figures that cross devices are the per-slab counts, from which each slab gets its offset.
 *    Two forms: ONE process driving several devices (mc_march_sharded: a context per device, host threads, the counts are
 *    plain host reads), and one process PER device (mc_comm_* + mc_march_rank: the counts travel by an RCCL all-gather over
 *    xGMI; librccl.so is loaded on first use, a process that never calls mc_comm_* does not need it). */

/* Layers [*z_begin, *z_end) of slab `part` when layers [0, n_layers) are cut into `parts` near-equal contiguous slabs
 * (the first n_layers % parts slabs get one layer more). */
void mc_shard_layers(int n_layers, int parts, int part, int *z_begin, int *z_end);

typedef struct mc_shard {
    int32_t z_begin, z_end;  /* cell layers this slab swept                                                           */
    uint64_t tri_offset;     /* triangles of all slabs below it = index of its first triangle in the whole grid's list */
    uint64_t vert_offset;    /* MC_FLAG_INDEXED: vertices owned by all slabs below it; ALREADY added to its tri_list   */
                             /* (mc_index_rebase), which therefore indexes the concatenated vertex_list                */
    uint64_t n_tris_total;   /* the whole grid's counts (the same in every entry)                                      */
    uint64_t n_verts_total;
} mc_shard;

/* One process, n contexts (one per device of the caller's device list; a device may appear more than once -- its contexts'
 * sweeps then overlap on it).  p describes the WHOLE sweep (its z_begin / z_end give the range that is cut, normally 0, -1);
 * bounds: NULL = near-equal slabs (mc_shard_layers), else n + 1 ascending layer indices (a caller that balances slabs by
 * measured cost passes its own).  Slab i runs on ctxs[i] -- all slabs at once, one host thread each -- and leaves results[i]
 * exactly as mc_march would; shards[i] (optional) receives its offsets.  With MC_FLAG_INDEXED every slab is welded as a part
 * of the whole grid (MC_FLAG_SEAM is implied) and re-based, so the slabs' vertex_list / tri_list / normals concatenated in
 * slab order are the single sweep's Poly_Data bit for bit.  Constraints and seed settings are per context: set them on
 * every context (mc_set_constraint, mc_set_seed, mc_seed_mode).  Seed mode follows one connected component through the
 * whole grid (marching.cpp:310-331), which needs every layer on one device: with seed mode on (on ctxs[0]) the whole range
 * is swept by ctxs[0] and the other contexts get EMPTY slabs -- same results layout, same mesh as mc_march.  On an error
 * the first failing slab's code and text are reported. */
int mc_march_sharded(mc_context *const *ctxs, int n, const mc_params *p, const int32_t *bounds, mc_result *results,
                     mc_shard *shards);
/* The slabs' results of the last mc_march_sharded on these contexts, concatenated in slab order into host memory: what
 * mc_copy_vertices / mc_copy_indexed / mc_copy_codes give for a single sweep.  Any output pointer may be NULL. */
int mc_copy_sharded_vertices(mc_context *const *ctxs, int n, float *host, uint64_t max_tris);
int mc_copy_sharded_indexed(mc_context *const *ctxs, int n, float *vertex_list, uint32_t *tri_list, float *normals,
                            uint64_t max_verts, uint64_t max_tris);
int mc_copy_sharded_codes(mc_context *const *ctxs, int n, uint8_t *host, uint64_t max_bytes);

/* One process per device.  A communicator spans `world` processes; rank 0 makes an id (mc_comm_get_id: ncclGetUniqueId),
 * the application hands its 128 bytes to the other ranks by whatever channel it has (MPI, a file, torch.distributed's
 * store), and every rank calls mc_comm_create with its context (ncclCommInitRank on the context's device).
 * MC_ERR_HIP when librccl.so cannot be loaded or RCCL reports an error. */
#define MC_COMM_ID_BYTES 128
typedef struct mc_comm mc_comm;
int mc_comm_get_id(uint8_t id[MC_COMM_ID_BYTES]);
int mc_comm_create(mc_context *ctx, const uint8_t id[MC_COMM_ID_BYTES], int world, int rank, mc_comm **out);
void mc_comm_destroy(mc_comm *comm);
/* This rank's slab of the sweep p describes (layers cut by mc_shard_layers(., world, rank), or by `bounds` -- world + 1
 * ascending layer indices, the same on every rank), then ONE ncclAllGather of {n_tris, n_verts} (16 bytes per rank) on a
 * side stream ordered behind the sweep, from which `shard` gets this rank's offsets and the totals; with MC_FLAG_INDEXED
 * the slab is welded with MC_FLAG_SEAM and its tri_list re-based.  Collective: every rank of the communicator calls it. */
int mc_march_rank(mc_comm *comm, const mc_params *p, const int32_t *bounds, mc_result *res, mc_shard *shard);
/* The steady-state form of the same exchange, without the host in the loop (a caller that replays a captured sweep,
 * mc_graph_replay_async, once per frame): mc_comm_gather_async enqueues -- behind everything already enqueued on the
 * context's stream -- a copy of the sweep's device-side {n_tris, n_active} words (mc_result.d_totals) and their all-gather
 * on the communicator's side stream, and orders the context's NEXT sweep behind the copy (its scan clears those words);
 * mc_comm_wait blocks until every gather enqueued so far is done and hands out the LAST one: counts[2 * r] = n_tris and
 * counts[2 * r + 1] = n_active of rank r (2 * world entries).  d_totals: the sweep's mc_result.d_totals. */
int mc_comm_gather_async(mc_comm *comm, const uint64_t *d_totals);
int mc_comm_wait(mc_comm *comm, uint64_t *counts);

#ifdef __cplusplus
}
#endif
#endif /* MC_HIP_H */
