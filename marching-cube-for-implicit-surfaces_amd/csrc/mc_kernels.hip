// mc_kernels.hip -- hand-written gfx950 (MI355X, wave64) kernels of the marching-cubes sweep.
//
// This file is compiled at run time by hiprtc, once per equation: the block between the
// MC_F markers below is replaced by the device function the expression compiler generates
// (mc_expr.cpp: emit_hip), so f(x,y,z) is straight-line VALU code inside the sweep and the
// compiler hoists every sub-expression that does not depend on the walk direction out of
// the inner loop.  It also compiles stand-alone (hipcc -c, sample f = unit sphere) so the
// kernels can be inspected / syntax-checked without a GPU.  MUST be built with
// -ffp-contract=off: cube codes are only bit-exact if every float op rounds once
// (SURVEY.md section 0 item 10).
//
// Replaces, in the reference: the z/y/x loop of Marching::recalculate
// (Source/marching.cpp:372-383), Marching::calculate_step (:456-595), Marching::interp
// (:437-446), Marching::evaluate (:209-224) and Evaluator::evaluate (Source/evaluator.cpp:53).
//
// Data layout in HBM (all owned by the context, mc_runtime.cpp):
//   axis   float[n1+1]        lattice coordinate c[i] (c[0]=-1, c[i+1]=c[i]+step, float adds)
//   axs    float[3][n1+1]     scale_x*c[i], scale_y*c[i], scale_z*c[i] (marching.cpp:211)
//   codes  u8, pitched        raw cube code per cell; row = (z-z_begin)*n1 + y, pitch % 128 == 0
//   segcb  uint2[nseg]        per SEGMENT (= 256 x-consecutive cells of one row; seg = row*nchunk + chunk):
//                             {triangles | active cells << 16, index of its first record in recs}
//   recs   u32[cap_recs]      one RECORD per active cell: cell | code<<8 | flip<<16 | triangles<<17 |
//                             triangle prefix inside the segment<<20.  DENSE: a tile collects its records in
//                             LDS and appends them with one atomic bump of rec_cursor, so a segment's records
//                             are contiguous and the whole buffer is 4 B per active cell.  (A 1 KB slot per
//                             segment -- 5 GB of address space at 1025^3, touched 12 bytes at a time -- cost
//                             0.10 ms in mc_classify alone: measured by confining the writes to a window.)
//   grpsum u64[ngroups]       per GROUP of 64 consecutive segments: triangles | active cells << 32
//                             (64-bit atomic adds by mc_classify, one per run of a tile's rows that share a group;
//                             left zero by the previous sweep's scan)
//   grpoff uint2[ngroups+1]   exclusive scan of grpsum: {triangle offset, active-cell offset}
//   tail   u32[rows]          only when the last chunk is 1..4 cells wide (n1 = 2^k+1): those cells' codes,
//                             one dword per row, instead of a 128-byte line per row in `codes`
//   verts  float[T][3][6]     {x,y,z,nx,ny,nz} per vertex, 72 B per triangle, reference order
#define MC_TRIG_FN __device__ __forceinline__
#ifndef MC_JIT
#include <hip/hip_runtime.h>
#include "../../include/mc_tables_data.h"
#include "../../include/mc_trig.h"
#else
#include "mc_tables_data.h"
#include "mc_trig.h"
#endif

typedef unsigned long long u64;
typedef unsigned int u32;
typedef unsigned char u8;
typedef u32 u32x4_t __attribute__((ext_vector_type(4)));
typedef u32 u32x2_t __attribute__((ext_vector_type(2)));

// ------------------------------------------------------------------ power rule P1
// `^` in the reference is pow(float,float) -> libm powf (evaluator.cpp:133).  Literal
// integer exponents become an IEEE product chain (double, one final rounding to float;
// n == 2 is emitted as a plain float multiply, which is the same value); everything else
// is (float)pow(double,double).  See DESIGN.md "P1" for how this relates to glibc's powf.
template <int N>
__device__ __forceinline__ float mc_pow_int(float a) {
    constexpr int M = N < 0 ? -N : N;
    const double p = (double)a;
    double r = p;
#pragma unroll
    for (int i = 2; i <= M; ++i) r = r * p;
    if (N < 0) r = 1.0 / r;
    return (float)r;
}
// not inlined: the double-precision pow body is ~1k instructions; f may call it several times and
// the kernels evaluate f at dozens of sites
__device__ __attribute__((noinline)) float mc_pow_general(float a, float b) { return (float)pow((double)a, (double)b); }

// ------------------------------------------------------------------ extension E1: enclosures of sin / cos
// [lo, hi] contains every value mc_sinf COMPUTES on [l, h].  mc_sinf is within 1e-7 (measured 9.3e-8) of
// the true sine and never exceeds 1 in magnitude (it clamps), and the true sine is monotone between extrema
// (pi/2 + n*pi): if no extremum can lie in [l, h] the computed endpoint values, widened by 3e-7,
// bound the range; an extremum that MAY lie inside contributes its +-1.  "May": the test runs in float (double
// instructions run at a fraction of the float rate, and this function is what an interval evaluation of a trigonometric
// f mostly consists of) with a guard band that is wider than the rounding of its three float operations -- a widened
// test only ever adds a bound of +-1, never drops one: a true extremum n in [l, h] has ka <= n <= kb; with kb > ka both
// bounds are set, with kb == ka == n the parity is n's.  Arguments are finite and below 8192 here (finite_on_domain).
__device__ __forceinline__ void mc_trig_iv(float l, float h, float shift, int which, float& lo, float& hi) {
    // extrema of sin at (n + 1/2) pi, of cos at n pi: maxima for even n, minima for odd n
    const float a = l * 0.318309886183790671538f - shift, b = h * 0.318309886183790671538f - shift;
    const float eps = 1e-6f * (1.0f + __builtin_fabsf(a) + __builtin_fabsf(b));  // |a|, |b| < 2609: their error is below 2e-7 (1 + |a|)
    const float ka = __builtin_ceilf(a - eps), kb = __builtin_floorf(b + eps);
    const float vl = mc_trig_eval(l, which), vh = mc_trig_eval(h, which);
    lo = __builtin_fminf(vl, vh) - 3e-7f;
    hi = __builtin_fmaxf(vl, vh) + 3e-7f;
    if (kb >= ka) {
        const bool two = kb > ka;
        const bool even = (((int)ka) & 1) == 0;
        if (two || even) hi = 1.0f;
        if (two || !even) lo = -1.0f;
    }
}
__device__ __forceinline__ void mc_sin_iv(float l, float h, float& lo, float& hi) { mc_trig_iv(l, h, 0.5f, 0, lo, hi); }
__device__ __forceinline__ void mc_cos_iv(float l, float h, float& lo, float& hi) { mc_trig_iv(l, h, 0.0f, 1, lo, hi); }

//@@MC_F_BEGIN  (replaced by generated code when JIT-compiled)
#ifdef MC_INTERP
// ------------------------------------------------------------------ the interpreter build (cold start)
// Evaluator::set_equation is instant in the reference (evaluator.cpp:15-17); specialising this file for an equation takes
// hiprtc about a second.  So the library also carries this file compiled AHEAD of time with f as an INTERPRETER of the same
// DAG (build.py: -DMC_INTERP): the first sweeps of an equation nobody has compiled yet run on it while a host thread
// compiles the specialised kernels (mc_runtime.hip: get_compiled).  The program is the DAG in topological order, one word
// per operation, in constant memory; every lane walks the same words, so the walk is scalar loads and uniform branches,
// and the values live in a small register file indexed by uniform numbers.  The same float operations on the same
// operands in the same order as the generated mc_f (no contraction in this build either): the same bits
// (tests: MC_FLAG_INTERP).  What the interpreter build does not have is everything that is generated PER equation beside
// mc_f -- enclosures (it takes the sampling walk), staged / tabulated forms -- and the finiteness proof (it keeps the NaN
// bookkeeping).
#define MC_INTERP_MAXI 120   // operations of one program
#define MC_INTERP_REGS 16    // live values (the host allocates registers by liveness: mc_runtime.hip, interp_program)
#define MC_INTERP_CONSTS 64
struct McInterpProg {
    u32 n;        // operations
    u32 root;     // the result's operand byte when n == 0 (f is a variable or a constant)
    u32 cons_op;  // constraint programs: 0 '>=', 1 '<=', 2 '>', 3 '<'
    float cons_rhs;
    u32 code[MC_INTERP_MAXI];  // op | dst << 8 | a << 16 | b << 24; operand byte: 0..63 register, 64 / 65 / 66 = x / y / z,
                               // 128 + k = constant k; POWI: b = exponent + 32
    float cval[MC_INTERP_CONSTS];
};
struct McInterpAll {
    u32 ncons, pad[3];
    McInterpProg f, g[3];  // f and the left-hand sides of the enabled constraints
};
__device__ __constant__ McInterpAll c_interp;

__device__ __forceinline__ float mc_pow_int_rt(float a, int n) {  // mc_pow_int<N> with N at run time: the same chain
    const int m = n < 0 ? -n : n;
    const double p = (double)a;
    double r = p;
    for (int i = 2; i <= m; ++i) r = r * p;
    if (n < 0) r = 1.0 / r;
    return (float)r;
}
// The register file lives in LDS, one column per thread of the workgroup (the largest workgroup of this file has 512
// threads: mc_emit_direct): a register number is uniform, so an access is one conflict-free ds_read / ds_write at
// [register][thread].  (In private memory -- a dynamically indexed array goes to scratch -- every operand was a trip to
// memory: the interpreter build swept 1025^3 cells in 45 ms; the values are only live inside one call of mc_f.)  The
// interpreter build's workgroups have at most 256 threads (build.py: MC_WPB_C = MC_WPB_E = 4, the index kernels' 4).
#define MC_INTERP_THREADS 256
static_assert(64 * MC_WPB_C <= MC_INTERP_THREADS && 64 * MC_WPB_E <= MC_INTERP_THREADS && 64 * MC_WPB_ES <= MC_INTERP_THREADS,
              "the interpreter's register file has one column per thread of a workgroup");
__device__ __forceinline__ float mc_interp_run(const McInterpProg& P, float x, float y, float z) {
    __shared__ float s_ir[MC_INTERP_REGS + 3][MC_INTERP_THREADS];  // the registers, then x, y, z (a variable is an operand like a register)
    float* r = &s_ir[0][threadIdx.x & (MC_INTERP_THREADS - 1)];
#define MC_IR(i) r[(i) * MC_INTERP_THREADS]
    MC_IR(MC_INTERP_REGS + 0) = x;
    MC_IR(MC_INTERP_REGS + 1) = y;
    MC_IR(MC_INTERP_REGS + 2) = z;
    auto fetch = [&](u32 o) -> float {
        if (o < 128u) return MC_IR(o < 64u ? (o & (MC_INTERP_REGS - 1)) : MC_INTERP_REGS + ((o - 64u) & 3u) % 3u);
        return P.cval[(o - 128u) & (MC_INTERP_CONSTS - 1)];
    };
    float last = fetch(P.root);
    const u32 n = P.n;
    for (u32 i = 0; i < n; ++i) {
        const u32 w = P.code[i];
        const u32 op = w & 0xFFu, d = (w >> 8) & 0xFFu, ao = (w >> 16) & 0xFFu, bo = w >> 24;
        const float a = fetch(ao);
        float v;
        if (op == 8u) v = a * a;                                   // POWI 2
        else if (op == 9u) v = mc_pow_int_rt(a, (int)bo - 32);     // POWI n
        else if (op == 5u) v = -a;
        else if (op == 6u) v = mc_sinf(a);
        else if (op == 7u) v = mc_cosf(a);
        else {
            const float b = fetch(bo);
            v = op == 0u ? a + b : op == 1u ? a - b : op == 2u ? a * b : op == 3u ? a / b : mc_pow_general(a, b);
        }
        MC_IR(d & (MC_INTERP_REGS - 1)) = v;
        last = v;
    }
    return last;
#undef MC_IR
}
__device__ __forceinline__ float mc_f(float x, float y, float z) { return mc_interp_run(c_interp.f, x, y, z); }
#define MC_CONS 1
// marching.cpp:255-280 check_constraints at a (scaled) point; NaN fails, as in C++
__device__ __forceinline__ bool mc_ok(float x, float y, float z) {
    bool ok = true;
    const u32 nc = c_interp.ncons;
    for (u32 i = 0; i < nc; ++i) {
        const float g = mc_interp_run(c_interp.g[i], x, y, z);
        const u32 op = c_interp.g[i].cons_op;
        const float rhs = c_interp.g[i].cons_rhs;
        ok = ok & (op == 0u ? g >= rhs : op == 1u ? g <= rhs : op == 2u ? g > rhs : g < rhs);
    }
    return ok;
}
#else
__device__ __forceinline__ float mc_f(float x, float y, float z) {
    const float t0 = z * z;
    const float t1 = t0 - 1.0f;
    const float t2 = y * y;
    const float t3 = t2 + t1;
    const float t4 = x * x;
    return t4 + t3;  // x^2+(y^2+(z^2-1)): the reference's right-to-left reduction
}
#define MC_FINITE 1
#define MC_HAVE_IV 1
// enclosure of every value mc_f computes on a box (generated by mc_expr.cpp: emit_hip_interval)
__device__ __forceinline__ void mc_f_iv(float xl, float xh, float yl, float yh, float zl, float zh, float& lo, float& hi) {
    const float za = __builtin_fabsf(zl), zb = __builtin_fabsf(zh), ya = __builtin_fabsf(yl), yb = __builtin_fabsf(yh);
    const float xa = __builtin_fabsf(xl), xb = __builtin_fabsf(xh);
    const float zm = __builtin_fmaxf(za, zb), zn = (zl <= 0.0f && zh >= 0.0f) ? 0.0f : __builtin_fminf(za, zb);
    const float ym = __builtin_fmaxf(ya, yb), yn = (yl <= 0.0f && yh >= 0.0f) ? 0.0f : __builtin_fminf(ya, yb);
    const float xm = __builtin_fmaxf(xa, xb), xn = (xl <= 0.0f && xh >= 0.0f) ? 0.0f : __builtin_fminf(xa, xb);
    lo = xn * xn + (yn * yn + (zn * zn - 1.0f));
    hi = xm * xm + (ym * ym + (zm * zm - 1.0f));
}
#ifdef MC_CHECK_CONS  // stand-alone compile check of the constraint paths: x > -0.5
#define MC_CONS 1
__device__ __forceinline__ bool mc_ok(float x, float y, float z) { return x > -0.5f; }
__device__ __forceinline__ void mc_ok_iv(float xl, float xh, float yl, float yh, float zl, float zh, bool& allok, bool& dead) {
    allok = xl > -0.5f;
    dead = !(xh > -0.5f);
}
#endif
#endif  // MC_INTERP
//@@MC_F_END

// ------------------------------------------------------------------ parameters
struct McParams {
    const float* axis;  // [n1+1]
    const float* axs;   // [3][n1+1]
    u64 pitch;          // bytes per code row
    int n1;             // cells per axis
    int nchunk;         // ceil(n1/256) segments per row
    int z_begin;        // first cell layer of the slab
    int nz;             // layers in the slab
    int tile_h;         // classify: rows per wave tile (1..63)
    int ntile_y;        // ceil(n1/tile_h)
    u32 nseg;           // nz*n1*nchunk
    u32 flags;          // MC_FLAG_*
    float iso, step;
    float sx, sy, sz;
    int own_z_lo;       // indexed mesh: cell layers >= own_z_lo exist when the OWNER of a vertex is looked for.  z_begin: the slab
                        // is welded on its own; 0 (MC_FLAG_SEAM): as a part of the whole grid -- layers below the swept range are
                        // there (their cells own the keys they share with it) although they have no record here
    u64 cap_tris;       // capacity of the vertex buffer in triangles
    u32* codes_tail;    // tail plane: one dword per (z,y) row = codes of cells main_cells..n1-1 (when tail_cells)
    int main_cells;     // cells per row held by the pitched code plane (n1 when there is no tail plane)
    int tail_cells;     // 0, or 1..4: width of the last chunk when it is handled by tail tiles
    int nchunk_main;    // chunks swept by the 256-wide tiles (nchunk, or nchunk-1 with a tail plane)
    int ntile_t;        // tail tiles per layer: ceil(n1/64), one row per lane
    u32* rec_cursor;    // record allocator, clean at the start of every sweep: word 0 = "a region overflowed", then
                        // MC_NCUR bump cursors 128 bytes apart (word 32*(1+k)), one per region of recs
    u64 cap_recs;       // capacity of the record buffer in records (MC_NCUR equal regions)
    const u32* overflow;  // the overflow word for the kernels BEHIND the scan (the scan moves it into the sweep's totals and
                          // clears the allocator for the next sweep)
    const float* tab;     // MC_TAB: f's one-variable sub-expressions per lattice index (mc_tabulate), else unused
    int nz_a;             // classify: the slab's first nz_a layers are cut into tiles of tile_h rows, the others -- the LAST tiles
    int tile_h2;          // of the launch -- into tiles of tile_h2 <= tile_h rows: the waves that start last are the ones the
    int ntile_y2;         // chip drains on, and a short tile is a short drain (nz_a == nz: one height).  ceil(n1 / tile_h2)
    int layer_order;      // classify: 0 = the layers in z order, 1 = from the slab's middle outwards (mc_runtime tries both and keeps
                          // the faster one per equation and grid)
};

#define MC_SEG 256          // cells per segment (4 per lane)
#define MC_REC_CAP 512      // records a wave buffers in LDS before appending them to the global array
// The record array is split into MC_NCUR regions, each with its own bump cursor; tile t appends to region
// t % MC_NCUR.  One cursor for the whole sweep serialised 87 k atomics on one address (measured: +0.08 ms).
#define MC_NCUR 256
// waves per workgroup.  A workgroup's slot is held until its slowest wave is done, and tiles / groups differ a
// lot in work, so small workgroups keep more waves resident (mc_runtime passes the same numbers to the launch).
#ifndef MC_WPB_C
#define MC_WPB_C 4
#endif
#ifndef MC_WPB_E
#define MC_WPB_E 4   // mc_emit_direct
#endif
#ifndef MC_WPB_ES
#define MC_WPB_ES 1  // mc_emit: its waves are independent (no table in LDS, no barrier); a workgroup keeps its LDS until its
                     // slowest wave is done, so one wave per workgroup wastes none
#endif
// Cache policy of the stores that write PROVEN code rows (buffer-store aux bits: 1 = sc0, 2 = nt, 16 = sc1): ~1 GB per
// 1025^3 sweep that nothing reads again.  Marked non-temporal they leave the write path a little sooner (classify of the
// 1025^3 sphere, three runs each in one session: 0.235-0.241 ms with 0, 0.228-0.231 with nt, the same with sc0 | nt but the
// emit kernel behind it 2 % slower; equation_3 at 1025^3: 0.237 -> 0.217).  NOT for the undecided rows: those are stored
// whole and their listed dwords written over them a moment later, which only works while the line sits in L2 (with nt
// there: 0.32 ms); nor for the emit kernels' vertex stores (two pieces per 24-byte vertex that L2 merges: 0.25 -> 0.32 ms).
#ifndef MC_CULLED_STORE_AUX
#define MC_CULLED_STORE_AUX 2
#endif
#ifndef MC_LIST_CAP
// triangles staged per wave in the emit kernel: one chunk of 64 records holds at most 64 * 5.  The kernel
// is latency-bound, so LDS is kept small for occupancy (measured: 0.274 ms at 768, 0.269 at 384, 0.256 at 320)
#define MC_LIST_CAP 320
#endif

__device__ __constant__ u64 c_tri_row[256] = MC_TRI_ROW_INIT;       // marching_lookup.h:64-320, nibble-packed
// per (row, edge) one byte: bit t = triangle t of the row has a corner on that edge -- derived from the rows at compile time
// (mc_vnormal: which of a cell's triangles touch a vertex's lattice edge)
struct McEdgeTri {
    u32 w[256 * 12 / 4];
    constexpr McEdgeTri() : w{} {
        constexpr u64 rows[256] = MC_TRI_ROW_INIT;
        for (int i = 0; i < 256 * 12; ++i) {
            const u64 tr = rows[i / 12];
            const u32 e = (u32)(i % 12);
            u32 m = 0u;
            for (int t = 0; t < 5; ++t) {
                const u32 trip = (u32)(tr >> (12 * t)) & 0xFFFu;  // (0xF ends the row: no edge has that number)
                if ((trip & 15u) == e || ((trip >> 4) & 15u) == e || (trip >> 8) == e) m |= 1u << t;
            }
            w[i / 4] |= m << (8 * (i % 4));
        }
    }
};
__device__ __constant__ McEdgeTri c_edgetri = McEdgeTri();
__device__ __constant__ u8 c_tri_count[256] = MC_TRI_COUNT_INIT;
__device__ __constant__ u8 c_amb_face[256] = MC_AMB_FACE_INIT;      // :329-587 (alt row is always 255-c)
__device__ __constant__ unsigned short c_face_corner[6] = MC_FACE_CORNER_INIT;  // :25-32
__device__ __constant__ u8 c_edge_corner[12] = MC_EDGE_CORNER_INIT; // :10-23

// corner i of a cell (marching.cpp:471-472): x offset bit, y offset bit, z offset bit
__device__ __forceinline__ int cx_bit(int v) { return (0x66 >> v) & 1; }
__device__ __forceinline__ int cy_bit(int v) { return (0xCC >> v) & 1; }
__device__ __forceinline__ int cz_bit(int v) { return v >> 2; }

// Marching::evaluate (marching.cpp:209-224): f(scale_x*x, scale_y*y, scale_z*z)
__device__ __forceinline__ float mc_F(const McParams& p, float x, float y, float z) {
#ifdef MC_UNIT_SCALE  // all three scale factors are exactly 1.0f (the default): 1.0f * x == x bit for bit
    (void)p;
    return mc_f(x, y, z);
#else
    return mc_f(p.sx * x, p.sy * y, p.sz * z);
#endif
}

// ------------------------------------------------------------------ MC_TAB: tabulated one-variable sub-expressions
// Equations with expensive sub-expressions of ONE variable (sin / cos, divisions, higher powers; mc_expr.cpp:
// emit_hip_tabulated) come with mc_f_ux / uy / uz (those sub-expressions) and mc_f_t (f from coordinates and their
// values).  At lattice coordinates such a sub-expression has n1 + 1 distinct values per axis, so mc_tabulate evaluates
// them once per (equation, lattice): table (v, w, k)[i] = sub-expression k of variable v at the argument mc_F passes for
// lattice index i -- w = 0: s_v * c[i];  w = 1: s_v * (c[i] + h);  w = 2: s_v * (c[i] - h), h = step / 2, the arguments
// of the normal's central differences (float operations exactly as mc_F and mc_grad_normal perform them, so a table
// value IS the value the kernels would have computed).  mc_classify's back-end and mc_emit read instead of evaluating:
// of the 13 sin / cos argument reductions a gyroid vertex costs, 3 are left (the interpolated coordinate and its +- h).
#ifdef MC_TAB
#define MC_TAB_NK (MC_TAB_NX > MC_TAB_NY ? (MC_TAB_NX > MC_TAB_NZ ? MC_TAB_NX : MC_TAB_NZ) : (MC_TAB_NY > MC_TAB_NZ ? MC_TAB_NY : MC_TAB_NZ))
__device__ __forceinline__ u32 mc_tab_stride(int n1) { return ((u32)n1 + 1u + 63u) & ~63u; }
// table of sub-expression k of variable v, argument kind w
__device__ __forceinline__ const float* mc_tab_ptr(const float* tab, u32 stride, int v, int w, int k) {
    return tab + (u32)((v * 3 + w) * MC_TAB_NK + k) * stride;
}
#ifndef MC_ONLY_INDEX_KERNELS  // (the module of the index kernels, compiled when MC_FLAG_INDEXED is first used, leaves the sweep kernels out)
extern "C" __global__ __launch_bounds__(256) void mc_tabulate(const McParams* __restrict__ P, float* __restrict__ tab) {
    const McParams p = *P;
    const int i = (int)(blockIdx.x * 256u + threadIdx.x);
    if (i > p.n1) return;
    const u32 stride = mc_tab_stride(p.n1);
    const float c = p.axis[i], h = 0.5f * p.step;
#pragma unroll
    for (int w = 0; w < 3; ++w) {
        const float cc = w == 0 ? c : w == 1 ? c + h : c - h;
#ifdef MC_UNIT_SCALE
        const float ax = cc, ay = cc, az = cc;
#else
        const float ax = p.sx * cc, ay = p.sy * cc, az = p.sz * cc;
#endif
        float UX[MC_TAB_NX], UY[MC_TAB_NY], UZ[MC_TAB_NZ];
        mc_f_ux(ax, UX);
        mc_f_uy(ay, UY);
        mc_f_uz(az, UZ);
#pragma unroll
        for (int k = 0; k < MC_TAB_NX; ++k) tab[(u32)((0 * 3 + w) * MC_TAB_NK + k) * stride + (u32)i] = UX[k];
#pragma unroll
        for (int k = 0; k < MC_TAB_NY; ++k) tab[(u32)((1 * 3 + w) * MC_TAB_NK + k) * stride + (u32)i] = UY[k];
#pragma unroll
        for (int k = 0; k < MC_TAB_NZ; ++k) tab[(u32)((2 * 3 + w) * MC_TAB_NK + k) * stride + (u32)i] = UZ[k];
    }
}
#endif  // MC_ONLY_INDEX_KERNELS
#endif

// Wavefront (64-lane) inclusive prefix sum in 7 DPP adds: row_shr 1,2,3 / 4 / 8 inside the
// 16-lane rows, then row_bcast15 / row_bcast31 across rows (gfx9 wave64 DPP controls).
__device__ __forceinline__ u32 wave_inclusive_scan(u32 v) {
    u32 x = v;
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x113, 0xf, 0xf, false);
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xe, false);
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xc, false);
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false);
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false);
    return x;
}
// same ladder with max instead of add (values are unsigned; 0 is the identity)
__device__ __forceinline__ u32 wave_inclusive_max(u32 v) {
    u32 x = v;
    x = max(x, (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false));
    x = max(x, (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false));
    x = max(x, (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x113, 0xf, 0xf, false));
    x = max(x, (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xe, false));
    x = max(x, (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xc, false));
    x = max(x, (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false));
    x = max(x, (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false));
    return x;
}
// per-lane select by a wave mask held in an SGPR pair: one v_cndmask, no exec juggling
__device__ __forceinline__ u32 select_by_mask(u64 m, u32 if_set, u32 if_clear) {
    // readfirstlane folds away when the mask already lives in SGPRs; it matters when register
    // pressure (e.g. around the non-inlined pow call) made the compiler keep it in VGPRs
    const u64 mm = ((u64)(u32)__builtin_amdgcn_readfirstlane((int)(u32)(m >> 32)) << 32) |
                   (u32)__builtin_amdgcn_readfirstlane((int)(u32)m);
    u32 r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(if_clear), "v"(if_set), "s"(mm));
    return r;
}
// number of set bits of a wave mask below this lane
__device__ __forceinline__ u32 mask_rank(u64 m) {
    return __builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0u));
}
__device__ __forceinline__ float readlane_f(float v, int l) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}
__device__ __forceinline__ float max3f(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); }
__device__ __forceinline__ float min3f(float a, float b, float c) { return __builtin_fminf(__builtin_fminf(a, b), c); }

// Ambiguity test of calculate_step (marching.cpp:523-549): sample f at the centre of the
// listed face; true = take the alternative row 255-code.
// ux / uy / uz point at the unscaled coordinates of the cell's lower corner (global table or an
// LDS copy of the tile's slice); element [1] is the upper corner.
template <typename PX, typename PY, typename PZ>
__device__ __forceinline__ bool amb_flip(const McParams& p, int face, PX ux, PY uy, PZ uz) {
    // the 6 x 16-bit face-corner table lives in two 64-bit immediates, not in memory: a global load here
    // (even one lane's) would make the wave wait for every code store it still has in flight -- vmcnt
    // retires in order -- and that serialised the whole back-end behind the walk's stores (measured:
    // compute 0.19 ms + stores 0.19 ms = 0.38 ms instead of the larger of the two)
    constexpr unsigned short kfc[6] = MC_FACE_CORNER_INIT;
    constexpr u64 kfc_lo = (u64)kfc[0] | ((u64)kfc[1] << 16) | ((u64)kfc[2] << 32) | ((u64)kfc[3] << 48);
    constexpr u64 kfc_hi = (u64)kfc[4] | ((u64)kfc[5] << 16);
    const u32 fc = (u32)((face < 4 ? kfc_lo >> (16 * face) : kfc_hi >> (16 * (face - 4))) & 0xFFFFull);
    float mx = 0.0f, my = 0.0f, mz = 0.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int v = (fc >> (4 * i)) & 0xF;
        mx += ux[cx_bit(v)];
        my += uy[cy_bit(v)];
        mz += uz[cz_bit(v)];
    }
    mx *= 0.25f;  // == (float)((double)m / 4.0): an exact scaling
    my *= 0.25f;
    mz *= 0.25f;
    return mc_F(p, mx, my, mz) > p.iso;
}

// =============================================================== K1: classify
// One wave = one tile: a 256-cell x-chunk (4 consecutive cells per lane) of one z layer, up to 63 rows high
// (or, for the 1..4-cell last chunk of a 2^k+1 grid, 64 rows with one ROW per lane: "tail tiles").
//
//  1. Row test, lane = row: one interval evaluation (mc_f_iv, generated from the same DAG as mc_f) over the box
//     of each row proves most rows all-above / all-below iso without evaluating a sample.
//  2. The rows are handled class by class (set-bit iteration over 64-bit masks: the CU has ONE scalar unit and a
//     per-row decision tree cost more scalar instructions than the work): aligned blocks of 4 proven rows ->
//     one dwordx4 store per lane; other proven rows -> one dword store; undecided rows -> one more interval
//     evaluation per LANE (its 4 cells), giving the lanes proven all-above and the lanes still undecided
//     ("listed"); those two masks are parked in lane `row` of four VGPRs (v_writelane).
//  3. Back-end (mc_backend), lane-parallel over the listed dwords: exact evaluation of the 20 lattice samples of
//     each, cube codes, triangle counts (LDS LUT), ambiguity test, per-segment prefix sums, one RECORD per active
//     cell (buffered in LDS, appended to the dense record array with one scalar atomic per tile), and the
//     undecided code rows written whole.
//
// Equations whose values cannot be bounded (possible NaN / inf: the host decides, finite_on_domain) take the
// sampling walk instead of 1-2: the previous row's samples stay in registers (2 evaluations per cell) and rows /
// lanes are classified by a min/max tree over the lane's 8 new samples; the back-end is the same.
//
// What bounds it (measured; 1025^3 sphere: 1.21 GB of stores, 76 M vector + 67 M scalar
// wave-instructions): no single unit.  Its waves wait for data 48 % of their time (half of that in the prologue, whose
// loads queue behind the kernel's own stores), wait for their turn to issue 23 %, execute 30 %; the vector ALU is busy
// 57 % -- all of it while back-ends run -- and the store path's FIFOs are full 1 % of the time.  It needs its 5 waves
// per SIMD (4: +16 %) and gains nothing from 6.  The rules kept throughout: no global LOAD after the first store (vmcnt
// retires in order: a load would wait for every store in flight), whole 128-byte lines only, as few store INSTRUCTIONS as
// possible (dwordx4 over 4 rows), wave-uniform work on the scalar unit only where it is cheaper than on the VALU.
#ifndef MC_CLASSIFY_MINW
#define MC_CLASSIFY_MINW 1
#endif
#ifndef MC_BLOCK_MIN_ROWS
#define MC_BLOCK_MIN_ROWS 8  // undecided rows a tile must have for the block-level evaluation (MC_BLOCK_LEVEL) to be worth one more
#endif

struct McTileCtx {
    int ch, y0, iz, lz, lane;
    u32 region;  // this tile's region of the record array
    u64 seg0;  // segment index of tile row 0; + nchunk per row
};

// LDS copies of the tile's coordinate-table slices for the back-end (see mc_backend)
struct McTileLds {
    const float* xs;   // [261] scaled x of samples X0 .. X0+256 (clamped to n1), then padding
    const float* ux;   // [261] unscaled
    const float* ys;   // [65]  scaled y of sample rows y0 .. y0+64 (clamped)
    const float* uy;   // [65]  unscaled
    float zk, zk1, uz0, uz1;
#ifdef MC_TAB  // f's one-variable sub-expressions at those samples (mc_tabulate): [k][264] for x, [k][72] for y, planes z / z+1
    const float* tx;
    const float* ty;
    float tz0[MC_TAB_NZ], tz1[MC_TAB_NZ];
#endif
};

// Row state the walk leaves behind for the back-end, lane j = tile row j (v_writelane by the walk):
// the lanes of row j whose 4 cells are listed (mix) and those proven all-above iso (all).
struct McRowMasks {
    u32 mixlo, mixhi, alllo, allhi;
};

// index of the k-th (0-based) set bit of hi:lo; k < popcount
__device__ __forceinline__ int nth_set_bit64(u32 lo, u32 hi, u32 k) {
    u32 c = (u32)__builtin_popcount(lo);
    const bool up = k >= c;
    u32 m = up ? hi : lo;
    k = up ? k - c : k;
    int base = up ? 32 : 0;
#pragma unroll
    for (int w = 16; w >= 1; w >>= 1) {
        c = (u32)__builtin_popcount(m & ((1u << w) - 1u));
        const bool u = k >= c;
        m = u ? (m >> w) : m;
        k = u ? k - c : k;
        base += u ? w : 0;
    }
    return base;
}

// Back-end of a tile, lane-parallel over its listed dwords ("entries", ordered by (row, lane)): exact
// evaluation of the entry's 20 lattice samples, cube codes, triangle counts, ambiguity test, records and
// per-segment counts; then the code rows the walk left pending are stored whole.
//
// The entry -> (row, lane) map is computed here, per 64-entry chunk, from the per-row masks: an
// exclusive scan of the rows' popcounts (lane = row) gives each row's first entry; the rows drop a
// marker at that position of a 64-slot LDS strip and a max-scan spreads it; the lane inside the row is
// the k-th set bit of the row's mask.  (Staging (row, lane) pairs from the walk, one row at a time,
// cost ~25 scalar + ~10 vector instructions per row; the walk is bound by the CU's one scalar unit.)
// A chunk ends at a row boundary (a row has at most 64 entries), so each row's dwords sit in one chunk's
// registers when its code row is assembled.
//
// It issues NO vector-memory load: the coordinates it needs come from LDS copies of the tile's table
// slices (tl) made before the walk's first store.  vmcnt retires in order, so a load issued here would
// have to wait for every code store the wave still has in flight.
//
// TAIL = true: tail tile (lane = row, one dword per row, chunk lane 0): the dwords go to tailbuf[row]
// (LDS) for the tile's one coalesced store instead of into whole code rows.
// Records: collected in LDS (recbuf, MC_REC_CAP) in entry order -- segment by segment -- and appended to the
// dense global array by flush_records(): one atomic bump per tile (rarely more), coalesced copy.  Returns in
// lane j the global index of row j's first record.
template <bool TAIL = false>
__device__ __forceinline__ u32 mc_backend(const McParams& p, const McTileCtx& t, const McTileLds& tl, const unsigned short* s_lut,
                                          u32* marker, u32* seg_cnt, u32* recbuf, u32* rowoff, const McRowMasks& rm, u64 rowPend,
                                          u32 vmask_row, u8* __restrict__ codes, u32* __restrict__ recs, u32* tailbuf = nullptr) {
    const int n1 = p.n1;
    const float zk = tl.zk, zk1 = tl.zk1;
    const float iso = p.iso;
    const int xl0 = t.ch * MC_SEG + t.lane * 4;  // this lane's cells when it acts as a ROW lane
    // lane = row: entries of the row, and the index of its first entry
    const u32 cnt = (u32)__builtin_popcount(rm.mixlo) + (u32)__builtin_popcount(rm.mixhi);
    const u32 incl = wave_inclusive_scan(cnt);
    const u32 off = incl - cnt;
    u8* __restrict__ tilebase = codes + ((u64)t.lz * n1 + t.y0) * p.pitch;
    u64 pend = rowPend;
    // the dwordx4 stores of 4 pending rows at once: this lane's row of the group, the lane its 16-byte piece starts at, the
    // valid bytes of the 4 dwords of the piece, and its byte offset inside the group (pieces beyond the row pitch get an
    // offset the range check rejects)
    const int q4 = t.lane & 3, k4 = t.lane & ~3;
    u32 vm4[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int xi = t.ch * MC_SEG + (k4 + i) * 4;
        vm4[i] = (xi + 3 < n1) ? 0xFFFFFFFFu : (xi + 2 < n1) ? 0x00FFFFFFu : (xi + 1 < n1) ? 0x0000FFFFu : (xi < n1) ? 0x000000FFu : 0u;
    }
    const u32 xbyte4 = (u32)(t.ch * MC_SEG + k4 * 4);
    const u32 woff4 = (xbyte4 < (u32)p.pitch) ? (u32)q4 * (u32)p.pitch + xbyte4 : 0x80000000u;
    u32 e0 = 0;
    u32 nbuf = 0;          // records waiting in recbuf
    u64 epochRows = 0ull;  // rows whose records are in recbuf
    u32 rowbase = 0u;      // lane = row: global index of the row's first record
    auto flush_records = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (nbuf) {
            // SCALAR atomic: its result comes back through lgkmcnt.  A vector atomic's would come through
            // vmcnt, which retires in order -- the wave would sit until every code store it has in flight had
            // landed before it could even start copying its records (measured: 0.36 -> 0.44 ms).
            u32 gb = 0u;
            u32* const cur = p.rec_cursor + 32u * (1u + t.region);
            asm volatile("s_atomic_add %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(gb) : "s"(cur), "0"(nbuf) : "memory");
            const u32 rsize = (u32)(p.cap_recs / MC_NCUR);
            // past the region's end nothing is written and the overflow word is raised: the host grows the
            // buffer and sweeps again; mc_emit sees the word and stays out
            if (gb + nbuf <= rsize) {
                gb += t.region * rsize;
                for (u32 i = (u32)t.lane; i < nbuf; i += 64u) recs[gb + i] = recbuf[i];
            } else if (t.lane == 0) {
                p.rec_cursor[0] = 1u;
            }
            if ((epochRows >> t.lane) & 1ull) rowbase = gb + rowoff[t.lane];
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        nbuf = 0u;
        epochRows = 0ull;
    };
    while (pend) {
        // the rows this chunk completes: the leading pending rows that end within 64 entries (ends are monotone)
        const u64 fit = __ballot(incl <= e0 + 64u) & pend;
        const int jl = 63 - __builtin_clzll(fit);  // fit != 0: the first pending row starts at e0 and has <= 64 entries
        const u32 e1 = (u32)__builtin_amdgcn_readlane((int)incl, jl);
        const u32 ntake = e1 - e0;
        u32 dw = 0;
        int ej = 0, eln = 0;   // this entry lane's row and chunk lane
        bool evalid = false;
        if (ntake) {
            if (nbuf + 256u > MC_REC_CAP) flush_records();  // a chunk adds at most 64 * 4 records
            marker[t.lane] = 0u;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (((fit >> t.lane) & 1ull) && cnt) marker[off - e0] = (u32)t.lane + 1u;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const bool valid = (u32)t.lane < ntake;
            const int row1 = (int)wave_inclusive_max(marker[t.lane]);  // row + 1 of the entry on this lane
            const int jj = valid ? row1 - 1 : 0;
            const int j = valid ? row1 - 1 : 1023;
            // (all three shuffles outside any lane-dependent control flow: a ds_bpermute reads 0 from a lane
            // that is switched off, and the row lanes are not the entry lanes)
            const u32 roff = (u32)__shfl((int)off, jj, 64);
            const u32 rmlo = (u32)__shfl((int)rm.mixlo, jj, 64), rmhi = (u32)__shfl((int)rm.mixhi, jj, 64);
            const u32 k = (e0 + (u32)t.lane) - roff;
            const int ln = valid ? nth_set_bit64(rmlo, rmhi, k) : 0;
            const int x0 = t.ch * MC_SEG + ln * 4;
            // the 20 lattice samples of the lane's 4 cells: same mc_f, same operands, same compare as
            // everywhere else (marching.cpp:475-479, :497-505)
            const float yl = tl.ys[jj], yu = tl.ys[jj + 1];
            // Each sample's verdict goes STRAIGHT to the code bits it is: sample c (column x0+c), row r, plane pl is corner
            // 0/4/3/7 (r, pl = 00/01/10/11) of cell c and corner 1/5/2/6 of cell c-1 (marching.cpp:471-472), i.e. ONE constant
            // per sample with both bits at their places in the lane's code dword, OR-ed in when f > iso -- no code is ever
            // assembled from sample bits (that was 72 of the ~430 vector instructions of a chunk).
#define MC_SAMPLE_BITS(c_, asleft_, asright_) \
    ((((c_) < 4) ? ((u32)(asleft_) << (8 * ((c_) & 3))) : 0u) | (((c_) > 0) ? ((u32)(asright_) << (8 * (((c_) + 3) & 3))) : 0u))
#ifdef MC_CONS
            u32 ob = 0;  // bit (4*c + 2*r + pl): the sample is inside every enabled constraint (marching.cpp:255-280)
#endif
#ifdef MC_TAB
            float UYl[MC_TAB_NY], UYu[MC_TAB_NY];
#pragma unroll
            for (int k = 0; k < MC_TAB_NY; ++k) {
                UYl[k] = tl.ty[k * 72 + jj];
                UYu[k] = tl.ty[k * 72 + jj + 1];
            }
#endif
#pragma unroll
            for (int c = 0; c < 5; ++c) {
                const float x = tl.xs[ln * 4 + c];
                const u32 k00 = MC_SAMPLE_BITS(c, 0x01u, 0x02u), k01 = MC_SAMPLE_BITS(c, 0x10u, 0x20u);  // (lower row, plane z / z+1)
                const u32 k10 = MC_SAMPLE_BITS(c, 0x08u, 0x04u), k11 = MC_SAMPLE_BITS(c, 0x80u, 0x40u);  // (upper row, plane z / z+1)
#ifdef MC_TAB  // the same values mc_f would compute: its one-variable sub-expressions come from the tables
                float UX[MC_TAB_NX];
#pragma unroll
                for (int k = 0; k < MC_TAB_NX; ++k) UX[k] = tl.tx[k * 264 + ln * 4 + c];
                dw |= mc_f_t(x, yl, zk, UX, UYl, tl.tz0) > iso ? k00 : 0u;
                dw |= mc_f_t(x, yl, zk1, UX, UYl, tl.tz1) > iso ? k01 : 0u;
                dw |= mc_f_t(x, yu, zk, UX, UYu, tl.tz0) > iso ? k10 : 0u;
                dw |= mc_f_t(x, yu, zk1, UX, UYu, tl.tz1) > iso ? k11 : 0u;
#else
                dw |= mc_f(x, yl, zk) > iso ? k00 : 0u;
                dw |= mc_f(x, yl, zk1) > iso ? k01 : 0u;
                dw |= mc_f(x, yu, zk) > iso ? k10 : 0u;
                dw |= mc_f(x, yu, zk1) > iso ? k11 : 0u;
#endif
#ifdef MC_CONS
                ob |= (mc_ok(x, yl, zk) ? 1u : 0u) << (4 * c + 0);
                ob |= (mc_ok(x, yl, zk1) ? 1u : 0u) << (4 * c + 1);
                ob |= (mc_ok(x, yu, zk) ? 1u : 0u) << (4 * c + 2);
                ob |= (mc_ok(x, yu, zk1) ? 1u : 0u) << (4 * c + 3);
#endif
            }
#undef MC_SAMPLE_BITS
            {
                // cells beyond the end of the grid (ragged last chunk) hold no code; with constraints, a cell with a corner
                // outside one is skipped (marching.cpp:476): no triangles, code 0
                u32 keep = 0u;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    bool k = x0 + c < n1;
#ifdef MC_CONS
                    k = k && ((ob >> (4 * c)) & 0xFFu) == 0xFFu;
#endif
                    keep |= k ? 0xFFu << (8 * c) : 0u;
                }
                dw &= keep;
            }
            if (!valid) dw = 0u;
            ej = jj;
            eln = ln;
            evalid = valid;

            // per cell: triangle count and ambiguity flip; meta nibble c = count | flip<<3
            u32 meta = 0, packed = 0;  // packed = triangles | active cells << 16 of this entry
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int code = (int)((dw >> (8 * c)) & 0xFF);
                if (code != 0 && code != 255) {
                    const u32 lut = s_lut[code];
                    u32 nt = lut & 0xFFu, flip = 0;
                    const int face = (int)(lut >> 8);
                    if (face != 0xFF) {
                        const float uz[2] = {tl.uz0, tl.uz1};
                        if (amb_flip(p, face, tl.ux + (ln * 4 + c), tl.uy + jj, uz)) {
                            nt = s_lut[255 - code] & 0xFFu;
                            flip = 1;
                        }
                    }
                    meta |= (nt | (flip << 3)) << (4 * c);
                    packed += nt + (1u << 16);
                }
            }
            // prefix inside each segment (= tile row j): wave scan minus the scan value at the
            // segment's first entry; both halves of `packed` are non-decreasing, so a max-scan of
            // "exclusive value at segment heads" propagates the base to the followers.
            const u32 pincl = wave_inclusive_scan(packed);
            const u32 excl = pincl - packed;
            const int jprev = __builtin_amdgcn_update_dpp(-1, j, 0x138, 0xf, 0xf, false);  // wave_shr:1
            const bool head = valid && (t.lane == 0 || jprev != j);
            const u32 base = wave_inclusive_max(head ? excl : 0u);
            const u32 pre = excl - base;
            if (valid) {
                // segment totals: order-independent LDS adds; slot j is read by lane j at tile end
                if (packed) __hip_atomic_fetch_add(&seg_cnt[j], packed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                u32 bpos = nbuf + (excl >> 16), tpre = pre & 0xFFFFu;  // records are buffered in entry order
                if (head) rowoff[j] = bpos;                           // the row's (= segment's) first record
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const u32 m = (meta >> (4 * c)) & 0xFu;
                    const u32 nt = m & 7u;
                    if (nt) {
                        recbuf[bpos++] = (u32)(ln * 4 + c) | (((dw >> (8 * c)) & 0xFFu) << 8) | ((m >> 3) << 16) | (nt << 17) |
                                         (tpre << 20);
                        tpre += nt;
                    }
                }
                if (TAIL) tailbuf[j] = dw;
            }
            nbuf += (u32)__builtin_amdgcn_readlane((int)pincl, 63) >> 16;
        }
        epochRows |= fit;
        if (!TAIL) {
            // The pending code rows this chunk completes.  Each row is stored WHOLE (256 B, full 128-byte lines) with
            // what the walk proved -- lanes all-above as ~0, the others, listed ones included, as 0 -- and then the
            // entry lanes overwrite their own dwords with ONE scattered store for the whole chunk: same wave, same
            // addresses, issued right behind, so the line is still being assembled in L2 when the dwords arrive.
            // (Measured: 0.264 ms against 0.283 for gathering each row's listed dwords with ds_bpermute before its
            // store; but 0.296 when the rows were stored by the walk, microseconds earlier -- by then a line may have
            // left L2 and the late dword becomes a read-modify-write in HBM.)
            // Aligned groups of 4 such rows go out with ONE dwordx4 store (lane 4k+s writes the 16 bytes of lanes 4k..4k+3 of
            // row j+s, as the walk stores its culled rows): the stores are bound by their NUMBER -- about 34 cycles of the
            // CU's address unit each -- and the pending rows, one dword store per row, were 58 % of the kernel's stores
            // (1.57 M of 2.69 M per 1025^3 sweep).
            u64 g4 = fit & (fit >> 1) & (fit >> 2) & (fit >> 3) & 0x1111111111111111ull;
            u64 f = fit & ~(g4 | (g4 << 1) | (g4 << 2) | (g4 << 3));
            while (g4) {
                const int jr = __builtin_ctzll(g4);
                g4 &= g4 - 1ull;
                const u32 alo = (u32)__shfl((int)rm.alllo, jr + q4, 64), ahi = (u32)__shfl((int)rm.allhi, jr + q4, 64);
                const u32 bits = ((k4 & 32) ? ahi : alo) >> (k4 & 31);
                typedef u32 u32x4 __attribute__((ext_vector_type(4)));
                u32x4 v;
                v.x = (bits & 1u) ? vm4[0] : 0u;
                v.y = (bits & 2u) ? vm4[1] : 0u;
                v.z = (bits & 4u) ? vm4[2] : 0u;
                v.w = (bits & 8u) ? vm4[3] : 0u;
                __builtin_amdgcn_raw_buffer_store_b128(
                    v, __builtin_amdgcn_make_buffer_rsrc(tilebase + (u64)((u32)jr * (u32)p.pitch), 0, (int)(4u * (u32)p.pitch), 0x00020000), woff4, 0, 0);
            }
            while (f) {
                const int jr = __builtin_ctzll(f);
                f &= f - 1ull;
                const u64 all = ((u64)(u32)__builtin_amdgcn_readlane((int)rm.allhi, jr) << 32) | (u32)__builtin_amdgcn_readlane((int)rm.alllo, jr);
                const u32 v = select_by_mask(all, vmask_row, 0u);
                __builtin_amdgcn_raw_buffer_store_b32(
                    v, __builtin_amdgcn_make_buffer_rsrc(tilebase + (u64)((u32)jr * (u32)p.pitch), 0, (int)p.pitch, 0x00020000), (u32)xl0, 0, 0);
            }
            // (a listed lane beyond the end of the grid -- ragged last chunk, degenerate clamped cells -- has dw == 0 and
            // must not write: its offset lies past the row, in the rows that follow)
            // A buffer store like the row stores above, so that both travel the same queue in issue order; lanes that
            // must not write get an offset the range check rejects.
            {
                const bool wr = evalid && t.ch * MC_SEG + eln * 4 < n1;
                const u32 eoff = wr ? (u32)ej * (u32)p.pitch + (u32)(t.ch * MC_SEG + eln * 4) : 0x80000000u;
                __builtin_amdgcn_raw_buffer_store_b32(dw, __builtin_amdgcn_make_buffer_rsrc(tilebase, 0, (int)(64u * (u32)p.pitch), 0x00020000),
                                                      eoff, 0, 0);
            }
        }
        pend &= ~fit;
        e0 = e1;
    }
    flush_records();
    return rowbase;
}


// Tile end: the per-segment counts of the tile's rows (lane = row) go into the per-GROUP sums the scan runs over (group =
// 64 consecutive segments).  Consecutive rows fall into the same group 64 / nchunk times in a row, so the counts are summed
// per run of equal group inside the wave first -- segmented sum: inclusive scan minus the scan value at the run's head,
// spread by a max-scan (both fields only grow) -- and only the last lane of a run issues the 64-bit atomic: ~6 atomics
// per tile instead of up to 63 (1.2 M -> 0.3 M per 1025^3 sweep; each is a 64-byte request at the memory side).
// c = triangles | active cells << 16 of the lane's segment (0 for lanes beyond the tile), sg = its segment index.
__device__ __forceinline__ void mc_add_group_sums(u64* __restrict__ grpsum, u32 c, u64 sg, int lane) {
    if (__ballot(c != 0u) == 0ull) return;  // wave-uniform
    const u32 key = (u32)(sg >> 6);
    const u32 v = (c & 0xFFFFu) | ((c >> 16) << 17);  // 64 rows: triangles < 2^17, cells < 2^15
    const u32 incl = wave_inclusive_scan(v);
    const u32 excl = incl - v;
    const u32 kprev = (u32)__shfl_up((int)key, 1, 64), knext = (u32)__shfl_down((int)key, 1, 64);
    const bool head = lane == 0 || kprev != key, last = lane == 63 || knext != key;
    const u32 base = wave_inclusive_max(head ? excl : 0u);
    const u32 run = incl - base;
    if (last && run) atomicAdd(&grpsum[key], (u64)(run & 0x1FFFFu) | ((u64)(run >> 17) << 32));
}

#ifndef MC_ONLY_INDEX_KERNELS  // (the module of the index kernels, compiled when MC_FLAG_INDEXED is first used, leaves the sweep kernels out)
extern "C" __global__ __launch_bounds__(64 * MC_WPB_C, MC_CLASSIFY_MINW) void mc_classify(const McParams* __restrict__ P, u8* __restrict__ codes,
                                                                         uint2* __restrict__ segcb, u32* __restrict__ recs,
                                                                         u64* __restrict__ grpsum) {
    __shared__ unsigned short s_lut[256];  // triangle count | ambiguity face << 8
    __shared__ u32 s_marker[MC_WPB_C][64];   // back-end: first-entry markers of a 64-entry chunk
    __shared__ u32 s_rowoff[MC_WPB_C][64];   // back-end: position of each row's first record in s_recbuf
    __shared__ u32 s_recbuf[MC_WPB_C][MC_REC_CAP];  // back-end: the tile's records before they are appended to recs
    __shared__ u32 s_tailbuf[MC_WPB_C][64];  // tail tiles: the rows' code dwords
    __shared__ u32 s_segcnt[MC_WPB_C][64];
    __shared__ float s_tab[MC_WPB_C][2 * 264 + 2 * 72];  // per wave: xs[264] ux[264] ys[72] uy[72] (table slices)
#ifdef MC_TAB
    __shared__ float s_tabu[MC_WPB_C][MC_TAB_NX * 264 + MC_TAB_NY * 72];  // per wave: slices of f's x and y sub-expression tables
#endif
#pragma unroll
    for (int i = (int)threadIdx.x; i < 256; i += 64 * MC_WPB_C)
        s_lut[i] = (unsigned short)(c_tri_count[i] | (c_amb_face[i] << 8));
    __syncthreads();

    const McParams p = *P;
    const int lane = threadIdx.x & 63;
    // the wave index is wave-uniform, but the compiler only knows that if told: without the
    // readfirstlane every tile coordinate (and the whole walk's scalar algebra) lands in VGPRs
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // (tile numbers fit 32 bits -- the grid size does -- and 32-bit divisions are a third of the scalar code of 64-bit ones)
    const u32 tile = blockIdx.x * (u32)MC_WPB_C + (u32)w;
    const u32 ntiles_a = (u32)p.nchunk_main * (u32)p.ntile_y * (u32)p.nz_a;
    const u32 ntiles_main = ntiles_a + (u32)p.nchunk_main * (u32)p.ntile_y2 * (u32)(p.nz - p.nz_a);
    const u32 ntiles = ntiles_main + (u32)p.ntile_t * (u32)p.nz;
    if (tile >= ntiles) return;  // whole wave
    const bool is_tail = tile >= ntiles_main;  // wave-uniform
    const int n1 = p.n1;
    int ch, ty, lz, y0, ny;
    // (chunk fastest: the 4 waves of a workgroup cover 1 KB of each code row together.  Layer-fastest, which
    // gives the 4 waves equal work, measured the same: 0.380 vs 0.378 ms.)
    if (!is_tail) {
        const bool late = tile >= ntiles_a;  // (wave-uniform) one of the short tiles of the slab's last layers
        const u32 tr = late ? tile - ntiles_a : tile;
        const u32 nty = (u32)(late ? p.ntile_y2 : p.ntile_y);
        const int th = late ? p.tile_h2 : p.tile_h;
        const u32 t2 = tr / (u32)p.nchunk_main;
        ch = (int)(tr - t2 * (u32)p.nchunk_main);
        const u32 lr = t2 / nty;
        ty = (int)(t2 - lr * nty);
        // Launch order of the layers.  A launch ends when its longest waves do.  A surface that is closed inside the domain has
        // its long tiles in the middle layers: started first they are the floor of the kernel's time, started in the middle of
        // the launch (equation_3 513^3: at 13 us, 40 us long, in a kernel of 55) they are that plus their start -- so those
        // sweeps run from the slab's middle outwards (mid, mid + 1, mid - 1, ...), the outer layers, short tiles, filling the
        // end.  Where the heavy layers lie elsewhere (Goursat: +10 %) z order is better; the host measures both.
        const int li = (int)lr + (late ? p.nz_a : 0), mid = (p.nz - 1) >> 1;
        lz = p.layer_order == 0 ? li : (li & 1) ? mid + ((li + 1) >> 1) : mid - (li >> 1);
        y0 = ty * th;
        ny = min(th, n1 - y0);
    } else {  // tail tile: the last chunk's 1..4 cells of 64 consecutive rows, one row per lane
        const u32 tt = tile - ntiles_main;
        ch = p.nchunk - 1;
        lz = (int)(tt / (u32)p.ntile_t);
        ty = (int)(tt - (u32)lz * (u32)p.ntile_t);
        y0 = ty * 64;
        ny = min(64, n1 - y0);
    }
    const int iz = p.z_begin + lz;
    const int x0 = ch * MC_SEG + lane * 4;
    const float iso = p.iso;

    u32* seg_cnt = s_segcnt[w];
    u32* marker = s_marker[w];
    u32* recbuf = s_recbuf[w];
    u32* rowoff = s_rowoff[w];
    seg_cnt[lane] = 0u;

    const float* __restrict__ ax = p.axs;
    const float* __restrict__ ay = p.axs + (n1 + 1);
    const float* __restrict__ az = p.axs + 2 * (n1 + 1);
    // y samples of the tile's 64 sample rows live in one VGPR (lane = row); the walk reads
    // them with v_readlane, so the inner loop issues no memory load at all.
    const float yv = ay[min(y0 + lane, n1)];

    float xs[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) xs[c] = ax[min(x0 + c, n1)];
    const float xe = ax[min(ch * MC_SEG + MC_SEG, n1)];  // first sample of the next chunk
    const float zk = az[iz], zk1 = az[iz + 1];

    // LDS copies of the tile's coordinate slices for the record pass (see mc_record_pass)
    McTileLds tl;
    {
        float* tab = s_tab[w];
        float* lxs = tab;
        float* lux = tab + 264;
        float* lys = tab + 528;
        float* luy = tab + 600;
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const int i = k * 64 + lane;  // 0..319, only 0..256 are used
            if (i < 264) {
                const int gi = min(ch * MC_SEG + i, n1);
                lxs[i] = ax[gi];
                lux[i] = p.axis[gi];
            }
        }
        lys[lane] = yv;
        luy[lane] = p.axis[min(y0 + lane, n1)];
        if (lane == 0) {
            lys[64] = ay[min(y0 + 64, n1)];
            luy[64] = p.axis[min(y0 + 64, n1)];
        }
        tl.xs = lxs;
        tl.ux = lux;
        tl.ys = lys;
        tl.uy = luy;
        tl.zk = zk;
        tl.zk1 = zk1;
        tl.uz0 = p.axis[iz];
        tl.uz1 = p.axis[iz + 1];
#ifdef MC_TAB
        {
            const u32 ts = mc_tab_stride(n1);
            float* ltx = s_tabu[w];
            float* lty = s_tabu[w] + MC_TAB_NX * 264;
#pragma unroll
            for (int kk = 0; kk < MC_TAB_NX; ++kk) {
                const float* __restrict__ t = mc_tab_ptr(p.tab, ts, 0, 0, kk);
#pragma unroll
                for (int k = 0; k < 5; ++k) {
                    const int i = k * 64 + lane;
                    if (i < 264) ltx[kk * 264 + i] = t[min(ch * MC_SEG + i, n1)];
                }
            }
#pragma unroll
            for (int kk = 0; kk < MC_TAB_NY; ++kk) {
                const float* __restrict__ t = mc_tab_ptr(p.tab, ts, 1, 0, kk);
                lty[kk * 72 + lane] = t[min(y0 + lane, n1)];
                if (lane == 0) lty[kk * 72 + 64] = t[min(y0 + 64, n1)];
            }
#pragma unroll
            for (int kk = 0; kk < MC_TAB_NZ; ++kk) {
                const float* __restrict__ t = mc_tab_ptr(p.tab, ts, 2, 0, kk);
                // (made scalar HERE: a vector register would be waited for at its first use, in the back-end, behind every
                // code store the walk has issued by then -- vmcnt retires in order)
                tl.tz0[kk] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, t[iz])));
                tl.tz1[kk] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, t[iz + 1])));
            }
            tl.tx = ltx;
            tl.ty = lty;
        }
#endif
    }

    if (is_tail) {
        // ---- tail tile: lane = row, cells ch*256 .. n1-1 (1..4 of them) of that row.  One interval
        // evaluation per lane decides most rows; the others go through the record pass (entry = row,
        // chunk lane 0), which leaves their code dword in LDS; one coalesced store writes the 64 rows.
        McTileCtx tt;
        tt.ch = ch;
        tt.y0 = y0;
        tt.iz = iz;
        tt.lz = lz;
        tt.lane = lane;
        tt.seg0 = ((u64)lz * n1 + y0) * p.nchunk + ch;
        tt.region = (u32)tile & (MC_NCUR - 1u);
        const bool rvalid = lane < ny;
        const u32 vm = p.tail_cells >= 4 ? 0xFFFFFFFFu : ((1u << (8 * p.tail_cells)) - 1u);
        u64 laneAll = 0ull, mixedL;
#if defined(MC_HAVE_IV) && defined(MC_FINITE) && !defined(MC_NO_CULL)
        {
            const float xa = ax[min(ch * MC_SEG, n1)], xb = ax[min(ch * MC_SEG + 4, n1)];
            const float yb = ay[min(y0 + lane + 1, n1)];
            float lo, hi;
            mc_f_iv(__builtin_fminf(xa, xb), __builtin_fmaxf(xa, xb), __builtin_fminf(yv, yb), __builtin_fmaxf(yv, yb),
                    __builtin_fminf(zk, zk1), __builtin_fmaxf(zk, zk1), lo, hi);
#ifdef MC_CONS
            bool allok, dead;
            mc_ok_iv(__builtin_fminf(xa, xb), __builtin_fmaxf(xa, xb), __builtin_fminf(yv, yb), __builtin_fmaxf(yv, yb),
                     __builtin_fminf(zk, zk1), __builtin_fmaxf(zk, zk1), allok, dead);
            laneAll = __ballot(rvalid && lo > iso && allok);
            mixedL = __ballot(rvalid && hi > iso && !dead) & ~laneAll;
#else
            laneAll = __ballot(rvalid && lo > iso);
            mixedL = __ballot(rvalid && hi > iso) & ~laneAll;
#endif
        }
#else
        mixedL = __ballot(rvalid);  // no enclosure for this equation: the record pass evaluates every row
#endif
        u32* tailbuf = s_tailbuf[w];
        McRowMasks rmt;
        rmt.mixlo = (u32)((mixedL >> lane) & 1ull);  // lane = row: its one entry is chunk lane 0
        rmt.mixhi = 0u;
        rmt.alllo = 0u;
        rmt.allhi = 0u;
        u32 rbase = 0u;
        if (mixedL) rbase = mc_backend<true>(p, tt, tl, s_lut, marker, seg_cnt, recbuf, rowoff, rmt, mixedL, 0u, codes, recs, tailbuf);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const u64 sgt = tt.seg0 + (u64)lane * p.nchunk;
        const u32 ct = rvalid ? seg_cnt[lane] : 0u;
        if (rvalid) {
            const u32 own = select_by_mask(mixedL, tailbuf[lane], select_by_mask(laneAll, vm, 0u));
            p.codes_tail[(u64)lz * n1 + y0 + lane] = own;
            segcb[sgt] = make_uint2(ct, rbase);
        }
        mc_add_group_sums(grpsum, ct, sgt, lane);
        return;
    }

    // sample column x = xe for the tile's 64 sample rows (lane = row): bit j of E0 / E1 is the
    // "x+4" neighbour of lane 63 in row j.  ENone / EFull bit j: both rows j, j+1 of both planes
    // are below / above iso in that column.
    const u64 E0 = __ballot(mc_f(xe, yv, zk) > iso);
    const u64 E1 = __ballot(mc_f(xe, yv, zk1) > iso);
    const u64 ENone = ~((E0 | E1) | ((E0 | E1) >> 1));
#ifdef MC_CONS
    // "every sample above iso" proves code 255 only where every sample is also inside the constraints
    const u64 EOk = __ballot(mc_ok(xe, yv, zk) && mc_ok(xe, yv, zk1));
    const u64 EFull = (E0 & E1 & EOk) & ((E0 & E1 & EOk) >> 1);
#else
    const u64 EFull = (E0 & E1) & ((E0 & E1) >> 1);
#endif

    // EXACT row culling.  mc_f_iv bounds every value mc_f can compute on a box (interval arithmetic
    // is exact for round-to-nearest code because each IEEE operation is monotone), so one interval
    // evaluation per tile row -- lane = row, box = the chunk's x extent x 2 sample rows x 2 planes --
    // proves most uniform rows all-below / all-above iso without sampling them.  Rows it cannot
    // decide take the sampling path below, so the result is identical either way.
    u64 rowNone = 0ull, rowFull = 0ull;
#if defined(MC_HAVE_IV) && defined(MC_FINITE) && !defined(MC_NO_CULL)
    {
        const float xa = ax[min(ch * MC_SEG, n1)];
        const float yb = ay[min(y0 + lane + 1, n1)];
        float lo, hi;
        mc_f_iv(__builtin_fminf(xa, xe), __builtin_fmaxf(xa, xe), __builtin_fminf(yv, yb), __builtin_fmaxf(yv, yb),
                __builtin_fminf(zk, zk1), __builtin_fmaxf(zk, zk1), lo, hi);
#ifdef MC_CONS
        // cells of a box proven wholly outside a constraint are all skipped (code 0, like "none");
        // "all above iso" is code 255 only if the box is proven wholly inside every constraint
        bool allok, dead;
        mc_ok_iv(__builtin_fminf(xa, xe), __builtin_fmaxf(xa, xe), __builtin_fminf(yv, yb), __builtin_fmaxf(yv, yb),
                 __builtin_fminf(zk, zk1), __builtin_fmaxf(zk, zk1), allok, dead);
        rowFull = __ballot(lo > iso && allok);
        rowNone = __ballot(!(hi > iso) || dead);
#else
        rowFull = __ballot(lo > iso);
        rowNone = __ballot(!(hi > iso));
#endif
    }
#endif
    const u64 rowCull = rowNone | rowFull;

    const u32 vmask = (x0 + 3 < n1) ? 0xFFFFFFFFu : (x0 + 2 < n1) ? 0x00FFFFFFu : (x0 + 1 < n1) ? 0x0000FFFFu
                      : (x0 < n1) ? 0x000000FFu : 0u;

    McTileCtx tc;
    tc.ch = ch;
    tc.y0 = y0;
    tc.iz = iz;
    tc.lz = lz;
    tc.lane = lane;
    tc.seg0 = ((u64)lz * n1 + y0) * p.nchunk + ch;
    tc.region = (u32)tile & (MC_NCUR - 1u);
    // what the walk hands to the back-end: per-row lane masks (lane j = row j) and the rows whose code row
    // it has not stored
    McRowMasks rm = {0u, 0u, 0u, 0u};
    u64 rowPend = 0ull;
#define MC_SET_ROW(j_, mixed_, all_)                                                                          \
    {   /* v_writelane: with an SGPR value the lane select must be M0 (constant-bus limit); M0 is restored */ \
        const u64 mx_ = (mixed_), al_ = (all_);                                                               \
        u32 m0save_;                                                                                          \
        asm("s_mov_b32 %4, m0\n\ts_mov_b32 m0, %5\n\ts_nop 0\n\tv_writelane_b32 %0, %6, m0\n\t"                \
            "v_writelane_b32 %1, %7, m0\n\tv_writelane_b32 %2, %8, m0\n\tv_writelane_b32 %3, %9, m0\n\t"        \
            "s_mov_b32 m0, %4"                                                                                \
            : "+v"(rm.mixlo), "+v"(rm.mixhi), "+v"(rm.alllo), "+v"(rm.allhi), "=&s"(m0save_)                  \
            : "s"((int)(j_)), "s"((u32)mx_), "s"((u32)(mx_ >> 32)), "s"((u32)al_), "s"((u32)(al_ >> 32)));     \
    }

    // row base of the code plane as a wave-uniform pointer + 32-bit lane offset
    u8* const tilebase = codes + ((u64)lz * n1 + y0) * p.pitch;
    u8* __restrict__ rowbase = tilebase;
    const u32 xoff = (u32)x0;
    // Code stores go through a buffer descriptor over ONE row, rebuilt per step by advancing its
    // (scalar) base: lanes beyond the end of the row are dropped by the hardware range check
    // instead of by exec-mask juggling, and the address needs no vector arithmetic.  (The range
    // check covers voffset + soffset, so the row cannot advance through soffset: measured -- such
    // stores are dropped from the second row on.)  The descriptor ends at the row PITCH (a
    // multiple of 128 B), not at the last cell: the lanes between n1 and the end of the last
    // 128-byte line store zeros, so that line is written whole.  Partial-line stores there cost
    // 0.14 ms at 1025^3 (measured: every one is a read-modify-write of a cold line).
    const int rowbytes = (int)p.pitch;
    auto store_codes = [&](u32 v) {
        __builtin_amdgcn_raw_buffer_store_b32(v, __builtin_amdgcn_make_buffer_rsrc(rowbase, 0, rowbytes, 0x00020000), xoff, 0, MC_CULLED_STORE_AUX);
        rowbase += p.pitch;
    };
#if defined(MC_HAVE_IV) && defined(MC_FINITE) && !defined(MC_NO_CULL)
    // ---- interval walk: no sample is evaluated here at all.  A row the tile-level test could
    // not decide is classified per LANE by one more interval evaluation over the lane's own box
    // (its 4 cells: 5 x samples x 2 rows x 2 planes).  Lanes proven all-below / all-above store
    // 0 / ~0; the undecided ones (the truly mixed lanes plus the few where the enclosure is
    // loose) go to the record pass, which evaluates their samples exactly.
    {
        const float xa4 = ax[min(x0, n1)], xb4 = ax[min(x0 + 4, n1)];
        const float lxl = __builtin_fminf(xa4, xb4), lxh = __builtin_fmaxf(xa4, xb4);
        const float zl = __builtin_fminf(zk, zk1), zh = __builtin_fmaxf(zk, zk1);
        // Four consecutive culled rows are stored with ONE dwordx4 store per lane (lane 4k+s writes
        // the 16 bytes of lanes 4k..4k+3 in row j+s): the code stores are issue-bound in the
        // texture-address unit (about 34 cycles per dword store per CU, measured), and a wide store
        // moves 4x the bytes per instruction.
        const int q = lane & 3, k4 = lane & ~3;
        u32 vm4[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int xi = ch * MC_SEG + (k4 + i) * 4;
            vm4[i] = (xi + 3 < n1) ? 0xFFFFFFFFu : (xi + 2 < n1) ? 0x00FFFFFFu : (xi + 1 < n1) ? 0x0000FFFFu
                     : (xi < n1) ? 0x000000FFu : 0u;
        }
        // byte offset of the lane's 16-byte piece inside a 4-row block; pieces that start beyond the
        // row pitch get an offset the range check rejects
        const u32 xbyte = (u32)(ch * MC_SEG + k4 * 4);
        const u32 woff = (xbyte < (u32)p.pitch) ? (u32)q * (u32)p.pitch + xbyte : 0x80000000u;
        const int wbytes = (int)(3u * (u32)p.pitch + (u32)p.pitch);
        // The rows are handled class by class, each class by iterating the set bits of its mask: the CU has
        // ONE scalar unit, and a per-row "which kind of row is this" decision tree cost more scalar
        // instructions than the work itself (measured: 135 M scalar vs 100 M vector instructions per sweep).
        // lxl / lxh come from loads issued at the top of the tile; consume them BEFORE the first store: the
        // compiler otherwise waits for them at their first use, after the store loops, and with in-order
        // vmcnt that wait is "until every store above has landed" (measured: the tile's compute then starts
        // only when its culled rows are in memory -- compute + store time instead of the larger of the two)
        asm volatile("" ::"v"(lxl), "v"(lxh), "v"(yv));
        const u64 rowsValid = (1ull << ny) - 1ull;  // ny <= 63 here
        const u64 cullRow = rowCull & rowsValid;
        // BLOCK level (MC_BLOCK_LEVEL: the host defines it for polynomials of higher degree, jit_source in mc_runtime.hip).  The row test
        // looks at boxes 256 cells wide; an enclosure grows with its box, so on small grids and for f of higher degree it
        // leaves whole tiles undecided that hold no cell of the surface (513^3: 42 % of the kernel's wave time on equation_3
        // went into tiles whose lane-level evaluations then found nothing).  ONE more evaluation covers the tile with 8 x 8
        // boxes of 32 cells x 8 rows (lane = box): a block of 8 rows whose 8 boxes are all decided is finished -- its rows
        // are stored like culled rows, with the boxes' values -- and only the others take the lane-level evaluations.
        // Exact like the rest: a box is decided only when the enclosure proves every sample of its cells on one side of
        // iso.  What it buys (equation_3 513^3): 85 % of those tiles decided, 17 % off the SUM of the waves' lifetimes --
        // nothing off the time of one sweep alone, which is set by the tiles that do hold the surface, but 7.5 % off the step
        // with three sweeps in flight, whose kernels take the freed wave slots.  Cheap f gains nothing (sphere 513^3: +1.5 %).
        // In a tile the staged enclosure (MC_IV_NY) handles, this evaluation is the plain mc_f_iv.
        u64 rowBlk = 0ull;   // bit j: row j is decided by its block of 8 rows
        u64 blkFull = 0ull;  // bit 8 * rb + cb: every sample of column block cb of row block rb is above iso
#ifdef MC_BLOCK_LEVEL
        if (__builtin_popcountll(rowsValid & ~cullRow) >= MC_BLOCK_MIN_ROWS) {
            const int cb = lane & 7, rb = lane >> 3;
            const float bxa = tl.xs[cb * 32], bxb = tl.xs[cb * 32 + 32];
            const float bya = __shfl(yv, min(8 * rb, 63), 64), byb = __shfl(yv, min(8 * rb + 8, ny), 64);
            float lo, hi;
            mc_f_iv(__builtin_fminf(bxa, bxb), __builtin_fmaxf(bxa, bxb), __builtin_fminf(bya, byb), __builtin_fmaxf(bya, byb), zl, zh, lo, hi);
            bool full = lo > iso, none = !(hi > iso);
#ifdef MC_CONS
            bool allok, dead;
            mc_ok_iv(__builtin_fminf(bxa, bxb), __builtin_fmaxf(bxa, bxb), __builtin_fminf(bya, byb), __builtin_fmaxf(bya, byb), zl, zh, allok, dead);
            full = full && allok;
            none = none || dead;
#endif
            const bool outside = ch * MC_SEG + cb * 32 >= n1;  // the box holds no cell of the grid (ragged last chunk)
            blkFull = __ballot(full && !outside);
            u64 dec = __ballot(full || none || outside);
            dec &= dec >> 4;
            dec &= dec >> 2;
            dec &= dec >> 1;
            dec &= 0x0101010101010101ull;                   // bit 8 * rb: the 8 boxes of row block rb are all decided
            rowBlk = ((dec << 8) - dec) & rowsValid & ~cullRow;  // ... spread over its rows 8 * rb .. 8 * rb + 7
        }
#endif
        const u64 cull = cullRow | rowBlk;
        const u32 lane8 = (u32)lane >> 3;
        // (1) aligned blocks of 4 decided rows (they lie in one block of 8 rows)
        u64 m4 = cull & (cull >> 1) & (cull >> 2) & (cull >> 3) & 0x1111111111111111ull;
        const u64 blockRows = m4 | (m4 << 1) | (m4 << 2) | (m4 << 3);
        while (m4) {
            const int j = __builtin_ctzll(m4);
            m4 &= m4 - 1ull;
            const u32 rf4 = (u32)((rowFull & cullRow) >> j) & 0xFu, rb4 = (u32)(rowBlk >> j) & 0xFu;
            const u32 patt = (u32)(blkFull >> (8 * (j >> 3))) & 0xFFu;
            const u32 c = (((rf4 >> q) | ((rb4 >> q) & (patt >> lane8))) & 1u) ? 0xFFFFFFFFu : 0u;
            typedef u32 u32x4 __attribute__((ext_vector_type(4)));
            u32x4 v;
            v.x = c & vm4[0];
            v.y = c & vm4[1];
            v.z = c & vm4[2];
            v.w = c & vm4[3];
            __builtin_amdgcn_raw_buffer_store_b128(
                v, __builtin_amdgcn_make_buffer_rsrc(tilebase + (u64)((u32)j * (u32)p.pitch), 0, wbytes, 0x00020000), woff, 0, MC_CULLED_STORE_AUX);
        }
        // (2) the other decided rows
        u64 m1 = cull & ~blockRows;
        while (m1) {
            const int j = __builtin_ctzll(m1);
            m1 &= m1 - 1ull;
            const u32 patt = (u32)(blkFull >> (8 * (j >> 3))) & 0xFFu;
            const u32 on = (u32)((rowFull & cullRow) >> j) | ((u32)(rowBlk >> j) & (patt >> lane8));
            const u32 v = (on & 1u) ? vmask : 0u;
            __builtin_amdgcn_raw_buffer_store_b32(
                v, __builtin_amdgcn_make_buffer_rsrc(tilebase + (u64)((u32)j * (u32)p.pitch), 0, rowbytes, 0x00020000), xoff, 0, MC_CULLED_STORE_AUX);
        }
        // (3) undecided rows: the lane-level masks go to lane j of rm, the row itself to the back-end.
        // Two adjacent undecided rows share ONE evaluation over the box of both (3 sample rows): half
        // the interval evaluations, for a few more lanes handed to the back-end (which is exact, so a
        // lane listed needlessly just yields uniform codes and no record).
        rowPend = rowsValid & ~cull;
        const u64 lanesIn = __ballot(x0 < n1);  // lanes that hold cells of the grid (a ragged last chunk has fewer)
#ifdef MC_IV_NY
        // f has expensive sub-expressions of y alone (sin / cos ...): their enclosures are computed ONCE per tile, lane =
        // row (mc_f_iv_y), and each step of the loop below only reads its row's values (v_readlane) and combines them
        // with the lane's x and z parts, which the compiler keeps across the loop (mc_f_iv_rest).  The pairing is the
        // loop's, in closed form: inside a run of undecided rows the 1st, 3rd, ... are heads, a head pairs with the row
        // above it when that one is undecided too.
        const u64 below = ~rowPend & ((1ull << lane) - 1ull);
        const int runlen = below ? lane - (63 - __builtin_clzll(below)) - 1 : lane;
        const bool headL = ((rowPend >> lane) & 1ull) && !(runlen & 1);
        const bool pairL = headL && (((rowPend >> 1) >> lane) & 1ull);
        float Yv[MC_IV_NY];
        {
            const float y1 = __shfl_down(yv, 1, 64), y2 = __shfl_down(yv, 2, 64);
            const float yt = pairL ? y2 : y1;
            mc_f_iv_y(__builtin_fminf(yv, yt), __builtin_fmaxf(yv, yt), zl, zh, Yv);
        }
        const u64 pairs = __ballot(pairL);
        u64 mu = __ballot(headL);
#else
        u64 mu = rowPend;
#endif
        while (mu) {
            const int j = __builtin_ctzll(mu);
            mu &= mu - 1ull;
#ifdef MC_IV_NY
            const int pair = (int)((pairs >> j) & 1ull);
#else
            const int pair = (int)((mu >> ((j + 1) & 63)) & 1ull);  // row j+1 is undecided too (j + 1 < ny then)
            if (pair) mu &= mu - 1ull;
#endif
            const float ya = readlane_f(yv, j), yb = readlane_f(yv, j + 1 + pair);
            float lo, hi;
#ifdef MC_IV_NY
            float Yu[MC_IV_NY];
#pragma unroll
            for (int k = 0; k < MC_IV_NY; ++k) Yu[k] = readlane_f(Yv[k], j);
            mc_f_iv_rest(lxl, lxh, __builtin_fminf(ya, yb), __builtin_fmaxf(ya, yb), zl, zh, Yu, lo, hi);
#else
            mc_f_iv(lxl, lxh, __builtin_fminf(ya, yb), __builtin_fmaxf(ya, yb), zl, zh, lo, hi);
#endif
#ifdef MC_CONS
            bool allok, dead;
            mc_ok_iv(lxl, lxh, __builtin_fminf(ya, yb), __builtin_fmaxf(ya, yb), zl, zh, allok, dead);
            const u64 laneAll = __ballot(lo > iso && allok);
            const u64 mixedL = __ballot(hi > iso && !dead) & ~laneAll & lanesIn;
#else
            const u64 laneAll = __ballot(lo > iso);
            const u64 mixedL = __ballot(hi > iso) & ~laneAll & lanesIn;
#endif
            MC_SET_ROW(j, mixedL, laneAll)
            if (pair) MC_SET_ROW(j + 1, mixedL, laneAll)
        }
    }
#else
    // ---- sampling walk (equations the interval code cannot bound: division by a variable,
    // general powers, possible NaN / inf)
    // lower (r0) / upper (r1) sample rows, plane z (a) and plane z+1 (c).  Written as straight
    // code: as lambdas the closure did not get scalarised and lived in scratch memory.
    const u64 lanesInS = __ballot(x0 < n1);  // lanes that hold cells of the grid
    float r0a[4], r0c[4], r1a[4], r1c[4];
    u64 gtPrev = 0, gePrev = 0;  // lower row, per lane: some sample > iso / every sample > iso
    bool haveLower = false;      // the lower sample row of the next step is in r0a / r0c
    bool okPrev0 = true;         // MC_CONS: lower row, the lane's first sample column is inside the constraints

    // per-lane uniformity of one sample row on the vector unit.  fmax/fmin skip NaN operands, which
    // is right for "some sample > iso" (NaN > iso is false, marching.cpp:498) but not for "every
    // sample > iso": a NaN among samples that otherwise all exceed iso shows as a NaN sum.
#ifndef MC_FINITE
#define MC_ROW_NANFIX(ra, rc, ge)                                                                            \
    if (ge != 0ull) {                                                                                        \
        const float sm_ = ((ra[0] + ra[1]) + (ra[2] + ra[3])) + ((rc[0] + rc[1]) + (rc[2] + rc[3]));          \
        ge &= ~__ballot(sm_ != sm_);                                                                         \
    }
#else
#define MC_ROW_NANFIX(ra, rc, ge)
#endif
#ifdef MC_CONS
    // constraints in the sampling walk: "every sample above iso" (the proof of code 255) additionally
    // needs every sample inside the constraints; "no sample above iso" gives code 0 either way.  ok0 =
    // the lane's first sample column (its left neighbour's x+4 column) is inside, both planes.
#define MC_ROW_CONS(yy_, ge, ok0)                                                                            \
    {                                                                                                        \
        bool a_ = true;                                                                                      \
        _Pragma("unroll") for (int c = 0; c < 4; ++c) {                                                      \
            const bool o_ = mc_ok(xs[c], yy_, zk) && mc_ok(xs[c], yy_, zk1);                                 \
            if (c == 0) ok0 = o_;                                                                            \
            a_ = a_ && o_;                                                                                   \
        }                                                                                                    \
        ge &= __ballot(a_);                                                                                  \
    }
#else
#define MC_ROW_CONS(yy_, ge, ok0)
#endif
#define MC_EVAL_ROW(y_, ra, rc, gt, ge, ok0)                                                                     \
    {                                                                                                        \
        const float yy_ = (y_);                                                                              \
        _Pragma("unroll") for (int c = 0; c < 4; ++c) {                                                      \
            ra[c] = mc_f(xs[c], yy_, zk);                                                                    \
            rc[c] = mc_f(xs[c], yy_, zk1);                                                                   \
        }                                                                                                    \
        const float mx_ = max3f(max3f(ra[0], ra[1], ra[2]), max3f(ra[3], rc[0], rc[1]), __builtin_fmaxf(rc[2], rc[3])); \
        const float mn_ = min3f(min3f(ra[0], ra[1], ra[2]), min3f(ra[3], rc[0], rc[1]), __builtin_fminf(rc[2], rc[3])); \
        gt = __ballot(mx_ > iso);                                                                            \
        ge = __ballot(mn_ > iso);                                                                            \
        MC_ROW_NANFIX(ra, rc, ge)                                                                            \
        MC_ROW_CONS(yy_, ge, ok0)                                                                            \
    }

    for (int j = 0; j < ny; ++j) {
        if (!haveLower) MC_EVAL_ROW(readlane_f(yv, j), r0a, r0c, gtPrev, gePrev, okPrev0)
        u64 gtNew, geNew;
        bool okNew0 = true;
        MC_EVAL_ROW(readlane_f(yv, j + 1), r1a, r1c, gtNew, geNew, okNew0)
        const u64 anyOwn = gtNew | gtPrev, allOwn = geNew & gePrev;
        const bool none = anyOwn == 0ull && ((ENone >> j) & 1ull);
        const bool full = allOwn == ~0ull && ((EFull >> j) & 1ull);
        if (__builtin_expect(none || full, 1)) {
            store_codes(full ? vmask : 0u);
        } else {
            // per-lane classification: the lane's x+4 neighbour column is lane+1's sample 0
            // (lane 63: column E).  The empty asm pins the four compares to this branch.
            float l0 = r0a[0], l1 = r0c[0], u0 = r1a[0], u1 = r1c[0];
            asm volatile("" : "+v"(l0), "+v"(l1), "+v"(u0), "+v"(u1));
            const u64 n0 = __ballot(l0 > iso), n1m = __ballot(l1 > iso), n2 = __ballot(u0 > iso), n3 = __ballot(u1 > iso);
            u64 topAny = (~ENone >> j) << 63, topAll = (EFull >> j) << 63;
            asm("" : "+s"(topAny), "+s"(topAll));  // keep the halves apart (no 64-bit funnel shift on the SALU)
            const u64 nbAny = ((n0 | n1m | n2 | n3) >> 1) | topAny;
#ifdef MC_CONS
            const u64 nbAll = (((n0 & n1m & n2 & n3) & __ballot(okPrev0 && okNew0)) >> 1) | topAll;
#else
            const u64 nbAll = ((n0 & n1m & n2 & n3) >> 1) | topAll;
#endif
            const u64 laneAll = allOwn & nbAll;
            const u64 mixedL = (anyOwn | nbAny) & ~laneAll & lanesInS;  // lanes with corners on both sides of iso
            if (mixedL) {  // the back-end writes this row whole
                MC_SET_ROW(j, mixedL, laneAll)
                rowPend |= 1ull << j;
                rowbase += p.pitch;
            } else {
                store_codes(select_by_mask(laneAll, vmask, 0u));
            }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {  // the upper row becomes the next step's lower row
            r0a[c] = r1a[c];
            r0c[c] = r1c[c];
        }
        gtPrev = gtNew;
        gePrev = geNew;
        okPrev0 = okNew0;
        haveLower = true;
    }
#undef MC_EVAL_ROW
#undef MC_ROW_NANFIX
#undef MC_ROW_CONS

#endif  // interval / sampling walk

#undef MC_SET_ROW
    u32 rbase = 0u;
    if (rowPend) rbase = mc_backend(p, tc, tl, s_lut, marker, seg_cnt, recbuf, rowoff, rm, rowPend, vmask, codes, recs);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const u64 sg = tc.seg0 + (u64)lane * p.nchunk;
    const u32 c = lane < ny ? seg_cnt[lane] : 0u;
    if (lane < ny) segcb[sg] = make_uint2(c, rbase);
    // group sums for the scan (group = 64 consecutive segments): triangles | active cells << 32
    mc_add_group_sums(grpsum, c, sg, lane);
}
#endif  // MC_ONLY_INDEX_KERNELS

// =============================================================== K3: emit
// ---- geometry of the 12 cell edges (marching_lookup.h:10-23 over the corner layout of marching.cpp:471-472)
// Edge e lies on ONE lattice edge: its axis, and the offset of that lattice edge's LOWER end inside the cell.
#define MC_EDGE_OX 0x622u    // edges whose lower end has x offset 1: 1, 5, 9, 10
#define MC_EDGE_OY 0xC44u    // ... y offset 1: 2, 6, 10, 11
#define MC_EDGE_OZ 0x0F0u    // ... z offset 1: 4, 5, 6, 7
#define MC_EDGE_DOWN 0x0CCu  // edges the table walks from their upper to their lower end: 2, 3, 6, 7
__device__ __forceinline__ int edge_axis(int e) { return e >= 8 ? 2 : (e & 1); }

// the edges of a cell that carry an intersection: the two corner bits differ (marching.cpp:560-566)
__device__ __forceinline__ u32 crossed_edges(u32 c) {
    const u32 t = c ^ (c >> 1);  // bit i: corner i against corner i+1 (edges 0, 1, 2, 4, 5, 6)
    const u32 u = c ^ (c >> 4);  // bit i: corner i against corner i+4 (edges 8..11)
    return (t & 0x77u) | ((((c >> 3) ^ c) & 1u) << 3) | ((((c >> 7) ^ (c >> 4)) & 1u) << 7) | ((u & 0xFu) << 8);
}

// Marching::interp (marching.cpp:437-446) for one axis.  The fallback `x_s + 0.5*(x_e - x_s)` is evaluated in double by
// the reference; one add of two floats rounded to double and then to float equals the float add (53 >= 2*24+2).
__device__ __forceinline__ float mc_interp(float iso, float xs, float xe, float vs, float ve) {
    const float t = (iso - vs) / (ve - vs);
    const float dx = xe - xs;
    const float v = t * dx;
    return (__builtin_isinf(v) || __builtin_isnan(v)) ? xs + 0.5f * dx : xs + v;
}

// n = g/|g|, g = central difference of F at the point, h = step/2.  false: zero or non-finite gradient.
__device__ __forceinline__ bool mc_unit_gradient(float gx, float gy, float gz, float& nx, float& ny, float& nz);
__device__ __forceinline__ bool mc_grad_normal(const McParams& p, float x, float y, float z, float h, float& nx, float& ny, float& nz) {
    const float gx = mc_F(p, x + h, y, z) - mc_F(p, x - h, y, z);
    const float gy = mc_F(p, x, y + h, z) - mc_F(p, x, y - h, z);
    const float gz = mc_F(p, x, y, z + h) - mc_F(p, x, y, z - h);
    return mc_unit_gradient(gx, gy, gz, nx, ny, nz);
}
__device__ __forceinline__ bool mc_unit_gradient(float gx, float gy, float gz, float& nx, float& ny, float& nz) {
    const float len2 = (gx * gx + gy * gy) + gz * gz;
    if (len2 >= 1e-30f && !__builtin_isinf(len2)) {
        // v_rsq_f32 (1 ulp) instead of an IEEE sqrt and an IEEE divide: the normal is a tolerance quantity
        // 
        const float inv = __builtin_amdgcn_rsqf(len2);
        nx = gx * inv;
        ny = gy * inv;
        nz = gz * inv;
        return true;
    }
    const float len = __builtin_sqrtf(len2);
    if (len > 0.0f && !__builtin_isinf(len)) {  // tiny gradient: rsq would flush it
        const float inv = 1.0f / len;
        nx = gx * inv;
        ny = gy * inv;
        nz = gz * inv;
        return true;
    }
    return false;
}

#ifdef MC_TAB
// mc_emit's phase C for the vertices of a chunk that lie on edges along AX (compile-time: the lanes of a pass all take the
// same path), one lane per vertex.  f's one-variable sub-expressions come from the tables wherever a coordinate is a
// lattice coordinate -- both corners, and two of the three axes of the normal's central differences (coordinate, + h,
// - h: table kinds 0, 1, 2); only the interpolated coordinate (and its +- h) is evaluated.  Same operations on the same
// operands as mc_F / mc_grad_normal, so the same bits.
template <int AX>
__device__ __forceinline__ void mc_emit_axis(const McParams& p, const float* __restrict__ axis, const unsigned char* __restrict__ list, u32 cnt,
                                             const unsigned short* item, const u32* recw, const u32* segrec, float* vc, float iso, float h,
                                             bool want_normals, int lane) {
    const u32 ts = mc_tab_stride(p.n1);
    const float* __restrict__ tab = p.tab;
#ifdef MC_UNIT_SCALE
    const float sx = 1.0f, sy = 1.0f, sz = 1.0f;
#else
    const float sx = p.sx, sy = p.sy, sz = p.sz;
#endif
    for (u32 j0 = 0; j0 < cnt; j0 += 64u) {
        const u32 j = j0 + (u32)lane;
        if (j < cnt) {
            const u32 i = list[j];
            const u32 it = item[i];
            const int e = (int)(it >> 6);
            const u32 rw = recw[it & 63u];
            const u32 s2 = segrec[rw & 63u];
            const int bx = (int)(((s2 >> 22) & 7u) * (u32)MC_SEG + ((rw >> 6) & 0xFFu)) + (int)((MC_EDGE_OX >> e) & 1u);
            const int by = (int)(s2 & 2047u) + (int)((MC_EDGE_OY >> e) & 1u);
            const int bz = (int)((s2 >> 11) & 2047u) + (int)((MC_EDGE_OZ >> e) & 1u);
            const int bi = AX == 0 ? bx : AX == 1 ? by : bz;
            const float x0 = axis[bx], y0 = axis[by], z0 = axis[bz];
            const float c0 = AX == 0 ? x0 : AX == 1 ? y0 : z0;  // the edge's lower end on its axis
            const float c1 = axis[bi + 1];                        // ... its upper end
            float UX[MC_TAB_NX], UY[MC_TAB_NY], UZ[MC_TAB_NZ];    // at the lower corner
            float UX1[MC_TAB_NX], UY1[MC_TAB_NY], UZ1[MC_TAB_NZ]; // at the upper corner (differs on AX only)
#pragma unroll
            for (int k = 0; k < MC_TAB_NX; ++k) {
                UX[k] = mc_tab_ptr(tab, ts, 0, 0, k)[bx];
                UX1[k] = AX == 0 ? mc_tab_ptr(tab, ts, 0, 0, k)[bx + 1] : UX[k];
            }
#pragma unroll
            for (int k = 0; k < MC_TAB_NY; ++k) {
                UY[k] = mc_tab_ptr(tab, ts, 1, 0, k)[by];
                UY1[k] = AX == 1 ? mc_tab_ptr(tab, ts, 1, 0, k)[by + 1] : UY[k];
            }
#pragma unroll
            for (int k = 0; k < MC_TAB_NZ; ++k) {
                UZ[k] = mc_tab_ptr(tab, ts, 2, 0, k)[bz];
                UZ1[k] = AX == 2 ? mc_tab_ptr(tab, ts, 2, 0, k)[bz + 1] : UZ[k];
            }
            const float x1 = AX == 0 ? c1 : x0, y1 = AX == 1 ? c1 : y0, z1 = AX == 2 ? c1 : z0;
            const float v0 = mc_f_t(sx * x0, sy * y0, sz * z0, UX, UY, UZ), v1 = mc_f_t(sx * x1, sy * y1, sz * z1, UX1, UY1, UZ1);
            const float pu = mc_interp(iso, c0, c1, v0, v1);  // marching.cpp:557-583 for an edge walked upwards
            const float pd = mc_interp(iso, c1, c0, v1, v0);  // ... downwards (edges 2, 3, 6, 7)
            const float qx = AX == 0 ? pu : x0, qy = AX == 1 ? pu : y0, qz = AX == 2 ? pu : z0;
            float nx = 0.0f, ny = 0.0f, nz = 0.0f;
            if (want_normals) {
                // sub-expressions at (q, q + h, q - h) per axis: evaluated on AX, read on the two lattice axes
                float UXp[MC_TAB_NX], UXm[MC_TAB_NX], UYp[MC_TAB_NY], UYm[MC_TAB_NY], UZp[MC_TAB_NZ], UZm[MC_TAB_NZ];
                if (AX == 0) {
                    mc_f_ux(sx * qx, UX);
                    mc_f_ux(sx * (qx + h), UXp);
                    mc_f_ux(sx * (qx - h), UXm);
                } else {
#pragma unroll
                    for (int k = 0; k < MC_TAB_NX; ++k) {
                        UXp[k] = mc_tab_ptr(tab, ts, 0, 1, k)[bx];
                        UXm[k] = mc_tab_ptr(tab, ts, 0, 2, k)[bx];
                    }
                }
                if (AX == 1) {
                    mc_f_uy(sy * qy, UY);
                    mc_f_uy(sy * (qy + h), UYp);
                    mc_f_uy(sy * (qy - h), UYm);
                } else {
#pragma unroll
                    for (int k = 0; k < MC_TAB_NY; ++k) {
                        UYp[k] = mc_tab_ptr(tab, ts, 1, 1, k)[by];
                        UYm[k] = mc_tab_ptr(tab, ts, 1, 2, k)[by];
                    }
                }
                if (AX == 2) {
                    mc_f_uz(sz * qz, UZ);
                    mc_f_uz(sz * (qz + h), UZp);
                    mc_f_uz(sz * (qz - h), UZm);
                } else {
#pragma unroll
                    for (int k = 0; k < MC_TAB_NZ; ++k) {
                        UZp[k] = mc_tab_ptr(tab, ts, 2, 1, k)[bz];
                        UZm[k] = mc_tab_ptr(tab, ts, 2, 2, k)[bz];
                    }
                }
                const float fx = sx * qx, fy = sy * qy, fz = sz * qz;
                const float gx = mc_f_t(sx * (qx + h), fy, fz, UXp, UY, UZ) - mc_f_t(sx * (qx - h), fy, fz, UXm, UY, UZ);
                const float gy = mc_f_t(fx, sy * (qy + h), fz, UX, UYp, UZ) - mc_f_t(fx, sy * (qy - h), fz, UX, UYm, UZ);
                const float gz = mc_f_t(fx, fy, sz * (qz + h), UX, UY, UZp) - mc_f_t(fx, fy, sz * (qz - h), UX, UY, UZm);
                if (!mc_unit_gradient(gx, gy, gz, nx, ny, nz)) nx = __builtin_nanf("");  // marker: see D
            }
            float4* o = (float4*)(vc + 8u * i);
            o[0] = make_float4(qx, qy, qz, pd);
            o[1] = make_float4(nx, ny, nz, 0.0f);
        }
    }
}
#ifdef MC_TAB_SYM
// The same for equations whose three axes carry the SAME one-variable functions (MC_TAB_SYM, mc_expr.cpp; the gyroid): the
// vertices of all three edge directions in ONE pass, one lane per vertex.  The per-axis passes each fill their 64 lanes with
// a third of the chunk's ~125 vertices -- three steps at 42 lanes; this is two at 63.  What a lane evaluates (the functions
// at the interpolated coordinate and at +- h) it evaluates with mc_f_ux whatever its axis -- the lists are the same functions,
// mc_tab_sym_y / _z reorder them -- and what it reads from the tables it reads for all three axes, keeping per axis the one
// that applies: the same operations on the same operands as mc_emit_axis<AX>, lane by lane.
__device__ __forceinline__ void mc_emit_sym(const McParams& p, const float* __restrict__ axis, u32 cnt, const unsigned short* item,
                                            const u32* recw, const u32* segrec, float* vc, float iso, float h, bool want_normals, int lane) {
    const u32 ts = mc_tab_stride(p.n1);
    const float* __restrict__ tab = p.tab;
#ifdef MC_UNIT_SCALE
    const float sx = 1.0f, sy = 1.0f, sz = 1.0f;
#else
    const float sx = p.sx, sy = p.sy, sz = p.sz;
#endif
    for (u32 i0 = 0; i0 < cnt; i0 += 64u) {
        const u32 i = i0 + (u32)lane;
        if (i < cnt) {
            const u32 it = item[i];
            const int e = (int)(it >> 6);
            const u32 rw = recw[it & 63u];
            const u32 s2 = segrec[rw & 63u];
            const int bx = (int)(((s2 >> 22) & 7u) * (u32)MC_SEG + ((rw >> 6) & 0xFFu)) + (int)((MC_EDGE_OX >> e) & 1u);
            const int by = (int)(s2 & 2047u) + (int)((MC_EDGE_OY >> e) & 1u);
            const int bz = (int)((s2 >> 11) & 2047u) + (int)((MC_EDGE_OZ >> e) & 1u);
            const int ax = edge_axis(e);
            const int dx = ax == 0 ? 1 : 0, dy = ax == 1 ? 1 : 0, dz = ax == 2 ? 1 : 0;
            const float x0 = axis[bx], y0 = axis[by], z0 = axis[bz];
            const float x1 = axis[bx + dx], y1 = axis[by + dy], z1 = axis[bz + dz];  // (the upper corner differs on the edge's axis only)
            const float c0 = ax == 0 ? x0 : ax == 1 ? y0 : z0;
            const float c1 = ax == 0 ? x1 : ax == 1 ? y1 : z1;
            float UX[MC_TAB_NX], UY[MC_TAB_NY], UZ[MC_TAB_NZ], UX1[MC_TAB_NX], UY1[MC_TAB_NY], UZ1[MC_TAB_NZ];
#pragma unroll
            for (int k = 0; k < MC_TAB_NX; ++k) {
                UX[k] = mc_tab_ptr(tab, ts, 0, 0, k)[bx];
                UX1[k] = mc_tab_ptr(tab, ts, 0, 0, k)[bx + dx];
            }
#pragma unroll
            for (int k = 0; k < MC_TAB_NY; ++k) {
                UY[k] = mc_tab_ptr(tab, ts, 1, 0, k)[by];
                UY1[k] = mc_tab_ptr(tab, ts, 1, 0, k)[by + dy];
            }
#pragma unroll
            for (int k = 0; k < MC_TAB_NZ; ++k) {
                UZ[k] = mc_tab_ptr(tab, ts, 2, 0, k)[bz];
                UZ1[k] = mc_tab_ptr(tab, ts, 2, 0, k)[bz + dz];
            }
            const float v0 = mc_f_t(sx * x0, sy * y0, sz * z0, UX, UY, UZ), v1 = mc_f_t(sx * x1, sy * y1, sz * z1, UX1, UY1, UZ1);
            const float pu = mc_interp(iso, c0, c1, v0, v1);  // marching.cpp:557-583 for an edge walked upwards
            const float pd = mc_interp(iso, c1, c0, v1, v0);  // ... downwards (edges 2, 3, 6, 7)
            const float qx = ax == 0 ? pu : x0, qy = ax == 1 ? pu : y0, qz = ax == 2 ? pu : z0;
            float nx = 0.0f, ny = 0.0f, nz = 0.0f;
            if (want_normals) {
                // the functions at (q, q + h, q - h) on the edge's axis: evaluated (in x's order, then put into the axis' own)
                const float sa = ax == 0 ? sx : ax == 1 ? sy : sz;
                float C0[MC_TAB_NX], Cp[MC_TAB_NX], Cm[MC_TAB_NX];
                mc_f_ux(sa * pu, C0);
                mc_f_ux(sa * (pu + h), Cp);
                mc_f_ux(sa * (pu - h), Cm);
                float Y0[MC_TAB_NY], Yp[MC_TAB_NY], Ym[MC_TAB_NY], Z0[MC_TAB_NZ], Zp[MC_TAB_NZ], Zm[MC_TAB_NZ];
                mc_tab_sym_y(C0, Y0);
                mc_tab_sym_y(Cp, Yp);
                mc_tab_sym_y(Cm, Ym);
                mc_tab_sym_z(C0, Z0);
                mc_tab_sym_z(Cp, Zp);
                mc_tab_sym_z(Cm, Zm);
                // per axis: those when it is the edge's axis, the tables (lattice coordinate, + h, - h: kinds 0, 1, 2) when not
                float UXq[MC_TAB_NX], UXp[MC_TAB_NX], UXm[MC_TAB_NX], UYq[MC_TAB_NY], UYp[MC_TAB_NY], UYm[MC_TAB_NY], UZq[MC_TAB_NZ], UZp[MC_TAB_NZ],
                    UZm[MC_TAB_NZ];
#pragma unroll
                for (int k = 0; k < MC_TAB_NX; ++k) {
                    const float tp = mc_tab_ptr(tab, ts, 0, 1, k)[bx], tm = mc_tab_ptr(tab, ts, 0, 2, k)[bx];
                    UXq[k] = ax == 0 ? C0[k] : UX[k];
                    UXp[k] = ax == 0 ? Cp[k] : tp;
                    UXm[k] = ax == 0 ? Cm[k] : tm;
                }
#pragma unroll
                for (int k = 0; k < MC_TAB_NY; ++k) {
                    const float tp = mc_tab_ptr(tab, ts, 1, 1, k)[by], tm = mc_tab_ptr(tab, ts, 1, 2, k)[by];
                    UYq[k] = ax == 1 ? Y0[k] : UY[k];
                    UYp[k] = ax == 1 ? Yp[k] : tp;
                    UYm[k] = ax == 1 ? Ym[k] : tm;
                }
#pragma unroll
                for (int k = 0; k < MC_TAB_NZ; ++k) {
                    const float tp = mc_tab_ptr(tab, ts, 2, 1, k)[bz], tm = mc_tab_ptr(tab, ts, 2, 2, k)[bz];
                    UZq[k] = ax == 2 ? Z0[k] : UZ[k];
                    UZp[k] = ax == 2 ? Zp[k] : tp;
                    UZm[k] = ax == 2 ? Zm[k] : tm;
                }
                const float fx = sx * qx, fy = sy * qy, fz = sz * qz;
                const float gx = mc_f_t(sx * (qx + h), fy, fz, UXp, UYq, UZq) - mc_f_t(sx * (qx - h), fy, fz, UXm, UYq, UZq);
                const float gy = mc_f_t(fx, sy * (qy + h), fz, UXq, UYp, UZq) - mc_f_t(fx, sy * (qy - h), fz, UXq, UYm, UZq);
                const float gz = mc_f_t(fx, fy, sz * (qz + h), UXq, UYq, UZp) - mc_f_t(fx, fy, sz * (qz - h), UXq, UYq, UZm);
                if (!mc_unit_gradient(gx, gy, gz, nx, ny, nz)) nx = __builtin_nanf("");  // marker: see D
            }
            float4* o = (float4*)(vc + 8u * i);
            o[0] = make_float4(qx, qy, qz, pd);
            o[1] = make_float4(nx, ny, nz, 0.0f);
        }
    }
}
#endif
#endif

// One wave = one GROUP of 64 consecutive segments; its records (= active cells, written by mc_classify) are taken in
// CHUNKS of up to 64, lane = record, in sweep order.
//
//  A. record -> owning segment (binary search over the group's active-cell offsets, LDS), cell coordinates, code.
//  B. Who computes which vertex.  A crossed lattice edge is shared by up to 4 cells, 2 of them in the same z layer
//     (4 for an edge along z), and every one of them emits it in 1-2 triangles: ~6 output vertices per lattice edge.
//     Inside a chunk each lattice edge is computed ONCE: by the cell that has it at its low-x / low-y side when that
//     cell is in the chunk (lane + 1 for the x neighbour; the y and xy neighbours by a binary search over the chunk's
//     sorted cell keys), else by the cell itself.  Per lane: the set S of edges it computes and their first slot.
//     Then, still one lane per record: every corner of the cell's triangles is resolved to the slot of its vertex
//     (table row -> edge -> owner lane and edge -> slot) and the triangle is written to the chunk's list as three
//     slots -- the per-record part of what an output vertex needs is done 64 lanes wide, once.
//  C. One lane per computed VERTEX: two corner evaluations, the intersection point seen from the lower end (p_up) and
//     from the upper end (p_down; the table walks edges 2, 3, 6, 7 downwards and the reference's interp is not symmetric:
//     the soup must carry the bits of the direction its own cell uses), the gradient normal at p_up (ONE normal per
//     lattice edge) -> 8 floats in LDS: x y z (with p_up on the edge's axis), p_down, nx ny nz.
//  D. One lane per OUTPUT vertex (3 per triangle, reference emission order): list entry -> slot -> two 16-byte LDS
//     reads -> 24 bytes stored.  No evaluation of f, no table walk here.
#ifndef MC_VCAP
#define MC_VCAP 160  // vertices a chunk may compute (32 bytes each in LDS); a denser chunk is cut short
#endif
static_assert(MC_VCAP >= 96 && MC_VCAP <= 255, "8 records compute at most 96 vertices; slots are kept in 8 bits");
// What mc_emit needs before its first load returns travels as kernel arguments (the kernarg segment is in SGPRs when
// the wave starts); only iso and the vertex capacity -- which a replayed graph changes per frame -- come from *P.
struct McEmitK {
    const float* axis;    // lattice coordinates c[0..n1]
    const u32* overflow;  // mc_classify's "a record region overflowed" word
    u32 nseg;
    int n1, nchunk, z_begin;
    u32 flags;
    float step, sx, sy, sz;
    const float* tab;     // MC_TAB: McParams::tab
};
#ifndef MC_ONLY_INDEX_KERNELS  // (the module of the index kernels, compiled when MC_FLAG_INDEXED is first used, leaves the sweep kernels out)
extern "C" __global__ __launch_bounds__(64 * MC_WPB_ES) void mc_emit(const McParams* __restrict__ P, const u32* __restrict__ recs,
                                                                     const uint2* __restrict__ segcb, const uint2* __restrict__ grpoff,
                                                                     float* __restrict__ verts, const McEmitK k) {
    __shared__ __attribute__((aligned(16))) float s_vc[MC_WPB_ES][MC_VCAP * 8];  // computed vertices
    __shared__ u32 s_seg[MC_WPB_ES][64];    // per segment of the group: iy | iz << 11 | chunk << 22 | row-in-group << 25
    __shared__ u32 s_act[MC_WPB_ES][66];    // ... exclusive active-cell offsets (+ the total)
    __shared__ u32 s_tri[MC_WPB_ES][64];    // ... first triangle
    __shared__ u32 s_rbase[MC_WPB_ES][64];  // ... first record
    __shared__ u32 s_key[MC_WPB_ES][66];    // per record of the chunk: row-in-group << 11 | ix, ascending; ~0 past the end
    __shared__ u32 s_rec[MC_WPB_ES][64];    // ... segment | cellx << 6
    __shared__ u32 s_own[MC_WPB_ES][64];    // ... S | first slot << 12
    __shared__ u32 s_list[MC_WPB_ES][320];  // triangles of the chunk: slot of corner k << 8k | (axis taken from p_down, 3 = none) << 24 + 2k
    __shared__ unsigned short s_item[MC_WPB_ES][MC_VCAP];  // vertices to compute: lane | edge << 6
#if defined(MC_TAB) && !defined(MC_TAB_SYM)
    __shared__ unsigned char s_axl[MC_WPB_ES][3 * MC_VCAP];  // ... their indices, sorted by the axis of the edge
#endif
    // No table in LDS, no workgroup barrier: the waves of a workgroup are independent (a workgroup is only a launch
    // container; it holds its LDS until its slowest wave is done, so it is kept small).  The case table row and the
    // lattice coordinates are read from memory where they are needed (L1 / L2 hits), always BEFORE a chunk's stores.
    // (Two waves per group -- one running phases A-C, loads and arithmetic only, the other phase D, LDS reads and stores
    // only, double-buffered through LDS with a barrier per chunk -- were measured SLOWER: gyroid 1.43 ms against 1.28, sphere
    // 0.37 against 0.32.  The phases are chains of dependent latencies; what hides them is the number of independent chains
    // per CU, and the split halves the waves that run any one phase.)
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // make wave-uniformity visible
    const u32 ngroups = (k.nseg + 63u) / 64u;
    // MC_ES_SAMEGROUP (small grids: mc_runtime): the workgroup's waves all serve ONE group and take its 64-record chunks in
    // turn.  A group's chunks are a chain of dependent phases in one wave -- 4.6 us each -- and on a grid whose whole emit
    // fits the chip at once the kernel is as long as the longest chain (equation_3 513^3: 2 423 waves, all started in
    // the first microseconds, the longest 7 chunks = 35 of the kernel's 36 us).
#ifdef MC_ES_SAMEGROUP
    const u32 group = blockIdx.x, usub = (u32)w, ustep = 64u * (u32)MC_WPB_ES;
#else
    const u32 group = blockIdx.x * (u32)MC_WPB_ES + (u32)w, usub = 0u, ustep = 64u;
#endif
    if (group >= ngroups) return;  // (the grid is rounded up to whole workgroups)
    // every load that does not depend on another is issued up front: the group's offsets, per-segment counts, the
    // overflow word and the two words of *P
    const u32 seg_first = group * 64u;
    const u32 seg = seg_first + (u32)lane;
    const uint2 g0 = grpoff[group], g1 = grpoff[group + 1u];
    const uint2 cb = segcb[min(seg, k.nseg - 1u)];  // {triangles | active << 16, first record}
    const u32 rec_overflow = k.overflow[0];
    const float iso = P->iso;
    const u64 cap_tris = P->cap_tris;
    const u32 cnt = seg < k.nseg ? cb.x : 0u;
    // rec_overflow: mc_classify ran out of record space (the host grows the buffer and sweeps again)
    if (g0.y == g1.y || rec_overflow != 0u) return;  // no active cell in these 64 segments
    if (64u * usub >= g1.y - g0.y) return;            // (MC_ES_SAMEGROUP) fewer chunks than waves: nothing for this one
    McParams p;  // (mc_F reads the scale factors from it)
    p.sx = k.sx;
    p.sy = k.sy;
    p.sz = k.sz;
    p.n1 = k.n1;
    p.tab = k.tab;
    const float* __restrict__ axis = k.axis;

    // the scan gives the group's first triangle; the prefix inside the group is a wavefront scan of the per-segment
    // counts (triangles | active cells << 16)
    const u32 ctri = cnt & 0xFFFFu, cact = cnt >> 16;
    const u32 itri = wave_inclusive_scan(ctri), iact = wave_inclusive_scan(cact);
    const u32 nrec = (u32)__builtin_amdgcn_readlane((int)iact, 63);  // records of the group

    u32* list = s_list[w];
    unsigned short* item = s_item[w];
    float* vc = s_vc[w];
    u32* segrec = s_seg[w];
    u32* actoff = s_act[w];
    u32* trioff = s_tri[w];
    u32* rbase = s_rbase[w];
    u32* key = s_key[w];
    u32* recw = s_rec[w];
    u32* own = s_own[w];
    const int n1 = k.n1;
    {
        const u32 sg = min(seg, k.nseg - 1u);
        const u32 rowidx = sg / (u32)k.nchunk;
        const u32 ch = sg - rowidx * (u32)k.nchunk;
        const u32 lz = rowidx / (u32)n1;
        const u32 iy = rowidx - lz * (u32)n1;
        const u32 row_first = seg_first / (u32)k.nchunk;
        segrec[lane] = iy | ((u32)(k.z_begin + (int)lz) << 11) | (ch << 22) | ((rowidx - row_first) << 25);
        actoff[lane] = iact - cact;
        trioff[lane] = g0.x + (itri - ctri);
        rbase[lane] = cb.y;
        if (lane == 63) actoff[64] = nrec;
        if (lane < 2) key[64 + lane] = 0xFFFFFFFFu;
    }
    const float h = 0.5f * k.step;
    const bool want_normals = (k.flags & 1u) != 0u;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();

    // ---- A. a lane's record: owning segment = the largest s with actoff[s] <= r (empty segments repeat the value)
    auto find_segment = [&](u32 r) {
        u32 lo = 0, hi = 64;
#pragma unroll
        for (int it = 0; it < 6; ++it) {
            const u32 mid = (lo + hi) >> 1;
            if (actoff[mid] <= r) lo = mid; else hi = mid;
        }
        return lo;
    };
    // (a wave's records come in units of 64: [ustart, uend); a unit is worked off in one chunk, or in several when it is cut)
    u32 ustart = 64u * usub;
    if (ustart >= nrec) return;
    u32 uend = min(ustart + 64u, nrec);
    u32 lo = find_segment(ustart + (u32)lane);
    u32 rec = ustart + (u32)lane < nrec ? recs[rbase[lo] + (ustart + (u32)lane - actoff[lo])] : 0u;
    for (u32 r0 = ustart; r0 < nrec;) {
        const u32 gtri0 = trioff[lo] + (rec >> 20);
        const u32 sr = segrec[lo];
        const u32 code = (rec >> 8) & 0xFFu;
        const u32 ix = ((sr >> 22) & 7u) * (u32)MC_SEG + (rec & 0xFFu);
        const u32 iy = sr & 2047u;
        const u32 mykey = ((sr >> 25) << 11) | ix;
        const u32 cm = crossed_edges(code);

        // ---- B. ownership inside the chunk; a chunk that would compute more than MC_VCAP vertices is cut in halves
        u32 nvalid = min(64u, uend - r0);
        u32 S, sb, M, vx, vy, vxy, ny_lane;
        for (;;) {
            const bool valid = (u32)lane < nvalid;
            key[lane] = valid ? mykey : 0xFFFFFFFFu;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // x neighbour (ix + 1, same row): the next record if there is one
            vx = key[lane + 1] == mykey + 1u ? 1u : 0u;
            // y neighbour (ix, row + 1) -- not across the top of a layer --: lower bound over the 64 sorted keys
            const u32 ty = mykey + (1u << 11);
            u32 pos = 0;
#pragma unroll
            for (int st = 32; st >= 1; st >>= 1)
                if (key[pos + st - 1] < ty) pos += st;
            // (with constraints the y neighbour can be a skipped cell -- no record -- while the xy neighbour is there:
            // the lower bound then lands on the xy neighbour itself)
            vy = (iy + 1u < (u32)n1 && key[pos] == ty) ? 1u : 0u;
            vxy = (iy + 1u < (u32)n1 && key[pos + vy] == ty + 1u) ? 1u : 0u;
            ny_lane = pos;
            // edges some other lane of the chunk computes: 1, 5, 9 = the x neighbour's 3, 7, 8; 2, 6, 11 = the y
            // neighbour's 0, 4, 8; 10 = the xy neighbour's 8 (else the x neighbour's 11, else the y neighbour's 9)
            const u32 remote = (vx ? 0x222u : 0u) | (vy ? 0x844u : 0u) | ((vx | vy | vxy) ? 0x400u : 0u);
            S = valid ? (cm & ~remote) : 0u;
            const u32 c = (u32)__builtin_popcount(S);
            const u32 incl = wave_inclusive_scan(c);
            sb = incl - c;
            M = (u32)__builtin_amdgcn_readlane((int)incl, 63);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (M <= (u32)MC_VCAP) break;
            // too many: keep the records whose vertices fit with a little room (a record that loses a neighbour to the
            // cut computes up to 4 more itself), at least 8 (8 records compute at most 96 vertices)
            const u32 fit = (u32)__builtin_popcountll(__ballot(incl + 8u <= (u32)MC_VCAP));
            nvalid = max(8u, min(fit, nvalid - 1u));
        }
        const bool valid = (u32)lane < nvalid;
        const u32 nt = valid ? (rec >> 17) & 7u : 0u;
        // the chunk's triangles are one contiguous range of the global order
        const u32 first = (u32)__builtin_amdgcn_readfirstlane((int)gtri0);
        const u32 chunk_t = (u32)__builtin_amdgcn_readlane((int)(gtri0 + nt), (int)nvalid - 1) - first;
        const u32 myow = S | (sb << 12);
        recw[lane] = lo | ((rec & 0xFFu) << 6);
        own[lane] = myow;
        {
            u32 m = S, j = sb;
            while (m) {
                const u32 e = (u32)__builtin_ctz(m);
                m &= m - 1u;
                item[j++] = (unsigned short)((u32)lane | (e << 6));
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (nt) {
            // Slot of the vertex on each of this cell's 12 edges: its own (edge set S, first slot sb) or the neighbour's
            // that computes it -- 1, 5, 9 = the x neighbour's 3, 7, 8; 2, 6, 11 = the y neighbour's 0, 4, 8; 10 = the xy
            // neighbour's 8, else the x neighbour's 11, else the y neighbour's 9.  Edge numbers are compile-time here:
            // each slot is `first slot + popcount(edge set below the edge)`, two instructions.
            const u32 owx = own[min((u32)lane + 1u, 63u)], owy = own[ny_lane], owxy = own[min(ny_lane + vy, 63u)];
#define MC_SLOT(OW, E) ((((OW) >> 12) & 0xFFu) + (u32)__builtin_popcount((OW) & ((1u << (E)) - 1u)))
#define MC_PICK(E, REMOTE) (((S >> (E)) & 1u) ? MC_SLOT(myow, E) : (REMOTE))
            const u32 s10 = vxy ? MC_SLOT(owxy, 8) : vx ? MC_SLOT(owx, 11) : MC_SLOT(owy, 9);
            const u32 sl0 = MC_SLOT(myow, 0) | (MC_PICK(1, MC_SLOT(owx, 3)) << 8) | (MC_PICK(2, MC_SLOT(owy, 0)) << 16) | (MC_SLOT(myow, 3) << 24);
            const u32 sl1 = MC_SLOT(myow, 4) | (MC_PICK(5, MC_SLOT(owx, 7)) << 8) | (MC_PICK(6, MC_SLOT(owy, 4)) << 16) | (MC_SLOT(myow, 7) << 24);
            const u32 sl2 = MC_SLOT(myow, 8) | (MC_PICK(9, MC_SLOT(owx, 8)) << 8) | (MC_PICK(10, s10) << 16) | (MC_PICK(11, MC_SLOT(owy, 8)) << 24);
#undef MC_PICK
#undef MC_SLOT
            const u32 rowi = ((rec >> 16) & 1u) ? 255u - code : code;  // marching.cpp:542-547
            const u64 trow = c_tri_row[rowi];
            for (u32 t = 0; t < nt; ++t) {
                const u32 tri12 = (u32)(trow >> (12u * t)) & 0xFFFu;  // the triangle's three edges (marching.cpp:586-594)
                u32 ent = 0;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const u32 e = (tri12 >> (4 * k)) & 0xFu;
                    const u32 pack = e < 4u ? sl0 : e < 8u ? sl1 : sl2;
                    const u32 slot = (pack >> (8u * (e & 3u))) & 0xFFu;
                    // an edge the table walks downwards (2, 3, 6, 7: x, y, x, y) takes its axis coordinate from p_down
                    const u32 dir = ((MC_EDGE_DOWN >> e) & 1u) ? (e & 1u) : 3u;
                    ent |= (slot << (8 * k)) | (dir << (24 + 2 * k));
                }
                list[gtri0 - first + t] = ent;
            }
        }

        // The next chunk's records are fetched NOW: vmcnt retires in order, so a load issued behind this chunk's vertex
        // stores (phase D) would wait until they have all landed; issued here its latency hides under phase C, and the
        // wait for it in front of D finds the previous chunk's stores long gone.
        u32 r0_n = r0 + nvalid;  // the rest of this unit, or this wave's next unit
        if (r0_n >= uend) {
            ustart += ustep;
            r0_n = ustart;
            uend = min(ustart + 64u, nrec);
        }
        const u32 rn = r0_n + (u32)lane;
        const u32 lo_n = find_segment(rn);
        u32 rec_n = 0u;
        if (rn < nrec) rec_n = recs[rbase[lo_n] + (rn - actoff[lo_n])];

        // ---- C. one lane per computed vertex
#if defined(MC_TAB_SYM)
        mc_emit_sym(p, axis, M, item, recw, segrec, vc, iso, h, want_normals, lane);
#elif defined(MC_TAB)
        {
            // the chunk's vertices by the axis of their edge, then one pass per axis (mc_emit_axis)
            unsigned char* axl = s_axl[w];
            u32 cn0 = 0u, cn1 = 0u, cn2 = 0u;
            for (u32 i0 = 0; i0 < M; i0 += 64u) {
                const u32 i = i0 + (u32)lane;
                const bool v = i < M;
                const int ax = v ? edge_axis((int)(item[i] >> 6)) : 3;
                const u64 m0 = __ballot(ax == 0), m1 = __ballot(ax == 1), m2 = __ballot(ax == 2);
                if (ax == 0) axl[cn0 + mask_rank(m0)] = (unsigned char)i;
                if (ax == 1) axl[MC_VCAP + cn1 + mask_rank(m1)] = (unsigned char)i;
                if (ax == 2) axl[2 * MC_VCAP + cn2 + mask_rank(m2)] = (unsigned char)i;
                cn0 += (u32)__builtin_popcountll(m0);
                cn1 += (u32)__builtin_popcountll(m1);
                cn2 += (u32)__builtin_popcountll(m2);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            mc_emit_axis<0>(p, axis, axl, cn0, item, recw, segrec, vc, iso, h, want_normals, lane);
            mc_emit_axis<1>(p, axis, axl + MC_VCAP, cn1, item, recw, segrec, vc, iso, h, want_normals, lane);
            mc_emit_axis<2>(p, axis, axl + 2 * MC_VCAP, cn2, item, recw, segrec, vc, iso, h, want_normals, lane);
        }
#else
        for (u32 i0 = 0; i0 < M; i0 += 64u) {
            const u32 i = i0 + (u32)lane;
            if (i < M) {
                const u32 it = item[i];
                const int e = (int)(it >> 6);
                const u32 rw = recw[it & 63u];
                const u32 s2 = segrec[rw & 63u];
                const int bx = (int)(((s2 >> 22) & 7u) * (u32)MC_SEG + ((rw >> 6) & 0xFFu)) + (int)((MC_EDGE_OX >> e) & 1u);
                const int by = (int)(s2 & 2047u) + (int)((MC_EDGE_OY >> e) & 1u);
                const int bz = (int)((s2 >> 11) & 2047u) + (int)((MC_EDGE_OZ >> e) & 1u);
                const int ax = edge_axis(e);
                const float x0 = axis[bx], y0 = axis[by], z0 = axis[bz];
                const float c0 = ax == 0 ? x0 : ax == 1 ? y0 : z0;              // the edge's lower end on its axis
                const float c1 = axis[(ax == 0 ? bx : ax == 1 ? by : bz) + 1];  // ... its upper end
                const float x1 = ax == 0 ? c1 : x0, y1 = ax == 1 ? c1 : y0, z1 = ax == 2 ? c1 : z0;
                // the same mc_F, the same operands as the classification of the two corners (marching.cpp:475-479)
                const float v0 = mc_F(p, x0, y0, z0), v1 = mc_F(p, x1, y1, z1);
                const float pu = mc_interp(iso, c0, c1, v0, v1);  // marching.cpp:557-583 for an edge walked upwards
                const float pd = mc_interp(iso, c1, c0, v1, v0);  // ... downwards (edges 2, 3, 6, 7)
                // the two other coordinates are lattice coordinates (x_s + t*0 in the reference)
                const float qx = ax == 0 ? pu : x0, qy = ax == 1 ? pu : y0, qz = ax == 2 ? pu : z0;
                float nx = 0.0f, ny = 0.0f, nz = 0.0f;
                if (want_normals && !mc_grad_normal(p, qx, qy, qz, h, nx, ny, nz)) nx = __builtin_nanf("");  // marker: see D
                float4* o = (float4*)(vc + 8u * i);
                o[0] = make_float4(qx, qy, qz, pd);
                o[1] = make_float4(nx, ny, nz, 0.0f);
            }
        }
#endif
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();

        asm volatile("s_waitcnt vmcnt(0)" : "+v"(rec_n)::"memory");  // the prefetched records, before the first store of D
        // ---- D. one lane per output vertex
        const u32 nverts = 3u * chunk_t;
        const u64 room = cap_tris > (u64)first ? (cap_tris - (u64)first) * 72ull : 0ull;  // bytes of the vertex array from `first` on
        const __amdgpu_buffer_rsrc_t vrsrc = __builtin_amdgcn_make_buffer_rsrc(verts + (u64)first * 18ull, 0, (int)(room < 0x7FFFFFF0ull ? room : 0x7FFFFFF0ull), 0x00020000);
        for (u32 v0 = 0; v0 < nverts; v0 += 64u) {
            const u32 vid = v0 + (u32)lane;
            if (vid < nverts) {
                const u32 tri = vid / 3u;
                const int k = (int)(vid - 3u * tri);
                const u32 ent = list[tri];
                const u32 slot = (ent >> (8 * k)) & 0xFFu, dir = (ent >> (24 + 2 * k)) & 3u;
                const float4 a = ((const float4*)vc)[2u * slot], b = ((const float4*)vc)[2u * slot + 1u];
                const float qx = dir == 0u ? a.w : a.x, qy = dir == 1u ? a.w : a.y, qz = dir == 2u ? a.w : a.z;
                float nx = b.x, ny = b.y, nz = b.z;
                if (nx != nx) {  // degenerate gradient: the triangle's own normal cross(B-A, C-A)
                    float px[3], py[3], pz[3];
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        const float4 q = ((const float4*)vc)[2u * ((ent >> (8 * c)) & 0xFFu)];
                        const u32 d = (ent >> (24 + 2 * c)) & 3u;
                        px[c] = d == 0u ? q.w : q.x;
                        py[c] = d == 1u ? q.w : q.y;
                        pz[c] = d == 2u ? q.w : q.z;
                    }
                    const float e1x = px[1] - px[0], e1y = py[1] - py[0], e1z = pz[1] - pz[0];
                    const float e2x = px[2] - px[0], e2y = py[2] - py[0], e2z = pz[2] - pz[0];
                    const float cxn = e1y * e2z - e1z * e2y;
                    const float cyn = e1z * e2x - e1x * e2z;
                    const float czn = e1x * e2y - e1y * e2x;
                    const float l = __builtin_sqrtf((cxn * cxn + cyn * cyn) + czn * czn);
                    nx = ny = nz = 0.0f;
                    if (l > 0.0f && !__builtin_isinf(l)) {
                        const float inv = 1.0f / l;
                        nx = cxn * inv;
                        ny = cyn * inv;
                        nz = czn * inv;
                    }
                }
                // 24 bytes per vertex through a buffer descriptor over the chunk's part of the vertex array: a scalar base
                // and a 32-bit lane offset (no 64-bit address arithmetic per lane), and the hardware's range check drops
                // what lies beyond the buffer's capacity (the host then grows it and sweeps again)
                typedef float f32x4 __attribute__((ext_vector_type(4)));
                typedef float f32x2 __attribute__((ext_vector_type(2)));
                f32x4 s0;
                s0.x = qx; s0.y = qy; s0.z = qz; s0.w = nx;
                f32x2 s1;
                s1.x = ny; s1.y = nz;
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, s0), vrsrc, vid * 24u, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_t, s1), vrsrc, vid * 24u + 16u, 0, 0);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        r0 = r0_n;
        lo = lo_n;
        rec = rec_n;
    }
}
#endif  // MC_ONLY_INDEX_KERNELS

struct McVert {
    float x, y, z;
};

// position of triangle-vertex `slot` (0..14) of table row `row` in cell (ix,iy,iz): marching.cpp:557-583 (edge
// interpolation from corner v1 to corner v2 of the edge table) with Marching::interp (:437-446).  The reference calls
// interp for x, y and z; on the two axes the edge does not run along it returns x_s + t*0 = the lattice coordinate, so
// only the edge's own axis is interpolated here -- from its lower end (p_up) and from its upper end (p_down): the
// table walks edges 2, 3, 6, 7 downwards and interp is not symmetric, the soup carries the bits of the table's direction.
// qn: the point with p_up on the axis, which is where DESIGN.md N1 takes the normal (one normal per lattice edge).
__device__ __forceinline__ McVert mc_vertex(const McParams& p, const float* s_axis, const u64* s_row, int row, int slot, int ix,
                                            int iy, int iz, McVert& qn) {
    const int e = (int)((s_row[row] >> (4 * slot)) & 0xF);
    const int bx = ix + (int)((MC_EDGE_OX >> e) & 1u), by = iy + (int)((MC_EDGE_OY >> e) & 1u), bz = iz + (int)((MC_EDGE_OZ >> e) & 1u);
    const int ax = edge_axis(e);
    const float x0 = s_axis[bx], y0 = s_axis[by], z0 = s_axis[bz];
    const float c0 = ax == 0 ? x0 : ax == 1 ? y0 : z0;                // the edge's lower end on its axis
    const float c1 = s_axis[(ax == 0 ? bx : ax == 1 ? by : bz) + 1];  // ... its upper end
    // the same mc_F, the same operands as the classification of the two corners (marching.cpp:475-479)
    const float v0 = mc_F(p, x0, y0, z0);
    const float v1 = mc_F(p, ax == 0 ? c1 : x0, ax == 1 ? c1 : y0, ax == 2 ? c1 : z0);
    const float pu = mc_interp(p.iso, c0, c1, v0, v1);
    const float pa = ((MC_EDGE_DOWN >> e) & 1u) ? mc_interp(p.iso, c1, c0, v1, v0) : pu;
    qn.x = ax == 0 ? pu : x0;
    qn.y = ax == 1 ? pu : y0;
    qn.z = ax == 2 ? pu : z0;
    McVert r;
    r.x = ax == 0 ? pa : x0;
    r.y = ax == 1 ? pa : y0;
    r.z = ax == 2 ? pa : z0;
    return r;
}

// mc_emit_direct -- the emit kernel for CHEAP f (mc_runtime picks it when f costs less than the bookkeeping that
// sharing a vertex between its ~6 output copies needs: measured).
// A workgroup = 8 consecutive GROUPS of 64 segments; the unit of work is a CHUNK of 64 records of one of them, taken by
// whichever wave is free (an LDS counter per group; a wave serves its own group first -- see the comment at the barrier
// below).  Phase 1, one lane per RECORD (= active cell, written by mc_classify): find the owning segment by binary search
// over the group's active-cell offsets (LDS), read the record, and expand its triangles into 4-byte work items in LDS --
// the list index is the triangle's position in the reference's emission order.  Phase 2, one lane per output VERTEX: edge
// lookup (nibble-packed table row in LDS), two corner evaluations, the interpolation, the central-difference gradient of f
// for the normal, 24-byte store.
#ifndef MC_ONLY_INDEX_KERNELS  // (the module of the index kernels, compiled when MC_FLAG_INDEXED is first used, leaves the sweep kernels out)
extern "C" __global__ __launch_bounds__(64 * MC_WPB_E) void mc_emit_direct(const McParams* __restrict__ P, const u32* __restrict__ recs,
                                                                     const uint2* __restrict__ segcb, const uint2* __restrict__ grpoff,
                                                                     float* __restrict__ verts) {
    __shared__ u64 s_row[256];
    __shared__ u32 s_list[MC_WPB_E][MC_LIST_CAP];
    __shared__ u32 s_seg[MC_WPB_E][64];
    __shared__ u32 s_act[MC_WPB_E][66];
    __shared__ u32 s_tri[MC_WPB_E][64];
    __shared__ u32 s_rbase[MC_WPB_E][64];
    __shared__ u32 s_next[MC_WPB_E];  // per group of the workgroup: the next chunk of records nobody has taken yet
    // the whole lattice coordinate table (n1+1 <= 2002 floats): the vertex phase gathers 6
    // coordinates per vertex, and vmcnt retires in order -- a global gather issued after the
    // previous iteration's vertex stores would wait for those stores to land
    extern __shared__ float s_axis[];  // dynamic: n1+1 floats (mc_runtime passes the size)
    // The kernel is latency-bound (a group holds little work), so every load that does not depend
    // on another is issued up front: the group's offsets and per-segment counts here, the tables
    // below, all in flight together; a block none of whose 4 groups has an active cell leaves
    // before it touches the tables.
    const McParams p = *P;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // make wave-uniformity visible
    const u32 ngroups = (p.nseg + 63u) / 64u;
    const u32 group = min(blockIdx.x * (u32)MC_WPB_E + (u32)w, ngroups - 1u);  // the grid is rounded up to whole blocks
    const bool in_range = blockIdx.x * (u32)MC_WPB_E + (u32)w < ngroups;
    const u32 seg_first = group * 64u;
    const u32 seg = seg_first + (u32)lane;
    const uint2 g0 = grpoff[group], g1 = grpoff[group + 1u];
    const uint2 cb = seg < p.nseg ? segcb[seg] : make_uint2(0u, 0u);  // {triangles | active << 16, first record}
    const u32 cnt = cb.x;
    const u32 rec_overflow = p.overflow[0];
    {
        // issue every table load before the first wait: a copy loop would pay one full memory
        // latency per iteration (the compiler waits for each load before its LDS store)
        // (the axis buffer is padded to 2052 floats by mc_runtime, so the 16-byte loads need no clamp)
        constexpr int NT = 64 * MC_WPB_E, NV = (512 + NT - 1) / NT, NR = (256 + NT - 1) / NT;
        const float4* __restrict__ gaxis = (const float4*)P->axis;
        const int n1p = P->n1;
        float4 t[NV];
        u64 trow[NR];
#pragma unroll
        for (int k = 0; k < NV; ++k) t[k] = gaxis[min((int)threadIdx.x + NT * k, 511)];
#pragma unroll
        for (int k = 0; k < NR; ++k) trow[k] = c_tri_row[((int)threadIdx.x + NT * k) & 255];
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int i = (int)threadIdx.x + NT * k;
            if (i < 512 && 4 * i <= n1p) ((float4*)s_axis)[i] = t[k];
        }
#pragma unroll
        for (int k = 0; k < NR; ++k)
            if ((int)threadIdx.x + NT * k < 256) s_row[(int)threadIdx.x + NT * k] = trow[k];
    }
    // rec_overflow: mc_classify ran out of record space (the host grows the buffer and sweeps again)
    const bool mine = in_range && g0.y != g1.y && rec_overflow == 0u;  // 64 segments with an active cell
    if (!__syncthreads_or(mine ? 1 : 0)) return;  // also the barrier that publishes the tables
    // (from here on every wave of the workgroup stays: a wave whose own group is empty helps the others, below)
    const u32 ctri = mine ? cnt & 0xFFFFu : 0u, cact = mine ? cnt >> 16 : 0u;
    // the scan gives the group's first triangle; the prefix inside the group is a wavefront scan of
    // the per-segment counts (triangles | active cells << 16)
    const u32 itri = wave_inclusive_scan(ctri), iact = wave_inclusive_scan(cact);
    const int n1 = p.n1;
    {
        const u32 sg = min(seg, p.nseg - 1u);
        const u32 rowidx = sg / (u32)p.nchunk;
        const u32 ch = sg - rowidx * (u32)p.nchunk;
        const u32 lz = rowidx / (u32)n1;
        const u32 iy = rowidx - lz * (u32)n1;
        s_seg[w][lane] = iy | ((u32)(p.z_begin + (int)lz) << 11) | (ch << 22);
        s_act[w][lane] = iact - cact;            // group-local first record of the segment
        s_tri[w][lane] = g0.x + (itri - ctri);   // its first triangle
        s_rbase[w][lane] = cb.y;
        if (lane == 63) {
            s_act[w][64] = iact;                 // records of the group
            s_next[w] = 0u;
        }
    }
    const float h = 0.5f * p.step;
    const bool want_normals = (p.flags & 1u) != 0u;
    // The groups of a workgroup differ in work by up to 10x (a group where the surface runs along the rows holds hundreds of
    // records, its neighbours a few dozen), and a workgroup keeps its LDS and its wave slots until its last wave is done:
    // round 2's "one wave, one group" left 19 of a CU's 32 wave slots busy on average (SQ_WAVE_CYCLES / SQ_BUSY_CYCLES).
    // Now the unit of work is a CHUNK of 64 records of any of the workgroup's groups: a wave takes the chunks of its own
    // group first (an LDS counter per group), then those of the groups of its neighbours, until all are taken.
    __syncthreads();  // every wave's segment tables and chunk counter are in LDS

    u32* list = s_list[w];
    u32 nlist = 0;     // triangles staged
    u32 listbase = 0;  // global index of list[0]
    const u32* segrec = s_seg[w];  // the tables of the group the current chunk belongs to

    // drains the staged triangles: one lane per vertex
    auto flush = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const u32 nverts = 3u * nlist;
        const u64 room = p.cap_tris > (u64)listbase ? (p.cap_tris - (u64)listbase) * 72ull : 0ull;  // bytes of the vertex array from `listbase` on
        const __amdgpu_buffer_rsrc_t vrsrc = __builtin_amdgcn_make_buffer_rsrc(verts + (u64)listbase * 18ull, 0, (int)(room < 0x7FFFFFF0ull ? room : 0x7FFFFFF0ull), 0x00020000);
        for (u32 v0 = 0; v0 < nverts; v0 += 64u) {
            const u32 vid = v0 + (u32)lane;
            if (vid < nverts) {
                const u32 tri = vid / 3u;
                const int k = (int)(vid - 3u * tri);
                const u32 e = list[tri];
                const u32 sr = segrec[e & 63u];
                const int cellx = (int)((e >> 6) & 255u);
                const int code = (int)((e >> 14) & 255u);
                const int row = ((e >> 22) & 1u) ? 255 - code : code;
                const int t = (int)((e >> 23) & 7u);
                const int iy = (int)(sr & 2047u), iz = (int)((sr >> 11) & 2047u);
                const int ix = (int)(sr >> 22) * MC_SEG + cellx;
                McVert qn;  // where the normal is taken: the edge's intersection point seen from its LOWER end
                const McVert q = mc_vertex(p, s_axis, s_row, row, 3 * t + k, ix, iy, iz, qn);
                float nx = 0.0f, ny = 0.0f, nz = 0.0f;
                if (want_normals) {
                    // n = g/|g|, g = central difference of F at that point, h = step/2
                    const float gx = mc_F(p, qn.x + h, qn.y, qn.z) - mc_F(p, qn.x - h, qn.y, qn.z);
                    const float gy = mc_F(p, qn.x, qn.y + h, qn.z) - mc_F(p, qn.x, qn.y - h, qn.z);
                    const float gz = mc_F(p, qn.x, qn.y, qn.z + h) - mc_F(p, qn.x, qn.y, qn.z - h);
                    const float len2 = (gx * gx + gy * gy) + gz * gz;
                    const float len = __builtin_sqrtf(len2);
                    if (len2 >= 1e-30f && !__builtin_isinf(len2)) {
                        // v_rsq_f32 (1 ulp) instead of an IEEE sqrt and an IEEE divide: the normal is a
                        // tolerance quantity, and this kernel is VALU-bound
                        const float inv = __builtin_amdgcn_rsqf(len2);
                        nx = gx * inv;
                        ny = gy * inv;
                        nz = gz * inv;
                    } else if (len > 0.0f && !__builtin_isinf(len)) {  // tiny gradient: rsq would flush it
                        const float inv = 1.0f / len;
                        nx = gx * inv;
                        ny = gy * inv;
                        nz = gz * inv;
                    } else {  // degenerate gradient: the triangle's own normal cross(B-A, C-A)
                        const McVert a = mc_vertex(p, s_axis, s_row, row, 3 * t + 0, ix, iy, iz, qn);
                        const McVert b = mc_vertex(p, s_axis, s_row, row, 3 * t + 1, ix, iy, iz, qn);
                        const McVert c = mc_vertex(p, s_axis, s_row, row, 3 * t + 2, ix, iy, iz, qn);
                        const float e1x = b.x - a.x, e1y = b.y - a.y, e1z = b.z - a.z;
                        const float e2x = c.x - a.x, e2y = c.y - a.y, e2z = c.z - a.z;
                        const float cxn = e1y * e2z - e1z * e2y;
                        const float cyn = e1z * e2x - e1x * e2z;
                        const float czn = e1x * e2y - e1y * e2x;
                        const float l = __builtin_sqrtf((cxn * cxn + cyn * cyn) + czn * czn);
                        if (l > 0.0f && !__builtin_isinf(l)) {
                            const float inv = 1.0f / l;
                            nx = cxn * inv;
                            ny = cyn * inv;
                            nz = czn * inv;
                        }
                    }
                }
                // 24 bytes per vertex through a buffer descriptor over the staged part of the vertex array: scalar base,
                // 32-bit lane offset, and the range check drops what lies beyond the buffer's capacity
                typedef float f32x4 __attribute__((ext_vector_type(4)));
                typedef float f32x2 __attribute__((ext_vector_type(2)));
                f32x4 s0;
                s0.x = q.x; s0.y = q.y; s0.z = q.z; s0.w = nx;
                f32x2 s1;
                s1.x = ny; s1.y = nz;
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, s0), vrsrc, vid * 24u, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_t, s1), vrsrc, vid * 24u + 16u, 0, 0);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        listbase += nlist;
        nlist = 0;
    };


    // The next chunk is taken -- and its records are requested -- BEFORE the current chunk's vertices are stored: vmcnt
    // retires in order, so the load is back long before the stores have drained, and the wave never waits a memory round
    // trip between two chunks.
    struct Unit {
        int wv;        // whose group
        u32 r0, nrec;  // first record of the chunk, records of the group
        u32 lo, rec;   // lane = record: its segment, the record word
    };
    auto take = [&](Unit& u) __attribute__((always_inline)) {
        u.rec = 0u;
        u.lo = 0u;
        // which groups of the workgroup still hold a chunk nobody has taken: lane i looks at group (w + i) mod MC_WPB_E (a
        // power of two), the wave's own first.  The counters only grow: a group seen exhausted stays so, one seen open may be
        // gone by the time of the atomic.  (Round 4: walking the groups one by one, an LDS atomic and a read each, cost a
        // wave eight dependent LDS round trips before it could leave.)
        const int gi = (w + lane) & (MC_WPB_E - 1);
        u64 cand = __ballot(lane < MC_WPB_E && s_next[gi] < s_act[gi][64]);
        while (cand) {
            const int di = __builtin_ctzll(cand);
            cand &= cand - 1ull;
            u.wv = (w + di) & (MC_WPB_E - 1);
            const u32* actoff = s_act[u.wv];
            u.nrec = actoff[64];
            u32 r0 = 0u;
            if (lane == 0) r0 = __hip_atomic_fetch_add(&s_next[u.wv], 64u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            u.r0 = (u32)__builtin_amdgcn_readfirstlane((int)r0);
            if (u.r0 >= u.nrec) continue;
            const u32 r = u.r0 + (u32)lane;
            // owning segment: the largest s with actoff[s] <= r (empty segments repeat the value)
            u32 lo = 0, hi = 64;
#pragma unroll
            for (int it = 0; it < 6; ++it) {
                const u32 mid = (lo + hi) >> 1;
                if (actoff[mid] <= r) lo = mid; else hi = mid;
            }
            u.lo = lo;
            if (r < u.nrec) u.rec = recs[s_rbase[u.wv][lo] + (r - actoff[lo])];
            return true;
        }
        return false;
    };
    Unit cur, nxt;
    bool have = take(cur);
    while (have) {
        const bool have_next = take(nxt);  // (its record load is in flight across this chunk's work)
        segrec = s_seg[cur.wv];
        const u32 rec = cur.rec, lo = cur.lo;
        const u32 gtri0 = s_tri[cur.wv][lo] + (rec >> 20);
        const u32 nt = (rec >> 17) & 7u;  // (lanes beyond the group's records hold rec = 0)
        // the chunk's triangles are one contiguous range of the global order
        const int lv = (int)min(63u, cur.nrec - 1u - cur.r0);
        listbase = (u32)__builtin_amdgcn_readfirstlane((int)gtri0);
        nlist = (u32)__builtin_amdgcn_readlane((int)(gtri0 + nt), lv) - listbase;
        const u32 base = gtri0 - listbase;
        const u32 item = lo | ((rec & 0xFFu) << 6) | (((rec >> 8) & 0xFFu) << 14) | (((rec >> 16) & 1u) << 22);
        for (u32 t = 0; t < nt; ++t) list[base + t] = item | (t << 23);
        // (drained chunk by chunk: staging consecutive chunks of a group together, up to MC_LIST_CAP triangles, saves the
        // partly filled last step of each drain but makes the units coarser -- measured the same or slower)
        if (nlist) flush();
        cur = nxt;
        have = have_next;
    }
}
#endif  // MC_ONLY_INDEX_KERNELS

#ifndef MC_NO_INDEX_KERNELS
// =============================================================== indexed mesh (MC_FLAG_INDEXED)
// Poly_Data as the reference builds it (marching.cpp:599-654): vertex_list / tri_list with the vertices welded.  The
// reference welds through a std::set whose comparator calls two points equal when they are closer than 1e-6 on every
// axis (marching.h:32-55); the first point inserted keeps its coordinates and its index.  Points of different cells
// can only meet on a shared lattice edge or at a lattice corner, so the same mesh follows from a closed form
// (tests/weld_model.py states it in Python and checks it against a replay of the std::set):
//
//   key(vertex)   = the lattice CORNER it sits on, when the intersection point of its lattice edge -- computed in the +axis
//                   direction -- is closer than 1e-6 to an end of that edge; else the lattice EDGE itself
//   owner(key)    = the first cell of the sweep (z, y, x) that produces a vertex with this key, and in that cell the
//                   lowest-numbered such edge; the vertex keeps the position THAT cell computes for it
//   index(vertex) = keys owned by earlier cells + rank of the key among the owner's own edges (edge order 0..11)
//
// mc_resolve evaluates owner(key) from lattice indices and at most 9 samples of f; the kernels below are three sweeps
// over the records (one wave per group, lane = record, like mc_emit): mark the owned edges, number and write the
// vertices, write the triangles' indices.
__device__ __forceinline__ int mc_snap(float iso, float c0, float c1, float v0, float v1) {
    const float pu = mc_interp(iso, c0, c1, v0, v1);
    // marching.h:38-40 close_enough: float difference, compared as a double with 0.000001
    if (__builtin_fabs((double)(pu - c0)) < 0.000001) return 1;  // sits on the lower end
    if (__builtin_fabs((double)(pu - c1)) < 0.000001) return 2;  // ... on the upper end
    return 0;
}

// (internal flag, set by mc_runtime) seed mode: the records of cells outside the seed's component have lost their triangles
// (mc_seed_filter) -- the reference never visits those cells, so they insert no vertex (marching.cpp:310-331)
#define MC_FLAG_SEEDED 0x40000000u
__device__ __forceinline__ u32 mc_find_record_opt(const McParams& p, const u32* __restrict__ recs, const uint2* __restrict__ segcb, int qx, int qy,
                                                  int qz);

// a cell that can hold vertices: inside the sweep's ownership range (own_z_lo .. the slab's last layer) and -- with
// constraints -- not skipped (marching.cpp:476; a skipped cell's code byte reads 0, and a cell that contains a crossed edge
// never has code 0 or 255 otherwise).  A cell BELOW the swept layers (MC_FLAG_SEAM: the previous slab's) has no code byte
// here: whether a constraint skips it is evaluated at its 8 corners, as the sweep that owns it does (marching.cpp:255-280).
__device__ __forceinline__ bool mc_cell_ok(const McParams& p, const u8* __restrict__ codes, const u32* __restrict__ recs,
                                           const uint2* __restrict__ segcb, int qx, int qy, int qz) {
    if (qx < 0 || qy < 0 || qx >= p.n1 || qy >= p.n1 || qz < p.own_z_lo || qz >= p.z_begin + p.nz) return false;
    if (p.flags & MC_FLAG_SEEDED) {  // (whole-grid sweeps only: every cell has its record here)
        const u32 q = mc_find_record_opt(p, recs, segcb, qx, qy, qz);
        return q != 0xFFFFFFFFu && ((recs[q] >> 17) & 7u) != 0u;
    }
#ifdef MC_CONS
    if (qz < p.z_begin) {
        const float* __restrict__ ax = p.axs;
        const float* __restrict__ ay = p.axs + (p.n1 + 1);
        const float* __restrict__ az = p.axs + 2 * (p.n1 + 1);
        bool ok = true;
#pragma unroll
        for (int v = 0; v < 8; ++v) ok = ok && mc_ok(ax[qx + cx_bit(v)], ay[qy + cy_bit(v)], az[qz + cz_bit(v)]);
        return ok;
    }
    const u64 row = (u64)(qz - p.z_begin) * (u64)p.n1 + (u64)qy;
    const u32 c = qx < p.main_cells ? (u32)codes[row * p.pitch + (u64)qx] : (p.codes_tail[row] >> (8 * (qx - p.main_cells))) & 0xFFu;
    return c != 0u && c != 255u;
#else
    (void)codes;
    return true;
#endif
}

// edge number of the lattice edge along `ax` whose lower end sits at offsets (d0, d1) on the two other axes (in axis order)
__device__ __forceinline__ int mc_edge_of(int ax, int d0, int d1) {
    return ax == 0 ? 2 * d0 + 4 * d1 : ax == 1 ? 3 - 2 * d0 + 4 * d1 : 8 + (d1 ? 3 - d0 : d0);
}

// owner of the vertex on edge e of cell (ix, iy, iz): cell (qx, qy, qz) and its edge qe
// corner: the key is a lattice corner (several lattice edges' vertices are welded there), not the lattice edge
// edge_key_known: the caller knows (mc_vmark has recorded it) that the key is the lattice edge itself -- the usual case: the
// owner then follows from lattice indices alone, no sample of f, no interpolation
__device__ __forceinline__ void mc_resolve(const McParams& p, const u8* __restrict__ codes, const u32* __restrict__ recs,
                                           const uint2* __restrict__ segcb, int ix, int iy, int iz, int e, int& qx, int& qy,
                                           int& qz, int& qe, bool& corner, bool edge_key_known = false, const float* cell6 = nullptr) {
    const float* __restrict__ axis = p.axis;
    const int ax = edge_axis(e);
    int b[3] = {ix + (int)((MC_EDGE_OX >> e) & 1u), iy + (int)((MC_EDGE_OY >> e) & 1u), iz + (int)((MC_EDGE_OZ >> e) & 1u)};
    int sn = 0;
    float v0 = 0.0f, v1 = 0.0f;
    if (!edge_key_known) {
        // cell6 = the cell's six lattice coordinates {x0, x1, y0, y1, z0, z1}, fetched once by the caller for all of the cell's
        // edges: the same values the loads below return, without a memory round trip per edge (an edge's lower end has
        // offset 0 on the edge's own axis)
        // (selects with constant indices: a dynamically indexed private array would live in scratch)
        const float x0 = cell6 ? (((MC_EDGE_OX >> e) & 1u) ? cell6[1] : cell6[0]) : axis[b[0]],
                    y0 = cell6 ? (((MC_EDGE_OY >> e) & 1u) ? cell6[3] : cell6[2]) : axis[b[1]],
                    z0 = cell6 ? (((MC_EDGE_OZ >> e) & 1u) ? cell6[5] : cell6[4]) : axis[b[2]];
        const int ba = ax == 0 ? b[0] : ax == 1 ? b[1] : b[2];
        const float c0 = cell6 ? (ax == 0 ? cell6[0] : ax == 1 ? cell6[2] : cell6[4]) : axis[ba];
        const float c1 = cell6 ? (ax == 0 ? cell6[1] : ax == 1 ? cell6[3] : cell6[5]) : axis[ba + 1];
        v0 = mc_F(p, x0, y0, z0);
        v1 = mc_F(p, ax == 0 ? c1 : x0, ax == 1 ? c1 : y0, ax == 2 ? c1 : z0);
        sn = mc_snap(p.iso, c0, c1, v0, v1);
    }
    corner = sn != 0;
    if (sn == 0) {
        // lattice-edge key: the first of the (up to) four cells around the edge, in sweep order
        const int a0 = ax == 0 ? 1 : 0, a1 = ax == 2 ? 1 : 2;
#pragma unroll
        for (int d1 = 1; d1 >= 0; --d1)
#pragma unroll
            for (int d0 = 1; d0 >= 0; --d0) {
                int q[3] = {b[0], b[1], b[2]};
                q[a0] -= d0;
                q[a1] -= d1;
                if (mc_cell_ok(p, codes, recs, segcb, q[0], q[1], q[2])) {
                    qx = q[0];
                    qy = q[1];
                    qz = q[2];
                    qe = mc_edge_of(ax, d0, d1);
                    return;
                }
            }
        qx = ix; qy = iy; qz = iz; qe = e;  // not reached: the cell itself is one of the four
        return;
    }
    // lattice-corner key C
    int C[3] = {b[0], b[1], b[2]};
    if (sn == 2) C[ax] += 1;
    const float fc = sn == 1 ? v0 : v1;
    const float cx = axis[C[0]], cy = axis[C[1]], cz = axis[C[2]];
    // which of the six lattice edges at C carry an intersection that sits on C: bit 2a + da, da = 0 the edge towards
    // +a (inside the cells whose offset from C on that axis is 0), da = 1 the edge towards -a
    u32 Sm = 0;
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int da = 0; da < 2; ++da) {
            const int n = C[a] + (da == 0 ? 1 : -1);
            const int lo_lim = a == 2 ? p.own_z_lo : 0, hi_lim = a == 2 ? p.z_begin + p.nz : p.n1;
            if (n < lo_lim || n > hi_lim) continue;  // no such sample in this sweep's lattice
            const float cn = axis[n], cc = a == 0 ? cx : a == 1 ? cy : cz;
            const float fn = mc_F(p, a == 0 ? cn : cx, a == 1 ? cn : cy, a == 2 ? cn : cz);
            if ((fc > p.iso) == (fn > p.iso)) continue;
            const float pu = da == 0 ? mc_interp(p.iso, cc, cn, fc, fn) : mc_interp(p.iso, cn, cc, fn, fc);
            if (__builtin_fabs((double)(pu - cc)) < 0.000001) Sm |= 1u << (2 * a + da);
        }
#pragma unroll
    for (int dz = 1; dz >= 0; --dz)
#pragma unroll
        for (int dy = 1; dy >= 0; --dy)
#pragma unroll
            for (int dx = 1; dx >= 0; --dx) {
                // the cell C - (dx, dy, dz) holds, of C's six edges, the one towards -a where its offset is 1, towards +a where 0
                const u32 mx = (Sm >> (0 + dx)) & 1u, my = (Sm >> (2 + dy)) & 1u, mz = (Sm >> (4 + dz)) & 1u;
                if (!(mx | my | mz)) continue;
                if (!mc_cell_ok(p, codes, recs, segcb, C[0] - dx, C[1] - dy, C[2] - dz)) continue;
                // their edge numbers in that cell (the lower end of an edge along a has offset 0 on a); lowest wins
                int best = 12;
                if (mx) best = min(best, mc_edge_of(0, dy, dz));
                if (my) best = min(best, mc_edge_of(1, dx, dz));
                if (mz) best = min(best, mc_edge_of(2, dx, dy));
                qx = C[0] - dx;
                qy = C[1] - dy;
                qz = C[2] - dz;
                qe = best;
                return;
            }
    qx = ix; qy = iy; qz = iz; qe = e;  // not reached: the cell itself holds a snapping edge at C
}

// the group's segment tables for the sweeps over the records (same layout as in mc_emit)
struct McGroup {
    u32* segrec;  // iy | iz << 11 | chunk << 22
    u32* actoff;  // exclusive active-cell offsets, [64] = total
    u32* trioff;  // first triangle
    u32* rbase;   // first record
    u32 nrec;
    u32 tri0;
};
__device__ __forceinline__ bool mc_group_setup(const McParams& p, const uint2* __restrict__ segcb, const uint2* __restrict__ grpoff, u32 group,
                                               int lane, McGroup& g) {
    const u32 seg = group * 64u + (u32)lane;
    const uint2 g0 = grpoff[group], g1 = grpoff[group + 1u];
    if (g0.y == g1.y) return false;
    const uint2 cb = seg < p.nseg ? segcb[seg] : make_uint2(0u, 0u);
    const u32 ctri = cb.x & 0xFFFFu, cact = cb.x >> 16;
    const u32 itri = wave_inclusive_scan(ctri), iact = wave_inclusive_scan(cact);
    g.nrec = (u32)__builtin_amdgcn_readlane((int)iact, 63);
    g.tri0 = g0.x;
    const u32 sg = min(seg, p.nseg - 1u);
    const u32 rowidx = sg / (u32)p.nchunk;
    const u32 ch = sg - rowidx * (u32)p.nchunk;
    const u32 lz = rowidx / (u32)p.n1;
    const u32 iy = rowidx - lz * (u32)p.n1;
    g.segrec[lane] = iy | ((u32)(p.z_begin + (int)lz) << 11) | (ch << 22);
    g.actoff[lane] = iact - cact;
    g.trioff[lane] = g0.x + (itri - ctri);
    g.rbase[lane] = cb.y;
    if (lane == 63) g.actoff[64] = g.nrec;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    return true;
}
// record number r of the group: its index in recs, the record, its cell and first triangle
__device__ __forceinline__ bool mc_group_record(const McGroup& g, const u32* __restrict__ recs, u32 r, u32& ridx, u32& rec, int& ix, int& iy,
                                                int& iz, u32& gtri0) {
    u32 lo = 0, hi = 64;
#pragma unroll
    for (int it = 0; it < 6; ++it) {
        const u32 mid = (lo + hi) >> 1;
        if (g.actoff[mid] <= r) lo = mid; else hi = mid;
    }
    const bool valid = r < g.nrec;
    ridx = valid ? g.rbase[lo] + (r - g.actoff[lo]) : 0u;
    rec = valid ? recs[ridx] : 0u;
    const u32 sr = g.segrec[lo];
    ix = (int)(((sr >> 22) & 7u) * (u32)MC_SEG + (rec & 0xFFu));
    iy = (int)(sr & 2047u);
    iz = (int)((sr >> 11) & 2047u);
    gtri0 = g.trioff[lo] + (rec >> 20);
    return valid;
}

#define MC_WPB_I 4  // waves per workgroup of the three indexing kernels (mc_runtime launches them with the same number)
#define MC_GROUP_LDS                                                        \
    __shared__ u32 s_seg[MC_WPB_I][64], s_act[MC_WPB_I][66], s_tri[MC_WPB_I][64], s_rb[MC_WPB_I][64]; \
    const int lane = threadIdx.x & 63;                                      \
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); \
    const McParams p = *P;                                                  \
    const u32 ngroups = (p.nseg + 63u) / 64u;                               \
    const u32 group = blockIdx.x * (u32)MC_WPB_I + (u32)w;                  \
    if (group >= ngroups || p.overflow[0] != 0u) return;                    \
    McGroup g;                                                              \
    g.segrec = s_seg[w];                                                    \
    g.actoff = s_act[w];                                                    \
    g.trioff = s_tri[w];                                                    \
    g.rbase = s_rb[w];                                                      \
    if (!mc_group_setup(p, segcb, grpoff, group, lane, g)) return;

// The same for kernels whose chunks of 64 records are independent of each other: the waves of a workgroup SHARE the chunks
// of its four groups (an LDS counter per group; a wave takes its own group's chunks first, then its neighbours') -- groups
// differ in work by 10x and a workgroup keeps its resources until its last wave is done (see mc_emit_direct).  The kernel
// body runs once per chunk with `g` the chunk's group and `r0` its first record.
#define MC_GROUP_LDS_SHARED_BEGIN                                                                                       \
    __shared__ u32 s_seg[MC_WPB_I][64], s_act[MC_WPB_I][66], s_tri[MC_WPB_I][64], s_rb[MC_WPB_I][64], s_next[MC_WPB_I]; \
    const int lane = threadIdx.x & 63;                                                                                  \
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));                                             \
    const McParams p = *P;                                                                                              \
    const u32 ngroups = (p.nseg + 63u) / 64u;                                                                           \
    const u32 group = blockIdx.x * (u32)MC_WPB_I + (u32)w;                                                              \
    if (p.overflow[0] != 0u) return;                                                                                    \
    {                                                                                                                   \
        McGroup g0_;                                                                                                    \
        g0_.segrec = s_seg[w];                                                                                          \
        g0_.actoff = s_act[w];                                                                                          \
        g0_.trioff = s_tri[w];                                                                                          \
        g0_.rbase = s_rb[w];                                                                                            \
        if (group >= ngroups || !mc_group_setup(p, segcb, grpoff, group, lane, g0_)) {                                  \
            if (lane == 63) s_act[w][64] = 0u;                                                                          \
        }                                                                                                               \
        if (lane == 0) s_next[w] = 0u;                                                                                  \
    }                                                                                                                   \
    __syncthreads();                                                                                                    \
    for (int dw_ = 0; dw_ < MC_WPB_I; ++dw_) {                                                                          \
        const int wv = (w + dw_) & (MC_WPB_I - 1);                                                                      \
        McGroup g;                                                                                                      \
        g.segrec = s_seg[wv];                                                                                           \
        g.actoff = s_act[wv];                                                                                           \
        g.trioff = s_tri[wv];                                                                                           \
        g.rbase = s_rb[wv];                                                                                             \
        g.nrec = s_act[wv][64];                                                                                         \
        g.tri0 = 0u;                                                                                                    \
        for (;;) {                                                                                                      \
            u32 r0 = 0u;                                                                                                \
            if (lane == 0) r0 = __hip_atomic_fetch_add(&s_next[wv], 64u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); \
            r0 = (u32)__builtin_amdgcn_readfirstlane((int)r0);                                                          \
            if (r0 >= g.nrec) break;
#define MC_GROUP_LDS_SHARED_END \
        }                       \
    }

// I1: per record, the edges whose vertex this cell owns (first of the sweep to produce the key); per group, their number
extern "C" __global__ __launch_bounds__(64 * MC_WPB_I) void mc_vmark(const McParams* __restrict__ P, const u32* __restrict__ recs,
                                                                      const uint2* __restrict__ segcb, const uint2* __restrict__ grpoff,
                                                                      const u8* __restrict__ codes, u32* __restrict__ recown,
                                                                      u64* __restrict__ grpv) {
    __shared__ u32 s_cnt[MC_WPB_I];  // vertices owned by the records of each of the workgroup's groups
    if ((threadIdx.x & 63) == 0) s_cnt[threadIdx.x >> 6] = 0u;  // (in front of the macro's barrier)
    // Round 4: one lane per (record, crossed edge) PAIR, listed in LDS like mc_emit_direct's triangles.  With a lane per
    // record a cell's edges were resolved one after the other -- 4.2 on average, as many steps as the wave's busiest cell has
    // (6-7) -- and the kernel is bound by its vector instructions (87 M, SQ_ACTIVE_INST_VALU = 0.14 of its 0.16 ms per SIMD).
    __shared__ unsigned short s_pair[MC_WPB_I][64 * 12];  // record lane | edge << 6
    __shared__ uint2 s_cell[MC_WPB_I][64];                // per record of the chunk: {ix | iy << 16, iz}
    __shared__ float s_c6[MC_WPB_I][6][64];               // ... its six lattice coordinates {x0, x1, y0, y1, z0, z1}
    __shared__ u32 s_acc[MC_WPB_I][64];                   // ... owned edges | corner-keyed edges << 16, OR-ed in by the pair lanes
    const int wme = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    MC_GROUP_LDS_SHARED_BEGIN
    {
        u32 ridx, rec, gtri0;
        int ix, iy, iz;
        const bool valid = mc_group_record(g, recs, r0 + (u32)lane, ridx, rec, ix, iy, iz, gtri0);
        const bool live = valid && ((rec >> 17) & 7u);
        if (valid && !live) recown[ridx] = 0u;  // seed mode: a cell outside the seed's component (no triangles left)
        unsigned short* pair = s_pair[wme];
        u32 m = live ? crossed_edges((rec >> 8) & 0xFFu) : 0u;
        {
            // the cell's six lattice coordinates, once and together, for the snap test of every crossed edge
            const int jx = live ? ix : 0, jy = live ? iy : 0, jz = live ? iz : 0;
            float c6[6] = {p.axis[jx], p.axis[jx + 1], p.axis[jy], p.axis[jy + 1], p.axis[jz], p.axis[jz + 1]};
            asm volatile("" : "+v"(c6[0]), "+v"(c6[1]), "+v"(c6[2]), "+v"(c6[3]), "+v"(c6[4]), "+v"(c6[5]));
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // (the pairs of the chunk before have been read)
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int i = 0; i < 6; ++i) s_c6[wme][i][lane] = c6[i];
        }
        s_cell[wme][lane] = make_uint2((u32)ix | ((u32)iy << 16), (u32)iz);
        s_acc[wme][lane] = 0u;
        const u32 cnt = (u32)__builtin_popcount(m), incl = wave_inclusive_scan(cnt);
        const u32 total = (u32)__builtin_amdgcn_readlane((int)incl, 63);
        for (u32 j = incl - cnt; m; ++j) {
            const u32 e = (u32)__builtin_ctz(m);
            m &= m - 1u;
            pair[j] = (unsigned short)((u32)lane | (e << 6));
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (u32 k0 = 0; k0 < total; k0 += 64u) {
            const u32 k = k0 + (u32)lane;
            if (k < total) {
                const u32 it = pair[k];
                const u32 rl = it & 63u;
                const int e = (int)(it >> 6);
                const uint2 cc = s_cell[wme][rl];
                const int cx = (int)(cc.x & 0xFFFFu), cy = (int)(cc.x >> 16), cz = (int)cc.y;
                const float c6[6] = {s_c6[wme][0][rl], s_c6[wme][1][rl], s_c6[wme][2][rl], s_c6[wme][3][rl], s_c6[wme][4][rl], s_c6[wme][5][rl]};
                int qx, qy, qz, qe;
                bool corner;
                mc_resolve(p, codes, recs, segcb, cx, cy, cz, e, qx, qy, qz, qe, corner, false, c6);
                const u32 bits = ((qx == cx && qy == cy && qz == cz && qe == e) ? 1u << e : 0u) | (corner ? 0x10000u << e : 0u);
                if (bits) __hip_atomic_fetch_or(&s_acc[wme][rl], bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        u32 ownm = 0;
        if (live) {
            const u32 acc = s_acc[wme][lane];
            ownm = acc & 0xFFFu;
            recown[ridx] = acc;  // bits 0..11: owned edges; 16..27: the crossed edges whose key is a lattice corner
        }
        const u32 sum = wave_inclusive_scan((u32)__builtin_popcount(ownm));
        if (lane == 63 && sum) __hip_atomic_fetch_add(&s_cnt[wv], sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    MC_GROUP_LDS_SHARED_END
    __syncthreads();
    if (lane == 0 && group < ngroups) grpv[group] = (u64)s_cnt[w];
}

// I2: number the vertices (group offset from the scan + prefix over the group's records) and write the owned ones
extern "C" __global__ __launch_bounds__(64 * MC_WPB_I) void mc_vwrite(const McParams* __restrict__ P, const u32* __restrict__ recs,
                                                                       const uint2* __restrict__ segcb, const uint2* __restrict__ grpoff,
                                                                       const u32* __restrict__ recown, const uint2* __restrict__ grpvoff,
                                                                       u32* __restrict__ recvb, float* __restrict__ vlist, u64 cap_verts) {
    MC_GROUP_LDS
    const float* __restrict__ axis = p.axis;
    u32 carry = grpvoff[group].x;
    for (u32 r0 = 0; r0 < g.nrec; r0 += 64u) {
        u32 ridx, rec, gtri0;
        int ix, iy, iz;
        const bool valid = mc_group_record(g, recs, r0 + (u32)lane, ridx, rec, ix, iy, iz, gtri0);
        const u32 ownm = valid ? recown[ridx] & 0xFFFu : 0u;
        const u32 c = (u32)__builtin_popcount(ownm);
        const u32 incl = wave_inclusive_scan(c);
        u32 vb = carry + incl - c;
        carry += (u32)__builtin_amdgcn_readlane((int)incl, 63);
        if (valid) {
            recvb[ridx] = vb;
            u32 m = ownm;
            while (m) {
                const int e = __builtin_ctz(m);
                m &= m - 1u;
                // the intersection point as THIS cell's edge direction computes it (marching.cpp:557-583)
                const int bx = ix + (int)((MC_EDGE_OX >> e) & 1u), by = iy + (int)((MC_EDGE_OY >> e) & 1u), bz = iz + (int)((MC_EDGE_OZ >> e) & 1u);
                const int ax = edge_axis(e);
                const float x0 = axis[bx], y0 = axis[by], z0 = axis[bz];
                const float c0 = ax == 0 ? x0 : ax == 1 ? y0 : z0;
                const float c1 = axis[(ax == 0 ? bx : ax == 1 ? by : bz) + 1];
                const float v0 = mc_F(p, x0, y0, z0);
                const float v1 = mc_F(p, ax == 0 ? c1 : x0, ax == 1 ? c1 : y0, ax == 2 ? c1 : z0);
                const float pa = ((MC_EDGE_DOWN >> e) & 1u) ? mc_interp(p.iso, c1, c0, v1, v0) : mc_interp(p.iso, c0, c1, v0, v1);
                if ((u64)vb < cap_verts) {
                    float* o = vlist + 3ull * vb;
                    o[0] = ax == 0 ? pa : x0;
                    o[1] = ax == 1 ? pa : y0;
                    o[2] = ax == 2 ? pa : z0;
                }
                ++vb;
            }
        }
    }
}

// the record of cell (qx, qy, qz), which has one: its segment's records are contiguous and ascending in x.  Up to four
// records are fetched at once and the one wanted is picked (a segment of a curved surface holds 1-3), longer ones are searched
__device__ __forceinline__ u32 mc_find_record(const McParams& p, const u32* __restrict__ recs, const uint2* __restrict__ segcb, int qx, int qy,
                                              int qz) {
    const u32 seg = (u32)(((qz - p.z_begin) * p.n1 + qy) * p.nchunk + (qx >> 8));
    const uint2 cb = segcb[seg];
    u32 lo = cb.y, n = cb.x >> 16;
    const u32 want = (u32)(qx & 255);
    if (n <= 4u) {
        u32 rb = recs[lo + 1u], rc_ = recs[lo + 2u], rd = recs[lo + 3u];  // (the buffer has slack behind its last record)
        asm volatile("" : "+v"(rb), "+v"(rc_), "+v"(rd));  // the three loads travel together (see mc_vn_cell: MC_TOGETHER)
        return (n > 3u && (rd & 0xFFu) <= want) ? lo + 3u : (n > 2u && (rc_ & 0xFFu) <= want) ? lo + 2u : (n > 1u && (rb & 0xFFu) <= want) ? lo + 1u : lo;
    }
    while (n > 1u) {  // lower bound
        const u32 half = n >> 1;
        if ((recs[lo + half - 1u] & 0xFFu) < want) {
            lo += half;
            n -= half;
        } else {
            n = half;
        }
    }
    return lo;
}

// I3: tri_list -- for every triangle corner the index of its vertex (marching.cpp:618-624, :646-654)
extern "C" __global__ __launch_bounds__(64 * MC_WPB_I) void mc_vindex(const McParams* __restrict__ P, const u32* __restrict__ recs,
                                                                       const uint2* __restrict__ segcb, const uint2* __restrict__ grpoff,
                                                                       const u8* __restrict__ codes, const u32* __restrict__ recown,
                                                                       const u32* __restrict__ recvb, u32* __restrict__ tlist, u64 cap_tris,
                                                                       u32* __restrict__ segtri) {
    __shared__ u32 s_eidx[MC_WPB_I][64 * 13];  // per lane: the vertex index of each of its 12 edges (stride 13: no bank conflicts)
    // Round 4: one lane per (record, crossed edge) PAIR.  With a lane per record the edges of a cell were walked one after the
    // other -- 4.2 on average, as many steps as the wave's busiest cell has (6-7), every step a chain of three dependent
    // look-ups (the owner's segment, its record, its vertex numbers).  The pairs are listed in LDS (like mc_emit_direct's
    // triangles) and taken 64 at a time; a pair's cell comes from its record's lane by ds_bpermute (no LDS of its own: LDS is
    // what limits the waves per CU here, and the kernel needs them).
    __shared__ unsigned short s_pair[MC_WPB_I][64 * 12];  // record lane | edge << 6
    const int wme = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    u32* eidx = s_eidx[wme] + 13 * (threadIdx.x & 63);
    MC_GROUP_LDS_SHARED_BEGIN
    // lane = segment: its first triangle, for mc_vnormal (a record's first triangle = its segment's + the record's prefix);
    // written with the group's first chunk
    if (r0 == 0u && (blockIdx.x * (u32)MC_WPB_I + (u32)wv) * 64u + (u32)lane < p.nseg)
        segtri[(blockIdx.x * (u32)MC_WPB_I + (u32)wv) * 64u + (u32)lane] = g.trioff[lane];
    {
        u32 ridx, rec, gtri0;
        int ix, iy, iz;
        const bool valid = mc_group_record(g, recs, r0 + (u32)lane, ridx, rec, ix, iy, iz, gtri0);
        const bool live = valid && ((rec >> 17) & 7u);
        const u32 code = (rec >> 8) & 0xFFu;
        u32 myrow = live ? recown[ridx] : 0u, myvb = live ? recvb[ridx] : 0u;
        asm volatile("" : "+v"(myrow), "+v"(myvb));  // (one level)
        unsigned short* pair = s_pair[wme];
        const u32 cxy = (u32)ix | ((u32)iy << 16);
        u32 m = live ? crossed_edges(code) : 0u;
        const u32 cnt = (u32)__builtin_popcount(m), incl = wave_inclusive_scan(cnt);
        const u32 total = (u32)__builtin_amdgcn_readlane((int)incl, 63);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // (the pairs of the chunk before have been read)
        __builtin_amdgcn_wave_barrier();
        for (u32 j = incl - cnt; m; ++j) {
            const u32 e = (u32)__builtin_ctz(m);
            m &= m - 1u;
            pair[j] = (unsigned short)((u32)lane | (e << 6));
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (u32 k0 = 0; k0 < total; k0 += 64u) {
            const u32 k = k0 + (u32)lane;
            const bool act = k < total;
            const u32 it = act ? pair[k] : 0u;
            const int rl = (int)(it & 63u);
            const int e = (int)(it >> 6);
            // (the shuffles outside any lane-dependent control flow: a ds_bpermute reads 0 from a lane that is switched off)
            const u32 pxy = (u32)__shfl((int)cxy, rl, 64), pz = (u32)__shfl(iz, rl, 64);
            const u32 prow = (u32)__shfl((int)myrow, rl, 64), pvb = (u32)__shfl((int)myvb, rl, 64);
            if (act) {
                const int cx = (int)(pxy & 0xFFFFu), cy = (int)(pxy >> 16), cz = (int)pz;
                int qx, qy, qz, qe;
                bool corner;
                mc_resolve(p, codes, recs, segcb, cx, cy, cz, e, qx, qy, qz, qe, corner, ((prow >> (16 + e)) & 1u) == 0u);
                u32 o = prow & 0xFFFu, vb = pvb, idx;
                if (qz < p.z_begin) {
                    // MC_FLAG_SEAM: the owner is a cell of the layer below the swept range (the lower plane of the ghost layer):
                    // it has no record here and its vertex no index; only ghost cells meet this, and their triangles are
                    // never handed out
                    idx = 0xFFFFFFFFu;
                } else {
                    if (!(qx == cx && qy == cy && qz == cz)) {
                        const u32 q = mc_find_record(p, recs, segcb, qx, qy, qz);
                        u32 oq = recown[q], vq = recvb[q];
                        asm volatile("" : "+v"(oq), "+v"(vq));  // (one level)
                        o = oq & 0xFFFu;
                        vb = vq;
                    }
                    idx = vb + (u32)__builtin_popcount(o & ((1u << qe) - 1u));
                }
                s_eidx[wme][13u * (u32)rl + (u32)e] = idx;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (live) {
            const u32 row = ((rec >> 16) & 1u) ? 255u - code : code;  // marching.cpp:542-547
            const u64 tr = c_tri_row[row];
            const u32 nt = (rec >> 17) & 7u;
            for (u32 t = 0; t < nt; ++t) {
                const u64 gt = (u64)gtri0 + t;
                if (gt < cap_tris) {
                    u32* o = tlist + 3ull * gt;
                    o[0] = eidx[(tr >> (12u * t)) & 0xFull];
                    o[1] = eidx[(tr >> (12u * t + 4u)) & 0xFull];
                    o[2] = eidx[(tr >> (12u * t + 8u)) & 0xFull];
                }
            }
        }
    }
    MC_GROUP_LDS_SHARED_END
}

// the record of cell (qx, qy, qz), or ~0 when that cell is outside the slab or has none (no surface / skipped)
__device__ __forceinline__ u32 mc_find_record_opt(const McParams& p, const u32* __restrict__ recs, const uint2* __restrict__ segcb, int qx, int qy,
                                                  int qz) {
    if (qx < 0 || qy < 0 || qx >= p.n1 || qy >= p.n1 || qz < p.z_begin || qz >= p.z_begin + p.nz) return 0xFFFFFFFFu;
    const u32 seg = (u32)(((qz - p.z_begin) * p.n1 + qy) * p.nchunk + (qx >> 8));
    const uint2 cb = segcb[seg];
    u32 lo = cb.y, n = cb.x >> 16;
    if (n == 0u) return 0xFFFFFFFFu;
    const u32 want = (u32)(qx & 255);
    while (n > 1u) {  // lower bound
        const u32 half = n >> 1;
        if ((recs[lo + half - 1u] & 0xFFu) < want) {
            lo += half;
            n -= half;
        } else {
            n = half;
        }
    }
    return (recs[lo] & 0xFFu) == want ? lo : 0xFFFFFFFFu;
}

// I4: CalculateNormal (Source/normal.h:3-41) on the welded mesh, as a GATHER: every vertex visits the cells that can hold a
// triangle on it -- the 4 cells around its lattice edge, or, when the vertex is welded to a lattice corner, the 12 cells
// around the edge's two ends -- in sweep order, their triangles in table order, and adds cross(B-A, C-A) for every corner
// that IS the vertex: exactly the additions the reference performs on vNormal[i], in the reference's order (its loop runs
// over the triangles in emission order), so the sums -- and the normalised result, glm's v * (1 / sqrt(dot(v, v))) -- are
// the reference's bits.  (A scatter with float atomics was 3.1 ms of the 3.7 ms the indexed mesh took at 1025^3: 89 M
// single-float atomics to scattered addresses run at a seventeenth of the rate contiguous ones do, and their sums depended
// on arrival order.)
//
// The kernel is a chain of dependent loads per (vertex, cell): segment -> its records (binary search) -> the record's
// triangles -> their indices -> their positions.  Round 2 walked it one lane per RECORD -- owned edges one after the other,
// their cells one after the other: ~100 dependent round trips per chunk of records, 1.16 of the 1.70 ms the indexed mesh took
// at 1025^3.  Now the vertices of a chunk are listed, and FOUR lanes serve one vertex, one per candidate cell: the four
// chains run side by side and leave (normal, hits) per triangle in LDS; the vertex's first lane then adds them up in the
// reference's order.  Vertices welded to a lattice corner (12 cells; rare unless the surface passes through lattice points)
// take a second pass, one lane per vertex.
#define MC_VN_CAP 256  // vertices of one kind (edge key / corner key) listed at a time; 16 records own at most 192
// what cell (qx, qy, qz) adds to vertex v: its triangles in table order; out[t] = {cross(B-A, C-A), corners that are v}
// what one cell contributes to a vertex's sum: the face normals of the first two of its triangles that touch the vertex
// (.w = how many corners of the triangle are the vertex, 0 = none), and -- rarely -- more of them (bit t of `more`; hp: two bits
// per triangle, the corner count; t0: the cell's first triangle)
struct McVnCell {
    float4 a, b;
    u32 more, hp, t0;
};
// The triangles of cell (qx, qy, qz) that have vertex v at a corner -> out (in triangle order: a, b, then `more`).
// match_edge >= 0: v's key is a lattice EDGE, which is edge `match_edge` of this cell -- the cell's table row says which of its
// triangles touch it, nothing is read from tri_list; match_edge < 0 (v is welded to a lattice corner, several edges lead to
// it): the triangles' corner indices are compared with v.  The face normals come from mc_tnormal (mc_scan.hip).
__device__ __forceinline__ void mc_vn_cell(const McParams& p, const u32* __restrict__ recs, const uint2* __restrict__ segcb,
                                           const u32* __restrict__ segtri, const u32* __restrict__ tlist, const float4* __restrict__ tnrm,
                                           const u8* s_edgetri, u64 cap_tris, int qx, int qy, int qz, u32 v, int match_edge,
                                           McVnCell& out) {
    out.a = out.b = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    out.more = 0u;
    out.hp = 0u;
    out.t0 = 0u;
    // the cell's record: segment -> its records (ascending in x).  Four dependent loads fewer than a search that ends in the
    // record index: the segment's first triangle comes with its counts, up to four records are fetched at once and the one
    // wanted is picked (a segment of the sphere holds 1-3), longer segments are searched
    // The loads of one level must TRAVEL TOGETHER: left to itself the compiler sinks each load to its first use behind the
    // branches in between (it saves a load on the paths that leave early) and the chain becomes one memory round trip per
    // LOAD instead of one per LEVEL -- the four records of a segment were fetched one after the other, each behind a wait,
    // the five face normals likewise (round 4, seen in the ISA).  MC_TOGETHER pins the values of a level at one point: the
    // loads are issued back to back, one wait covers them.
#define MC_TOGETHER2(a, b) asm volatile("" : "+v"(a), "+v"(b))
#define MC_TOGETHER4(a, b, c, d) asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d))
    if (qx < 0 || qy < 0 || qx >= p.n1 || qy >= p.n1 || qz < p.z_begin || qz >= p.z_begin + p.nz) return;
    const u32 seg = (u32)(((qz - p.z_begin) * p.n1 + qy) * p.nchunk + (qx >> 8));
    uint2 cb = segcb[seg];
    u32 st = segtri[seg];
    MC_TOGETHER2(cb.y, st);
    const u32 n = cb.x >> 16, want = (u32)(qx & 255);
    if (n == 0u) return;
    u32 rec;
    if (n <= 4u) {
        u32 ra = recs[cb.y], rb = recs[cb.y + 1u], rc_ = recs[cb.y + 2u], rd = recs[cb.y + 3u];  // (the buffer has slack behind its last record)
        MC_TOGETHER4(ra, rb, rc_, rd);
        rec = (ra & 0xFFu) == want ? ra : (n > 1u && (rb & 0xFFu) == want) ? rb : (n > 2u && (rc_ & 0xFFu) == want) ? rc_ : (n > 3u && (rd & 0xFFu) == want) ? rd : 0xFFFFFFFFu;
        if (rec == 0xFFFFFFFFu) return;
    } else {
        u32 lo = cb.y, m = n;
        while (m > 1u) {  // lower bound
            const u32 half = m >> 1;
            if ((recs[lo + half - 1u] & 0xFFu) < want) {
                lo += half;
                m -= half;
            } else {
                m = half;
            }
        }
        rec = recs[lo];
        if ((rec & 0xFFu) != want) return;
    }
    const u32 nt = (rec >> 17) & 7u;
    const u32 t0 = st + (rec >> 20);
    u32 hm = 0u;
    if (match_edge >= 0) {
        // which of the cell's triangles touch that edge: one byte per (table row, edge), made once per workgroup (mc_vnormal).
        // (Comparing the row's fifteen nibbles with the edge here, per cell and round, was a sixth of the kernel's vector
        // instructions -- and those are what bounds it, round 4.)  A triangle touches an edge once.
        const u32 code = (rec >> 8) & 0xFFu;
        hm = s_edgetri[(((rec >> 16) & 1u) ? 255u - code : code) * 12u + (u32)match_edge];  // marching.cpp:542-547
        const u64 room = cap_tris > (u64)t0 ? cap_tris - (u64)t0 : 0ull;  // triangles beyond the buffer's capacity do not exist
        if (room < 5ull) hm &= (1u << (u32)room) - 1u;
        hm &= (1u << nt) - 1u;  // (seed mode: a record outside the seed's component keeps its code and has no triangles)
        out.hp = (hm & 1u) | ((hm & 2u) << 1) | ((hm & 4u) << 2) | ((hm & 8u) << 3) | ((hm & 16u) << 4);
    } else {
        u32 hits[5];
        // (every triangle's three indices, wanted or not -- triangle 0 of the list stands in for the ones that are not, its
        // line is in every lane's cache --, so that the fifteen loads are one level)
        u32 ia[5], ib[5], ic[5];
#pragma unroll
        for (int t = 0; t < 5; ++t) {
            const bool wanted = (u32)t < nt && (u64)(t0 + (u32)t) < cap_tris;
            const u32* tri = tlist + 3ull * (wanted ? t0 + (u32)t : 0u);
            ia[t] = tri[0];
            ib[t] = tri[1];
            ic[t] = tri[2];
        }
        MC_TOGETHER4(ia[0], ib[0], ic[0], ia[1]);
        MC_TOGETHER4(ib[1], ic[1], ia[2], ib[2]);
        MC_TOGETHER4(ic[2], ia[3], ib[3], ic[3]);
        MC_TOGETHER4(ia[4], ib[4], ic[4], ia[0]);
#pragma unroll
        for (int t = 0; t < 5; ++t) {
            const bool wanted = (u32)t < nt && (u64)(t0 + (u32)t) < cap_tris;
            hits[t] = wanted ? (ia[t] == v ? 1u : 0u) + (ib[t] == v ? 1u : 0u) + (ic[t] == v ? 1u : 0u) : 0u;
        }
#pragma unroll
        for (int t = 0; t < 5; ++t) {
            hm |= hits[t] ? 1u << t : 0u;
            out.hp |= hits[t] << (2 * t);
        }
    }
    // The face normals of the FIRST TWO triangles that touch the vertex, one level; a cell's triangles touch one of its edges
    // once or twice, a third / fourth / fifth is left to the caller (out.more, in triangle order behind these two).  The
    // kernel pays for every gather and for every slot of the sum's chain: five loads and five slots per cell, most of them
    // empty, were a quarter of its time.
    const u32 rest = hm & (hm - 1u);
    const u32 ta = hm ? (u32)__builtin_ctz(hm) : 0u, tb = rest ? (u32)__builtin_ctz(rest) : 0u;
    float4 fa = tnrm[hm ? t0 + ta : 0u], fb = tnrm[rest ? t0 + tb : 0u];
    MC_TOGETHER4(fa.x, fa.w, fb.x, fb.w);
    if (hm && fa.w != 0.0f) out.a = make_float4(fa.x, fa.y, fa.z, __builtin_bit_cast(float, (out.hp >> (2u * ta)) & 3u));
    if (rest && fb.w != 0.0f) out.b = make_float4(fb.x, fb.y, fb.z, __builtin_bit_cast(float, (out.hp >> (2u * tb)) & 3u));
    out.more = rest & (rest - 1u);
    out.t0 = t0;
#undef MC_TOGETHER2
#undef MC_TOGETHER4
}
// vNormal[i] = normal + vNormal[i], once per corner that is the vertex (normal.h:22-31): almost always 0 or 1 times -- the
// second and third block are skipped by the whole wave unless a degenerate triangle has the vertex at two corners
__device__ __forceinline__ void mc_vn_add(const float4 c, float& sx, float& sy, float& sz) {
    const u32 hits = __builtin_bit_cast(u32, c.w);
    if (hits >= 1u) {
        sx = c.x + sx;
        sy = c.y + sy;
        sz = c.z + sz;
    }
    if (hits >= 2u) {
        sx = c.x + sx;
        sy = c.y + sy;
        sz = c.z + sz;
    }
    if (hits >= 3u) {
        sx = c.x + sx;
        sy = c.y + sy;
        sz = c.z + sz;
    }
}
// ... a whole cell's contribution, in triangle order
__device__ __forceinline__ void mc_vn_add_cell(const McVnCell& c, const float4* __restrict__ tnrm, float& sx, float& sy, float& sz) {
    mc_vn_add(c.a, sx, sy, sz);
    mc_vn_add(c.b, sx, sy, sz);
    u32 more = c.more;
    while (more) {  // (a third, fourth, fifth triangle of one cell on one vertex: rare)
        const u32 t = (u32)__builtin_ctz(more);
        more &= more - 1u;
        const float4 f = tnrm[c.t0 + t];
        if (f.w != 0.0f) mc_vn_add(make_float4(f.x, f.y, f.z, __builtin_bit_cast(float, (c.hp >> (2u * t)) & 3u)), sx, sy, sz);
    }
}
// The same for the lanes where `on` holds, written WITHOUT lane-dependent control flow: the sum of an edge-keyed vertex
// travels through the four lanes of its cells (mc_vnormal), one lane adding per step, and as branches every step cost the wave
// the whole of mc_vn_add_cell -- four times ~95 instructions per round of 16 vertices, half of the kernel, which is bound by
// its vector instructions (SQ_ACTIVE_INST_VALU = 0.37 of its 0.49 ms per SIMD, round 4).  A lane that does not add adds +0.0f:
// x + 0.0f is x for every x the sum can hold (it starts at +0.0f and round-to-nearest never makes it -0.0f; NaN and inf
// stay), so the bits are the reference's.  A second / third hit of one triangle and a third triangle of one cell are rare:
// those parts run only when some lane of the wave needs them.
__device__ __forceinline__ void mc_vn_add_masked(bool on, const float4 c, float& sx, float& sy, float& sz) {
    const u32 hits = on ? __builtin_bit_cast(u32, c.w) : 0u;
    {
        const bool h = hits >= 1u;
        sx = (h ? c.x : 0.0f) + sx;
        sy = (h ? c.y : 0.0f) + sy;
        sz = (h ? c.z : 0.0f) + sz;
    }
    if (__ballot(hits >= 2u)) {  // (wave-uniform)
        const bool h2 = hits >= 2u, h3 = hits >= 3u;
        sx = (h2 ? c.x : 0.0f) + sx;
        sy = (h2 ? c.y : 0.0f) + sy;
        sz = (h2 ? c.z : 0.0f) + sz;
        sx = (h3 ? c.x : 0.0f) + sx;
        sy = (h3 ? c.y : 0.0f) + sy;
        sz = (h3 ? c.z : 0.0f) + sz;
    }
}
__device__ __forceinline__ void mc_vn_add_cell_masked(bool on, const McVnCell& c, const float4* __restrict__ tnrm, float& sx, float& sy, float& sz) {
    mc_vn_add_masked(on, c.a, sx, sy, sz);
    mc_vn_add_masked(on, c.b, sx, sy, sz);
    if (__ballot(on && c.more != 0u)) {  // (wave-uniform; a third, fourth, fifth triangle of one cell on one vertex: rare)
        u32 more = on ? c.more : 0u;
        while (more) {
            const u32 t = (u32)__builtin_ctz(more);
            more &= more - 1u;
            const float4 f = tnrm[c.t0 + t];
            if (f.w != 0.0f) mc_vn_add(make_float4(f.x, f.y, f.z, __builtin_bit_cast(float, (c.hp >> (2u * t)) & 3u)), sx, sy, sz);
        }
    }
}
// The four cells around a lattice edge, ALL ON ONE LANE (round 4): lane = vertex, the loads of a level issued together for the
// four cells (eight, sixteen, eight loads: the chain is three round trips per 64 vertices), no divergent branch but the rare
// long segment.  The round-3 form put the four cells on four lanes -- 16 vertices per round, the sum handed from lane to lane
// -- which is the same latency per vertex and four times the instructions, and instructions are what bounds the kernel.
// qx / qy / qz: the cells in sweep order; me: the vertex's lattice edge as an edge of each cell (mc_resolve's rule).
__device__ __forceinline__ void mc_vn_cells4(const McParams& p, const u32* __restrict__ recs, const uint2* __restrict__ segcb,
                                             const u32* __restrict__ segtri, const float4* __restrict__ tnrm, const u8* s_edgetri, u64 cap_tris,
                                             const int (&qx)[4], const int (&qy)[4], const int (&qz)[4], const int (&me)[4], bool act,
                                             McVnCell (&out)[4]) {
    bool in[4];
    u32 seg[4], st[4], want[4];
    uint2 cb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        in[j] = act && qx[j] >= 0 && qy[j] >= 0 && qx[j] < p.n1 && qy[j] < p.n1 && qz[j] >= p.z_begin && qz[j] < p.z_begin + p.nz;
        seg[j] = in[j] ? (u32)(((qz[j] - p.z_begin) * p.n1 + qy[j]) * p.nchunk + (qx[j] >> 8)) : 0u;  // (segment 0 stands in: a valid address)
        want[j] = (u32)(qx[j] & 255);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        cb[j] = segcb[seg[j]];
        st[j] = segtri[seg[j]];
    }
    asm volatile("" : "+v"(cb[0].x), "+v"(cb[0].y), "+v"(cb[1].x), "+v"(cb[1].y), "+v"(cb[2].x), "+v"(cb[2].y), "+v"(cb[3].x), "+v"(cb[3].y));
    asm volatile("" : "+v"(st[0]), "+v"(st[1]), "+v"(st[2]), "+v"(st[3]));
    u32 n[4], r[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        n[j] = in[j] ? cb[j].x >> 16 : 0u;
        const u32 base = n[j] ? cb[j].y : 0u;  // (record 0 stands in; the buffer has slack behind its last record)
#pragma unroll
        for (int i = 0; i < 4; ++i) r[j][i] = recs[base + (u32)i];
    }
    asm volatile("" : "+v"(r[0][0]), "+v"(r[0][1]), "+v"(r[0][2]), "+v"(r[0][3]), "+v"(r[1][0]), "+v"(r[1][1]), "+v"(r[1][2]), "+v"(r[1][3]));
    asm volatile("" : "+v"(r[2][0]), "+v"(r[2][1]), "+v"(r[2][2]), "+v"(r[2][3]), "+v"(r[3][0]), "+v"(r[3][1]), "+v"(r[3][2]), "+v"(r[3][3]));
    u32 rec[4];
    bool found[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const u32 w = want[j];
        rec[j] = (r[j][0] & 0xFFu) == w ? r[j][0] : (n[j] > 1u && (r[j][1] & 0xFFu) == w) ? r[j][1] : (n[j] > 2u && (r[j][2] & 0xFFu) == w) ? r[j][2] : r[j][3];
        found[j] = n[j] >= 1u && n[j] <= 4u && ((r[j][0] & 0xFFu) == w || (n[j] > 1u && (r[j][1] & 0xFFu) == w) || (n[j] > 2u && (r[j][2] & 0xFFu) == w) ||
                                                (n[j] > 3u && (r[j][3] & 0xFFu) == w));
    }
    if (__ballot(n[0] > 4u || n[1] > 4u || n[2] > 4u || n[3] > 4u)) {  // (wave-uniform) a segment of more than four records: searched
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (n[j] > 4u) {
                u32 lo = cb[j].y, m = n[j];
                while (m > 1u) {  // lower bound
                    const u32 half = m >> 1;
                    if ((recs[lo + half - 1u] & 0xFFu) < want[j]) {
                        lo += half;
                        m -= half;
                    } else {
                        m = half;
                    }
                }
                rec[j] = recs[lo];
                found[j] = (rec[j] & 0xFFu) == want[j];
            }
    }
    u32 hm[4], t0[4], ta[4], tb[4], rest[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const u32 nt = (rec[j] >> 17) & 7u, code = (rec[j] >> 8) & 0xFFu;
        t0[j] = st[j] + (rec[j] >> 20);
        u32 h = s_edgetri[(((rec[j] >> 16) & 1u) ? 255u - code : code) * 12u + (u32)me[j]];  // marching.cpp:542-547
        const u64 room = cap_tris > (u64)t0[j] ? cap_tris - (u64)t0[j] : 0ull;  // triangles beyond the buffer's capacity do not exist
        if (room < 5ull) h &= (1u << (u32)room) - 1u;
        h &= (1u << nt) - 1u;  // (seed mode: a record outside the seed's component keeps its code and has no triangles)
        hm[j] = found[j] ? h : 0u;
        rest[j] = hm[j] & (hm[j] - 1u);
        ta[j] = hm[j] ? (u32)__builtin_ctz(hm[j]) : 0u;
        tb[j] = rest[j] ? (u32)__builtin_ctz(rest[j]) : 0u;
    }
    float4 fa[4], fb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        fa[j] = tnrm[hm[j] ? t0[j] + ta[j] : 0u];
        fb[j] = tnrm[rest[j] ? t0[j] + tb[j] : 0u];
    }
    asm volatile("" : "+v"(fa[0].x), "+v"(fa[0].w), "+v"(fb[0].x), "+v"(fb[0].w), "+v"(fa[1].x), "+v"(fa[1].w), "+v"(fb[1].x), "+v"(fb[1].w));
    asm volatile("" : "+v"(fa[2].x), "+v"(fa[2].w), "+v"(fb[2].x), "+v"(fb[2].w), "+v"(fa[3].x), "+v"(fa[3].w), "+v"(fb[3].x), "+v"(fb[3].w));
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        // (a triangle touches an edge once: hit count 1 where the triangle exists and is not degenerate)
        out[j].a = make_float4(fa[j].x, fa[j].y, fa[j].z, __builtin_bit_cast(float, (hm[j] && fa[j].w != 0.0f) ? 1u : 0u));
        out[j].b = make_float4(fb[j].x, fb[j].y, fb[j].z, __builtin_bit_cast(float, (rest[j] && fb[j].w != 0.0f) ? 1u : 0u));
        out[j].more = rest[j] & (rest[j] - 1u);
        out[j].hp = (hm[j] & 1u) | ((hm[j] & 2u) << 1) | ((hm[j] & 4u) << 2) | ((hm[j] & 8u) << 3) | ((hm[j] & 16u) << 4);
        out[j].t0 = t0[j];
    }
}
extern "C" __global__ __launch_bounds__(64 * MC_WPB_I) void mc_vnormal(const McParams* __restrict__ P, const u32* __restrict__ recs,
                                                                        const uint2* __restrict__ segcb, const uint2* __restrict__ grpoff,
                                                                        const u32* __restrict__ recown, const u32* __restrict__ recvb,
                                                                        const u32* __restrict__ segtri, const u32* __restrict__ tlist,
                                                                        const float4* __restrict__ tnrm, float* __restrict__ vnrm, u64 nverts,
                                                                        u64 cap_tris) {
    // (round 4's counters: 95 M vector instructions, half of the kernel's time per SIMD -- 223 M and three quarters before a
    // vertex's four cells moved onto one lane --, the rest the three round trips of a look-up per 64 vertices)
    // per (case-table row, edge): bit t = triangle t of the row has a corner on that edge (mc_vn_cell; c_edgetri: made at compile
    // time from marching_lookup.h:64-320's rows); filled in front of the macro's barrier
    __shared__ u32 s_edgetri32[256 * 12 / 4];
    for (int i = (int)threadIdx.x; i < 256 * 12 / 4; i += 64 * MC_WPB_I) s_edgetri32[i] = c_edgetri.w[i];
    const u8* s_edgetri = (const u8*)s_edgetri32;
    __shared__ unsigned short s_item[MC_WPB_I][2 * MC_VN_CAP];  // the listed vertices: record lane | edge << 6; edge keys from the front, corner keys from the back
    __shared__ uint2 s_rc[MC_WPB_I][64];                         // per record of the chunk: {ix | iy << 16, iz}
    __shared__ uint2 s_rv[MC_WPB_I][64];                         // ... {first vertex, owned edges}
    MC_GROUP_LDS_SHARED_BEGIN
    unsigned short* item = s_item[w];
    uint2* rc = s_rc[w];
    uint2* rv = s_rv[w];
    {
        u32 ridx, rec, gtri0;
        int ix, iy, iz;
        const bool valid = mc_group_record(g, recs, r0 + (u32)lane, ridx, rec, ix, iy, iz, gtri0);
        const u32 ow = valid ? recown[ridx] : 0u;
        rc[lane] = make_uint2((u32)ix | ((u32)iy << 16), (u32)iz);
        rv[lane] = make_uint2(valid ? recvb[ridx] : 0u, ow & 0xFFFu);
        // the chunk's vertices are listed all at once when they fit, else 16 records at a time (which always fit)
        int nsub = 1;
        for (int sub = 0; sub < nsub; ++sub) {
            const int lo = nsub == 1 ? 0 : 16 * sub, hi = nsub == 1 ? 64 : lo + 16;
            const u32 own = (lane >= lo && lane < hi) ? ow & 0xFFFu : 0u, corn = (ow >> 16) & own;
            const u32 ne = (u32)__builtin_popcount(own & ~corn), nc = (u32)__builtin_popcount(corn);
            const u32 ie = wave_inclusive_scan(ne), ic = wave_inclusive_scan(nc);
            const u32 NE = (u32)__builtin_amdgcn_readlane((int)ie, 63), NC = (u32)__builtin_amdgcn_readlane((int)ic, 63);
            if (nsub == 1 && (NE > (u32)MC_VN_CAP || NC > (u32)MC_VN_CAP)) {
                nsub = 4;
                sub = -1;
                continue;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            {
                u32 m = own & ~corn, j = ie - ne;
                while (m) {
                    const u32 e = (u32)__builtin_ctz(m);
                    m &= m - 1u;
                    item[j++] = (unsigned short)((u32)lane | (e << 6));
                }
                m = corn;
                j = 2u * MC_VN_CAP - 1u - (ic - nc);
                while (m) {
                    const u32 e = (u32)__builtin_ctz(m);
                    m &= m - 1u;
                    item[j--] = (unsigned short)((u32)lane | (e << 6));
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // ---- vertices whose key is their lattice edge: 64 per round, lane = vertex; its four cells in sweep order on the one
            // lane (mc_vn_cells4), their triangles added in that order -- the reference's additions in the reference's order
            for (u32 k0 = 0; k0 < NE; k0 += 64u) {
                const u32 k = k0 + (u32)lane;
                const bool act = k < NE;
                const u32 it = act ? item[k] : 0u;
                const int e = (int)(it >> 6);
                const uint2 c = rc[it & 63u];
                const uint2 o = rv[it & 63u];
                const u32 v = o.x + (u32)__builtin_popcount(o.y & ((1u << e) - 1u));
                const int ax = edge_axis(e);
                // lower end of the lattice edge; the cells around it: offsets -1 / 0 on the two other axes (a0 = the faster of the
                // two, a1 the slower), the slower axis first.  Written per axis with constant offsets: a private array indexed by
                // a0 / a1 would live in scratch
                const int bx = (int)(c.x & 0xFFFFu) + (int)((MC_EDGE_OX >> e) & 1u), by = (int)(c.x >> 16) + (int)((MC_EDGE_OY >> e) & 1u),
                          bz = (int)c.y + (int)((MC_EDGE_OZ >> e) & 1u);
                int qx[4], qy[4], qz[4], me[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int d1 = 1 - (j >> 1), d0 = 1 - (j & 1);
                    qx[j] = bx - (ax != 0 ? d0 : 0);                           // a0 is x unless the edge runs along x
                    qy[j] = by - (ax == 0 ? d0 : 0) - (ax == 2 ? d1 : 0);      // y: a0 for an x edge, a1 for a z edge
                    qz[j] = bz - (ax != 2 ? d1 : 0);                           // a1 is z unless the edge runs along z
                    me[j] = mc_edge_of(ax, d0, d1);  // (in that cell the vertex's lattice edge is this edge: mc_resolve's rule)
                }
                McVnCell cell[4];
                mc_vn_cells4(p, recs, segcb, segtri, tnrm, s_edgetri, cap_tris, qx, qy, qz, me, act, cell);
                float sx = 0.0f, sy = 0.0f, sz = 0.0f;
#pragma unroll
                for (int j = 0; j < 4; ++j) mc_vn_add_cell_masked(act, cell[j], tnrm, sx, sy, sz);
                if (act && (u64)v < nverts) {
                    const float d = (sx * sx + sy * sy) + sz * sz;
                    const float inv = 1.0f / __builtin_sqrtf(d);  // glm::normalize: v * inversesqrt(dot(v, v)), inversesqrt = 1 / sqrt
                    vnrm[3ull * v] = sx * inv;
                    vnrm[3ull * v + 1] = sy * inv;
                    vnrm[3ull * v + 2] = sz * inv;
                }
            }
            // ---- vertices welded to a lattice corner: one lane per vertex, the 12 cells around the edge's two ends in sweep order
            for (u32 k0 = 0; k0 < NC; k0 += 64u) {
                const u32 k = k0 + (u32)lane;
                if (k >= NC) continue;
                const u32 it = item[2u * MC_VN_CAP - 1u - k];
                const int e = (int)(it >> 6);
                const uint2 c = rc[it & 63u];
                const uint2 o = rv[it & 63u];
                const u32 v = o.x + (u32)__builtin_popcount(o.y & ((1u << e) - 1u));
                const int ax = edge_axis(e);
                const int bx = (int)(c.x & 0xFFFFu) + (int)((MC_EDGE_OX >> e) & 1u), by = (int)(c.x >> 16) + (int)((MC_EDGE_OY >> e) & 1u),
                          bz = (int)c.y + (int)((MC_EDGE_OZ >> e) & 1u);
                float sx = 0.0f, sy = 0.0f, sz = 0.0f;
                for (int dz = -1; dz <= (ax == 2 ? 1 : 0); ++dz)
                    for (int dy = -1; dy <= (ax == 1 ? 1 : 0); ++dy)
                        for (int dx = -1; dx <= (ax == 0 ? 1 : 0); ++dx) {
                            McVnCell cc;
                            mc_vn_cell(p, recs, segcb, segtri, tlist, tnrm, s_edgetri, cap_tris, bx + dx, by + dy, bz + dz, v, -1, cc);
                            mc_vn_add_cell(cc, tnrm, sx, sy, sz);
                        }
                if ((u64)v < nverts) {
                    const float d = (sx * sx + sy * sy) + sz * sz;
                    const float inv = 1.0f / __builtin_sqrtf(d);
                    vnrm[3ull * v] = sx * inv;
                    vnrm[3ull * v + 1] = sy * inv;
                    vnrm[3ull * v + 2] = sz * inv;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    MC_GROUP_LDS_SHARED_END
}

#endif  // MC_NO_INDEX_KERNELS
// =============================================================== evaluate points
// Evaluator::evaluate(x,y,z) (evaluator.cpp:53) for a batch of points: out[i] = f(xyz[3i..3i+2]).
#ifndef MC_ONLY_INDEX_KERNELS  // (the module of the index kernels, compiled when MC_FLAG_INDEXED is first used, leaves the sweep kernels out)
extern "C" __global__ __launch_bounds__(256) void mc_eval(const float* __restrict__ xyz, float* __restrict__ out, u64 n) {
    const u64 i = (u64)blockIdx.x * 256ull + threadIdx.x;
    if (i < n) out[i] = mc_f(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]);
}
#endif  // MC_ONLY_INDEX_KERNELS
