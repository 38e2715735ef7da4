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
//                             (64-bit atomic adds by mc_classify; zeroed before every sweep)
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
// bound the range; an extremum that may lie inside (tested in double with a guard band) contributes
// its +-1.  Arguments are finite and below 8192 here (finite_on_domain, mc_expr.cpp).
__device__ __forceinline__ void mc_trig_iv(float l, float h, double shift, int which, float& lo, float& hi) {
    // extrema of sin at (n + 1/2) pi, of cos at n pi: maxima for even n, minima for odd n
    const double a = (double)l * 0.31830988618379067154 - shift, b = (double)h * 0.31830988618379067154 - shift;
    const double eps = 1e-9 * (1.0 + __builtin_fabs(a) + __builtin_fabs(b));
    const double ka = __builtin_ceil(a - eps), kb = __builtin_floor(b + eps);
    const float vl = mc_trig_eval(l, which), vh = mc_trig_eval(h, which);
    lo = __builtin_fminf(vl, vh) - 3e-7f;
    hi = __builtin_fmaxf(vl, vh) + 3e-7f;
    if (kb >= ka) {
        const bool two = kb > ka;
        const bool even = __builtin_fmod(ka, 2.0) == 0.0;
        if (two || even) hi = 1.0f;
        if (two || !even) lo = -1.0f;
    }
}
__device__ __forceinline__ void mc_sin_iv(float l, float h, float& lo, float& hi) { mc_trig_iv(l, h, 0.5, 0, lo, hi); }
__device__ __forceinline__ void mc_cos_iv(float l, float h, float& lo, float& hi) { mc_trig_iv(l, h, 0.0, 1, lo, hi); }

//@@MC_F_BEGIN  (replaced by generated code when JIT-compiled)
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
    float pad;
    u64 cap_tris;       // capacity of the vertex buffer in triangles
    u32* codes_tail;    // tail plane: one dword per (z,y) row = codes of cells main_cells..n1-1 (when tail_cells)
    int main_cells;     // cells per row held by the pitched code plane (n1 when there is no tail plane)
    int tail_cells;     // 0, or 1..4: width of the last chunk when it is handled by tail tiles
    int nchunk_main;    // chunks swept by the 256-wide tiles (nchunk, or nchunk-1 with a tail plane)
    int ntile_t;        // tail tiles per layer: ceil(n1/64), one row per lane
    u32* rec_cursor;    // record allocator, zeroed before every sweep: word 0 = "a region overflowed", then
                        // MC_NCUR bump cursors 128 bytes apart (word 32*(1+k)), one per region of recs
    u64 cap_recs;       // capacity of the record buffer in records (MC_NCUR equal regions)
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
#define MC_WPB_E 4
#endif
#ifndef MC_LIST_CAP
// triangles staged per wave in the emit kernel: one chunk of 64 records holds at most 64 * 5.  The kernel
// is latency-bound, so LDS is kept small for occupancy (measured: 0.274 ms at 768, 0.269 at 384, 0.256 at 320)
#define MC_LIST_CAP 320
#endif

__device__ __constant__ u64 c_tri_row[256] = MC_TRI_ROW_INIT;       // marching_lookup.h:64-320, nibble-packed
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
// What bounds it (measured): the 1.26 GB of stores at ~5 TB/s, with the instruction stream
// (89 M vector + 69 M scalar wave-instructions per 1025^3 sweep) just below that.  Hence the rules kept throughout:
// no global LOAD after the first store (vmcnt retires in order: a load would wait for every store in flight),
// whole 128-byte lines only, wave-uniform work on the scalar unit only where it is cheaper than on the VALU.
#ifndef MC_CLASSIFY_MINW
#define MC_CLASSIFY_MINW 1
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
#ifdef MC_DBG_NO_FLUSH  // timing probe only (wrong results): no allocation, no copy
            const bool probe_skip = true;
#else
            const bool probe_skip = false;
#endif
            u32 gb = 0u;
            if (!probe_skip) {
                u32* const cur = p.rec_cursor + 32u * (1u + t.region);
                asm volatile("s_atomic_add %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(gb) : "s"(cur), "0"(nbuf) : "memory");
            }
            const u32 rsize = (u32)(p.cap_recs / MC_NCUR);
            // past the region's end nothing is written and the overflow word is raised: the host grows the
            // buffer and sweeps again; mc_emit sees the word and stays out
            if (probe_skip) {
            } else if (gb + nbuf <= rsize) {
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
            u32 sb = 0;  // bit (4*c + 2*r + pl): sample x0+c, row r (0 lower / 1 upper), plane pl
#ifdef MC_CONS
            u32 ob = 0;  // same layout: the sample is inside every enabled constraint (marching.cpp:255-280)
#endif
#pragma unroll
            for (int c = 0; c < 5; ++c) {
                const float x = tl.xs[ln * 4 + c];
                sb |= (mc_f(x, yl, zk) > iso ? 1u : 0u) << (4 * c + 0);
                sb |= (mc_f(x, yl, zk1) > iso ? 1u : 0u) << (4 * c + 1);
                sb |= (mc_f(x, yu, zk) > iso ? 1u : 0u) << (4 * c + 2);
                sb |= (mc_f(x, yu, zk1) > iso ? 1u : 0u) << (4 * c + 3);
#ifdef MC_CONS
                ob |= (mc_ok(x, yl, zk) ? 1u : 0u) << (4 * c + 0);
                ob |= (mc_ok(x, yl, zk1) ? 1u : 0u) << (4 * c + 1);
                ob |= (mc_ok(x, yu, zk) ? 1u : 0u) << (4 * c + 2);
                ob |= (mc_ok(x, yu, zk1) ? 1u : 0u) << (4 * c + 3);
#endif
            }
            // cube code bit i <-> corner i (marching.cpp:471-472): with s = nibble of sample c and
            // n = nibble of sample c+1:  0:(x0,y0,z0)=s.0  1:(x1,y0,z0)=n.0  2:(x1,y1,z0)=n.2  3:(x0,y1,z0)=s.2
            //                            4:(x0,y0,z1)=s.1  5:(x1,y0,z1)=n.1  6:(x1,y1,z1)=n.3  7:(x0,y1,z1)=s.3
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const u32 sN = (sb >> (4 * c)) & 0xFu, nN = (sb >> (4 * c + 4)) & 0xFu;
                const u32 code = (sN & 1u) | ((nN & 1u) << 1) | (((nN >> 2) & 1u) << 2) | (((sN >> 2) & 1u) << 3) |
                                 (((sN >> 1) & 1u) << 4) | (((nN >> 1) & 1u) << 5) | (((nN >> 3) & 1u) << 6) |
                                 (((sN >> 3) & 1u) << 7);
#ifdef MC_CONS
                // a cell with a corner outside a constraint is skipped (marching.cpp:476): no triangles, code 0
                if (((ob >> (4 * c)) & 0xFFu) != 0xFFu) continue;
#endif
                if (x0 + c < n1) dw |= code << (8 * c);
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
#ifdef MC_DBG_NO_ROWLOOP  // timing probe only (wrong results)
        if (false) {
#else
        if (!TAIL) {
#endif
            // The pending code rows this chunk completes.  Each row is stored WHOLE (256 B, full 128-byte lines) with
            // what the walk proved -- lanes all-above as ~0, the others, listed ones included, as 0 -- and then the
            // entry lanes overwrite their own dwords with ONE scattered store for the whole chunk: same wave, same
            // addresses, issued right behind, so the line is still being assembled in L2 when the dwords arrive.
            // (Measured: 0.264 ms against 0.283 for gathering each row's listed dwords with ds_bpermute before its
            // store; but 0.296 when the rows were stored by the walk, microseconds earlier -- by then a line may have
            // left L2 and the late dword becomes a read-modify-write in HBM.)
            u64 f = fit;
            while (f) {
                const int jr = __builtin_ctzll(f);
                f &= f - 1ull;
                const u64 all = ((u64)(u32)__builtin_amdgcn_readlane((int)rm.allhi, jr) << 32) | (u32)__builtin_amdgcn_readlane((int)rm.alllo, jr);
                const u32 v = select_by_mask(all, vmask_row, 0u);
#if !defined(MC_DBG_NO_STORE) && !defined(MC_DBG_NO_ROWSTORE)
                __builtin_amdgcn_raw_buffer_store_b32(
                    v, __builtin_amdgcn_make_buffer_rsrc(tilebase + (u64)((u32)jr * (u32)p.pitch), 0, (int)p.pitch, 0x00020000), (u32)xl0, 0, 0);
#else
                asm volatile("" ::"v"(v));
#endif
            }
#if !defined(MC_DBG_NO_STORE) && !defined(MC_DBG_NO_ROWSTORE)
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
#endif
        }
        pend &= ~fit;
        e0 = e1;
    }
    flush_records();
    return rowbase;
}


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
#pragma unroll
    for (int i = (int)threadIdx.x; i < 256; i += 64 * MC_WPB_C)
        s_lut[i] = (unsigned short)(c_tri_count[i] | (c_amb_face[i] << 8));
    __syncthreads();

    const McParams p = *P;
    const int lane = threadIdx.x & 63;
    // the wave index is wave-uniform, but the compiler only knows that if told: without the
    // readfirstlane every tile coordinate (and the whole walk's scalar algebra) lands in VGPRs
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const long long tile = (long long)blockIdx.x * MC_WPB_C + w;
    const long long ntiles_main = (long long)p.nchunk_main * p.ntile_y * p.nz;
    const long long ntiles = ntiles_main + (long long)p.ntile_t * p.nz;
    if (tile >= ntiles) return;  // whole wave
    const bool is_tail = tile >= ntiles_main;  // wave-uniform
    const int n1 = p.n1;
    int ch, ty, lz, y0, ny;
    // (chunk fastest: the 4 waves of a workgroup cover 1 KB of each code row together.  Layer-fastest, which
    // gives the 4 waves equal work, measured the same: 0.380 vs 0.378 ms.)
    if (!is_tail) {
        ch = (int)(tile % p.nchunk_main);
        const long long t2 = tile / p.nchunk_main;
        ty = (int)(t2 % p.ntile_y);
        lz = (int)(t2 / p.ntile_y);
        y0 = ty * p.tile_h;
        ny = min(p.tile_h, n1 - y0);
    } else {  // tail tile: the last chunk's 1..4 cells of 64 consecutive rows, one row per lane
        const long long tt = tile - ntiles_main;
        ch = p.nchunk - 1;
        ty = (int)(tt % p.ntile_t);
        lz = (int)(tt / p.ntile_t);
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
#ifndef MC_DBG_NO_RECORD
        if (mixedL) rbase = mc_backend<true>(p, tt, tl, s_lut, marker, seg_cnt, recbuf, rowoff, rmt, mixedL, 0u, codes, recs, tailbuf);
#endif
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (rvalid) {
            const u32 own = select_by_mask(mixedL, tailbuf[lane], select_by_mask(laneAll, vm, 0u));
#ifndef MC_DBG_NO_STORE
            p.codes_tail[(u64)lz * n1 + y0 + lane] = own;
#endif
            const u64 sg = tt.seg0 + (u64)lane * p.nchunk;
            const u32 c = seg_cnt[lane];
            segcb[sg] = make_uint2(c, rbase);
            if (c) atomicAdd(&grpsum[sg >> 6], (u64)(c & 0xFFFFu) | ((u64)(c >> 16) << 32));
        }
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
#if defined(MC_DBG_NO_STORE)
        asm volatile("" ::"v"(v));
#else
        __builtin_amdgcn_raw_buffer_store_b32(v, __builtin_amdgcn_make_buffer_rsrc(rowbase, 0, rowbytes, 0x00020000), xoff, 0, 0);
#endif
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
        const u64 cull = rowCull & rowsValid;
#ifndef MC_STORES_LAST
        // (1) aligned blocks of 4 culled rows
        u64 m4 = cull & (cull >> 1) & (cull >> 2) & (cull >> 3) & 0x1111111111111111ull;
        const u64 blockRows = m4 | (m4 << 1) | (m4 << 2) | (m4 << 3);
#ifdef MC_DBG_NO_CULLSTORE
        m4 = 0ull;
#endif
        while (m4) {
            const int j = __builtin_ctzll(m4);
            m4 &= m4 - 1ull;
            const u32 c = ((rowFull >> j) >> q) & 1ull ? 0xFFFFFFFFu : 0u;
#ifndef MC_DBG_NO_STORE
            typedef u32 u32x4 __attribute__((ext_vector_type(4)));
            u32x4 v;
            v.x = c & vm4[0];
            v.y = c & vm4[1];
            v.z = c & vm4[2];
            v.w = c & vm4[3];
            __builtin_amdgcn_raw_buffer_store_b128(
                v, __builtin_amdgcn_make_buffer_rsrc(tilebase + (u64)((u32)j * (u32)p.pitch), 0, wbytes, 0x00020000), woff, 0, 0);
#else
            asm volatile("" ::"v"(c));
#endif
        }
        // (2) the other culled rows
        u64 m1 = cull & ~blockRows;
#ifdef MC_DBG_NO_CULLSTORE
        m1 = 0ull;
#endif
        while (m1) {
            const int j = __builtin_ctzll(m1);
            m1 &= m1 - 1ull;
            const u32 v = ((rowFull >> j) & 1ull) ? vmask : 0u;
#ifndef MC_DBG_NO_STORE
            __builtin_amdgcn_raw_buffer_store_b32(
                v, __builtin_amdgcn_make_buffer_rsrc(tilebase + (u64)((u32)j * (u32)p.pitch), 0, rowbytes, 0x00020000), xoff, 0, 0);
#else
            asm volatile("" ::"v"(v));
#endif
        }
#endif
        // (3) undecided rows: the lane-level masks go to lane j of rm, the row itself to the back-end.
        // Two adjacent undecided rows share ONE evaluation over the box of both (3 sample rows): half
        // the interval evaluations, for a few more lanes handed to the back-end (which is exact, so a
        // lane listed needlessly just yields uniform codes and no record).
        rowPend = rowsValid & ~cull;
        const u64 lanesIn = __ballot(x0 < n1);  // lanes that hold cells of the grid (a ragged last chunk has fewer)
        u64 mu = rowPend;
        while (mu) {
            const int j = __builtin_ctzll(mu);
            mu &= mu - 1ull;
#ifndef MC_NO_PAIR
            const int pair = (int)((mu >> ((j + 1) & 63)) & 1ull);  // row j+1 is undecided too (j + 1 < ny then)
            if (pair) mu &= mu - 1ull;
#else
            const int pair = 0;
#endif
            const float ya = readlane_f(yv, j), yb = readlane_f(yv, j + 1 + pair);
            float lo, hi;
            mc_f_iv(lxl, lxh, __builtin_fminf(ya, yb), __builtin_fmaxf(ya, yb), zl, zh, lo, hi);
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
#ifdef MC_STORES_LAST
        // (1) aligned blocks of 4 culled rows
        u64 m4 = cull & (cull >> 1) & (cull >> 2) & (cull >> 3) & 0x1111111111111111ull;
        const u64 blockRows = m4 | (m4 << 1) | (m4 << 2) | (m4 << 3);
#ifdef MC_DBG_NO_CULLSTORE
        m4 = 0ull;
#endif
        while (m4) {
            const int j = __builtin_ctzll(m4);
            m4 &= m4 - 1ull;
            const u32 c = ((rowFull >> j) >> q) & 1ull ? 0xFFFFFFFFu : 0u;
#ifndef MC_DBG_NO_STORE
            typedef u32 u32x4 __attribute__((ext_vector_type(4)));
            u32x4 v;
            v.x = c & vm4[0];
            v.y = c & vm4[1];
            v.z = c & vm4[2];
            v.w = c & vm4[3];
            __builtin_amdgcn_raw_buffer_store_b128(
                v, __builtin_amdgcn_make_buffer_rsrc(tilebase + (u64)((u32)j * (u32)p.pitch), 0, wbytes, 0x00020000), woff, 0, 0);
#else
            asm volatile("" ::"v"(c));
#endif
        }
        // (2) the other culled rows
        u64 m1 = cull & ~blockRows;
#ifdef MC_DBG_NO_CULLSTORE
        m1 = 0ull;
#endif
        while (m1) {
            const int j = __builtin_ctzll(m1);
            m1 &= m1 - 1ull;
            const u32 v = ((rowFull >> j) & 1ull) ? vmask : 0u;
#ifndef MC_DBG_NO_STORE
            __builtin_amdgcn_raw_buffer_store_b32(
                v, __builtin_amdgcn_make_buffer_rsrc(tilebase + (u64)((u32)j * (u32)p.pitch), 0, rowbytes, 0x00020000), xoff, 0, 0);
#else
            asm volatile("" ::"v"(v));
#endif
        }
#endif
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
#ifndef MC_DBG_NO_RECORD
    if (rowPend) rbase = mc_backend(p, tc, tl, s_lut, marker, seg_cnt, recbuf, rowoff, rm, rowPend, vmask, codes, recs);
#endif
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (lane < ny) {
        const u64 sg = tc.seg0 + (u64)lane * p.nchunk;
        const u32 c = seg_cnt[lane];
        segcb[sg] = make_uint2(c, rbase);
        // group sums for the scan (group = 64 consecutive segments): triangles | active cells << 32
        if (c) atomicAdd(&grpsum[sg >> 6], (u64)(c & 0xFFFFu) | ((u64)(c >> 16) << 32));
    }
}

// =============================================================== K3: emit
struct McVert {
    float x, y, z;
};

// position of triangle-vertex `slot` (0..14) of table row `row` in cell (ix,iy,iz):
// marching.cpp:557-583 (edge interpolation from corner v1 to corner v2 of the edge table) with
// Marching::interp (:437-446) applied to x, y and z.  The reference evaluates the quotient
// (iso - v_s)/(v_e - v_s) once per axis with identical operands; it is computed once here.
// The fallback `x_s + 0.5*(x_e - x_s)` is evaluated in double by the reference; one add of two
// floats rounded to double and then to float equals the float add (53 >= 2*24+2).
__device__ __forceinline__ McVert mc_vertex(const McParams& p, const float* s_axis, const u64* s_row, const u8* s_edge,
                                            int row, int slot, int ix, int iy, int iz) {
    const int edge = (int)((s_row[row] >> (4 * slot)) & 0xF);
    const int ec = s_edge[edge];
    const int v1 = ec & 0xF, v2 = ec >> 4;
    const float xs = s_axis[ix + cx_bit(v1)], xe = s_axis[ix + cx_bit(v2)];
    const float ys = s_axis[iy + cy_bit(v1)], ye = s_axis[iy + cy_bit(v2)];
    const float zs = s_axis[iz + cz_bit(v1)], ze = s_axis[iz + cz_bit(v2)];
    const float vs = mc_F(p, xs, ys, zs);
    const float ve = mc_F(p, xe, ye, ze);
    const float t = (p.iso - vs) / (ve - vs);
    const float dx = xe - xs, dy = ye - ys, dz = ze - zs;
    const float vx = t * dx, vy = t * dy, vz = t * dz;
    McVert r;
    r.x = (__builtin_isinf(vx) || __builtin_isnan(vx)) ? xs + 0.5f * dx : xs + vx;
    r.y = (__builtin_isinf(vy) || __builtin_isnan(vy)) ? ys + 0.5f * dy : ys + vy;
    r.z = (__builtin_isinf(vz) || __builtin_isnan(vz)) ? zs + 0.5f * dz : zs + vz;
    return r;
}

// One wave = one GROUP of 64 consecutive segments.  Phase 1, one lane per RECORD (= active
// cell, written by mc_classify): find the owning segment by binary search over the group's
// active-cell offsets (LDS), read the record, and expand its triangles into 4-byte work items
// in LDS -- the list index is the triangle's position in the reference's emission order.
// Phase 2, one lane per output VERTEX: edge lookup (nibble-packed table row in LDS), two corner
// evaluations, the interpolation, the central-difference gradient of f for the normal, 24-byte
// store.
extern "C" __global__ __launch_bounds__(64 * MC_WPB_E) void mc_emit(const McParams* __restrict__ P, const u32* __restrict__ recs,
                                                                     const uint2* __restrict__ segcb, const uint2* __restrict__ grpoff,
                                                                     float* __restrict__ verts, unsigned short* __restrict__ trimeta) {
    __shared__ u64 s_row[256];
    __shared__ u8 s_edge[16];
    __shared__ u32 s_list[MC_WPB_E][MC_LIST_CAP];
    __shared__ u32 s_seg[MC_WPB_E][64];
    __shared__ u32 s_act[MC_WPB_E][66];
    __shared__ u32 s_tri[MC_WPB_E][64];
    __shared__ u32 s_rbase[MC_WPB_E][64];
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
    const u32 rec_overflow = p.rec_cursor[0];
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
        const u8 tedge = c_edge_corner[threadIdx.x < 12 ? threadIdx.x : 0];
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int i = (int)threadIdx.x + NT * k;
            if (i < 512 && 4 * i <= n1p) ((float4*)s_axis)[i] = t[k];
        }
#pragma unroll
        for (int k = 0; k < NR; ++k)
            if ((int)threadIdx.x + NT * k < 256) s_row[(int)threadIdx.x + NT * k] = trow[k];
        if (threadIdx.x < 12) s_edge[threadIdx.x] = tedge;
    }
    // rec_overflow: mc_classify ran out of record space (the host grows the buffer and sweeps again)
    const bool mine = in_range && g0.y != g1.y && rec_overflow == 0u;  // 64 segments with an active cell
    if (!__syncthreads_or(mine ? 1 : 0)) return;  // also the barrier that publishes the tables
    if (!mine) return;
    {
    // the scan gives the group's first triangle; the prefix inside the group is a wavefront scan of
    // the per-segment counts (triangles | active cells << 16)
    const u32 ctri = cnt & 0xFFFFu, cact = cnt >> 16;
    const u32 itri = wave_inclusive_scan(ctri), iact = wave_inclusive_scan(cact);
    const uint2 o0 = make_uint2(g0.x + (itri - ctri), iact - cact);  // {first triangle, group-local first record}
    const u32 act_base = 0u;
    const u32 nrec = (u32)__builtin_amdgcn_readlane((int)iact, 63);  // records of the group

    u32* list = s_list[w];
    u32* segrec = s_seg[w];
    u32* actoff = s_act[w];
    u32* trioff = s_tri[w];
    u32* rbase = s_rbase[w];
    const int n1 = p.n1;
    {
        const u32 sg = min(seg, p.nseg - 1u);
        const u32 rowidx = sg / (u32)p.nchunk;
        const u32 ch = sg - rowidx * (u32)p.nchunk;
        const u32 lz = rowidx / (u32)n1;
        const u32 iy = rowidx - lz * (u32)n1;
        segrec[lane] = iy | ((u32)(p.z_begin + (int)lz) << 11) | (ch << 22);
        actoff[lane] = o0.y - act_base;
        trioff[lane] = o0.x;
        rbase[lane] = cb.y;
        if (lane == 63) actoff[64] = nrec;
    }
    const float h = 0.5f * p.step;
    const bool want_normals = (p.flags & 1u) != 0u;
    const bool want_meta = (p.flags & 16u) != 0u && trimeta != nullptr;

    u32 nlist = 0;                                             // triangles staged
    u32 listbase = (u32)__builtin_amdgcn_readfirstlane((int)o0.x);  // global index of list[0]

    // drains the staged triangles: one lane per vertex
    auto flush = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
#ifdef MC_DBG_EMIT_NOFLUSH
        const u32 nverts = 0;
#else
        const u32 nverts = 3u * nlist;
#endif
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
                const McVert q = mc_vertex(p, s_axis, s_row, s_edge, row, 3 * t + k, ix, iy, iz);
                float nx = 0.0f, ny = 0.0f, nz = 0.0f;
                if (want_normals) {
                    // n = g/|g|, g = central difference of F at the vertex, h = step/2
                    const float gx = mc_F(p, q.x + h, q.y, q.z) - mc_F(p, q.x - h, q.y, q.z);
                    const float gy = mc_F(p, q.x, q.y + h, q.z) - mc_F(p, q.x, q.y - h, q.z);
                    const float gz = mc_F(p, q.x, q.y, q.z + h) - mc_F(p, q.x, q.y, q.z - h);
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
                        const McVert a = mc_vertex(p, s_axis, s_row, s_edge, row, 3 * t + 0, ix, iy, iz);
                        const McVert b = mc_vertex(p, s_axis, s_row, s_edge, row, 3 * t + 1, ix, iy, iz);
                        const McVert c = mc_vertex(p, s_axis, s_row, s_edge, row, 3 * t + 2, ix, iy, iz);
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
                const u64 gtri = (u64)listbase + tri;
#ifdef MC_DBG_EMIT_NOSTORE
                asm volatile("" ::"v"(q.x), "v"(q.y), "v"(q.z), "v"(nx), "v"(ny), "v"(nz));
                if (false) {
#else
                if (gtri < p.cap_tris) {
#endif
                    // three 8-byte stores per vertex.  (Staging the iteration's 1536 contiguous bytes
                    // in LDS and writing 16-byte pieces was measured slower: 0.31 vs 0.29 ms.)
                    float* o = verts + (gtri * 3ull + (u64)k) * 6ull;
#ifdef MC_EMIT_STORE_X4
                    typedef float f4u __attribute__((ext_vector_type(4), aligned(8)));
                    typedef float f2u __attribute__((ext_vector_type(2), aligned(8)));
                    f4u a;
                    a.x = q.x; a.y = q.y; a.z = q.z; a.w = nx;
                    f2u b;
                    b.x = ny; b.y = nz;
                    *(f4u*)o = a;
                    *(f2u*)(o + 4) = b;
#else
                    ((float2*)o)[0] = make_float2(q.x, q.y);
                    ((float2*)o)[1] = make_float2(q.z, nx);
                    ((float2*)o)[2] = make_float2(ny, nz);
#endif
                    // optional (MC_FLAG_TRI_META): table row actually used | triangle number inside its
                    // cell << 8, so a host can rebuild the reference's per-cell, per-edge vertex order
                    if (want_meta && k == 0) trimeta[gtri] = (unsigned short)(row | (t << 8));
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        listbase += nlist;
        nlist = 0;
    };

    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (u32 r0 = 0; r0 < nrec; r0 += 64u) {
        const u32 r = r0 + (u32)lane;
        const bool valid = r < nrec;
        // owning segment: the largest s with actoff[s] <= r (empty segments repeat the value)
        u32 lo = 0, hi = 64;
#pragma unroll
        for (int it = 0; it < 6; ++it) {
            const u32 mid = (lo + hi) >> 1;
            if (actoff[mid] <= r) lo = mid; else hi = mid;
        }
        u32 rec = 0, gtri0 = 0;
        if (valid) {
            rec = recs[rbase[lo] + (r - actoff[lo])];
            gtri0 = trioff[lo] + (rec >> 20);
        }
        const u32 nt = (rec >> 17) & 7u;
        // the chunk's triangles are one contiguous range of the global order
        const int lv = (int)min(63u, nrec - 1u - r0);
        const u32 first = (u32)__builtin_amdgcn_readfirstlane((int)gtri0);
        const u32 chunk_t = (u32)__builtin_amdgcn_readlane((int)(gtri0 + nt), lv) - first;
        if (nlist + chunk_t > MC_LIST_CAP) flush();
        const u32 base = (gtri0 - listbase) & 0xFFFFFFFFu;
        const u32 item = lo | ((rec & 0xFFu) << 6) | (((rec >> 8) & 0xFFu) << 14) | (((rec >> 16) & 1u) << 22);
        for (u32 t = 0; t < nt; ++t) list[base + t] = item | (t << 23);
        nlist += chunk_t;
    }
    if (nlist) flush();
    }
}

// =============================================================== evaluate points
// Evaluator::evaluate(x,y,z) (evaluator.cpp:53) for a batch of points: out[i] = f(xyz[3i..3i+2]).
extern "C" __global__ __launch_bounds__(256) void mc_eval(const float* __restrict__ xyz, float* __restrict__ out, u64 n) {
    const u64 i = (u64)blockIdx.x * 256ull + threadIdx.x;
    if (i < n) out[i] = mc_f(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]);
}
