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
//   segcnt u32[nseg]          per SEGMENT (= 256 x-consecutive cells of one row):
//                             triangles | active cells << 16; seg = row*nchunk + chunk
//   recs   u32[nseg][256]     per segment, its ACTIVE cells compacted in x order (only the first
//                             `active` entries are ever written / read): cell | code<<8 | flip<<16 |
//                             triangles<<17 | triangle prefix inside the segment<<20
//   segoff uint2[nseg+1]      exclusive scan of segcnt: {triangle offset, active-cell offset}
//   verts  float[T][3][6]     {x,y,z,nx,ny,nz} per vertex, 72 B per triangle, reference order
#ifndef MC_JIT
#include <hip/hip_runtime.h>
#include "../../include/mc_tables_data.h"
#else
#include "mc_tables_data.h"
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

//@@MC_F_BEGIN  (replaced by generated code when JIT-compiled)
__device__ __forceinline__ float mc_f(float x, float y, float z) {
    const float t0 = z * z;
    const float t1 = t0 - 1.0f;
    const float t2 = y * y;
    const float t3 = t2 + t1;
    const float t4 = x * x;
    return t4 + t3;  // x^2+(y^2+(z^2-1)): the reference's right-to-left reduction
}
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
};

#define MC_SEG 256          // cells per segment (4 per lane)
#define MC_LIST_CAP 768     // triangles staged per wave in the emit kernel (>= 64 records * 5)

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
    return mc_f(p.sx * x, p.sy * y, p.sz * z);
}

// acc = 2*acc + mask[lane]: one VALU op shifts a wave-mask bit into a per-lane register.
__device__ __forceinline__ void push_bit(u32& acc, u64 m) {
    asm("v_addc_co_u32_e64 %0, vcc, %0, %0, %1" : "+v"(acc) : "s"(m) : "vcc");
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
__device__ __forceinline__ bool amb_flip(const McParams& p, int face, int ix, int iy, int iz) {
    const u32 fc = c_face_corner[face];
    float mx = 0.0f, my = 0.0f, mz = 0.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int v = (fc >> (4 * i)) & 0xF;
        mx += p.axis[ix + cx_bit(v)];
        my += p.axis[iy + cy_bit(v)];
        mz += p.axis[iz + cz_bit(v)];
    }
    mx *= 0.25f;  // == (float)((double)m / 4.0): an exact scaling
    my *= 0.25f;
    mz *= 0.25f;
    return mc_F(p, mx, my, mz) > p.iso;
}

// =============================================================== K1: classify
// One wave = one tile: a 256-cell x-chunk (4 consecutive cells per lane) of one z layer,
// walked along y for up to 63 rows.  Every lattice sample of the two z planes is evaluated
// once per tile row and the previous row's samples stay in registers, so a cell costs 2
// evaluations instead of 8.  The kernel is bound by instruction issue, not by HBM (measured,
// profiles/: a VOP3 op such as v_cmp->SGPR, v_max3 or v_addc costs ~4 cycles per wave on its
// SIMD, a VOP2 add ~2.2, and scalar ops are not free either), so the design minimises
// instructions per 256-cell step:
//
//  * Uniform steps (every corner of all 256 cells on one side of iso -- ~70 % of the steps
//    of the 1024^3 sphere) are recognised with a min/max tree over the lane's 8 new samples
//    and two v_cmp, and store 0x00000000 / 0xFFFFFFFF: ~20 VALU + ~10 SALU ops.
//  * Mixed steps classify each LANE the same way with four more v_cmp (the lane's "x+4"
//    neighbour column) and a little mask algebra: lanes whose 4 cells are all-below /
//    all-above store 0 / ~0; the few remaining lanes (1-4 per step on a smooth surface) only
//    note their position in a per-wave LDS list.
//  * The expensive part -- assembling the 8-bit cube codes of those lanes, triangle-count
//    lookup, ambiguity test, per-segment prefix sums and the compact per-cell RECORDS the
//    emit kernel consumes -- runs lane-parallel over that list once per tile (a tile has ~35
//    listed lanes on a smooth surface) instead of wave-wide in each of its ~20 mixed steps.
#define MC_ENT_CAP 512  // (row, lane) positions staged per wave before the record pass runs

struct McTileCtx {
    int ch, y0, iz, lz, lane;
    u64 seg0;  // segment index of tile row 0; + nchunk per row
};

// lane-parallel pass over the staged positions (sorted by (row, lane)): one lane = one dword
// of 4 cells whose corners are not all on one side of iso
__device__ __forceinline__ void mc_record_pass(const McParams& p, const McTileCtx& t, const unsigned short* s_lut,
                                               const unsigned short* ent_pos, u32* seg_cnt, u32 nent,
                                               u8* __restrict__ codes, u32* __restrict__ recs, int& carry_j,
                                               u32& carry_val) {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int n1 = p.n1;
    const float* __restrict__ ax = p.axs;
    const float* __restrict__ ay = p.axs + (n1 + 1);
    const float* __restrict__ az = p.axs + 2 * (n1 + 1);
    const float zk = az[t.iz], zk1 = az[t.iz + 1];
    const float iso = p.iso;
    for (u32 e0 = 0; e0 < nent; e0 += 64u) {
        const u32 e = e0 + (u32)t.lane;
        const bool valid = e < nent;
        const u32 pos = valid ? ent_pos[e] : 0xFFFFu;
        const int j = valid ? (int)(pos >> 6) : 1023, ln = (int)(pos & 63u);
        const int jj = valid ? j : 0;
        const int x0 = t.ch * MC_SEG + ln * 4;
        // the 20 lattice samples of the lane's 4 cells: same mc_f, same operands, same compare as
        // the walk (marching.cpp:475-479, :497-505), so both agree on every shared sample
        const float yl = ay[t.y0 + jj], yu = ay[t.y0 + jj + 1];
        u32 sb = 0;  // bit (4*c + 2*r + pl): sample x0+c, row r (0 lower / 1 upper), plane pl
#pragma unroll
        for (int c = 0; c < 5; ++c) {
            const float x = ax[min(x0 + c, n1)];
            sb |= (mc_f(x, yl, zk) > iso ? 1u : 0u) << (4 * c + 0);
            sb |= (mc_f(x, yl, zk1) > iso ? 1u : 0u) << (4 * c + 1);
            sb |= (mc_f(x, yu, zk) > iso ? 1u : 0u) << (4 * c + 2);
            sb |= (mc_f(x, yu, zk1) > iso ? 1u : 0u) << (4 * c + 3);
        }
        // cube code bit i <-> corner i (marching.cpp:471-472): with s = nibble of sample c and
        // n = nibble of sample c+1:  0:(x0,y0,z0)=s.0  1:(x1,y0,z0)=n.0  2:(x1,y1,z0)=n.2  3:(x0,y1,z0)=s.2
        //                            4:(x0,y0,z1)=s.1  5:(x1,y0,z1)=n.1  6:(x1,y1,z1)=n.3  7:(x0,y1,z1)=s.3
        u32 dw = 0;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const u32 sN = (sb >> (4 * c)) & 0xFu, nN = (sb >> (4 * c + 4)) & 0xFu;
            const u32 code = (sN & 1u) | ((nN & 1u) << 1) | (((nN >> 2) & 1u) << 2) | (((sN >> 2) & 1u) << 3) |
                             (((sN >> 1) & 1u) << 4) | (((nN >> 1) & 1u) << 5) | (((nN >> 3) & 1u) << 6) |
                             (((sN >> 3) & 1u) << 7);
            if (x0 + c < n1) dw |= code << (8 * c);
        }
        if (valid && x0 < n1) *(u32*)(codes + ((u64)t.lz * n1 + t.y0 + j) * p.pitch + x0) = dw;

        // per cell: triangle count and ambiguity flip; meta nibble c = count | flip<<3
        u32 meta = 0, packed = 0;  // packed = triangles | active cells << 16 of this entry
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int code = valid ? (int)((dw >> (8 * c)) & 0xFF) : 0;
            if (code != 0 && code != 255) {
                const u32 lut = s_lut[code];
                u32 nt = lut & 0xFFu, flip = 0;
                const int face = (int)(lut >> 8);
                if (face != 0xFF)
                    if (amb_flip(p, face, x0 + c, t.y0 + j, t.iz)) {
                        nt = s_lut[255 - code] & 0xFFu;
                        flip = 1;
                    }
                meta |= (nt | (flip << 3)) << (4 * c);
                packed += nt + (1u << 16);
            }
        }
        // prefix inside each segment (= tile row j): wave scan minus the scan value at the
        // segment's first entry; both halves of `packed` are non-decreasing, so a max-scan of
        // "exclusive value at segment heads" propagates the base to the followers.
        const u32 incl = wave_inclusive_scan(packed);
        const u32 excl = incl - packed;
        const int jprev = __builtin_amdgcn_update_dpp(-1, j, 0x138, 0xf, 0xf, false);  // wave_shr:1
        const bool head = valid && (t.lane == 0 || jprev != j);
        const u32 base = wave_inclusive_max(head ? excl : 0u);
        const int j0 = __builtin_amdgcn_readfirstlane(j);
        u32 pre = excl - base;
        if (j == j0 && j0 == carry_j) pre += carry_val;  // segment continues from the previous 64 entries
        if (valid) {
            // segment totals: order-independent LDS adds; slot j is read by lane j at tile end
            if (packed) __hip_atomic_fetch_add(&seg_cnt[j], packed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            u32* __restrict__ rseg = recs + (t.seg0 + (u64)j * p.nchunk) * MC_SEG;
            u32 rank = pre >> 16, tpre = pre & 0xFFFFu;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const u32 m = (meta >> (4 * c)) & 0xFu;
                const u32 nt = m & 7u;
                if (nt) {
                    rseg[rank++] = (u32)(ln * 4 + c) | (((dw >> (8 * c)) & 0xFFu) << 8) | ((m >> 3) << 16) | (nt << 17) |
                                   (tpre << 20);
                    tpre += nt;
                }
            }
        }
        // carry for a segment that spans two 64-entry chunks
        const int lv = (int)min(63u, nent - 1u - e0);
        carry_j = __builtin_amdgcn_readlane(j, lv);
        carry_val = (u32)__builtin_amdgcn_readlane((int)(pre + packed), lv);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

extern "C" __global__ __launch_bounds__(256) void mc_classify(const McParams* __restrict__ P, u8* __restrict__ codes,
                                                               u32* __restrict__ segcnt, u32* __restrict__ recs) {
    __shared__ unsigned short s_lut[256];  // triangle count | ambiguity face << 8
    __shared__ unsigned short s_ent_pos[4][MC_ENT_CAP];
    __shared__ u32 s_segcnt[4][64];
    s_lut[threadIdx.x] = (unsigned short)(c_tri_count[threadIdx.x] | (c_amb_face[threadIdx.x] << 8));
    __syncthreads();

    const McParams p = *P;
    const int lane = threadIdx.x & 63;
    // the wave index is wave-uniform, but the compiler only knows that if told: without the
    // readfirstlane every tile coordinate (and the whole walk's scalar algebra) lands in VGPRs
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const long long tile = (long long)blockIdx.x * 4 + w;
    const long long ntiles = (long long)p.nchunk * p.ntile_y * p.nz;
    if (tile >= ntiles) return;  // whole wave
    const int ch = (int)(tile % p.nchunk);
    const long long t2 = tile / p.nchunk;
    const int ty = (int)(t2 % p.ntile_y);
    const int lz = (int)(t2 / p.ntile_y);
    const int iz = p.z_begin + lz;
    const int n1 = p.n1;
    const int x0 = ch * MC_SEG + lane * 4;
    const int y0 = ty * p.tile_h;
    const int ny = min(p.tile_h, n1 - y0);
    const float iso = p.iso;

    unsigned short* ent_pos = s_ent_pos[w];
    u32* seg_cnt = s_segcnt[w];
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

    // sample column x = xe for the tile's 64 sample rows (lane = row): bit j of E0 / E1 is the
    // "x+4" neighbour of lane 63 in row j.  ENone / EFull bit j: both rows j, j+1 of both planes
    // are below / above iso in that column.
    const u64 E0 = __ballot(mc_f(xe, yv, zk) > iso);
    const u64 E1 = __ballot(mc_f(xe, yv, zk1) > iso);
    const u64 ENone = ~((E0 | E1) | ((E0 | E1) >> 1));
    const u64 EFull = (E0 & E1) & ((E0 & E1) >> 1);

    const u32 vmask = (x0 + 3 < n1) ? 0xFFFFFFFFu : (x0 + 2 < n1) ? 0x00FFFFFFu : (x0 + 1 < n1) ? 0x0000FFFFu
                      : (x0 < n1) ? 0x000000FFu : 0u;

    McTileCtx tc;
    tc.ch = ch;
    tc.y0 = y0;
    tc.iz = iz;
    tc.lz = lz;
    tc.lane = lane;
    tc.seg0 = ((u64)lz * n1 + y0) * p.nchunk + ch;
    u32 nent = 0;
    int carry_j = -1;
    u32 carry_val = 0;

    // Two sample-row register sets (plane z / plane z+1 each) ping-pong between "lower row" and
    // "upper row" so the walk never copies registers.
    float r0a[4], r0c[4], r1a[4], r1c[4];
    u64 gtPrev, gePrev;  // lower row, per lane: some sample > iso / every sample > iso
    // row base of the code plane as a wave-uniform pointer + 32-bit lane offset (saddr store)
    u8* __restrict__ rowbase = codes + ((u64)lz * n1 + y0) * p.pitch;
    const u32 xoff = (u32)x0;

    auto eval_row = [&](float y, float (&ra)[4], float (&rc)[4], u64& gt, u64& ge) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            ra[c] = mc_f(xs[c], y, zk);
            rc[c] = mc_f(xs[c], y, zk1);
        }
        // per-lane uniformity on the vector unit.  fmax/fmin skip NaN operands, which is right
        // for "some sample > iso" (NaN > iso is false, marching.cpp:498) but not for "every
        // sample > iso": a NaN among samples that otherwise all exceed iso shows as a NaN sum.
        const float mx = max3f(max3f(ra[0], ra[1], ra[2]), max3f(ra[3], rc[0], rc[1]), __builtin_fmaxf(rc[2], rc[3]));
        const float mn = min3f(min3f(ra[0], ra[1], ra[2]), min3f(ra[3], rc[0], rc[1]), __builtin_fminf(rc[2], rc[3]));
        gt = __ballot(mx > iso);
        ge = __ballot(mn > iso);
#ifndef MC_FINITE  // MC_FINITE: the expression compiler proved f finite on the whole domain
        if (ge != 0ull) {
            const float sm = ((ra[0] + ra[1]) + (ra[2] + ra[3])) + ((rc[0] + rc[1]) + (rc[2] + rc[3]));
            ge &= ~__ballot(sm != sm);
        }
#endif
    };

    // Code stores go through a buffer descriptor over the valid bytes of ONE row, rebuilt per
    // step by advancing its (scalar) base: lanes beyond the end of the row are dropped by the
    // hardware range check instead of by exec-mask juggling, and the address needs no vector
    // arithmetic.  (The range check covers voffset + soffset, so the row cannot advance through
    // soffset: measured -- such stores are dropped from the second row on.)
    const int rowbytes = (int)(((u32)n1 + 3u) & ~3u);

    auto step = [&](int j, float (&la)[4], float (&lc)[4], float (&ua)[4], float (&uc)[4]) {
        u64 gtNew, geNew;
        eval_row(readlane_f(yv, j + 1), ua, uc, gtNew, geNew);
        const u64 anyOwn = gtNew | gtPrev, allOwn = geNew & gePrev;
        const bool none = anyOwn == 0ull && ((ENone >> j) & 1ull);
        const bool full = allOwn == ~0ull && ((EFull >> j) & 1ull);
        if (__builtin_expect(none || full, 1)) {
#ifdef MC_STORE_GLOBAL
            if (x0 < n1) *(u32*)(rowbase + xoff) = full ? vmask : 0u;
#else
            __builtin_amdgcn_raw_buffer_store_b32(full ? vmask : 0u,
                                                  __builtin_amdgcn_make_buffer_rsrc(rowbase, 0, rowbytes, 0x00020000), xoff, 0, 0);
#endif
        } else {
            // per-lane classification: the lane's x+4 neighbour column is lane+1's sample 0
            // (lane 63: column E).  The empty asm pins the four compares to this branch.
            float l0 = la[0], l1 = lc[0], u0 = ua[0], u1 = uc[0];
            asm volatile("" : "+v"(l0), "+v"(l1), "+v"(u0), "+v"(u1));
            const u64 n0 = __ballot(l0 > iso), n1m = __ballot(l1 > iso), n2 = __ballot(u0 > iso), n3 = __ballot(u1 > iso);
            u64 topAny = (~ENone >> j) << 63, topAll = (EFull >> j) << 63;
            asm("" : "+s"(topAny), "+s"(topAll));  // keep the halves apart (no 64-bit funnel shift on the SALU)
            const u64 nbAny = ((n0 | n1m | n2 | n3) >> 1) | topAny;
            const u64 nbAll = ((n0 & n1m & n2 & n3) >> 1) | topAll;
            const u64 laneAll = allOwn & nbAll;
            const u64 mixedL = (anyOwn | nbAny) & ~laneAll;  // lanes with corners on both sides of iso
            const u32 dw = __builtin_amdgcn_inverse_ballot_w64(laneAll) ? vmask : 0u;
#ifdef MC_STORE_GLOBAL
            if (!__builtin_amdgcn_inverse_ballot_w64(mixedL) && x0 < n1) *(u32*)(rowbase + xoff) = dw;
#else
            if (!__builtin_amdgcn_inverse_ballot_w64(mixedL))
                __builtin_amdgcn_raw_buffer_store_b32(dw, __builtin_amdgcn_make_buffer_rsrc(rowbase, 0, rowbytes, 0x00020000),
                                                      xoff, 0, 0);
#endif
            if (mixedL) {
                const u32 cnt = (u32)__builtin_popcountll(mixedL);
                if (nent + cnt > MC_ENT_CAP) {
                    mc_record_pass(p, tc, s_lut, ent_pos, seg_cnt, nent, codes, recs, carry_j, carry_val);
                    nent = 0;
                }
                if (__builtin_amdgcn_inverse_ballot_w64(mixedL))
                    ent_pos[nent + mask_rank(mixedL)] = (unsigned short)((j << 6) | lane);
                nent += cnt;
            }
        }
        rowbase += p.pitch;
        gtPrev = gtNew;
        gePrev = geNew;
    };

    eval_row(readlane_f(yv, 0), r0a, r0c, gtPrev, gePrev);
    int j = 0;
    for (; j + 1 < ny; j += 2) {
        step(j, r0a, r0c, r1a, r1c);
        step(j + 1, r1a, r1c, r0a, r0c);
    }
    if (j < ny) step(j, r0a, r0c, r1a, r1c);

    if (nent) mc_record_pass(p, tc, s_lut, ent_pos, seg_cnt, nent, codes, recs, carry_j, carry_val);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (lane < ny) segcnt[tc.seg0 + (u64)lane * p.nchunk] = seg_cnt[lane];
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
__device__ __forceinline__ McVert mc_vertex(const McParams& p, const u64* s_row, const u8* s_edge, int row, int slot,
                                            int ix, int iy, int iz) {
    const int edge = (int)((s_row[row] >> (4 * slot)) & 0xF);
    const int ec = s_edge[edge];
    const int v1 = ec & 0xF, v2 = ec >> 4;
    const float xs = p.axis[ix + cx_bit(v1)], xe = p.axis[ix + cx_bit(v2)];
    const float ys = p.axis[iy + cy_bit(v1)], ye = p.axis[iy + cy_bit(v2)];
    const float zs = p.axis[iz + cz_bit(v1)], ze = p.axis[iz + cz_bit(v2)];
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
extern "C" __global__ __launch_bounds__(256) void mc_emit(const McParams* __restrict__ P, const u32* __restrict__ recs,
                                                           const uint2* __restrict__ segoff, float* __restrict__ verts) {
    __shared__ u64 s_row[256];
    __shared__ u8 s_edge[16];
    __shared__ u32 s_list[4][MC_LIST_CAP];
    __shared__ u32 s_seg[4][64];
    __shared__ u32 s_act[4][66];
    __shared__ u32 s_tri[4][64];
    s_row[threadIdx.x] = c_tri_row[threadIdx.x];
    if (threadIdx.x < 12) s_edge[threadIdx.x] = c_edge_corner[threadIdx.x];
    __syncthreads();

    const McParams p = *P;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // make wave-uniformity visible
    const u32 group = blockIdx.x * 4u + (u32)w;
    const u32 seg_first = group * 64u;
    if (seg_first >= p.nseg) return;
    const u32 seg = seg_first + (u32)lane;
    const uint2 o0 = segoff[min(seg, p.nseg)];
    const uint2 o1 = segoff[min(seg + 1u, p.nseg)];
    const u32 act_base = (u32)__builtin_amdgcn_readfirstlane((int)o0.y);
    const u32 nrec = (u32)__builtin_amdgcn_readlane((int)o1.y, 63) - act_base;  // records of the group
    if (nrec == 0u) return;

    u32* list = s_list[w];
    u32* segrec = s_seg[w];
    u32* actoff = s_act[w];
    u32* trioff = s_tri[w];
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
        if (lane == 63) actoff[64] = o1.y - act_base;
    }
    const float h = 0.5f * p.step;
    const bool want_normals = (p.flags & 1u) != 0u;

    u32 nlist = 0;                                             // triangles staged
    u32 listbase = (u32)__builtin_amdgcn_readfirstlane((int)o0.x);  // global index of list[0]

    // drains the staged triangles: one lane per vertex
    auto flush = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const u32 nverts = 3u * nlist;
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
                const McVert q = mc_vertex(p, s_row, s_edge, row, 3 * t + k, ix, iy, iz);
                float nx = 0.0f, ny = 0.0f, nz = 0.0f;
                if (want_normals) {
                    // n = g/|g|, g = central difference of F at the vertex, h = step/2
                    const float gx = mc_F(p, q.x + h, q.y, q.z) - mc_F(p, q.x - h, q.y, q.z);
                    const float gy = mc_F(p, q.x, q.y + h, q.z) - mc_F(p, q.x, q.y - h, q.z);
                    const float gz = mc_F(p, q.x, q.y, q.z + h) - mc_F(p, q.x, q.y, q.z - h);
                    const float len = __builtin_sqrtf((gx * gx + gy * gy) + gz * gz);
                    if (len > 0.0f && !__builtin_isinf(len)) {
                        nx = gx / len;
                        ny = gy / len;
                        nz = gz / len;
                    } else {  // degenerate gradient: the triangle's own normal cross(B-A, C-A)
                        const McVert a = mc_vertex(p, s_row, s_edge, row, 3 * t + 0, ix, iy, iz);
                        const McVert b = mc_vertex(p, s_row, s_edge, row, 3 * t + 1, ix, iy, iz);
                        const McVert c = mc_vertex(p, s_row, s_edge, row, 3 * t + 2, ix, iy, iz);
                        const float e1x = b.x - a.x, e1y = b.y - a.y, e1z = b.z - a.z;
                        const float e2x = c.x - a.x, e2y = c.y - a.y, e2z = c.z - a.z;
                        const float cxn = e1y * e2z - e1z * e2y;
                        const float cyn = e1z * e2x - e1x * e2z;
                        const float czn = e1x * e2y - e1y * e2x;
                        const float l = __builtin_sqrtf((cxn * cxn + cyn * cyn) + czn * czn);
                        if (l > 0.0f && !__builtin_isinf(l)) {
                            nx = cxn / l;
                            ny = cyn / l;
                            nz = czn / l;
                        }
                    }
                }
                const u64 gtri = (u64)listbase + tri;
                if (gtri < p.cap_tris) {
                    float2* o = (float2*)(verts + (gtri * 3ull + (u64)k) * 6ull);
                    o[0] = make_float2(q.x, q.y);
                    o[1] = make_float2(q.z, nx);
                    o[2] = make_float2(ny, nz);
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
            rec = recs[((u64)seg_first + lo) * MC_SEG + (r - actoff[lo])];
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

// =============================================================== evaluate points
// Evaluator::evaluate(x,y,z) (evaluator.cpp:53) for a batch of points: out[i] = f(xyz[3i..3i+2]).
extern "C" __global__ __launch_bounds__(256) void mc_eval(const float* __restrict__ xyz, float* __restrict__ out, u64 n) {
    const u64 i = (u64)blockIdx.x * 256ull + threadIdx.x;
    if (i < n) out[i] = mc_f(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]);
}
