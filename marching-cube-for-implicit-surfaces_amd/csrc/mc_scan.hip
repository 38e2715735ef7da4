// mc_scan.hip -- equation-independent gfx950 kernels: the triangle-count scan (K2, built on
// hipcub block primitives) and two small packing kernels.  Compiled ahead of time by build.py
// into a code object (hipcc --genco) that is embedded in libmc_hip.so and loaded with
// hipModuleLoadData, like the hiprtc-specialised kernels of mc_kernels.hip.
//
// K2 turns the per-GROUP sums (group = 64 consecutive segments; mc_classify adds every tile's
// counts into them with one 64-bit atomic per non-empty segment) into exclusive group offsets.
// The emit kernel finishes the prefix inside its group with a wavefront scan, so the output
// order stays the reference's sweep order (Source/marching.cpp:375-383: z, then y, then x; then
// table order inside a cell) while the scan touches 64x fewer elements than a per-segment one.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

typedef unsigned long long u64;
typedef unsigned int u32;
typedef unsigned char u8;

#define SCAN_ITEMS 8
#define SCAN_BLOCK 256
#define SCAN_TILE (SCAN_ITEMS * SCAN_BLOCK)

// grpsum[i] = triangles | active cells << 32 per group.  Three small launches turn it into
// grpoff[i] = exclusive {triangle, active} offsets (u32) plus 64-bit totals.
extern "C" __global__ __launch_bounds__(SCAN_BLOCK) void mc_scan_reduce(const u64* __restrict__ segcnt, u32 nseg,
                                                             uint2* __restrict__ blocksum) {
    typedef hipcub::BlockReduce<u64, SCAN_BLOCK> Reduce;
    __shared__ typename Reduce::TempStorage tmp;
    const u32 base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
    u64 acc = 0;  // tris in the low half, active cells in the high half
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i)
        if (base + i < nseg) acc += segcnt[base + i];
    const u64 tot = Reduce(tmp).Sum(acc);
    if (threadIdx.x == 0) blocksum[blockIdx.x] = make_uint2((u32)tot, (u32)(tot >> 32));
}

extern "C" __global__ __launch_bounds__(SCAN_BLOCK) void mc_scan_blocks(const uint2* __restrict__ blocksum, u32 nblocks,
                                                             ulonglong2* __restrict__ blockoff, u64* __restrict__ totals,
                                                             uint2* __restrict__ segoff_last) {
    typedef hipcub::BlockScan<u64, SCAN_BLOCK> Scan;
    __shared__ typename Scan::TempStorage tmp;
    u64 carry_t = 0, carry_a = 0;
    for (u32 b0 = 0; b0 < nblocks; b0 += SCAN_BLOCK) {
        const u32 b = b0 + threadIdx.x;
        const uint2 v = b < nblocks ? blocksum[b] : make_uint2(0, 0);
        u64 et, ea, tt, ta;
        Scan(tmp).ExclusiveSum((u64)v.x, et, tt);
        __syncthreads();
        Scan(tmp).ExclusiveSum((u64)v.y, ea, ta);
        __syncthreads();
        if (b < nblocks) blockoff[b] = make_ulonglong2(carry_t + et, carry_a + ea);
        carry_t += tt;
        carry_a += ta;
    }
    if (threadIdx.x == 0) {
        totals[0] = carry_t;
        totals[1] = carry_a;
        *segoff_last = make_uint2((u32)carry_t, (u32)carry_a);
    }
}

extern "C" __global__ __launch_bounds__(SCAN_BLOCK) void mc_scan_final(const u64* __restrict__ segcnt, u32 nseg,
                                                            const ulonglong2* __restrict__ blockoff,
                                                            uint2* __restrict__ segoff) {
    typedef hipcub::BlockScan<u64, SCAN_BLOCK> Scan;
    __shared__ typename Scan::TempStorage tmp;
    const u32 base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
    u64 item[SCAN_ITEMS];
    u64 sum = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        u64 v = 0;
        if (base + i < nseg) v = segcnt[base + i];
        item[i] = sum;  // exclusive within the thread
        sum += v;
    }
    u64 excl;
    Scan(tmp).ExclusiveSum(sum, excl);
    const ulonglong2 bo = blockoff[blockIdx.x];
    const u64 bt = bo.x + (excl & 0xFFFFFFFFull);
    const u64 ba = bo.y + (excl >> 32);
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i)
        if (base + i < nseg)
            segoff[base + i] = make_uint2((u32)(bt + (item[i] & 0xFFFFFFFFull)), (u32)(ba + (item[i] >> 32)));
}

// positions only: verts[T*3][6] -> soup[T*3][3]
extern "C" __global__ __launch_bounds__(256) void mc_pack_soup(const float* __restrict__ verts, float* __restrict__ soup, u64 nverts) {
    const u64 i = (u64)blockIdx.x * 256ull + threadIdx.x;
    if (i < nverts) {
        soup[3 * i + 0] = verts[6 * i + 0];
        soup[3 * i + 1] = verts[6 * i + 1];
        soup[3 * i + 2] = verts[6 * i + 2];
    }
}

// pitched main plane (+ the tail plane: one dword per row holding cells main_cells..n1-1) -> compact rows of n1 bytes
extern "C" __global__ __launch_bounds__(256) void mc_pack_codes(const u8* __restrict__ codes, const u32* __restrict__ tail,
                                                     u64 pitch, int n1, int main_cells, u64 nrows, u8* __restrict__ out) {
    const u64 i = (u64)blockIdx.x * 256ull + threadIdx.x;
    const u64 total = nrows * (u64)n1;
    if (i < total) {
        const u64 r = i / (u64)n1;
        const int x = (int)(i - r * (u64)n1);
        out[i] = x < main_cells ? codes[r * pitch + (u64)x] : (u8)(tail[r] >> (8 * (x - main_cells)));
    }
}



// =============================================================== seed mode (Marching::seed_mode, marching.cpp:42-137, :310-331)
// The reference walks breadth-first from the cell that contains the seed to the face neighbours across every face
// that carries an intersection.  Here the dense sweep has already classified every cell, so the walk runs over the
// code volume: frontier in, frontier out, one launch per breadth-first level; mc_seed_filter then removes the
// triangles of unvisited cells from the records before the scan and mc_emit run.  Cell id = (z*n1 + y)*n1 + x.
#include "../../include/mc_tables_data.h"

__device__ __forceinline__ u32 seed_code(const u8* codes, const u32* tail, u64 pitch, int n1, int main_cells, int x, int y, int z) {
    const u64 row = (u64)z * (u64)n1 + (u64)y;
    return x < main_cells ? (u32)codes[row * pitch + (u64)x] : (tail[row] >> (8 * (x - main_cells))) & 0xFFu;
}

extern "C" __global__ __launch_bounds__(64) void mc_seed_init(u32* __restrict__ visited, u32* __restrict__ frontier, u32* __restrict__ counts,
                                                    u32 cell) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        visited[cell >> 5] = 1u << (cell & 31u);
        frontier[0] = cell;
        counts[0] = 1u;  // level 0 holds the seed's cell
        counts[1] = 0u;
        counts[2] = 0u;
    }
}

// One breadth-first level.  The frontier sizes live on the device (counts[level % 3]) so that the host can enqueue many
// levels without reading anything back: this launch reads counts[lvl % 3], appends to counts[(lvl+1) % 3] and zeroes
// counts[(lvl+2) % 3] for the launch after it; a fixed grid strides over the frontier, an empty frontier costs a launch.
// imax: the largest cell index a move may reach (marching.cpp:84-86: x0 + 0.5*step <= 1); the lower bound is index 0
extern "C" __global__ __launch_bounds__(256) void mc_seed_expand(const u8* __restrict__ codes, const u32* __restrict__ tail, u64 pitch, int n1,
                                                       int main_cells, int imax, u32* __restrict__ visited,
                                                       const u32* __restrict__ fin, u32* __restrict__ fout,
                                                       u32* __restrict__ counts, u32 lvl, u32 cap) {
    const u32 nin = min(counts[lvl % 3u], cap);
    u32* __restrict__ nout = counts + (lvl + 1u) % 3u;
    if (blockIdx.x == 0 && threadIdx.x == 0) counts[(lvl + 2u) % 3u] = 0u;
    for (u32 i = blockIdx.x * 256u + threadIdx.x; i < nin; i += gridDim.x * 256u) {
    const u32 cell = fin[i];
    const int x = (int)(cell % (u32)n1), y = (int)((cell / (u32)n1) % (u32)n1), z = (int)(cell / ((u32)n1 * (u32)n1));
    const u32 code = seed_code(codes, tail, pitch, n1, main_cells, x, y, z);
    if (code == 0u || code == 255u) continue;  // no surface in this cell: its edge list is empty (marching.cpp:508-510)
    constexpr unsigned short kfc[6] = MC_FACE_CORNER_INIT;  // marching_lookup.h:25-32
#pragma unroll
    for (int f = 0; f < 6; ++f) {
        // the face carries an intersection iff its four corners are not all on one side (marching.cpp:62-69)
        int ones = 0, bx = 0, by = 0, bz = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int v = (kfc[f] >> (4 * k)) & 0xF;
            ones += (code >> v) & 1u;
            bx += (0x66 >> v) & 1;
            by += (0xCC >> v) & 1;
            bz += v >> 2;
        }
        if (ones == 0 || ones == 4) continue;
        // marching_lookup.h:43-50 cube_face_normal = the axis on which the face's corners agree
        const int nx = x + (bx == 4 ? 1 : bx == 0 ? -1 : 0), ny = y + (by == 4 ? 1 : by == 0 ? -1 : 0),
                  nz = z + (bz == 4 ? 1 : bz == 0 ? -1 : 0);
        if (nx < 0 || ny < 0 || nz < 0 || nx > imax || ny > imax || nz > imax) continue;  // marching.cpp:84-86
        const u32 nc = ((u32)nz * (u32)n1 + (u32)ny) * (u32)n1 + (u32)nx;
        const u32 bit = 1u << (nc & 31u);
        if (!(atomicOr(&visited[nc >> 5], bit) & bit)) {  // marching.cpp:89-97: queue it unless it is in the set already
            const u32 slot = atomicAdd(nout, 1u);
            if (slot < cap) fout[slot] = nc;
        }
    }
    }
}

// one lane per segment: records of unvisited cells lose their triangles; the segment's triangle prefix, its
// counts and the group sums are rebuilt
extern "C" __global__ __launch_bounds__(256) void mc_seed_filter(u32* __restrict__ recs, uint2* __restrict__ segcb, const u32* __restrict__ visited,
                                                       u32 nseg, int nchunk, int n1, u64* __restrict__ grpsum,
                                                       const u32* __restrict__ overflow) {
    const u32 seg = blockIdx.x * 256u + threadIdx.x;
    if (seg >= nseg || overflow[0] != 0u) return;
    const uint2 cb = segcb[seg];
    const u32 act = cb.x >> 16;
    if (act == 0u) return;
    const u32 rowidx = seg / (u32)nchunk, ch = seg - rowidx * (u32)nchunk;
    const u32 rowcell = rowidx * (u32)n1 + ch * 256u;  // id of the segment's first cell (rowidx = z*n1 + y)
    u32 tris = 0u;
    for (u32 k = 0; k < act; ++k) {
        u32 r = recs[cb.y + k];
        const u32 cell = rowcell + (r & 0xFFu);
        u32 nt = (r >> 17) & 7u;
        if (!((visited[cell >> 5] >> (cell & 31u)) & 1u)) nt = 0u;
        r = (r & 0x1FFFFu) | (nt << 17) | (tris << 20);
        recs[cb.y + k] = r;
        tris += nt;
    }
    segcb[seg] = make_uint2(tris | (act << 16), cb.y);
    atomicAdd(&grpsum[seg >> 6], (u64)tris | ((u64)act << 32));
}
