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

