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

// grpsum[i] = triangles | active cells << 32 per group  ->  grpoff[i] = exclusive {triangle, active} offsets (u32) plus
// 64-bit totals, in ONE launch: a single-pass scan with decoupled look-back.  (Three launches -- reduce, scan of the block
// sums, final -- cost 13 us per sweep at 1025^3, mostly launch boundaries: a tenth of a slab's sweep once the grid is
// sharded over 8 GPUs.)
//
// Every workgroup takes a ticket (so that all workgroups with a lower number have started, whatever the dispatch order),
// reduces its tile of SCAN_TILE groups, publishes the tile's AGGREGATE, looks back over its predecessors -- one wave, 64
// predecessors per step -- until it meets one whose INCLUSIVE prefix is known, publishes its own inclusive prefix and
// scans its tile.  A status entry is two 8-byte words {state << 62 | triangles, state << 62 | active cells}, each
// written by one agent-scope store (the value travels with its tag: no fence, MI355X_MICROARCH.md "granule"); a reader
// accepts a pair only when both words carry the same non-zero state (the words are written in order, first to second,
// and read in the same order, so a torn pair shows two different states).  `status` and `ticket` are zero when the kernel
// starts (the host clears the control block once; every sweep leaves it clean, see below).  Spins are bounded: a workgroup
// that gives up raises totals[3], mirrored into the host's pinned totals like the counts, and the host checks that word
// after EVERY sweep and every batch of replays (mc_runtime.hip: mc_march, graph_collect): a give-up is MC_ERR_HIP, never
// silent offsets, and the context clears its control block before its next sweep.  (It cannot happen while every
// workgroup with a lower ticket is running, which the ticket order guarantees.)
#define SCAN_AGG 1ull
#define SCAN_INC 2ull
#define SCAN_SPIN_MAX (1u << 24)
//
// The kernel also leaves the control block CLEAN for the next sweep, so that no memset node is needed between sweeps (two
// fill kernels of ~4.5 us each per sweep, a tenth of a slab's sweep on 8 GPUs): every group sum is zeroed right after
// it has been read; mc_classify's record cursors are zeroed by slices; its overflow word is moved into totals[2] (where
// the kernels behind the scan read it) and cleared; and the workgroup that finishes LAST -- a second counter tells --
// clears the status words, the ticket and that counter: by then no workgroup looks at them any more.
extern "C" __global__ __launch_bounds__(SCAN_BLOCK) void mc_scan_onepass(u64* __restrict__ grpsum, u32 n, uint2* __restrict__ grpoff,
                                                              u64* __restrict__ totals, u64* status, u32* ticket,
                                                              u32* __restrict__ cursors, u32 cursor_words, u64* __restrict__ totals_host) {
    typedef hipcub::BlockScan<u64, SCAN_BLOCK> Scan;
    __shared__ typename Scan::TempStorage tmp;
    __shared__ u32 s_bid;
    __shared__ u64 s_pre[2];
    if (gridDim.x == 1u) {
        // a list of one tile (small grids): no ticket, no status words, no look-back, no second counter
        u64 carry_t = 0, carry_a = 0;
        for (u32 i = 1u + threadIdx.x; i < cursor_words; i += SCAN_BLOCK) cursors[i] = 0u;
        for (u32 t0 = 0; t0 < n; t0 += SCAN_TILE) {
            const u32 base = t0 + threadIdx.x * SCAN_ITEMS;
            u64 item[SCAN_ITEMS];
            u64 sum = 0;
#pragma unroll
            for (int i = 0; i < SCAN_ITEMS; ++i) {
                u64 v = 0;
                if (base + i < n) {
                    v = grpsum[base + i];
                    grpsum[base + i] = 0ull;
                }
                item[i] = sum;
                sum += v;
            }
            u64 excl, agg;
            Scan(tmp).ExclusiveSum(sum, excl, agg);
            __syncthreads();
            const u64 bt = carry_t + (excl & 0xFFFFFFFFull), ba = carry_a + (excl >> 32);
#pragma unroll
            for (int i = 0; i < SCAN_ITEMS; ++i)
                if (base + i < n) grpoff[base + i] = make_uint2((u32)(bt + (item[i] & 0xFFFFFFFFull)), (u32)(ba + (item[i] >> 32)));
            carry_t += agg & 0xFFFFFFFFull;
            carry_a += agg >> 32;
        }
        if (threadIdx.x == 0) {
            u64 ovf = 0ull;
            if (cursors) {
                ovf = (u64)cursors[0];
                cursors[0] = 0u;
            }
            totals[0] = carry_t;
            totals[1] = carry_a;
            totals[2] = ovf;
            grpoff[n] = make_uint2((u32)carry_t, (u32)carry_a);
            if (totals_host) {
                totals_host[0] = carry_t;
                totals_host[1] = carry_a;
                totals_host[2] = ovf;
            }
        }
        return;
    }
    if (threadIdx.x == 0) s_bid = atomicAdd(ticket, 1u);
    __syncthreads();
    const u32 b = s_bid;
    const u32 base = b * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
    u64 item[SCAN_ITEMS];
    u64 sum = 0;  // inside one tile both halves fit 32 bits: 2048 groups * 64 segments * (1280 triangles | 256 cells)
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        u64 v = 0;
        if (base + i < n) {
            v = grpsum[base + i];
            grpsum[base + i] = 0ull;  // clean for the next sweep's atomic adds
        }
        item[i] = sum;  // exclusive within the thread
        sum += v;
    }
    // mc_classify's record cursors (word 0, the overflow word, is handled by the last workgroup below)
    for (u32 i = 1u + b * SCAN_BLOCK + threadIdx.x; i < cursor_words; i += gridDim.x * SCAN_BLOCK) cursors[i] = 0u;
    u64 excl, agg;
    Scan(tmp).ExclusiveSum(sum, excl, agg);
    const u64 agg_t = agg & 0xFFFFFFFFull, agg_a = agg >> 32;
    if (threadIdx.x < 64) {  // wave 0: publish, look back, publish
        const int lane = (int)threadIdx.x;
        u64 pre_t = 0, pre_a = 0;
        if (b == 0) {
            if (lane == 0) {
                __hip_atomic_store(&status[0], (SCAN_INC << 62) | agg_t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&status[1], (SCAN_INC << 62) | agg_a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        } else {
            if (lane == 0) {
                __hip_atomic_store(&status[2 * b], (SCAN_AGG << 62) | agg_t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&status[2 * b + 1], (SCAN_AGG << 62) | agg_a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            int j0 = (int)b - 1;  // lane l looks at workgroup j0 - l
            bool done = false, failed = false;
            while (!done) {
                const int j = j0 - lane;
                u64 w0 = 0, w1 = 0;
                if (j >= 0) {
                    u32 spins = 0;
                    for (;;) {
                        w0 = __hip_atomic_load(&status[2 * j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        w1 = __hip_atomic_load(&status[2 * j + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if ((w0 >> 62) != 0ull && (w0 >> 62) == (w1 >> 62)) break;
                        if (++spins > SCAN_SPIN_MAX) {
                            failed = true;
                            w0 = w1 = SCAN_INC << 62;  // give up: stop the walk here
                            break;
                        }
                        __builtin_amdgcn_s_sleep(2);
                    }
                } else {
                    w0 = w1 = SCAN_INC << 62;  // before the first workgroup: an inclusive prefix of zero
                }
                const u64 incm = __ballot((w0 >> 62) == SCAN_INC);
                // sum the lanes up to and including the first one that holds an inclusive prefix
                const int stop = incm ? __builtin_ctzll(incm) : 63;
                const bool take = lane <= stop;
                u64 vt = take ? (w0 & 0x3FFFFFFFFFFFFFFFull) : 0ull, va = take ? (w1 & 0x3FFFFFFFFFFFFFFFull) : 0ull;
#pragma unroll
                for (int o = 32; o >= 1; o >>= 1) {
                    vt += __shfl_xor(vt, o, 64);
                    va += __shfl_xor(va, o, 64);
                }
                pre_t += vt;
                pre_a += va;
                done = incm != 0ull;
                j0 -= 64;
            }
            if (__ballot(failed) && lane == 0) {
                totals[3] = 1ull;
                if (totals_host) totals_host[3] = 1ull;
            }
            if (lane == 0) {
                __hip_atomic_store(&status[2 * b], (SCAN_INC << 62) | (pre_t + agg_t), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&status[2 * b + 1], (SCAN_INC << 62) | (pre_a + agg_a), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        if (lane == 0) {
            s_pre[0] = pre_t;
            s_pre[1] = pre_a;
            if (b == gridDim.x - 1u) {
                // totals: {triangles, active cells, mc_classify's "a record region overflowed" word, scan gave up};
                // mirrored into pinned host memory so that the host needs no copy node behind the sweep
                u64 ovf = 0ull;
                if (cursors) {
                    ovf = (u64)cursors[0];
                    cursors[0] = 0u;
                }
                totals[0] = pre_t + agg_t;
                totals[1] = pre_a + agg_a;
                totals[2] = ovf;
                grpoff[n] = make_uint2((u32)(pre_t + agg_t), (u32)(pre_a + agg_a));
                if (totals_host) {
                    totals_host[0] = pre_t + agg_t;
                    totals_host[1] = pre_a + agg_a;
                    totals_host[2] = ovf;
                }
            }
        }
    }
    __syncthreads();
    const u64 bt = s_pre[0] + (excl & 0xFFFFFFFFull);
    const u64 ba = s_pre[1] + (excl >> 32);
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i)
        if (base + i < n) grpoff[base + i] = make_uint2((u32)(bt + (item[i] & 0xFFFFFFFFull)), (u32)(ba + (item[i] >> 32)));
    // the workgroup that finishes last clears the scan's own state (its look-back is over, like everybody else's)
    __shared__ u32 s_last;
    __syncthreads();
    if (threadIdx.x == 0) s_last = atomicAdd(&ticket[1], 1u) == gridDim.x - 1u ? 1u : 0u;
    __syncthreads();
    if (s_last) {
        for (u32 i = threadIdx.x; i < 2u * gridDim.x; i += SCAN_BLOCK)
            __hip_atomic_store(&status[i], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (threadIdx.x == 0) {
            __hip_atomic_store(&ticket[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&ticket[1], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// MC_FLAG_SEAM: where the layers that are handed out begin and end inside the swept (ghost-extended) slab.  One wave per
// boundary: out[3 b .. 3 b + 2] = {triangles, active cells, indexed vertices} in front of segment bound[b] in sweep order --
// the group offsets of the scans plus the prefix inside the group (per-segment counts; for the vertices the owned-edge
// masks of the records in front).  grpvoff / recown may be null (no indexed mesh): vertices read 0.
extern "C" __global__ __launch_bounds__(64) void mc_bounds(const uint2* __restrict__ segcb, const uint2* __restrict__ grpoff,
                                                         const uint2* __restrict__ grpvoff, const u32* __restrict__ recown,
                                                         u32 nseg, u32 bound0, u32 bound1, u64* __restrict__ out) {
    const u32 s = blockIdx.x == 0u ? bound0 : bound1;
    const u32 ngroups = (nseg + 63u) / 64u;
    const u32 g = min(s / 64u, ngroups);       // s == nseg on a group boundary: the totals behind the last group
    const int lane = (int)threadIdx.x;
    const u32 seg = g * 64u + (u32)lane;
    u64 t = 0, a = 0, v = 0;
    if (g < ngroups && seg < s && seg < nseg) {
        const uint2 cb = segcb[seg];
        t = cb.x & 0xFFFFu;
        a = cb.x >> 16;
        if (recown)
            for (u32 i = 0; i < (cb.x >> 16); ++i) v += (u64)__builtin_popcount(recown[cb.y + i] & 0xFFFu);
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        t += __shfl_xor(t, o, 64);
        a += __shfl_xor(a, o, 64);
        v += __shfl_xor(v, o, 64);
    }
    if (lane == 0) {
        const uint2 go = grpoff[g];
        out[3u * blockIdx.x + 0u] = (u64)go.x + t;
        out[3u * blockIdx.x + 1u] = (u64)go.y + a;
        out[3u * blockIdx.x + 2u] = grpvoff ? (u64)grpvoff[g].x + v : 0ull;
    }
}

// tri_list += delta (mod 2^32): MC_FLAG_SEAM hands out indices relative to the slab's own first vertex (a vertex owned by the
// slab below reads as a "negative" number); mc_index_rebase adds the slab's offset in the whole grid's vertex_list
extern "C" __global__ __launch_bounds__(256) void mc_index_add(u32* __restrict__ tlist, u64 n, u32 delta) {
    const u64 i = (u64)blockIdx.x * 256ull + threadIdx.x;
    if (i < n) tlist[i] += delta;
}

// CalculateNormal (Source/normal.h:3-41), first half: the face normal of every triangle of the welded mesh, ONCE --
// glm::cross(B - A, C - A) from the welded positions, the reference's operation order (normal.h:17-20; this file is built
// with -ffp-contract=off like every other) -- so that mc_vnormal, which sums them per vertex in the reference's order, reads
// 16 bytes per triangle it visits instead of nine scattered floats (each triangle is visited ~3 times: round 4, 0.645 ->
// see DESIGN.md section 4 "Indexed mesh").  .w = 1: the triangle counts; 0: one of its corners lies outside the vertex list
// (a vertex owned by the layer below a ghost layer, or beyond a captured graph's capacity) -- such a triangle is skipped by
// the sum, as before.  total: the sweep's triangle count on the device (the scan's last offset).
extern "C" __global__ __launch_bounds__(256) void mc_tnormal(const u32* __restrict__ tlist, const float* __restrict__ vlist, float4* __restrict__ tnrm,
                                                   const uint2* __restrict__ total, u64 cap_tris, u64 nverts) {
    const u64 tri = (u64)blockIdx.x * 256ull + threadIdx.x;
    const u64 nt = (u64)total->x < cap_tris ? (u64)total->x : cap_tris;
    if (tri >= nt) return;
    // (the loads of a level pinned together -- left alone the compiler fetches the three indices one after the other, each
    // behind the range check of the one before: four memory round trips where two do)
    u32 i1 = tlist[3ull * tri], i2 = tlist[3ull * tri + 1], i3 = tlist[3ull * tri + 2];
    asm volatile("" : "+v"(i1), "+v"(i2), "+v"(i3));
    if (i1 >= nverts || i2 >= nverts || i3 >= nverts) {
        tnrm[tri] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        return;
    }
    float ax_ = vlist[3ull * i1], ay_ = vlist[3ull * i1 + 1], az_ = vlist[3ull * i1 + 2];
    float bx_ = vlist[3ull * i2], by_ = vlist[3ull * i2 + 1], bz_ = vlist[3ull * i2 + 2];
    float cx_ = vlist[3ull * i3], cy_ = vlist[3ull * i3 + 1], cz_ = vlist[3ull * i3 + 2];
    asm volatile("" : "+v"(ax_), "+v"(ay_), "+v"(az_), "+v"(bx_), "+v"(by_), "+v"(bz_), "+v"(cx_), "+v"(cy_), "+v"(cz_));
    const float bax = bx_ - ax_, bay = by_ - ay_, baz = bz_ - az_;
    const float cax = cx_ - ax_, cay = cy_ - ay_, caz = cz_ - az_;
    // glm::cross(x, y) = (x.y*y.z - y.y*x.z, x.z*y.x - y.z*x.x, x.x*y.y - y.x*x.y)
    tnrm[tri] = make_float4(bay * caz - cay * baz, baz * cax - caz * bax, bax * cay - cax * bay, 1.0f);
}

// one half of every vertex: verts[T*3][6] -> soup[T*3][3]; half = 0: positions, 3: normals
extern "C" __global__ __launch_bounds__(256) void mc_pack_soup(const float* __restrict__ verts, float* __restrict__ soup, u64 nverts, u32 half) {
    const u64 i = (u64)blockIdx.x * 256ull + threadIdx.x;
    if (i < nverts) {
        soup[3 * i + 0] = verts[6 * i + half + 0];
        soup[3 * i + 1] = verts[6 * i + half + 1];
        soup[3 * i + 2] = verts[6 * i + half + 2];
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
// The reference walks breadth-first from the cell that contains the seed to the face neighbours across every face that
// carries an intersection, and emits the cells it reaches.  What it reaches is a CONNECTED COMPONENT of the graph whose
// nodes are the cells with surface and whose edges are the faces with corners on both sides of iso (such a face makes both
// its cells surface cells, so the relation is symmetric) -- and the dense sweep has already listed exactly those cells: the
// records.  So instead of a walk (whose ~2 N dependent hops cost 14 ms at 1025^3, whatever the width of the machine) the
// component is LABELLED: a lock-free union-find over record indices (mc_cc_link: every record hooks itself to its +x, +y,
// +z neighbours across mixed faces; roots only ever decrease), a flatten pass, the seed cell's root (mc_cc_seed), and
// mc_seed_filter then takes the triangles of every other component's records away before the scan and the emit kernel run.
// Order: the reference emits in visitation order; here the cells come out in sweep order (set-equal).
//
// Bounds (marching.cpp:84-86): a move may only reach a cell whose three indices are all <= imax (x0 + 0.5*step <= 1); a
// cell beyond that is never entered, so it links to nothing -- it can only be emitted as the seed's own cell, whose
// in-bounds neighbours mc_cc_seed then adds as further roots.

// path halving; concurrent hooks only ever replace a parent by one of its ancestors
__device__ __forceinline__ u32 cc_find(u32* parent, u32 r) {
    for (;;) {
        const u32 p = __hip_atomic_load(&parent[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (p == r) return r;
        const u32 gp = __hip_atomic_load(&parent[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (gp == p) return p;
        __hip_atomic_store(&parent[r], gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        r = gp;
    }
}
// the same without writing: the flatten pass must be the only writer of what it publishes -- a concurrent path-halving
// store (an ancestor read a moment ago) could land on top of a root the record's own lane has just stored, and the filter
// would then take the record for a member of another component (seen: a few cells of 4.9 M dropped, differently every run)
__device__ __forceinline__ u32 cc_find_ro(const u32* parent, u32 r) {
    for (;;) {
        const u32 p = __hip_atomic_load(&parent[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (p == r) return r;
        r = p;
    }
}
// the larger root is hooked under the smaller one; a failed CAS means somebody else hooked it first: find again
__device__ __forceinline__ void cc_union(u32* parent, u32 a, u32 b) {
    for (;;) {
        a = cc_find(parent, a);
        b = cc_find(parent, b);
        if (a == b) return;
        if (a < b) {
            const u32 t = a;
            a = b;
            b = t;
        }
        if (atomicCAS(&parent[a], a, b) == a) return;
    }
}
// Most unions the link pass still attempts join two records that are in one set already.  That can be seen with PLAIN loads
// (L1 / L2 hits instead of device-scope loads at ~1 us apiece): whatever such a load returns is an ancestor -- an out-of-date
// one at worst -- and two records with a common ancestor are in one set, for good (sets only merge).  Only when the
// ancestors differ is the exact, device-scope union made, starting from them.
__device__ __forceinline__ void cc_union_fast(u32* parent, u32 a, u32 b) {
    const u32* pl = parent;
    for (;;) {
        const u32 pa = pl[a];
        if (pa == a) break;
        a = pa;
    }
    for (;;) {
        const u32 pb = pl[b];
        if (pb == b) break;
        b = pb;
    }
    if (a != b) cc_union(parent, a, b);
}
// index of the record of cell `cellx` of segment seg (records of a segment ascend in cellx), ~0u when it has none
__device__ __forceinline__ u32 cc_lookup(const u32* __restrict__ recs, const uint2* __restrict__ segcb, u32 seg, u32 cellx) {
    const uint2 cb = segcb[seg];
    const u32 act = cb.x >> 16;
    u32 lo = 0u, hi = act;
    while (lo < hi) {
        const u32 mid = (lo + hi) >> 1;
        if ((recs[cb.y + mid] & 0xFFu) < cellx) lo = mid + 1u;
        else hi = mid;
    }
    return (lo < act && (recs[cb.y + lo] & 0xFFu) == cellx) ? cb.y + lo : 0xFFFFFFFFu;
}
// corners of the +x / +y / +z face of a cell (corner i = bit i of the cube code, marching.cpp:471-472): the face carries an
// intersection iff its corners are not all on one side (marching.cpp:62-69)
#define CC_FACE_PX 0x66u
#define CC_FACE_PY 0xCCu
#define CC_FACE_PZ 0xF0u
__device__ __forceinline__ bool cc_mixed(u32 code, u32 face) { return (code & face) != 0u && (code & face) != face; }

// two look-ups side by side (the +y and the +z neighbour): their binary searches are chains of dependent loads, and a
// wave has nothing else to do meanwhile -- issued together they cost one chain, not two
__device__ __forceinline__ void cc_lookup2(const u32* __restrict__ recs, const uint2* __restrict__ segcb, bool wa, u32 sega, bool wb, u32 segb,
                                           u32 cellx, u32& ra, u32& rb) {
    // (every level's loads pinned together: the compiler otherwise puts each load behind the condition that guards it and
    // the two look-ups run one after the other, load by load -- seen in the ISA: 33 loads, 33 waits in mc_cc_link)
    uint2 ca = segcb[wa ? sega : 0u], cb = segcb[wb ? segb : 0u];
    asm volatile("" : "+v"(ca.x), "+v"(ca.y), "+v"(cb.x), "+v"(cb.y));
    if (!wa) ca = make_uint2(0u, 0u);
    if (!wb) cb = make_uint2(0u, 0u);
    const u32 acta = ca.x >> 16, actb = cb.x >> 16;
    if (acta <= 4u && actb <= 4u) {
        // the common case (a segment of a curved surface holds 1-3 records): the first four records of both segments at
        // once, the one wanted picked -- one round trip instead of a binary search's two or three (the record buffer has
        // slack behind its last record: mc_runtime.hip MC_REC_SLACK_BYTES)
        u32 a0 = recs[ca.y], a1 = recs[ca.y + 1u], a2 = recs[ca.y + 2u], a3 = recs[ca.y + 3u];
        u32 b0 = recs[cb.y], b1 = recs[cb.y + 1u], b2 = recs[cb.y + 2u], b3 = recs[cb.y + 3u];
        asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3));
        ra = (acta > 0u && (a0 & 0xFFu) == cellx) ? ca.y : (acta > 1u && (a1 & 0xFFu) == cellx) ? ca.y + 1u
             : (acta > 2u && (a2 & 0xFFu) == cellx) ? ca.y + 2u : (acta > 3u && (a3 & 0xFFu) == cellx) ? ca.y + 3u : 0xFFFFFFFFu;
        rb = (actb > 0u && (b0 & 0xFFu) == cellx) ? cb.y : (actb > 1u && (b1 & 0xFFu) == cellx) ? cb.y + 1u
             : (actb > 2u && (b2 & 0xFFu) == cellx) ? cb.y + 2u : (actb > 3u && (b3 & 0xFFu) == cellx) ? cb.y + 3u : 0xFFFFFFFFu;
        return;
    }
    u32 loa = 0u, hia = acta, lob = 0u, hib = actb;
    while (loa < hia || lob < hib) {
        const u32 ma = (loa + hia) >> 1, mb = (lob + hib) >> 1;
        const u32 va = loa < hia ? recs[ca.y + ma] & 0xFFu : 0u, vb = lob < hib ? recs[cb.y + mb] & 0xFFu : 0u;
        if (loa < hia) {
            if (va < cellx) loa = ma + 1u;
            else hia = ma;
        }
        if (lob < hib) {
            if (vb < cellx) lob = mb + 1u;
            else hib = mb;
        }
    }
    const u32 fa = loa < acta ? recs[ca.y + loa] & 0xFFu : 0x100u, fb = lob < actb ? recs[cb.y + lob] & 0xFFu : 0x100u;
    ra = fa == cellx ? ca.y + loa : 0xFFFFFFFFu;
    rb = fb == cellx ? cb.y + lob : 0xFFFFFFFFu;
}

// ---- the same union-find on a wave's window of records, in LDS (indices = positions in the window)
__device__ __forceinline__ u32 lds_find(u32* lab, u32 i) {
    for (;;) {
        const u32 p = __hip_atomic_load(&lab[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (p == i) return i;
        const u32 gp = __hip_atomic_load(&lab[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (gp == p) return p;
        __hip_atomic_store(&lab[i], gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        i = gp;
    }
}
__device__ __forceinline__ void lds_union(u32* lab, u32 a, u32 b) {
    for (;;) {
        a = lds_find(lab, a);
        b = lds_find(lab, b);
        if (a == b) return;
        if (a < b) {
            const u32 t = a;
            a = b;
            b = t;
        }
        if (atomicCAS(&lab[a], a, b) == a) return;
    }
}

#define CC_WIN 512u  // records of a group labelled together in LDS (a group of the 1025^3 sphere has ~60, of the gyroid ~250; a power of two)

// what the two kernels below share: the group's segments (lane = segment), their record offsets inside the group
// (s_off[0..64]) and first record indices, in LDS; returns the group's record count
__device__ __forceinline__ u32 cc_group(const uint2* __restrict__ segcb, u32 seg0, u32 nseg, u32 lane, u32* s_off, u32* s_first) {
    u32 act = 0u, first = 0u;
    if (seg0 + lane < nseg) {
        const uint2 cb = segcb[seg0 + lane];
        act = cb.x >> 16;
        first = cb.y;
    }
    u32 incl = act;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const u32 t = (u32)__shfl_up((int)incl, o, 64);
        if (lane >= (u32)o) incl += t;
    }
    s_off[lane + 1u] = incl;
    if (lane == 0u) s_off[0] = 0u;
    s_first[lane] = first;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    return (u32)__builtin_amdgcn_readlane((int)incl, 63);
}
// segment (0..63) of the group's k-th record: the last one whose offset is <= k
__device__ __forceinline__ u32 cc_segment_of(const u32* s_off, u32 k) {
    u32 s = 0u;
#pragma unroll
    for (u32 st = 32u; st >= 1u; st >>= 1)
        if (s + st < 64u && s_off[s + st] <= k) s += st;
    return s;
}

// Step 1, one wave per GROUP of 64 segments (12.8 rows of one layer at 1025 cells per row): the components of the group's
// own records -- +x links (next record of the segment, or the first record of the next segment of the row) and +y links
// into the group's later rows -- by a union-find in LDS, CC_WIN records at a time; then every record's global parent is
// set to its local root.  The global union-find (mc_cc_link) so starts from ~2 sets per group instead of one per record:
// 30x fewer hooks on the way to the one big component a surface usually is, and trees one level deep.
extern "C" __global__ __launch_bounds__(256) void mc_cc_local(const u32* __restrict__ recs, const uint2* __restrict__ segcb, u32* __restrict__ parent,
                                                    u32 nseg, int nchunk, int n1, int imax, const u32* __restrict__ overflow) {
    __shared__ u32 s_off[4][65];
    __shared__ u32 s_first[4][64];
    __shared__ u32 s_rec[4][CC_WIN];  // cell | code << 8 | segment << 16
    __shared__ u32 s_lab[4][CC_WIN];
    if (overflow[0] != 0u) return;  // a record region overflowed: the host sweeps again with a larger buffer
    const u32 lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    const u32 seg0 = (blockIdx.x * 4u + w) * 64u;
    if (seg0 >= nseg) return;  // whole wave
    u32* const off = s_off[w];
    u32* const lrec = s_rec[w];
    u32* const lab = s_lab[w];
    const u32 total = cc_group(segcb, seg0, nseg, lane, off, s_first[w]);
    for (u32 win0 = 0u; win0 < total; win0 += CC_WIN) {
        const u32 n = min(CC_WIN, total - win0);
        for (u32 i = lane; i < n; i += 64u) {
            const u32 k = win0 + i;
            const u32 sg = cc_segment_of(off, k);
            lrec[i] = (recs[s_first[w][sg] + (k - off[sg])] & 0xFFFFu) | (sg << 16);
            lab[i] = i;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (u32 i = lane; i < n; i += 64u) {
            const u32 v = lrec[i];
            const u32 cellx = v & 0xFFu, code = (v >> 8) & 0xFFu, sg = v >> 16;
            const u32 seg = seg0 + sg;
            const u32 rowidx = seg / (u32)nchunk, ch = seg - rowidx * (u32)nchunk;
            const int x = (int)(ch * 256u + cellx), y = (int)(rowidx % (u32)n1), z = (int)(rowidx / (u32)n1);
            if (x > imax || y > imax || z > imax) continue;  // never entered by a move: links to nothing
            if (cc_mixed(code, CC_FACE_PX) && x + 1 <= imax) {
                u32 j = 0xFFFFFFFFu;
                if (cellx < 255u) {
                    if (i + 1u < n && lrec[i + 1u] == ((v & 0xFFFF0000u) | (lrec[i + 1u] & 0xFF00u) | (cellx + 1u))) j = i + 1u;
                } else if (sg + 1u < 64u && (int)ch + 1 < nchunk) {
                    const u32 k2 = off[sg + 1u];
                    if (k2 < off[sg + 2u] && k2 >= win0 && k2 < win0 + n && (lrec[k2 - win0] & 0xFFu) == 0u) j = k2 - win0;
                }
                if (j != 0xFFFFFFFFu) lds_union(lab, i, j);
            }
            if (cc_mixed(code, CC_FACE_PY) && y + 1 <= imax && sg + (u32)nchunk < 64u) {
                const u32 sy = sg + (u32)nchunk;
                u32 lo = max(off[sy], win0), hi = min(off[sy + 1u], win0 + n);
                if (lo < hi) {
                    lo -= win0;
                    hi -= win0;
                    const u32 end = hi;
                    while (lo < hi) {
                        const u32 mid = (lo + hi) >> 1;
                        if ((lrec[mid] & 0xFFu) < cellx) lo = mid + 1u;
                        else hi = mid;
                    }
                    if (lo < end && (lrec[lo] & 0xFFu) == cellx) lds_union(lab, i, lo);
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (u32 i = lane; i < n; i += 64u) {
            u32 root = i;  // (read-only walk: the unions are over)
            for (;;) {
                const u32 p = lab[root];
                if (p == root) break;
                root = p;
            }
            const u32 sg = lrec[i] >> 16, sr = lrec[root] >> 16;
            parent[s_first[w][sg] + (win0 + i - off[sg])] = s_first[w][sr] + (win0 + root - off[sr]);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

// One lane per (pm, pn) pair of the chunk: a 128-slot LDS table claimed with a compare-and-swap; a lane that finds its
// slot taken yields only if the holder has the very same pair (a colliding pair is simply made twice).
__device__ __forceinline__ bool cc_leader(u32* tab, u32* key, u32 lane, bool has, u32 pm, u32 pn) {
    tab[lane] = 0u;
    tab[lane + 64u] = 0u;
    key[lane] = pm;
    key[lane + 64u] = pn;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    bool lead = has;
    if (has) {
        const u32 slot = ((pm * 0x9E3779B1u) ^ (pn * 0x85EBCA77u)) >> 25;
        const u32 old = atomicCAS(&tab[slot], 0u, lane + 1u);
        if (old != 0u && key[old - 1u] == pm && key[old + 63u] == pn) lead = false;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();  // (the table is used again right away)
    return lead;
}

// Step 2 (mode 1), one wave per group, one lane per record (64 at a time): the links mc_cc_local could not see -- +z, +y out
// of the group, +x out of the group's last segment (and, in a group of more than CC_WIN records, every +y and the links
// across a window's end) -- as unions on the global parents.  Of the chunk's links of one kind only ONE per pair (parent of
// the record, parent of its neighbour) is made, by whichever lane claims the pair first (cc_leader): equal parents mean
// one set (sets only merge), so that lane's union already joins all four.
// Step 3 (mode 2): parent[r] = root(r).
extern "C" __global__ __launch_bounds__(256) void mc_cc_link(const u32* __restrict__ recs, const uint2* __restrict__ segcb, u32* parent, u32 nseg,
                                                   int nchunk, int n1, int imax, const u32* __restrict__ overflow, int mode) {
    __shared__ u32 s_off[4][65];
    __shared__ u32 s_first[4][64];
    __shared__ u32 s_tab[4][128], s_key[4][128];
    __shared__ u32 s_total[4], s_next[4];
    if (overflow[0] != 0u) return;  // (the whole grid: uniform)
    const u32 lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    {
        const u32 seg0w = (blockIdx.x * 4u + w) * 64u;
        const u32 tot = seg0w < nseg ? cc_group(segcb, seg0w, nseg, lane, s_off[w], s_first[w]) : 0u;
        if (lane == 0u) {
            s_total[w] = tot;
            s_next[w] = 0u;
        }
    }
    __syncthreads();
    // the waves of a workgroup share the chunks of its four groups (an LDS counter per group; own group first): groups
    // differ in size by 10x and every chunk is a chain of dependent look-ups -- see mc_emit_direct (mc_kernels.hip)
    for (u32 dw = 0u; dw < 4u; ++dw) {
    const u32 wv = (w + dw) & 3u;
    const u32 seg0 = (blockIdx.x * 4u + wv) * 64u;
    const u32* const off = s_off[wv];
    const u32 total = s_total[wv];
    const bool single = total <= CC_WIN;
    for (;;) {  // (wave-uniform trip count: the shuffles below see every lane)
        u32 base = 0u;
        if (lane == 0u) base = __hip_atomic_fetch_add(&s_next[wv], 64u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        base = (u32)__builtin_amdgcn_readfirstlane((int)base);
        if (base >= total) break;
        const u32 k = base + lane;
        const bool valid = k < total;
        const u32 sg = valid ? cc_segment_of(off, k) : 0u;
        const u32 r = s_first[wv][sg] + (k - off[sg]);
        if (mode == 2) {
            if (valid) {
                // (roots are final: mode 1 has completed, in the launch before.  Plain loads: whatever this launch has
                // already overwritten with a root reads as that root or as the older link, and both lead to the root)
                u32 root = r;
                for (;;) {
                    const u32 pp = ((const u32*)parent)[root];
                    if (pp == root) break;
                    root = pp;
                }
                parent[r] = root;
            }
            continue;  // wave-uniform
        }
        const u32 seg = seg0 + sg;
        const u32 rec = valid ? recs[r] : 0u;
        const u32 cellx = rec & 0xFFu, code = (rec >> 8) & 0xFFu;
        const u32 rowidx = seg / (u32)nchunk, ch = seg - rowidx * (u32)nchunk;
        const int x = (int)(ch * 256u + cellx), y = (int)(rowidx % (u32)n1), z = (int)(rowidx / (u32)n1);
        const bool inb = valid && x <= imax && y <= imax && z <= imax;  // a cell beyond imax is never entered by a move
        // the links left to this step, and their far ends
        u32 rx = 0xFFFFFFFFu, ry, rz;
        if (inb && cc_mixed(code, CC_FACE_PX) && x + 1 <= imax) {
            if (cellx < 255u) {
                if (!single && ((k + 1u) & (CC_WIN - 1u)) == 0u && k + 1u < off[sg + 1u] && (recs[r + 1u] & 0xFFu) == cellx + 1u) rx = r + 1u;
            } else if ((int)ch + 1 < nchunk && (sg == 63u || !single)) {
                rx = cc_lookup(recs, segcb, seg + 1u, 0u);
            }
        }
        const bool wy = inb && cc_mixed(code, CC_FACE_PY) && y + 1 <= imax && (sg + (u32)nchunk >= 64u || !single);
        const bool wz = inb && cc_mixed(code, CC_FACE_PZ) && z + 1 <= imax;
        cc_lookup2(recs, segcb, wy, seg + (u32)nchunk, wz, seg + (u32)nchunk * (u32)n1, cellx, ry, rz);
        // parents, for the test above (any value read is an ancestor; equal ancestors = one set)
        const u32 none = 0xFFFFFFFFu;
        // (plain loads: a value that is out of date is still an ancestor -- mc_cc_local's, from the launch before, or a
        // later one -- and the test only ever PRUNES a union when two ancestors are equal, i.e. the sets were one already)
        // The four loads travel TOGETHER (a link that does not exist reads the record's own parent instead, and drops it):
        // left to itself the compiler puts each load behind the branch that guards it, four memory round trips in a row.
        const u32* __restrict__ pplain = parent;
        const u32 rsafe = valid ? r : 0u;
        u32 pm = pplain[rsafe];
        u32 px = pplain[rx != none ? rx : rsafe];
        u32 py = pplain[ry != none ? ry : rsafe];
        u32 pz = pplain[rz != none ? rz : rsafe];
        asm volatile("" : "+v"(pm), "+v"(px), "+v"(py), "+v"(pz));
        pm = valid ? pm : none;
        px = rx != none ? px : none;
        py = ry != none ? py : none;
        pz = rz != none ? pz : none;
        const bool ux = cc_leader(s_tab[w], s_key[w], lane, rx != none, pm, px);
        const bool uy = cc_leader(s_tab[w], s_key[w], lane, ry != none, pm, py);
        const bool uz = cc_leader(s_tab[w], s_key[w], lane, rz != none, pm, pz);
        if (ux) cc_union_fast(parent, r, rx);
        if (uy) cc_union_fast(parent, r, ry);
        if (uz) cc_union_fast(parent, r, rz);
    }
    }
}

// The roots the seed reaches (after the flatten pass: parent[r] IS the root): roots[0] = how many, roots[1..7].  The seed's
// own cell is always visited (marching.cpp:310-316); if it lies beyond imax it is linked to nothing, and the cells its
// expansion enters are its in-bounds neighbours across mixed faces.
extern "C" __global__ __launch_bounds__(64) void mc_cc_seed(const u32* __restrict__ recs, const uint2* __restrict__ segcb, const u32* __restrict__ parent,
                                                  u32* __restrict__ roots, int nchunk, int n1, int imax, int sx, int sy, int sz, int inside,
                                                  const u32* __restrict__ overflow) {
    if (threadIdx.x != 0u || blockIdx.x != 0u) return;
    u32 n = 0u;
    if (inside && overflow[0] == 0u) {
        const u32 seg = ((u32)sz * (u32)n1 + (u32)sy) * (u32)nchunk + (u32)(sx >> 8);
        const u32 r = cc_lookup(recs, segcb, seg, (u32)(sx & 255));
        if (r != 0xFFFFFFFFu) {  // (a seed cell without surface reaches nothing: its edge list is empty, marching.cpp:508-510)
            roots[++n] = parent[r];
            if (sx > imax || sy > imax || sz > imax) {
                const u32 code = (recs[r] >> 8) & 0xFFu;
                const u32 faces[6] = {CC_FACE_PX, 0xFFu ^ CC_FACE_PX, CC_FACE_PY, 0xFFu ^ CC_FACE_PY, CC_FACE_PZ, 0xFFu ^ CC_FACE_PZ};
                const int dx[6] = {1, -1, 0, 0, 0, 0}, dy[6] = {0, 0, 1, -1, 0, 0}, dz[6] = {0, 0, 0, 0, 1, -1};
                for (int f = 0; f < 6; ++f) {
                    const int nx = sx + dx[f], ny = sy + dy[f], nz = sz + dz[f];
                    if (!cc_mixed(code, faces[f]) || nx < 0 || ny < 0 || nz < 0 || nx > imax || ny > imax || nz > imax) continue;
                    const u32 rn = cc_lookup(recs, segcb, ((u32)nz * (u32)n1 + (u32)ny) * (u32)nchunk + (u32)(nx >> 8), (u32)(nx & 255));
                    if (rn != 0xFFFFFFFFu) roots[++n] = parent[rn];
                }
            }
        }
    }
    roots[0] = n;
}

// Records outside the seed's component lose their triangles; every segment's triangle prefixes, its counts and the group sums
// are rebuilt.  One wave per GROUP of 64 segments, lane = record in chunks of 64 (round 2 put one lane on a SEGMENT and
// walked its records one after the other -- 256 dependent steps where a row lies in the surface -- and issued one 64-bit
// atomic per segment; now the prefix inside a segment is a segmented wave scan plus a per-segment carry in LDS, and a group
// costs ONE atomic).
__device__ __forceinline__ u32 sf_scan(u32 v) {  // wavefront inclusive prefix sum (row_shr 1,2,3 / 4 / 8, row_bcast 15 / 31)
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
extern "C" __global__ __launch_bounds__(256) void mc_seed_filter(u32* __restrict__ recs, uint2* __restrict__ segcb, const u32* __restrict__ parent,
                                                       const u32* __restrict__ roots, u32 nseg, u64* __restrict__ grpsum,
                                                       const u32* __restrict__ overflow) {
    __shared__ u32 s_act[4][66];   // per segment of the group: its first record (group-local), [64] = records of the group
    __shared__ u32 s_tot[4][64];   // ... triangles kept so far
    __shared__ u32 s_rb[4][64];    // ... its first record in recs
    const int lane = threadIdx.x & 63, w = (int)(threadIdx.x >> 6);
    const u32 group = blockIdx.x * 4u + (u32)w;
    const u32 ngroups = (nseg + 63u) / 64u;
    if (group >= ngroups || overflow[0] != 0u) return;  // (whole waves; no workgroup barrier below)
    const u32 seg = group * 64u + (u32)lane;
    const uint2 cb = seg < nseg ? segcb[seg] : make_uint2(0u, 0u);
    const u32 act = cb.x >> 16;
    const u32 iact = sf_scan(act);
    const u32 nrec = (u32)__builtin_amdgcn_readlane((int)iact, 63);
    if (nrec == 0u) return;
    s_act[w][lane] = iact - act;
    s_tot[w][lane] = 0u;
    s_rb[w][lane] = cb.y;
    if (lane == 63) s_act[w][64] = nrec;
    const u32 nroots = roots[0];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (u32 r0 = 0; r0 < nrec; r0 += 64u) {
        const u32 r = r0 + (u32)lane;
        const bool valid = r < nrec;
        u32 lo = 0, hi = 64;  // owning segment: the largest s with s_act[s] <= r (empty segments repeat the value)
#pragma unroll
        for (int it = 0; it < 6; ++it) {
            const u32 mid = (lo + hi) >> 1;
            if (s_act[w][mid] <= r) lo = mid; else hi = mid;
        }
        const u32 idx = s_rb[w][lo] + (r - s_act[w][lo]);
        u32 rec = 0u, nt = 0u;
        if (valid) {
            rec = recs[idx];
            const u32 root = parent[idx];
            bool keep = false;
            for (u32 i = 1u; i <= nroots; ++i) keep = keep || root == roots[i];
            nt = keep ? (rec >> 17) & 7u : 0u;
        }
        // triangles kept in front of this record inside its segment: what earlier chunks left (s_tot) + the wave prefix since
        // the segment's first record of this chunk
        const u32 incl = sf_scan(nt), excl = incl - nt;
        const int sprev = __builtin_amdgcn_update_dpp(-1, (int)lo, 0x138, 0xf, 0xf, false);  // wave_shr:1
        const bool head = lane == 0 || sprev != (int)lo;
        u32 hb = head ? excl + 1u : 0u;  // max-scan of (prefix at the head + 1): non-decreasing along the wave
        {
            u32 x = hb;
            x = max(x, (u32)__builtin_amdgcn_update_dpp(0, (int)hb, 0x111, 0xf, 0xf, false));
            x = max(x, (u32)__builtin_amdgcn_update_dpp(0, (int)hb, 0x112, 0xf, 0xf, false));
            x = max(x, (u32)__builtin_amdgcn_update_dpp(0, (int)hb, 0x113, 0xf, 0xf, false));
            x = max(x, (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xe, false));
            x = max(x, (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xc, false));
            x = max(x, (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false));
            x = max(x, (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false));
            hb = x - 1u;
        }
        const u32 carry = s_tot[w][lo];
        const u32 tpre = carry + (excl - hb);
        if (valid) recs[idx] = (rec & 0x1FFFFu) | (nt << 17) | (tpre << 20);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // the last record of each segment run of this chunk leaves the segment's new total
        const int snext = __builtin_amdgcn_update_dpp(-1, (int)lo, 0x130, 0xf, 0xf, false);  // wave_shl:1
        if (valid && (lane == 63 || snext != (int)lo || r + 1u >= nrec)) s_tot[w][lo] = tpre + nt;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    const u32 tris = act ? s_tot[w][lane] : 0u;
    if (seg < nseg && act) segcb[seg] = make_uint2(tris | (act << 16), cb.y);
    const u32 gt = sf_scan(tris);
    if (lane == 63) atomicAdd(&grpsum[group], (u64)gt | ((u64)nrec << 32));
}
