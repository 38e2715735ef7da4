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
// starts (the host clears the control block once; every sweep leaves it clean, see below).  Spins are bounded: a workgroup that gives up raises totals[3] (the
// host reads totals[3] from the device copy only when the counts look wrong -- it cannot happen while every workgroup
// with a lower ticket is running, which the ticket order guarantees).
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
// code volume, in one launch (mc_seed_walk); mc_seed_filter then removes the triangles of unvisited cells from the
// records before the scan and the emit kernel run.  Cell id = (z*n1 + y)*n1 + x.
#include "../../include/mc_tables_data.h"

__device__ __forceinline__ u32 seed_code(const u8* codes, const u32* tail, u64 pitch, int n1, int main_cells, int x, int y, int z) {
    const u64 row = (u64)z * (u64)n1 + (u64)y;
    return x < main_cells ? (u32)codes[row * pitch + (u64)x] : (tail[row] >> (8 * (x - main_cells))) & 0xFFu;
}

// counters of the walk, one per 128-byte line: [0] next queue slot to take, [32] slots handed out to producers, [64] cells
// fully expanded, [96] a lane gave up (spin bound)
extern "C" __global__ __launch_bounds__(64) void mc_seed_init(u32* __restrict__ visited, u32* __restrict__ queue, u32* __restrict__ ctr,
                                                    u32 cell) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        visited[cell >> 5] = 1u << (cell & 31u);
        queue[0] = cell + 1u;  // a queue slot holds cell + 1; 0 = not written yet
        ctr[0] = 0u;
        ctr[32] = 1u;
        ctr[64] = 0u;
        ctr[96] = 0u;
    }
}

// The whole walk in ONE launch: an asynchronous work list instead of one launch per breadth-first level (the order in
// which the cells of a component are reached does not change the component).  A WAVE takes 64 consecutive queue slots at a
// time (one atomic add on the head per 64 cells: a single word takes ~90 atomics per microsecond, and 5 M cells were
// 40 ms of them when every lane dequeued for itself); lane l polls slot base + l until a producer has filled it, expands
// its cell -- for each face that carries an intersection the neighbour is marked in the visited bitmap with atomicOr and
// kept if it was new --, the wave's new cells get their slots with ONE atomic add on the tail (wave prefix sum) and are
// stored as cell + 1 (the value is its own "ready" flag), and the expanded cells are counted with one add.  The walk is
// over when every appended cell has been expanded (done == tail: no expansion is running, so nothing more can be
// appended); a lane that finds that, and whose slot lies beyond the tail, is through; a wave leaves when all its lanes are.
// One loop with a wave-uniform exit and plain if / else inside: a lane that waits must never sit in an inner spin loop of
// its own -- in lockstep the lane of the same wave that holds the work those lanes wait for would never get its turn.
// imax: the largest cell index a move may reach (marching.cpp:84-86: x0 + 0.5*step <= 1); the lower bound is index 0.
#define SEED_SPIN_MAX (1u << 21)  // ~1 s of polling: the whole walk takes milliseconds
extern "C" __global__ __launch_bounds__(64) void mc_seed_walk(const u8* __restrict__ codes, const u32* __restrict__ tail, u64 pitch, int n1,
                                                    int main_cells, int imax, u32* __restrict__ visited, u32* queue, u32 qcap, u32* ctr) {
    constexpr unsigned short kfc[6] = MC_FACE_CORNER_INIT;  // marching_lookup.h:25-32
    const u32 lane = threadIdx.x & 63u;
    u32 base = 0u;
    if (lane == 0u) base = atomicAdd(&ctr[0], 64u);
    base = (u32)__builtin_amdgcn_readfirstlane((int)base);
    bool consumed = false;  // this lane's slot of the current block has been expanded
    bool through = false;   // ... will never be filled: the walk is over
    u32 spins = 0;
    for (;;) {
        const u32 my = base + lane;
        u32 v = 0u;
        if (!consumed && !through && my < qcap) v = __hip_atomic_load(&queue[my], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool have = v != 0u;
        u32 fresh[6];
        u32 nfresh = 0u;
        if (have) {
            const u32 cell = v - 1u;
            const int x = (int)(cell % (u32)n1), y = (int)((cell / (u32)n1) % (u32)n1), z = (int)(cell / ((u32)n1 * (u32)n1));
            const u32 code = seed_code(codes, tail, pitch, n1, main_cells, x, y, z);
            const bool surf = code != 0u && code != 255u;  // (a cell without surface has an empty edge list: marching.cpp:508-510)
#pragma unroll
            for (int f = 0; f < 6; ++f) {
                // the face carries an intersection iff its four corners are not all on one side (marching.cpp:62-69)
                int ones = 0, bx = 0, by = 0, bz = 0;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int c = (kfc[f] >> (4 * k)) & 0xF;
                    ones += (code >> c) & 1u;
                    bx += (0x66 >> c) & 1;
                    by += (0xCC >> c) & 1;
                    bz += c >> 2;
                }
                // marching_lookup.h:43-50 cube_face_normal = the axis on which the face's corners agree
                const int nx = x + (bx == 4 ? 1 : bx == 0 ? -1 : 0), ny = y + (by == 4 ? 1 : by == 0 ? -1 : 0),
                          nz = z + (bz == 4 ? 1 : bz == 0 ? -1 : 0);
                const bool go = surf && ones != 0 && ones != 4 && nx >= 0 && ny >= 0 && nz >= 0 && nx <= imax && ny <= imax && nz <= imax;  // marching.cpp:84-86
                fresh[f] = 0xFFFFFFFFu;
                if (go) {
                    const u32 nc = ((u32)nz * (u32)n1 + (u32)ny) * (u32)n1 + (u32)nx;
                    const u32 bit = 1u << (nc & 31u);
                    if (!(atomicOr(&visited[nc >> 5], bit) & bit)) {  // marching.cpp:89-97: queue it unless it is in the set already
                        fresh[f] = nc;
                        ++nfresh;
                    }
                }
            }
            consumed = true;
            spins = 0;
        }
        // the wave's new cells: one atomic add on the tail, the slots handed out by a prefix sum
        u32 incl = nfresh;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const u32 t = (u32)__shfl_up((int)incl, o, 64);
            if (lane >= (u32)o) incl += t;
        }
        const u32 total = (u32)__builtin_amdgcn_readlane((int)incl, 63);
        const u64 worked = __ballot(have);
        if (total) {
            u32 tb = 0u;
            if (lane == 0u) tb = atomicAdd(&ctr[32], total);
            tb = (u32)__builtin_amdgcn_readfirstlane((int)tb);
            u32 slot = tb + incl - nfresh;
#pragma unroll
            for (int f = 0; f < 6; ++f)
                if (have && fresh[f] != 0xFFFFFFFFu) {
                    if (slot < qcap) __hip_atomic_store(&queue[slot], fresh[f] + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ++slot;
                }
            // (slots beyond the list's end -- cannot happen, it holds every record -- are counted as done right away so
            // that the walk still ends)
            const u32 lost = tb + total > qcap ? min(total, tb + total - qcap) : 0u;
            if (lost && lane == 0u) atomicAdd(&ctr[64], lost);
        }
        // after the appends: done == tail then really means "nothing is running"
        if (worked && lane == 0u) atomicAdd(&ctr[64], (u32)__builtin_popcountll(worked));
        if (!have && !consumed && !through) {
            const u32 done = __hip_atomic_load(&ctr[64], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const u32 tl = __hip_atomic_load(&ctr[32], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((done == tl && my >= tl) || my >= qcap) through = true;
            if (++spins > SEED_SPIN_MAX) {
                ctr[96] = 1u;
                through = true;
            }
        }
        const u64 open = __ballot(!consumed && !through);
        if (open == 0ull) {
            if (__ballot(through)) break;  // the end of the list lies inside (or before) this block: nothing comes after it
            // every slot of the block has been expanded: the next 64
            u32 nb = 0u;
            if (lane == 0u) nb = atomicAdd(&ctr[0], 64u);
            base = (u32)__builtin_amdgcn_readfirstlane((int)nb);
            consumed = false;
        } else if (worked == 0ull) {
            __builtin_amdgcn_s_sleep(8);
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
