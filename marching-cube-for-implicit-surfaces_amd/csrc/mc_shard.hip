// mc_shard.hip -- the multi-device sweep of libmc_hip.so (include/mc_hip.h, "multi-device sweep"): the grid's cell layers
// cut into one contiguous Z slab per device, swept concurrently, the counts turned into offsets.
//
// Replaces nothing in the reference -- Marching::recalculate (Source/marching.cpp:368-384) is single-threaded -- but is the
// form of that call SURVEY 8b / 8e ask for: a device list on the C++ side.  Why it is only this much code: the sweep is
// z-major (marching.cpp:375: z outermost), so contiguous slabs concatenate to the single sweep's order; f is analytic, so a
// slab evaluates its own top sample plane and no halo exists; and the indexed mesh of a slab welded as a part of the whole
// grid (MC_FLAG_SEAM, mc_runtime.hip) needs ONE number from the slabs below it, their vertex count.
//
// Two forms: one process with a context per device (host threads; counts are host reads), and one process per device with
// an RCCL all-gather of the counts (librccl.so is dlopen'ed on first use, so the library itself does not depend on it).
// Everything here goes through the public C ABI plus mc_internal.hpp's two hooks; no kernel lives in this file.
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/mc_hip.h"
#include "mc_internal.hpp"

namespace {

// z range p asks for, clamped like mc_march does (mc_runtime.hip: prepare)
bool sweep_range(const mc_params* p, int& n1, int& zb, int& ze) {
    n1 = mc_cells_per_axis(p->step);
    if (n1 <= 0) return false;
    zb = p->z_begin < 0 ? 0 : p->z_begin;
    ze = (p->z_end < 0 || p->z_end > n1) ? n1 : p->z_end;
    return true;
}

// the n + 1 slab bounds: the caller's, checked, or near-equal parts of [zb, ze)
int slab_bounds(const int32_t* bounds, int n, int zb, int ze, std::vector<int>& b) {
    b.resize((size_t)n + 1);
    if (bounds) {
        for (int i = 0; i <= n; ++i) b[(size_t)i] = bounds[i];
        if (b[0] < zb || b[(size_t)n] > ze) return mc_internal_fail(MC_ERR_ARG, "slab bounds [%d, %d) leave the sweep's layers [%d, %d)", b[0], b[(size_t)n], zb, ze);
        for (int i = 0; i < n; ++i)
            if (b[(size_t)i] > b[(size_t)i + 1]) return mc_internal_fail(MC_ERR_ARG, "slab bounds must ascend (bounds[%d] = %d > bounds[%d] = %d)", i, b[(size_t)i], i + 1, b[(size_t)i + 1]);
        return MC_OK;
    }
    for (int i = 0; i < n; ++i) {
        int lo, hi;
        mc_shard_layers(ze - zb, n, i, &lo, &hi);
        b[(size_t)i] = zb + lo;
        b[(size_t)i + 1] = zb + hi;
    }
    return MC_OK;
}

// ---------------------------------------------------------------- RCCL, loaded on first use
typedef struct { char internal[128]; } rccl_unique_id;  // rccl.h: ncclUniqueId, NCCL_UNIQUE_ID_BYTES = 128
static_assert(sizeof(rccl_unique_id) == MC_COMM_ID_BYTES, "MC_COMM_ID_BYTES is RCCL's unique-id size");
typedef void* rccl_comm_t;
struct Rccl {
    void* so = nullptr;
    int (*GetUniqueId)(rccl_unique_id*) = nullptr;
    int (*CommInitRank)(rccl_comm_t*, int, rccl_unique_id, int) = nullptr;
    int (*CommDestroy)(rccl_comm_t) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, rccl_comm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    std::string why;  // why it could not be loaded
};
constexpr int kRcclUint64 = 5;  // rccl.h: ncclUint64

Rccl* rccl() {
    static Rccl R;
    static std::once_flag once;
    std::call_once(once, [] {
        // A process that has RCCL loaded already (torch.distributed brings its own copy, soname librccl.so.1) must not
        // get a second one: two copies would resolve each other's internals.  So: the loaded copy if there is one, else
        // $MC_RCCL_LIB / the ROCm installation's, bound to its own symbols.
        const char* loaded[] = {"librccl.so.1", "librccl.so"};
        for (const char* nm : loaded) {
            R.so = dlopen(nm, RTLD_NOW | RTLD_NOLOAD);
            if (R.so) break;
        }
        const char* names[] = {getenv("MC_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so"};
        for (const char* nm : names) {
            if (R.so) break;
            if (!nm || !*nm) continue;
            R.so = dlopen(nm, RTLD_NOW | RTLD_LOCAL | RTLD_DEEPBIND);
            if (!R.so) R.why = dlerror();
        }
        if (!R.so) return;
        R.GetUniqueId = (decltype(R.GetUniqueId))dlsym(R.so, "ncclGetUniqueId");
        R.CommInitRank = (decltype(R.CommInitRank))dlsym(R.so, "ncclCommInitRank");
        R.CommDestroy = (decltype(R.CommDestroy))dlsym(R.so, "ncclCommDestroy");
        R.AllGather = (decltype(R.AllGather))dlsym(R.so, "ncclAllGather");
        R.GetErrorString = (decltype(R.GetErrorString))dlsym(R.so, "ncclGetErrorString");
        if (!R.GetUniqueId || !R.CommInitRank || !R.CommDestroy || !R.AllGather) {
            R.why = "librccl.so lacks ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclAllGather";
            dlclose(R.so);
            R.so = nullptr;
        }
    });
    return &R;
}

int rccl_fail(const char* what, int code) {
    Rccl* R = rccl();
    return mc_internal_fail(MC_ERR_HIP, "%s failed: %s", what, R->GetErrorString ? R->GetErrorString(code) : "RCCL error");
}

#define HIPCHK_S(call)                                                                                         \
    do {                                                                                                       \
        hipError_t e_ = (call);                                                                                \
        if (e_ != hipSuccess) return mc_internal_fail(MC_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_)); \
    } while (0)

}  // namespace

#define MC_GATHER_SLOTS 64  // gathers in flight before mc_comm_gather_async waits for the oldest

struct mc_comm {
    mc_context* ctx = nullptr;
    int device = 0, world = 1, rank = 0;
    rccl_comm_t comm = nullptr;
    hipStream_t side = nullptr;          // the copy of the counts and the all-gather run here, beside the sweeps
    hipEvent_t ev_sweep = nullptr, ev_copied = nullptr;
    uint64_t* d_send = nullptr;          // [MC_GATHER_SLOTS][2]
    uint64_t* d_recv = nullptr;          // [MC_GATHER_SLOTS][world][2]
    uint64_t* h_recv = nullptr;          // pinned [world][2]: the last gather, copied out by mc_comm_wait
    uint64_t* h_send = nullptr;          // pinned [2]: mc_march_rank's counts on their way up
    int slot = 0, in_flight = 0;
};

#pragma GCC visibility push(default)
extern "C" {

void mc_shard_layers(int n_layers, int parts, int part, int* z_begin, int* z_end) {
    if (parts < 1) parts = 1;
    if (part < 0) part = 0;
    if (part >= parts) part = parts - 1;
    if (n_layers < 0) n_layers = 0;
    const int base = n_layers / parts, rem = n_layers % parts;
    const int b = part * base + (part < rem ? part : rem);
    if (z_begin) *z_begin = b;
    if (z_end) *z_end = b + base + (part < rem ? 1 : 0);
}

int mc_march_sharded(mc_context* const* ctxs, int n, const mc_params* p, const int32_t* bounds, mc_result* results, mc_shard* shards) {
    if (!ctxs || n < 1 || !p || !p->equation || !results) return mc_internal_fail(MC_ERR_ARG, "null argument / empty device list");
    for (int i = 0; i < n; ++i) {
        if (!ctxs[i]) return mc_internal_fail(MC_ERR_ARG, "context %d of the list is null", i);
        for (int j = 0; j < i; ++j)
            if (ctxs[j] == ctxs[i]) return mc_internal_fail(MC_ERR_ARG, "context %d appears twice in the list: a context runs one sweep at a time (make one per slab)", i);
    }
    int n1, zb, ze;
    if (!sweep_range(p, n1, zb, ze)) return mc_internal_fail(MC_ERR_STEP, "grid step %g outside [0.001, 0.5] (Marching::set_grid_step_size)", (double)p->step);
    if (zb > ze) return mc_internal_fail(MC_ERR_ARG, "z_begin %d > z_end %d", zb, ze);
    std::vector<int> b;
    int r = slab_bounds(bounds, n, zb, ze, b);
    if (r) return r;
    // Seed mode (marching.cpp:310-331) follows ONE connected component through the whole grid: the component labelling needs
    // every layer on one device.  With seed mode on (on the first context: the facade keeps all of a Marching object's
    // contexts in the same state) the whole range is therefore swept by the first context and the others get empty slabs --
    // the same results layout, the same mesh as mc_march, no error for a caller that has a device list set.
    if (mc_internal_seed_on(ctxs[0])) {
        if (zb != 0 || ze != n1) return mc_internal_fail(MC_ERR_ARG, "seed mode needs the whole grid in one sweep (z_begin 0, z_end -1)");
        b[0] = zb;
        for (int i = 1; i <= n; ++i) b[(size_t)i] = ze;
    }
    const bool indexed = (p->flags & MC_FLAG_INDEXED) != 0;
    const uint32_t flags = indexed ? (p->flags | MC_FLAG_SEAM) : p->flags;
    std::vector<int> rc((size_t)n, MC_OK);
    std::vector<std::string> err((size_t)n);
    auto slab = [&](int i) {
        mc_params q = *p;
        q.flags = flags;
        q.z_begin = b[(size_t)i];
        q.z_end = b[(size_t)i + 1];
        rc[(size_t)i] = mc_march(ctxs[i], &q, &results[i]);
        if (rc[(size_t)i]) err[(size_t)i] = mc_last_error();  // (thread-local text: keep it for the caller's thread)
    };
    {
        // one host thread per slab (64 MB stacks: a slab's first sweep of an equation may run hiprtc), slab 0 on the caller's
        std::vector<McThread> th((size_t)n);
        for (int i = 1; i < n; ++i) {
            auto* fn = new std::function<void()>([&slab, i]() { slab(i); });
            if (!mc_thread_start(th[(size_t)i], fn)) {
                delete fn;
                slab(i);  // (no thread to be had: the slab runs here, after slab 0's)
            }
        }
        slab(0);
        for (int i = 1; i < n; ++i) mc_thread_join(th[(size_t)i]);
    }
    for (int i = 0; i < n; ++i)
        if (rc[(size_t)i]) return mc_internal_fail(rc[(size_t)i], "slab %d (layers [%d, %d)): %s", i, b[(size_t)i], b[(size_t)i + 1], err[(size_t)i].c_str());
    uint64_t toff = 0, voff = 0;
    std::vector<uint64_t> to((size_t)n), vo((size_t)n);
    for (int i = 0; i < n; ++i) {
        to[(size_t)i] = toff;
        vo[(size_t)i] = voff;
        toff += results[i].n_tris;
        voff += results[i].n_verts;
    }
    if (indexed) {
        if (voff > 0xFFFFFFFFull) return mc_internal_fail(MC_ERR_OVERFLOW, "%llu vertices in the whole grid exceed 2^32-1 (tri_list holds 32-bit indices)", (unsigned long long)voff);
        for (int i = 0; i < n; ++i)
            if ((r = mc_index_rebase(ctxs[i], vo[(size_t)i]))) return r;
    }
    if (shards)
        for (int i = 0; i < n; ++i) {
            shards[i].z_begin = results[i].z_begin;
            shards[i].z_end = results[i].z_end;
            shards[i].tri_offset = to[(size_t)i];
            shards[i].vert_offset = indexed ? vo[(size_t)i] : 0;
            shards[i].n_tris_total = toff;
            shards[i].n_verts_total = indexed ? voff : 0;
        }
    return MC_OK;
}

int mc_copy_sharded_vertices(mc_context* const* ctxs, int n, float* host, uint64_t max_tris) {
    if (!ctxs || n < 1 || !host) return mc_internal_fail(MC_ERR_ARG, "null argument");
    uint64_t at = 0;
    for (int i = 0; i < n; ++i) {
        const mc_result* L = ctxs[i] ? mc_internal_last(ctxs[i]) : nullptr;
        if (!L) return mc_internal_fail(MC_ERR_ARG, "context %d holds no sweep", i);
        if (at + L->n_tris > max_tris) return mc_internal_fail(MC_ERR_ARG, "buffer holds %llu triangles, the slabs have more", (unsigned long long)max_tris);
        if (L->n_tris) {
            const int r = mc_copy_vertices(ctxs[i], host + at * 18, max_tris - at);
            if (r) return r;
        }
        at += L->n_tris;
    }
    return MC_OK;
}

int mc_copy_sharded_indexed(mc_context* const* ctxs, int n, float* vertex_list, uint32_t* tri_list, float* normals, uint64_t max_verts,
                            uint64_t max_tris) {
    if (!ctxs || n < 1) return mc_internal_fail(MC_ERR_ARG, "null argument");
    uint64_t vat = 0, tat = 0;
    for (int i = 0; i < n; ++i) {
        const mc_result* L = ctxs[i] ? mc_internal_last(ctxs[i]) : nullptr;
        if (!L) return mc_internal_fail(MC_ERR_ARG, "context %d holds no sweep", i);
        if ((vertex_list || normals) && vat + L->n_verts > max_verts) return mc_internal_fail(MC_ERR_ARG, "buffer holds %llu vertices, the slabs have more", (unsigned long long)max_verts);
        if (tri_list && tat + L->n_tris > max_tris) return mc_internal_fail(MC_ERR_ARG, "buffer holds %llu triangles, the slabs have more", (unsigned long long)max_tris);
        const int r = mc_copy_indexed(ctxs[i], vertex_list ? vertex_list + vat * 3 : nullptr, tri_list ? tri_list + tat * 3 : nullptr,
                                      normals ? normals + vat * 3 : nullptr, max_verts - vat, max_tris - tat);
        if (r) return r;
        vat += L->n_verts;
        tat += L->n_tris;
    }
    return MC_OK;
}

int mc_copy_sharded_codes(mc_context* const* ctxs, int n, uint8_t* host, uint64_t max_bytes) {
    if (!ctxs || n < 1 || !host) return mc_internal_fail(MC_ERR_ARG, "null argument");
    uint64_t at = 0;
    for (int i = 0; i < n; ++i) {
        const mc_result* L = ctxs[i] ? mc_internal_last(ctxs[i]) : nullptr;
        if (!L) return mc_internal_fail(MC_ERR_ARG, "context %d holds no sweep", i);
        if (at + L->n_cells > max_bytes) return mc_internal_fail(MC_ERR_ARG, "buffer holds %llu bytes, the slabs have more cells", (unsigned long long)max_bytes);
        if (L->n_cells) {
            const int r = mc_copy_codes(ctxs[i], host + at, max_bytes - at);
            if (r) return r;
        }
        at += L->n_cells;
    }
    return MC_OK;
}

// ---------------------------------------------------------------- one process per device
int mc_comm_get_id(uint8_t id[MC_COMM_ID_BYTES]) {
    if (!id) return mc_internal_fail(MC_ERR_ARG, "null id");
    Rccl* R = rccl();
    if (!R->so) return mc_internal_fail(MC_ERR_HIP, "librccl.so could not be loaded: %s", R->why.c_str());
    rccl_unique_id u;
    const int rc = R->GetUniqueId(&u);
    if (rc) return rccl_fail("ncclGetUniqueId", rc);
    memcpy(id, u.internal, MC_COMM_ID_BYTES);
    return MC_OK;
}

void mc_comm_destroy(mc_comm* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->side) (void)hipStreamSynchronize(c->side);
    if (c->comm) (void)rccl()->CommDestroy(c->comm);
    if (c->d_send) (void)hipFree(c->d_send);
    if (c->d_recv) (void)hipFree(c->d_recv);
    if (c->h_recv) (void)hipHostFree(c->h_recv);
    if (c->h_send) (void)hipHostFree(c->h_send);
    if (c->ev_sweep) (void)hipEventDestroy(c->ev_sweep);
    if (c->ev_copied) (void)hipEventDestroy(c->ev_copied);
    if (c->side) (void)hipStreamDestroy(c->side);
    delete c;
}

int mc_comm_create(mc_context* ctx, const uint8_t id[MC_COMM_ID_BYTES], int world, int rank, mc_comm** out) {
    if (!out) return mc_internal_fail(MC_ERR_ARG, "null out");
    *out = nullptr;
    if (!ctx || !id) return mc_internal_fail(MC_ERR_ARG, "null argument");
    if (world < 1 || rank < 0 || rank >= world) return mc_internal_fail(MC_ERR_ARG, "rank %d outside a world of %d", rank, world);
    Rccl* R = rccl();
    if (!R->so) return mc_internal_fail(MC_ERR_HIP, "librccl.so could not be loaded: %s", R->why.c_str());
    mc_comm* c = new mc_comm();
    c->ctx = ctx;
    c->device = mc_internal_device(ctx);
    c->world = world;
    c->rank = rank;
    auto bail = [&](int code) {
        mc_comm_destroy(c);
        return code;
    };
#define HIPCHK_C(call)                                                                                                 \
    do {                                                                                                               \
        hipError_t e_ = (call);                                                                                        \
        if (e_ != hipSuccess) return bail(mc_internal_fail(MC_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_))); \
    } while (0)
    HIPCHK_C(hipSetDevice(c->device));
    HIPCHK_C(hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking));
    HIPCHK_C(hipEventCreateWithFlags(&c->ev_sweep, hipEventDisableTiming));
    HIPCHK_C(hipEventCreateWithFlags(&c->ev_copied, hipEventDisableTiming));
    HIPCHK_C(hipMalloc((void**)&c->d_send, (size_t)MC_GATHER_SLOTS * 2 * sizeof(uint64_t)));
    HIPCHK_C(hipMalloc((void**)&c->d_recv, (size_t)MC_GATHER_SLOTS * 2 * sizeof(uint64_t) * (size_t)world));
    HIPCHK_C(hipHostMalloc((void**)&c->h_recv, 2 * sizeof(uint64_t) * (size_t)world, hipHostMallocDefault));
    HIPCHK_C(hipHostMalloc((void**)&c->h_send, 2 * sizeof(uint64_t), hipHostMallocDefault));
#undef HIPCHK_C
    rccl_unique_id u;
    memcpy(u.internal, id, MC_COMM_ID_BYTES);
    const int rc = R->CommInitRank(&c->comm, world, u, rank);
    if (rc) return bail(rccl_fail("ncclCommInitRank", rc));
    *out = c;
    return MC_OK;
}

int mc_comm_gather_async(mc_comm* c, const uint64_t* d_totals) {
    if (!c || !d_totals) return mc_internal_fail(MC_ERR_ARG, "null argument");
    HIPCHK_S(hipSetDevice(c->device));
    hipStream_t sweep = (hipStream_t)mc_stream(c->ctx);
    if (c->in_flight == MC_GATHER_SLOTS) {  // every slot is in use: let the gathers drain (their results would be overwritten)
        HIPCHK_S(hipStreamSynchronize(c->side));
        c->in_flight = 0;
    }
    const int s = c->slot;
    uint64_t* send = c->d_send + (size_t)s * 2;
    uint64_t* recv = c->d_recv + (size_t)s * 2 * (size_t)c->world;
    // behind the sweep: copy its two count words aside; the context's next sweep (whose scan clears them) waits for the copy
    HIPCHK_S(hipEventRecord(c->ev_sweep, sweep));
    HIPCHK_S(hipStreamWaitEvent(c->side, c->ev_sweep, 0));
    HIPCHK_S(hipMemcpyAsync(send, d_totals, 2 * sizeof(uint64_t), hipMemcpyDeviceToDevice, c->side));
    HIPCHK_S(hipEventRecord(c->ev_copied, c->side));
    HIPCHK_S(hipStreamWaitEvent(sweep, c->ev_copied, 0));
    const int rc = rccl()->AllGather(send, recv, 2, kRcclUint64, c->comm, c->side);
    if (rc) return rccl_fail("ncclAllGather", rc);
    c->slot = (s + 1) % MC_GATHER_SLOTS;
    ++c->in_flight;
    return MC_OK;
}

int mc_comm_wait(mc_comm* c, uint64_t* counts) {
    if (!c) return mc_internal_fail(MC_ERR_ARG, "null communicator");
    HIPCHK_S(hipSetDevice(c->device));
    if (counts) {
        const int last = (c->slot + MC_GATHER_SLOTS - 1) % MC_GATHER_SLOTS;
        HIPCHK_S(hipMemcpyAsync(c->h_recv, c->d_recv + (size_t)last * 2 * (size_t)c->world, 2 * sizeof(uint64_t) * (size_t)c->world,
                                hipMemcpyDeviceToHost, c->side));
    }
    HIPCHK_S(hipStreamSynchronize(c->side));
    c->in_flight = 0;
    if (counts) memcpy(counts, c->h_recv, 2 * sizeof(uint64_t) * (size_t)c->world);
    return MC_OK;
}

int mc_march_rank(mc_comm* c, const mc_params* p, const int32_t* bounds, mc_result* res, mc_shard* shard) {
    if (!c || !p || !p->equation) return mc_internal_fail(MC_ERR_ARG, "null argument");
    int n1, zb, ze;
    if (!sweep_range(p, n1, zb, ze)) return mc_internal_fail(MC_ERR_STEP, "grid step %g outside [0.001, 0.5] (Marching::set_grid_step_size)", (double)p->step);
    if (zb > ze) return mc_internal_fail(MC_ERR_ARG, "z_begin %d > z_end %d", zb, ze);
    std::vector<int> b;
    int r = slab_bounds(bounds, c->world, zb, ze, b);
    if (r) return r;
    const bool indexed = (p->flags & MC_FLAG_INDEXED) != 0;
    mc_params q = *p;
    if (indexed) q.flags |= MC_FLAG_SEAM;
    q.z_begin = b[(size_t)c->rank];
    q.z_end = b[(size_t)c->rank + 1];
    mc_result mine{};
    // (a rank whose sweep fails still takes part in the all-gather -- with counts of ~0 -- so that the others do not hang)
    const int rsweep = mc_march(c->ctx, &q, &mine);
    const std::string sweep_err = rsweep ? mc_last_error() : "";
    HIPCHK_S(hipSetDevice(c->device));
    c->h_send[0] = rsweep ? ~0ull : mine.n_tris;
    c->h_send[1] = rsweep ? ~0ull : mine.n_verts;
    uint64_t* send = c->d_send;
    uint64_t* recv = c->d_recv;
    HIPCHK_S(hipStreamSynchronize(c->side));  // (no gather of the asynchronous form is left using slot 0)
    c->in_flight = 0;
    HIPCHK_S(hipMemcpyAsync(send, c->h_send, 2 * sizeof(uint64_t), hipMemcpyHostToDevice, c->side));
    const int rc = rccl()->AllGather(send, recv, 2, kRcclUint64, c->comm, c->side);
    if (rc) return rccl_fail("ncclAllGather", rc);
    HIPCHK_S(hipMemcpyAsync(c->h_recv, recv, 2 * sizeof(uint64_t) * (size_t)c->world, hipMemcpyDeviceToHost, c->side));
    HIPCHK_S(hipStreamSynchronize(c->side));
    if (rsweep) return mc_internal_fail(rsweep, "%s", sweep_err.c_str());
    uint64_t toff = 0, voff = 0, ttot = 0, vtot = 0;
    for (int k = 0; k < c->world; ++k) {
        const uint64_t t = c->h_recv[2 * k], v = c->h_recv[2 * k + 1];
        if (t == ~0ull) return mc_internal_fail(MC_ERR_HIP, "rank %d's sweep failed", k);
        if (k < c->rank) {
            toff += t;
            voff += v;
        }
        ttot += t;
        vtot += v;
    }
    if (indexed) {
        if (vtot > 0xFFFFFFFFull) return mc_internal_fail(MC_ERR_OVERFLOW, "%llu vertices in the whole grid exceed 2^32-1 (tri_list holds 32-bit indices)", (unsigned long long)vtot);
        if ((r = mc_index_rebase(c->ctx, voff))) return r;
    }
    if (res) *res = mine;
    if (shard) {
        shard->z_begin = mine.z_begin;
        shard->z_end = mine.z_end;
        shard->tri_offset = toff;
        shard->vert_offset = indexed ? voff : 0;
        shard->n_tris_total = ttot;
        shard->n_verts_total = indexed ? vtot : 0;
    }
    return MC_OK;
}

}  // extern "C"
#pragma GCC visibility pop
