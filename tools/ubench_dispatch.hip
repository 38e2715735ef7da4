// Developer micro-benchmark: what does it cost just to DISPATCH the workgroups of the sweep's kernels (empty bodies,
// the same grid shapes and LDS footprints)?
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_dispatch.hip -o tools/ubench_dispatch && tools/ubench_dispatch
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int LDS>
__global__ void empty(unsigned* out) {
    __shared__ unsigned s[LDS / 4];
    s[threadIdx.x % (LDS / 4)] = threadIdx.x;
    __syncthreads();
    if (s[0] == 0xDEADBEEFu) out[0] = 1u;  // never
}

template <int LDS>
int run(unsigned* d, unsigned blocks, unsigned threads, const char* name) {
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        CHECK(hipEventRecord(a));
        hipLaunchKernelGGL((empty<LDS>), dim3(blocks), dim3(threads), 0, 0, d);
        CHECK(hipEventRecord(b));
        CHECK(hipEventSynchronize(b));
        float ms;
        CHECK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
    }
    printf("%-44s %7u workgroups x %4u threads, %5d B LDS: %8.4f ms  (%.1f ns per workgroup)\n", name, blocks, threads, LDS, best, best * 1e6 / blocks);
    return 0;
}

int main() {
    unsigned* d;
    CHECK(hipMalloc(&d, 256));
    run<23552>(d, 21782, 256, "classify-like (1025^3: 87 k tiles / 4)");
    run<1024>(d, 21782, 256, "  the same with 1 KB of LDS");
    run<24576>(d, 10261, 512, "emit_direct-like (82 k groups / 8)");
    run<1024>(d, 10261, 512, "  the same with 1 KB of LDS");
    run<9024>(d, 82081, 64, "emit-like (82 k groups, one wave each)");
    run<1024>(d, 82081, 64, "  the same with 1 KB of LDS");
    run<1024>(d, 2723, 256, "classify-like, 1/8 slab");
    run<1024>(d, 1283, 512, "emit_direct-like, 1/8 slab");
    run<1024>(d, 1, 64, "one wave");
    return 0;
}
