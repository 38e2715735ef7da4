// Developer micro-benchmark: how fast can MI355X WRITE 1 GiB?  Store width x cache policy x workgroup shape.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_fill.hip -o gpurun_out/ubench_fill && gpurun_out/ubench_fill
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u32;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int AUX, int WIDE>
__global__ __launch_bounds__(256) void fill(unsigned char* p, size_t bytes_per_block, int iters) {
    unsigned char* base = p + (size_t)blockIdx.x * bytes_per_block;
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(base, 0, (int)bytes_per_block, 0x00020000);
    for (int i = 0; i < iters; ++i) {
        if (WIDE) {
            u32x4 v = {1u, 2u, 3u, (u32)i};
            __builtin_amdgcn_raw_buffer_store_b128(v, r, (u32)(i * 4096 + threadIdx.x * 16), 0, AUX);
        } else {
            __builtin_amdgcn_raw_buffer_store_b32((u32)i, r, (u32)(i * 1024 + threadIdx.x * 4), 0, AUX);
        }
    }
}

// mc_classify's store pattern without its arithmetic: a workgroup of 4 waves = the 4 x-chunks (256 B each) of a block of 63
// rows of 1024 B; a wave writes its 256-byte column 4 rows per dwordx4 store (lane 4k+s: the 16 bytes of lanes 4k..4k+3 in
// row j+s) -- or row by row with dword stores.
template <int AUX, int BLOCKS4>
__global__ __launch_bounds__(256) void tile_pattern(unsigned char* p, int rows_total) {
    const int lane = threadIdx.x & 63, ch = threadIdx.x >> 6;
    const int y0 = blockIdx.x * 63;
    const int ny = min(63, rows_total - y0);
    unsigned char* tilebase = p + (size_t)y0 * 1024;
    const int q = lane & 3, k4 = lane & ~3;
    if (BLOCKS4) {
        int j = 0;
        for (; j + 4 <= ny; j += 4) {
            u32x4 v = {1u, 2u, 3u, (u32)j};
            const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(tilebase + (size_t)j * 1024, 0, 4096, 0x00020000);
            __builtin_amdgcn_raw_buffer_store_b128(v, r, (u32)(q * 1024 + ch * 256 + k4 * 4), 0, AUX);
        }
        for (; j < ny; ++j) {
            const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(tilebase + (size_t)j * 1024, 0, 1024, 0x00020000);
            __builtin_amdgcn_raw_buffer_store_b32((u32)j, r, (u32)(ch * 256 + lane * 4), 0, AUX);
        }
    } else {
        for (int j = 0; j < ny; ++j) {
            const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(tilebase + (size_t)j * 1024, 0, 1024, 0x00020000);
            __builtin_amdgcn_raw_buffer_store_b32((u32)j, r, (u32)(ch * 256 + lane * 4), 0, AUX);
        }
    }
}

// the emit kernels' store pattern: 24 bytes per lane (one vertex) as a 16-byte and an 8-byte store, 1536 contiguous bytes
// per pair of instructions -- against the same bytes written as contiguous 16-byte pieces (what an LDS transposition in
// front of the stores would produce: 1.5 instructions per 64 vertices)
template <int TRANSPOSED>
__global__ __launch_bounds__(64) void vertex_pattern(unsigned char* p, int iters) {
    const int lane = threadIdx.x;
    unsigned char* base = p + (size_t)blockIdx.x * iters * 1536;
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(base, 0, iters * 1536, 0x00020000);
    typedef u32 u32x2 __attribute__((ext_vector_type(2)));
    for (int i = 0; i < iters; ++i) {
        u32x4 a = {1u, 2u, 3u, (u32)i};
        u32x2 b = {5u, 6u};
        if (TRANSPOSED) {
            __builtin_amdgcn_raw_buffer_store_b128(a, r, (u32)(i * 1536 + lane * 16), 0, 0);
            if (lane < 32) __builtin_amdgcn_raw_buffer_store_b128(a, r, (u32)(i * 1536 + 1024 + lane * 16), 0, 0);
        } else {
            __builtin_amdgcn_raw_buffer_store_b128(a, r, (u32)(i * 1536 + lane * 24), 0, 0);
            __builtin_amdgcn_raw_buffer_store_b64(b, r, (u32)(i * 1536 + lane * 24 + 16), 0, 0);
        }
    }
}
// one TRIANGLE per lane: three vertices of 24 bytes each, 72 contiguous bytes per lane, written as 3 x (16 + 8) bytes
__global__ __launch_bounds__(64) void triangle_pattern(unsigned char* p, int iters) {
    const int lane = threadIdx.x;
    unsigned char* base = p + (size_t)blockIdx.x * iters * 4608;
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(base, 0, iters * 4608, 0x00020000);
    typedef u32 u32x2 __attribute__((ext_vector_type(2)));
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            u32x4 a = {1u, 2u, 3u, (u32)i};
            u32x2 b = {5u, (u32)k};
            __builtin_amdgcn_raw_buffer_store_b128(a, r, (u32)(i * 4608 + lane * 72 + k * 24), 0, 0);
            __builtin_amdgcn_raw_buffer_store_b64(b, r, (u32)(i * 4608 + lane * 72 + k * 24 + 16), 0, 0);
        }
    }
}
int run_triangle(unsigned char* d, size_t total, int iters, const char* name) {
    const unsigned blocks = (unsigned)(total / ((size_t)iters * 4608));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        CHECK(hipEventRecord(a));
        hipLaunchKernelGGL(triangle_pattern, dim3(blocks), dim3(64), 0, 0, d, iters);
        CHECK(hipEventRecord(b));
        CHECK(hipEventSynchronize(b));
        float ms;
        CHECK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
    }
    printf("%-44s iters %3d %8.4f ms  %7.1f GB/s\n", name, iters, best, (double)blocks * iters * 4608 / best / 1e6);
    return 0;
}

template <int TRANSPOSED>
int run_vertex(unsigned char* d, size_t total, int iters, const char* name) {
    const unsigned blocks = (unsigned)(total / ((size_t)iters * 1536));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        CHECK(hipEventRecord(a));
        hipLaunchKernelGGL((vertex_pattern<TRANSPOSED>), dim3(blocks), dim3(64), 0, 0, d, iters);
        CHECK(hipEventRecord(b));
        CHECK(hipEventSynchronize(b));
        float ms;
        CHECK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
    }
    printf("%-44s iters %3d %8.4f ms  %7.1f GB/s\n", name, iters, best, (double)blocks * iters * 1536 / best / 1e6);
    return 0;
}

template <int AUX, int BLOCKS4>
int run_pattern(unsigned char* d, size_t total, const char* name) {
    const int rows = (int)(total / 1024);
    const unsigned blocks = (unsigned)((rows + 62) / 63);
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        CHECK(hipEventRecord(a));
        hipLaunchKernelGGL((tile_pattern<AUX, BLOCKS4>), dim3(blocks), dim3(256), 0, 0, d, rows);
        CHECK(hipEventRecord(b));
        CHECK(hipEventSynchronize(b));
        float ms;
        CHECK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
    }
    printf("%-44s %8.4f ms  %7.1f GB/s\n", name, best, (double)rows * 1024 / best / 1e6);
    return 0;
}

template <int AUX, int WIDE>
int run(unsigned char* d, size_t total, int iters, const char* name) {
    const size_t per_block = (size_t)iters * (WIDE ? 4096 : 1024);
    const unsigned blocks = (unsigned)(total / per_block);
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        CHECK(hipEventRecord(a));
        hipLaunchKernelGGL((fill<AUX, WIDE>), dim3(blocks), dim3(256), 0, 0, d, per_block, iters);
        CHECK(hipEventRecord(b));
        CHECK(hipEventSynchronize(b));
        float ms;
        CHECK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
    }
    printf("%-28s iters %3d  %8.4f ms  %7.1f GB/s\n", name, iters, best, (double)blocks * per_block / best / 1e6);
    return 0;
}

int main() {
    const size_t total = (size_t)1 << 30;
    unsigned char* d;
    CHECK(hipMalloc(&d, total));
    CHECK(hipMemset(d, 0, total));
    for (int iters : {4, 16, 64}) {
        run<0, 1>(d, total, iters, "b128 aux0");
        run<1, 1>(d, total, iters, "b128 sc0");
        run<2, 1>(d, total, iters, "b128 nt");
        run<3, 1>(d, total, iters, "b128 sc0|nt");
        run<16, 1>(d, total, iters, "b128 sc1");
        run<18, 1>(d, total, iters, "b128 sc1|nt");
        run<0, 0>(d, total, iters, "b32  aux0");
        run<2, 0>(d, total, iters, "b32  nt");
    }
    for (int iters : {6, 24}) {
        run_vertex<0>(d, total, iters, "emit vertex pattern, 16 + 8 bytes per lane");
        run_vertex<1>(d, total, iters, "the same bytes, contiguous 16-byte pieces");
    }
    run_triangle(d, total, 2, "one triangle per lane, 3 x (16 + 8) bytes");
    run_triangle(d, total, 8, "one triangle per lane, 3 x (16 + 8) bytes");
    run_pattern<0, 1>(d, total, "classify tile pattern, 4-row blocks, aux0");
    run_pattern<2, 1>(d, total, "classify tile pattern, 4-row blocks, nt");
    run_pattern<0, 0>(d, total, "classify tile pattern, row by row, aux0");
    run_pattern<2, 0>(d, total, "classify tile pattern, row by row, nt");
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    CHECK(hipEventRecord(a));
    CHECK(hipMemsetAsync(d, 1, total, 0));
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms;
    CHECK(hipEventElapsedTime(&ms, a, b));
    printf("hipMemsetAsync               %8.4f ms  %7.1f GB/s\n", ms, total / ms / 1e6);
    return 0;
}
