// Probe: does gfx950 execute scalar-memory atomics (s_atomic_add, returned through lgkmcnt)?
// Every wave of the grid adds its value to one counter with ONE scalar atomic and keeps the old value.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(unsigned* counter, unsigned* old_out) {
    const unsigned wave = blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
    unsigned v = __builtin_amdgcn_readfirstlane(wave % 7 + 1);
    unsigned old;
    asm volatile("s_atomic_add %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(old) : "s"(counter), "0"(v) : "memory");
    if ((threadIdx.x & 63) == 0) old_out[wave] = old;
}
int main() {
    unsigned *c, *o;
    const int blocks = 1024, waves = blocks * 4;
    hipMalloc(&c, 4); hipMalloc(&o, waves * 4); hipMemset(c, 0, 4);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, c, o);
    hipError_t e = hipDeviceSynchronize();
    unsigned total = 0; std::vector<unsigned> h(waves);
    hipMemcpy(&total, c, 4, hipMemcpyDeviceToHost); hipMemcpy(h.data(), o, waves * 4, hipMemcpyDeviceToHost);
    unsigned long long want = 0; for (int w = 0; w < waves; ++w) want += w % 7 + 1;
    // the old values must be distinct prefix sums: check max(old + v) == total
    unsigned mx = 0; for (int w = 0; w < waves; ++w) mx = h[w] + (w % 7 + 1) > mx ? h[w] + (w % 7 + 1) : mx;
    printf("sync=%s total=%u want=%llu max(old+v)=%u\n", hipGetErrorString(e), total, want, mx);
    return (total == want && mx == total) ? 0 : 1;
}
