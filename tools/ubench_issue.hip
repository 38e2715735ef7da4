// Micro-benchmark: issue cost of the instruction kinds the classify kernel is made of.
// Each kernel runs REP x 64 copies of one instruction pattern per wave, 8 waves per SIMD on
// every CU; prints cycles per wave-instruction per SIMD (assuming 2.4 GHz for the conversion,
// and the measured wall time).  Build: hipcc --offload-arch=gfx950 -O2 tools/ubench_issue.hip -o ubench_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP 4096
#define STR2(x) #x
#define STR(x) STR2(x)
#define R4(x) x x x x
#define R16(x) R4(R4(x))
#define R64(x) R4(R16(x))

template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, float a, unsigned long long m) {
    float v0 = threadIdx.x * a, v1 = v0 + 1.f, v2 = v0 + 2.f, v3 = v0 + 3.f, v4 = v0 + 4.f, v5 = v0 + 5.f, v6 = v0 + 6.f, v7 = v0 + 7.f;
    unsigned u0 = threadIdx.x, u1 = u0 + 1;
    unsigned long long s0 = m, s1 = m + 1, s2 = m + 2, s3 = m + 3;
    unsigned q0 = (unsigned)m, q1 = q0 + 1, q2 = q0 + 2, q3 = q0 + 3;
    for (int i = 0; i < REP; ++i) {
        if (KIND == 0) {  // independent v_add_f32 (8 chains)
            asm volatile(R4("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                         "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n")
                         : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(a));
        } else if (KIND == 1) {  // v_max3_f32
            asm volatile(R4("v_max3_f32 %0, %0, %1, %8\n v_max3_f32 %1, %1, %2, %8\n v_max3_f32 %2, %2, %3, %8\n v_max3_f32 %3, %3, %4, %8\n"
                         "v_max3_f32 %4, %4, %5, %8\n v_max3_f32 %5, %5, %6, %8\n v_max3_f32 %6, %6, %7, %8\n v_max3_f32 %7, %7, %0, %8\n")
                         : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(a));
        } else if (KIND == 2) {  // v_cmp -> SGPR pair
            asm volatile(R4("v_cmp_lt_f32_e64 %0, %4, %5\n v_cmp_lt_f32_e64 %1, %4, %6\n v_cmp_lt_f32_e64 %2, %4, %7\n v_cmp_lt_f32_e64 %3, %4, %8\n"
                         "v_cmp_lt_f32_e64 %0, %4, %6\n v_cmp_lt_f32_e64 %1, %4, %7\n v_cmp_lt_f32_e64 %2, %4, %8\n v_cmp_lt_f32_e64 %3, %4, %5\n")
                         : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : "s"(a), "v"(v0), "v"(v1), "v"(v2), "v"(v3));
        } else if (KIND == 3) {  // dependent v_addc chain with SGPR carry-in
            asm volatile(R4("v_addc_co_u32_e64 %0, vcc, %0, %0, %2\n v_addc_co_u32_e64 %0, vcc, %0, %0, %3\n v_addc_co_u32_e64 %0, vcc, %0, %0, %4\n v_addc_co_u32_e64 %0, vcc, %0, %0, %5\n"
                         "v_addc_co_u32_e64 %1, vcc, %1, %1, %2\n v_addc_co_u32_e64 %1, vcc, %1, %1, %3\n v_addc_co_u32_e64 %1, vcc, %1, %1, %4\n v_addc_co_u32_e64 %1, vcc, %1, %1, %5\n")
                         : "+v"(u0), "+v"(u1) : "s"(s0), "s"(s1), "s"(s2), "s"(s3) : "vcc");
        } else if (KIND == 4) {  // v_pk_add_f32
            asm volatile(R4("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n"
                         "v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n")
                         : "+v"(*(double*)&v0), "+v"(*(double*)&v2), "+v"(*(double*)&v4), "+v"(*(double*)&v6) : "v"(*(double*)&s0));
        } else if (KIND == 5) {  // SALU only: s_or_b64
            asm volatile(R4("s_or_b64 %0, %0, %1\n s_or_b64 %1, %1, %2\n s_or_b64 %2, %2, %3\n s_or_b64 %3, %3, %0\n"
                         "s_and_b64 %0, %0, %1\n s_and_b64 %1, %1, %2\n s_and_b64 %2, %2, %3\n s_and_b64 %3, %3, %0\n")
                         : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3));
        } else if (KIND == 6) {  // 1:1 mix of v_add and s_or
            asm volatile(R4("v_add_f32 %0, %0, %8\n s_or_b64 %9, %9, %10\n v_add_f32 %1, %1, %8\n s_or_b64 %10, %10, %11\n v_add_f32 %2, %2, %8\n s_or_b64 %11, %11, %12\n v_add_f32 %3, %3, %8\n s_or_b64 %12, %12, %9\n"
                         "v_add_f32 %4, %4, %8\n s_and_b64 %9, %9, %10\n v_add_f32 %5, %5, %8\n s_and_b64 %10, %10, %11\n v_add_f32 %6, %6, %8\n s_and_b64 %11, %11, %12\n v_add_f32 %7, %7, %8\n s_and_b64 %12, %12, %9\n")
                         : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7), "+v"(a), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3));
        } else if (KIND == 9) {  // v_max_f32 e32 + v_cmp e32 (vcc)
            asm volatile(R4("v_max_f32 %0, %0, %1\n v_max_f32 %1, %1, %2\n v_max_f32 %2, %2, %3\n v_max_f32 %3, %3, %0\n"
                         "v_cmp_lt_f32 vcc, %4, %0\n v_cmp_lt_f32 vcc, %4, %1\n v_cmp_lt_f32 vcc, %4, %2\n v_cmp_lt_f32 vcc, %4, %3\n")
                         : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "s"(a) : "vcc");
        } else if (KIND == 10) {  // v_readlane_b32
            asm volatile(R4("v_readlane_b32 %0, %4, 1\n v_readlane_b32 %1, %4, 2\n v_readlane_b32 %2, %4, 3\n v_readlane_b32 %3, %4, 4\n"
                         "v_readlane_b32 %0, %4, 5\n v_readlane_b32 %1, %4, 6\n v_readlane_b32 %2, %4, 7\n v_readlane_b32 %3, %4, 8\n")
                         : "+s"(q0), "+s"(q1), "+s"(q2), "+s"(q3) : "v"(v0));
        } else if (KIND == 11) {  // 2 SALU per VALU
            asm volatile(R4("v_add_f32 %0, %0, %8\n s_or_b64 %9, %9, %10\n s_and_b64 %10, %10, %11\n v_add_f32 %1, %1, %8\n s_or_b64 %11, %11, %12\n s_and_b64 %12, %12, %9\n"
                         "v_add_f32 %2, %2, %8\n s_lshr_b64 %9, %9, 1\n s_and_b64 %10, %10, %11\n v_add_f32 %3, %3, %8\n s_or_b64 %11, %11, %12\n s_lshr_b64 %12, %12, 1\n")
                         : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7), "+v"(a), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3));
        } else if (KIND == 7) {  // dependent v_add chain (1 chain)
            asm volatile(R4("v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n"
                         "v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n")
                         : "+v"(v0) : "v"(a));
        } else if (KIND == 8) {  // v_cmp e64 followed by dependent scalar op
            asm volatile(R4("v_cmp_lt_f32_e64 %0, %4, %5\n s_or_b64 %1, %1, %0\n v_cmp_lt_f32_e64 %2, %4, %6\n s_or_b64 %3, %3, %2\n"
                         "v_cmp_lt_f32_e64 %0, %4, %7\n s_or_b64 %1, %1, %0\n v_cmp_lt_f32_e64 %2, %4, %8\n s_or_b64 %3, %3, %2\n")
                         : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : "s"(a), "v"(v0), "v"(v1), "v"(v2), "v"(v3));
        }
    }
    if (v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7 + u0 + u1 == 12345.f) out[0] = v0;
    if (threadIdx.x == 0 && blockIdx.x == 0) ((unsigned long long*)out)[1] = s0 ^ s1 ^ s2 ^ s3 ^ q0 ^ q1 ^ q2 ^ q3;
}

template <int KIND>
void run(const char* name, int per_iter, int waves_per_simd) {
    float* d;
    hipMalloc(&d, 64);
    const int blocks = 256 * waves_per_simd;  // 256 CUs x (4 waves per block) x waves_per_simd blocks per CU
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<KIND><<<blocks, 256>>>(d, 1.0f, 3ull);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<KIND><<<blocks, 256>>>(d, 1.0f, 3ull);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double inst_per_simd = (double)REP * per_iter * waves_per_simd;  // wave-instructions issued on one SIMD
    printf("%-34s waves/SIMD %d: %8.3f ms  -> %.2f cycles per wave-instruction per SIMD (at 2.4 GHz)\n", name, waves_per_simd, ms,
           ms * 1e-3 * 2.4e9 / inst_per_simd);
    hipFree(d);
}

int main() {
    for (int w : {2, 8}) {
        run<0>("v_add_f32 x8 independent", 32, w);
        run<7>("v_add_f32 dependent chain", 32, w);
        run<1>("v_max3_f32", 32, w);
        run<2>("v_cmp_lt_f32_e64 -> sgpr", 32, w);
        run<3>("v_addc_co_u32_e64 sgpr carry (dep)", 32, w);
        run<4>("v_pk_add_f32", 32, w);
        run<5>("s_or/s_and_b64", 32, w);
        run<6>("v_add + s_or 1:1 (64 instr)", 64, w);
        run<8>("v_cmp + dependent s_or (16 instr)", 32, w);
        run<9>("v_max_f32 e32 / v_cmp e32 vcc", 32, w);
        run<10>("v_readlane_b32", 32, w);
        run<11>("v_add + 2 SALU (48 instr)", 48, w);
    }
    return 0;
}
