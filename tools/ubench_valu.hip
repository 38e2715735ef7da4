// Micro-benchmark: what ONE wave-instruction of each kind in the emit kernels' vertex loop costs a SIMD (gfx950).
// Every kernel runs REP x 32 copies of one instruction over four independent registers, W waves per SIMD on every CU;
// the figure printed is SIMD cycles per wave-instruction (wall time x clock / instructions issued on one SIMD; the clock
// is taken from hipDeviceProp, so read the RATIOS).  The point: the compiler counts instructions, the SIMD counts issue
// cycles -- and a VOP3-encoded select or bit-field extract is not priced like a v_add_f32.
// Build: hipcc --offload-arch=gfx950 -O2 tools/ubench_valu.hip -o tools/ubench_valu
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP 2048
#define R4(x) x x x x

// one pattern = 8 instructions over v0..v3 (%0..%3), a (%4, VGPR), s (%5, SGPR pair), u (%6, SGPR 32)
#define PATTERNS(X)                                                                                                         \
    X(0, "v_add_f32_e32", "v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4\n")                        \
    X(1, "v_mul_f32_e32", "v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4\n")                        \
    X(2, "v_and_b32_e32", "v_and_b32 %0, %0, %4\n v_and_b32 %1, %1, %4\n v_and_b32 %2, %2, %4\n v_and_b32 %3, %3, %4\n")                        \
    X(3, "v_lshrrev_b32_e32 (imm)", "v_lshrrev_b32 %0, 1, %0\n v_lshrrev_b32 %1, 1, %1\n v_lshrrev_b32 %2, 1, %2\n v_lshrrev_b32 %3, 1, %3\n")  \
    X(4, "v_add_u32_e32", "v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n")                        \
    X(5, "v_cndmask_b32_e32 (vcc)", "v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc\n") \
    X(6, "v_cndmask_b32_e64 (sgpr pair)", "v_cndmask_b32_e64 %0, %0, %4, %5\n v_cndmask_b32_e64 %1, %1, %4, %5\n v_cndmask_b32_e64 %2, %2, %4, %5\n v_cndmask_b32_e64 %3, %3, %4, %5\n") \
    X(7, "v_bfe_u32", "v_bfe_u32 %0, %0, 3, 8\n v_bfe_u32 %1, %1, 3, 8\n v_bfe_u32 %2, %2, 3, 8\n v_bfe_u32 %3, %3, 3, 8\n")                    \
    X(8, "v_lshl_add_u32", "v_lshl_add_u32 %0, %0, 2, %4\n v_lshl_add_u32 %1, %1, 2, %4\n v_lshl_add_u32 %2, %2, 2, %4\n v_lshl_add_u32 %3, %3, 2, %4\n") \
    X(9, "v_fma_f32", "v_fma_f32 %0, %0, %4, %4\n v_fma_f32 %1, %1, %4, %4\n v_fma_f32 %2, %2, %4, %4\n v_fma_f32 %3, %3, %4, %4\n")            \
    X(10, "v_fmac_f32_e32", "v_fmac_f32 %0, %4, %4\n v_fmac_f32 %1, %4, %4\n v_fmac_f32 %2, %4, %4\n v_fmac_f32 %3, %4, %4\n")                  \
    X(11, "v_mul_hi_u32", "v_mul_hi_u32 %0, %0, %4\n v_mul_hi_u32 %1, %1, %4\n v_mul_hi_u32 %2, %2, %4\n v_mul_hi_u32 %3, %3, %4\n")            \
    X(12, "v_mul_lo_u32", "v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %4\n")            \
    X(13, "v_mad_u32_u24", "v_mad_u32_u24 %0, %0, 3, %4\n v_mad_u32_u24 %1, %1, 3, %4\n v_mad_u32_u24 %2, %2, 3, %4\n v_mad_u32_u24 %3, %3, 3, %4\n") \
    X(14, "v_and_or_b32", "v_and_or_b32 %0, %0, %4, %4\n v_and_or_b32 %1, %1, %4, %4\n v_and_or_b32 %2, %2, %4, %4\n v_and_or_b32 %3, %3, %4, %4\n") \
    X(15, "v_add3_u32", "v_add3_u32 %0, %0, %4, %4\n v_add3_u32 %1, %1, %4, %4\n v_add3_u32 %2, %2, %4, %4\n v_add3_u32 %3, %3, %4, %4\n")      \
    X(16, "v_add_f32_e64 (|x| modifier)", "v_add_f32_e64 %0, |%0|, %4\n v_add_f32_e64 %1, |%1|, %4\n v_add_f32_e64 %2, |%2|, %4\n v_add_f32_e64 %3, |%3|, %4\n") \
    X(17, "v_add_f32 + 32-bit literal", "v_add_f32 %0, 0x3f9d70a4, %0\n v_add_f32 %1, 0x3f9d70a4, %1\n v_add_f32 %2, 0x3f9d70a4, %2\n v_add_f32 %3, 0x3f9d70a4, %3\n") \
    X(18, "v_add_f32 sgpr operand", "v_add_f32 %0, %6, %0\n v_add_f32 %1, %6, %1\n v_add_f32 %2, %6, %2\n v_add_f32 %3, %6, %3\n")               \
    X(19, "v_cmp_lt_f32_e32 (vcc)", "v_cmp_lt_f32 vcc, %0, %4\n v_cmp_lt_f32 vcc, %1, %4\n v_cmp_lt_f32 vcc, %2, %4\n v_cmp_lt_f32 vcc, %3, %4\n") \
    X(20, "v_cmp_lt_f32_e64 (sgpr pair)", "v_cmp_lt_f32_e64 %5, %0, %4\n v_cmp_lt_f32_e64 %5, %1, %4\n v_cmp_lt_f32_e64 %5, %2, %4\n v_cmp_lt_f32_e64 %5, %3, %4\n") \
    X(21, "v_rcp_f32", "v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n")                                            \
    X(22, "v_rsq_f32", "v_rsq_f32 %0, %0\n v_rsq_f32 %1, %1\n v_rsq_f32 %2, %2\n v_rsq_f32 %3, %3\n")                                            \
    X(23, "v_div_scale_f32", "v_div_scale_f32 %0, vcc, %0, %4, %0\n v_div_scale_f32 %1, vcc, %1, %4, %1\n v_div_scale_f32 %2, vcc, %2, %4, %2\n v_div_scale_f32 %3, vcc, %3, %4, %3\n") \
    X(24, "v_div_fmas_f32", "v_div_fmas_f32 %0, %0, %4, %4\n v_div_fmas_f32 %1, %1, %4, %4\n v_div_fmas_f32 %2, %2, %4, %4\n v_div_fmas_f32 %3, %3, %4, %4\n") \
    X(25, "v_div_fixup_f32", "v_div_fixup_f32 %0, %0, %4, %4\n v_div_fixup_f32 %1, %1, %4, %4\n v_div_fixup_f32 %2, %2, %4, %4\n v_div_fixup_f32 %3, %3, %4, %4\n") \
    X(26, "v_mov_b32 dpp row_shr:1", "v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n") \
    X(27, "v_add_f32 dpp row_shr:1", "v_add_f32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %1, %2, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %2, %3, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %3, %0, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n") \
    X(28, "v_max3_f32", "v_max3_f32 %0, %0, %4, %4\n v_max3_f32 %1, %1, %4, %4\n v_max3_f32 %2, %2, %4, %4\n v_max3_f32 %3, %3, %4, %4\n")      \
    X(29, "v_mov_b32_e32", "v_mov_b32 %0, %4\n v_mov_b32 %1, %4\n v_mov_b32 %2, %4\n v_mov_b32 %3, %4\n")                                        \
    X(30, "v_sub_f32 x2 + v_mul_f32 x2 (mix)", "v_sub_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_sub_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4\n")    \
    X(31, "v_cndmask e64 + v_add_f32 1:1", "v_cndmask_b32_e64 %0, %0, %4, %5\n v_add_f32 %1, %1, %4\n v_cndmask_b32_e64 %2, %2, %4, %5\n v_add_f32 %3, %3, %4\n") \
    X(32, "v_bfe_u32 + v_add_f32 1:1", "v_bfe_u32 %0, %0, 3, 8\n v_add_f32 %1, %1, %4\n v_bfe_u32 %2, %2, 3, 8\n v_add_f32 %3, %3, %4\n")       \
    X(33, "v_lshlrev_b32 (vgpr shift)", "v_lshlrev_b32 %0, %4, %0\n v_lshlrev_b32 %1, %4, %1\n v_lshlrev_b32 %2, %4, %2\n v_lshlrev_b32 %3, %4, %3\n") \
    X(34, "v_xor_b32_e32", "v_xor_b32 %0, %0, %4\n v_xor_b32 %1, %1, %4\n v_xor_b32 %2, %2, %4\n v_xor_b32 %3, %3, %4\n")                        \
    X(35, "v_perm_b32", "v_perm_b32 %0, %0, %4, %4\n v_perm_b32 %1, %1, %4, %4\n v_perm_b32 %2, %2, %4, %4\n v_perm_b32 %3, %3, %4, %4\n") \
    X(36, "mix fma add add add", "v_fma_f32 %0, %0, %4, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4\n")         \
    X(37, "mix fma fma add add", "v_fma_f32 %0, %0, %4, %4\n v_fma_f32 %1, %1, %4, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4\n")   \
    X(38, "mix fma add fma add", "v_fma_f32 %0, %0, %4, %4\n v_add_f32 %1, %1, %4\n v_fma_f32 %2, %2, %4, %4\n v_add_f32 %3, %3, %4\n")   \
    X(39, "v_cmp vcc + 3 cndmask_e32 vcc", "v_cmp_lt_f32 vcc, %0, %4\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc\n") \
    X(40, "mix rcp add add add", "v_rcp_f32 %0, %0\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4\n")               \
    X(41, "mix cndmask_e64 x3 + add", "v_cndmask_b32_e64 %0, %0, %4, %5\n v_cndmask_b32_e64 %1, %1, %4, %5\n v_cndmask_b32_e64 %2, %2, %4, %5\n v_add_f32 %3, %3, %4\n") \
    X(42, "mix bfe cndmask_e64 lshl_add fma", "v_bfe_u32 %0, %0, 3, 8\n v_cndmask_b32_e64 %1, %1, %4, %5\n v_lshl_add_u32 %2, %2, 2, %4\n v_fma_f32 %3, %3, %4, %4\n") \
    X(43, "dependent chain add x4 (one reg)", "v_add_f32 %0, %0, %4\n v_add_f32 %0, %0, %4\n v_add_f32 %0, %0, %4\n v_add_f32 %0, %0, %4\n") \
    X(44, "dependent chain fma x4 (one reg)", "v_fma_f32 %0, %0, %4, %4\n v_fma_f32 %0, %0, %4, %4\n v_fma_f32 %0, %0, %4, %4\n v_fma_f32 %0, %0, %4, %4\n") \
    X(45, "v_cndmask_b32_e64 with vcc as mask", "v_cndmask_b32_e64 %0, %0, %4, vcc\n v_cndmask_b32_e64 %1, %1, %4, vcc\n v_cndmask_b32_e64 %2, %2, %4, vcc\n v_cndmask_b32_e64 %3, %3, %4, vcc\n") \
    X(46, "v_cmp_e64 sgpr + 3 cndmask_e64 sgpr", "v_cmp_lt_f32_e64 %5, %0, %4\n v_cndmask_b32_e64 %1, %1, %4, %5\n v_cndmask_b32_e64 %2, %2, %4, %5\n v_cndmask_b32_e64 %3, %3, %4, %5\n") \
    X(47, "v_cmp vcc, s_nop 1, cndmask vcc, add, add", "v_cmp_lt_f32 vcc, %0, %4\n s_nop 1\n v_cndmask_b32 %1, %1, %4, vcc\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4\n") \
    X(48, "v_cmp_e64 sgpr, cndmask_e64, add, add", "v_cmp_lt_f32_e64 %5, %0, %4\n v_cndmask_b32_e64 %1, %1, %4, %5\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4\n") \
    X(49, "v_addc_co_u32 vcc chain", "v_add_co_u32 %0, vcc, %0, %4\n v_addc_co_u32 %1, vcc, %1, %4, vcc\n v_add_co_u32 %2, vcc, %2, %4\n v_addc_co_u32 %3, vcc, %3, %4, vcc\n") \
    X(50, "div_scale vcc + div_fmas + 2 add", "v_div_scale_f32 %0, vcc, %0, %4, %0\n v_div_fmas_f32 %1, %1, %4, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4\n")

template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, float af, unsigned long long m) {
    float v0 = threadIdx.x * af, v1 = v0 + 1.f, v2 = v0 + 2.f, v3 = v0 + 3.f;
    float a = af + 0.25f;
    unsigned long long s = m;
    unsigned u = (unsigned)m + 7u;
    for (int i = 0; i < REP; ++i) {
#define X(ID, NAME, TEXT)                                                                                                          \
        if (KIND == ID) asm volatile(R4(TEXT TEXT) : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(a), "+s"(s), "+s"(u)::"vcc");
        PATTERNS(X)
#undef X
    }
    if (v0 + v1 + v2 + v3 == 12345.f) out[0] = v0 + (float)s + (float)u;
}

// LDS reads: each lane reads its own dword / 8 bytes / 16 bytes (conflict-free), 32 reads per iteration, one wait per iteration
template <int BYTES>
__global__ __launch_bounds__(256) void klds(float* out, int n) {
    __shared__ float buf[256 * 4 + 64];
    for (int i = threadIdx.x; i < 256 * 4 + 64; i += 256) buf[i] = (float)i;
    __syncthreads();
    float acc = 0.f;
    const unsigned addr = threadIdx.x * BYTES;
    for (int i = 0; i < n; ++i) {
        if (BYTES == 4) {
            float x0, x1, x2, x3;
            asm volatile(R4("ds_read_b32 %0, %4\n ds_read_b32 %1, %4 offset:4\n ds_read_b32 %2, %4 offset:8\n ds_read_b32 %3, %4 offset:12\n ds_read_b32 %0, %4 offset:16\n ds_read_b32 %1, %4 offset:20\n ds_read_b32 %2, %4 offset:24\n ds_read_b32 %3, %4 offset:28\n")
                         "s_waitcnt lgkmcnt(0)\n" : "=&v"(x0), "=&v"(x1), "=&v"(x2), "=&v"(x3) : "v"(addr));
            acc += x0 + x1 + x2 + x3;
        } else if (BYTES == 8) {
            double x0, x1, x2, x3;
            asm volatile(R4("ds_read_b64 %0, %4\n ds_read_b64 %1, %4 offset:8\n ds_read_b64 %2, %4 offset:16\n ds_read_b64 %3, %4 offset:24\n ds_read_b64 %0, %4 offset:32\n ds_read_b64 %1, %4 offset:40\n ds_read_b64 %2, %4 offset:48\n ds_read_b64 %3, %4 offset:56\n")
                         "s_waitcnt lgkmcnt(0)\n" : "=&v"(x0), "=&v"(x1), "=&v"(x2), "=&v"(x3) : "v"(addr));
            acc += (float)(x0 + x1 + x2 + x3);
        }
    }
    if (acc == 12345.f) out[0] = acc;
}

static double g_clock_hz = 2.4e9;

template <int KIND>
void run(const char* name, int waves_per_simd) {
    float* d;
    (void)hipMalloc(&d, 64);
    const int blocks = 256 * waves_per_simd;  // 256 CUs x (4 waves per block) x waves_per_simd blocks per CU
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    k<KIND><<<blocks, 256>>>(d, 1.0f, 3ull);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<KIND><<<blocks, 256>>>(d, 1.0f, 3ull);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double inst_per_simd = (double)REP * 32 * waves_per_simd;
    printf("%-36s waves/SIMD %d: %8.3f ms  -> %5.2f cycles per wave-instruction per SIMD\n", name, waves_per_simd, ms,
           ms * 1e-3 * g_clock_hz / inst_per_simd);
    (void)hipFree(d);
}

template <int BYTES>
void run_lds(const char* name, int waves_per_simd) {
    float* d;
    (void)hipMalloc(&d, 64);
    const int blocks = 256 * waves_per_simd, n = 1024;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    klds<BYTES><<<blocks, 256>>>(d, n);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    klds<BYTES><<<blocks, 256>>>(d, n);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double inst_per_cu = (double)n * 32 * waves_per_simd * 4;
    printf("%-36s waves/SIMD %d: %8.3f ms  -> %5.2f cycles per wave-instruction per CU\n", name, waves_per_simd, ms,
           ms * 1e-3 * g_clock_hz / inst_per_cu);
    (void)hipFree(d);
}

int main() {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) == hipSuccess && prop.clockRate > 0) g_clock_hz = (double)prop.clockRate * 1e3;
    printf("clock taken as %.0f MHz\n", g_clock_hz / 1e6);
    for (int w : {8, 2, 1}) {
#define X(ID, NAME, TEXT) run<ID>(NAME, w);
        PATTERNS(X)
#undef X
        run_lds<4>("ds_read_b32", w);
        run_lds<8>("ds_read_b64", w);
    }
    return 0;
}
