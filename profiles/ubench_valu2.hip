// profiles/ubench_valu2.hip — measurement aid (not product code): per-instruction issue cost of
// integer VALU ops on gfx950 with the shader clock measured in-kernel (s_memtime / s_memrealtime).
// One instruction type per loop, 4 independent dependency chains per wave, 8 waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define REP8(x) x x x x x x x x
#define BODY4(ins) REP8(asm volatile(ins "\n" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(p), "+v"(q), "+v"(r), "+v"(s) :: "vcc", "s20", "s21", "s22", "s23");)

template <int KIND>
__global__ void __launch_bounds__(256) k(uint64_t* out, int iters, unsigned long long* clk) {
    uint32_t a = threadIdx.x * 2654435761u + 1, b = blockIdx.x * 40503u + 7, c = a ^ b, d = a + b;
    uint64_t p = ((uint64_t)a << 32) | b, q = ((uint64_t)c << 32) | d, r = p ^ q, s = p + q;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    asm volatile("s_mov_b32 vcc_lo, 0x55555555\n s_mov_b32 vcc_hi, 0x55555555" ::: "vcc");
    for (int i = 0; i < iters; i++) {
        if constexpr (KIND == 0) { BODY4("v_add_u32 %0, %0, %1\n v_add_u32 %1, %1, %2\n v_add_u32 %2, %2, %3\n v_add_u32 %3, %3, %0") }
        if constexpr (KIND == 1) { BODY4("v_and_b32 %0, %0, %1\n v_xor_b32 %1, %1, %2\n v_or_b32 %2, %2, %3\n v_sub_u32 %3, %3, %0") }
        if constexpr (KIND == 2) { BODY4("v_lshlrev_b32 %0, 3, %0\n v_lshrrev_b32 %1, 1, %1\n v_lshlrev_b32 %2, 5, %2\n v_lshrrev_b32 %3, 2, %3") }
        if constexpr (KIND == 3) { BODY4("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %0, vcc") }
        if constexpr (KIND == 4) { BODY4("v_add_co_u32 %0, s[20:21], %0, %1\n v_add_co_u32 %1, s[22:23], %1, %2\n v_add_co_u32 %2, s[20:21], %2, %3\n v_add_co_u32 %3, s[22:23], %3, %0") }
        if constexpr (KIND == 5) { BODY4("v_lshl_add_u64 %4, %4, 0, %5\n v_lshl_add_u64 %5, %5, 0, %6\n v_lshl_add_u64 %6, %6, 0, %7\n v_lshl_add_u64 %7, %7, 0, %4") }
        if constexpr (KIND == 6) { BODY4("v_mad_u64_u32 %4, s[20:21], %0, %1, %4\n v_mad_u64_u32 %5, s[22:23], %1, %2, %5\n v_mad_u64_u32 %6, s[20:21], %2, %3, %6\n v_mad_u64_u32 %7, s[22:23], %3, %0, %7") }
        if constexpr (KIND == 7) { BODY4("v_mul_lo_u32 %0, %0, %1\n v_mul_lo_u32 %1, %1, %2\n v_mul_lo_u32 %2, %2, %3\n v_mul_lo_u32 %3, %3, %0") }
        if constexpr (KIND == 8) { BODY4("v_cmp_lt_u64 s[20:21], %4, %5\n v_cmp_lt_u64 s[22:23], %5, %6\n v_cmp_lt_u64 s[20:21], %6, %7\n v_cmp_lt_u64 s[22:23], %7, %4") }
        if constexpr (KIND == 9) { BODY4("v_cmp_lt_u32 s[20:21], %0, %1\n v_cmp_lt_u32 s[22:23], %1, %2\n v_cmp_lt_u32 s[20:21], %2, %3\n v_cmp_lt_u32 s[22:23], %3, %0") }
        if constexpr (KIND == 10) { BODY4("v_add3_u32 %0, %0, %1, %2\n v_add3_u32 %1, %1, %2, %3\n v_add3_u32 %2, %2, %3, %0\n v_add3_u32 %3, %3, %0, %1") }
        if constexpr (KIND == 11) { BODY4("v_alignbit_b32 %0, %0, %1, 7\n v_alignbit_b32 %1, %1, %2, 9\n v_alignbit_b32 %2, %2, %3, 3\n v_alignbit_b32 %3, %3, %0, 5") }
        if constexpr (KIND == 12) { BODY4("v_mad_u32_u24 %0, %0, %1, %2\n v_mad_u32_u24 %1, %1, %2, %3\n v_mad_u32_u24 %2, %2, %3, %0\n v_mad_u32_u24 %3, %3, %0, %1") }
        if constexpr (KIND == 13) { BODY4("v_lshlrev_b64 %4, 7, %4\n v_lshrrev_b64 %5, 3, %5\n v_lshlrev_b64 %6, 5, %6\n v_lshrrev_b64 %7, 1, %7") }
        if constexpr (KIND == 14) { BODY4("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %0") }
        if constexpr (KIND == 15) { BODY4("v_addc_co_u32 %0, vcc, %0, %1, vcc\n v_addc_co_u32 %1, vcc, %1, %2, vcc\n v_addc_co_u32 %2, vcc, %2, %3, vcc\n v_addc_co_u32 %3, vcc, %3, %0, vcc") }
        if constexpr (KIND == 16) { BODY4("v_mul_hi_u32 %0, %0, %1\n v_mul_hi_u32 %1, %1, %2\n v_mul_hi_u32 %2, %2, %3\n v_mul_hi_u32 %3, %3, %0") }
        if constexpr (KIND == 17) { BODY4("v_pk_add_u16 %0, %0, %1\n v_pk_add_u16 %1, %1, %2\n v_pk_add_u16 %2, %2, %3\n v_pk_add_u16 %3, %3, %0") }
        if constexpr (KIND == 18) { BODY4("v_mov_b64 %4, %5\n v_mov_b64 %5, %6\n v_mov_b64 %6, %7\n v_mov_b64 %7, %4") }
        if constexpr (KIND == 19) { BODY4("v_sub_co_u32 %0, vcc, %0, %1\n v_subb_co_u32 %1, vcc, %1, %2, vcc\n v_sub_co_u32 %2, vcc, %2, %3\n v_subb_co_u32 %3, vcc, %3, %0, vcc") }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * 256 + threadIdx.x] = a + b + c + d + p + q + r + s;
    if (threadIdx.x == 0 && blockIdx.x == 5) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <int KIND>
void run(uint64_t* d, unsigned long long* clk, const char* name) {
    const int blocks = 256 * 8, iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, 200, clk);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, iters, clk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    double ghz = (double)h[0] / (double)h[1] * 0.1;                 // s_memrealtime ticks at 100 MHz
    double winst = 8.0 * iters * 32;                                // wave-instructions per SIMD
    double ns = ms * 1e6 / winst;
    printf("%-28s %8.3f ms  %6.3f ns/instr  clock %5.2f GHz  %5.2f cycles/instr\n", name, ms, ns, ghz, ns * ghz);
}

int main() {
    uint64_t* d; hipMalloc(&d, (size_t)2048 * 256 * 8);
    unsigned long long* clk; hipMalloc(&clk, 16);
    run<0>(d, clk, "v_add_u32"); run<1>(d, clk, "and/xor/or/sub_u32"); run<2>(d, clk, "v_lsh*_b32"); run<3>(d, clk, "v_cndmask_b32 (vcc)");
    run<4>(d, clk, "v_add_co_u32 (sgpr carry)"); run<5>(d, clk, "v_lshl_add_u64"); run<6>(d, clk, "v_mad_u64_u32"); run<7>(d, clk, "v_mul_lo_u32");
    run<8>(d, clk, "v_cmp_lt_u64"); run<9>(d, clk, "v_cmp_lt_u32"); run<10>(d, clk, "v_add3_u32"); run<11>(d, clk, "v_alignbit_b32");
    run<12>(d, clk, "v_mad_u32_u24"); run<13>(d, clk, "v_lsh*_b64"); run<14>(d, clk, "v_mov_b32"); run<15>(d, clk, "v_addc_co_u32 (vcc chain)");
    run<16>(d, clk, "v_mul_hi_u32"); run<17>(d, clk, "v_pk_add_u16"); run<18>(d, clk, "v_mov_b64"); run<19>(d, clk, "v_sub_co+v_subb_co pairs");
    return 0;
}
