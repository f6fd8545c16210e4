// profiles/ubench_valu.hip — measurement aid (not product code): issue throughput of the
// integer VALU instructions the Goldilocks kernels are made of, on gfx950.
// Build: hipcc --offload-arch=gfx950 -O3 -o ubench_valu ubench_valu.hip ; run on the GPU box.
// Prints cycles per wave-instruction per SIMD assuming the s_memtime clock (100 MHz ref is not
// used: we report ns per instruction per SIMD-resident wave set and the ratio to v_add_u32).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include <string>

#define REP16(x) x x x x x x x x x x x x x x x x

template <int KIND>
__global__ void __launch_bounds__(256) k(uint64_t* out, int iters) {
    uint32_t a = threadIdx.x * 2654435761u + 1, b = blockIdx.x * 40503u + 7, c = a ^ b, d = a + b;
    uint64_t p = ((uint64_t)a << 32) | b, q = ((uint64_t)c << 32) | d, r = p ^ q, s = p + q;
    for (int i = 0; i < iters; i++) {
        if constexpr (KIND == 0) { REP16(asm volatile("v_add_u32 %0, %0, %1\n v_add_u32 %2, %2, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
        if constexpr (KIND == 1) { REP16(asm volatile("v_mul_lo_u32 %0, %0, %1\n v_mul_lo_u32 %2, %2, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
        if constexpr (KIND == 2) { REP16(asm volatile("v_mul_hi_u32 %0, %0, %1\n v_mul_hi_u32 %2, %2, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
        if constexpr (KIND == 3) { REP16(asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_mad_u64_u32 %1, vcc, %3, %2, %1" : "+v"(p), "+v"(q), "+v"(a), "+v"(b) :: "vcc");) }
        if constexpr (KIND == 4) { REP16(asm volatile("v_lshl_add_u64 %0, %0, 0, %1\n v_lshl_add_u64 %2, %2, 0, %3" : "+v"(p), "+v"(q), "+v"(r), "+v"(s));) }
        if constexpr (KIND == 5) { REP16(asm volatile("v_add_co_u32 %0, vcc, %0, %1\n v_addc_co_u32 %2, vcc, %2, %3, vcc" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) :: "vcc");) }
        if constexpr (KIND == 6) { REP16(asm volatile("v_cmp_lt_u64 vcc, %0, %1\n v_cndmask_b32 %2, %2, %3, vcc" : "+v"(p), "+v"(q), "+v"(a), "+v"(b) :: "vcc");) }
        if constexpr (KIND == 7) { REP16(asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %2, %2, %3, vcc" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) :: "vcc");) }
        if constexpr (KIND == 8) { REP16(asm volatile("v_mul_u32_u24 %0, %0, %1\n v_mad_u32_u24 %2, %2, %3, %2" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
        if constexpr (KIND == 9) { REP16(asm volatile("v_lshlrev_b64 %0, 7, %0\n v_lshrrev_b64 %1, 3, %1" : "+v"(p), "+v"(q));) }
        if constexpr (KIND == 10) { REP16(asm volatile("v_cmp_lt_u32 vcc, %0, %1\n v_cmp_ge_u32 vcc, %2, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) :: "vcc");) }
        if constexpr (KIND == 11) { REP16(asm volatile("v_add3_u32 %0, %0, %1, %2\n v_alignbit_b32 %3, %3, %0, 5" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
        if constexpr (KIND == 12) { REP16(asm volatile("v_fma_f64 %0, %0, %1, %0\n v_fma_f64 %2, %2, %3, %2" : "+v"(p), "+v"(q), "+v"(r), "+v"(s));) }
        if constexpr (KIND == 13) { REP16(asm volatile("v_sub_co_u32 %0, vcc, %0, %1\n v_subbrev_co_u32 %2, vcc, 0, %2, vcc" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) :: "vcc");) }
        if constexpr (KIND == 14) { REP16(asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %1\n v_mad_u64_u32 %1, vcc, %3, %2, %0" : "+v"(p), "+v"(q), "+v"(a), "+v"(b) :: "vcc");) }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a + b + c + d + p + q + r + s;
}

template <int KIND>
double run(uint64_t* d, int blocks, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, 16);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    int blocks = 256 * 8;   // 8 blocks of 4 waves per CU = 8 waves per SIMD
    int iters = 2000;
    uint64_t* d; hipMalloc(&d, (size_t)blocks * 256 * 8);
    const char* names[] = {"v_add_u32 x2", "v_mul_lo_u32 x2", "v_mul_hi_u32 x2", "v_mad_u64_u32 x2 (indep)", "v_lshl_add_u64 x2",
                           "v_add_co+v_addc_co", "v_cmp_lt_u64+v_cndmask", "v_cndmask x2", "v_mul_u32_u24+v_mad_u32_u24",
                           "v_lshlrev_b64+v_lshrrev_b64", "v_cmp_u32 x2", "v_add3_u32+v_alignbit", "v_fma_f64 x2", "v_sub_co+v_subbrev_co",
                           "v_mad_u64_u32 x2 (dep chain)"};
    double ms[15];
    ms[0] = run<0>(d, blocks, iters); ms[1] = run<1>(d, blocks, iters); ms[2] = run<2>(d, blocks, iters); ms[3] = run<3>(d, blocks, iters);
    ms[4] = run<4>(d, blocks, iters); ms[5] = run<5>(d, blocks, iters); ms[6] = run<6>(d, blocks, iters); ms[7] = run<7>(d, blocks, iters);
    ms[8] = run<8>(d, blocks, iters); ms[9] = run<9>(d, blocks, iters); ms[10] = run<10>(d, blocks, iters); ms[11] = run<11>(d, blocks, iters);
    ms[12] = run<12>(d, blocks, iters); ms[13] = run<13>(d, blocks, iters); ms[14] = run<14>(d, blocks, iters);
    // wave-instructions per SIMD: 8 waves/SIMD * iters * 32 instr
    double winst = 8.0 * iters * 32;
    printf("%-34s %10s %14s %8s\n", "pair", "ms", "ns/wave-instr", "x add");
    for (int i = 0; i < 15; i++) printf("%-34s %10.3f %14.3f %8.2f\n", names[i], ms[i], ms[i] * 1e6 / winst, ms[i] / ms[0]);
    return 0;
}
