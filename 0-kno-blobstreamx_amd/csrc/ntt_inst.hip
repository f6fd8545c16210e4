// ntt_inst.hip — explicit instantiations of glp_ntt_pass_kernel for ONE tile size
// (compiled once per GLP_INST_LOG_R by the Makefile so the instantiations build in parallel).
#include <hip/hip_runtime.h>
#include "ntt_kernels.cuh"
#include "ntt_launch.h"

#ifndef GLP_INST_LOG_R
#error "compile with -DGLP_INST_LOG_R=<6..12>"
#endif

template <int MODE, bool INV, int LOG_E, bool PLAIN>
static hipError_t launch_kern(unsigned grid, unsigned block, size_t lds, hipStream_t st, const GlpNttPassArgs& a) {
    auto kern = glp_ntt_pass_kernel<GLP_INST_LOG_R, MODE, INV, LOG_E, PLAIN>;
    static bool attr_done = false;   // one ctx per process per GPU: no concurrent first call
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(block), lds, st, a);
    return hipGetLastError();
}
template <int MODE, bool INV, int LOG_E>
static hipError_t launch_one(unsigned grid, unsigned block, size_t lds, hipStream_t st, const GlpNttPassArgs& a) {
    if constexpr (MODE == GLP_STRIP && LOG_E == 5 && GLP_INST_LOG_R == 10) {
        // the strip kernel of the default plan for large 2^20 batches, tile width as a compile-time constant (LDS offsets as immediates:
        // 126 VGPRs and no spills instead of 128 + 8 spilled; 0.58 instead of 0.66 ms).  FINAL_T stays on the runtime-width kernel: its
        // compile-time form needs only 106 VGPRs, runs four waves per SIMD and measured SLOWER (0.53 vs 0.50 ms) than the 142-VGPR form at three
        constexpr int CTC = 4;
        if (glp_ntt_args_plain(a) && a.log_c == (unsigned)CTC) {
            auto kern = glp_ntt_pass_kernel<GLP_INST_LOG_R, MODE, INV, LOG_E, true, CTC>;
            static bool attr_done_ct = false;
            if (!attr_done_ct) {
                hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                if (e != hipSuccess) return e;
                attr_done_ct = true;
            }
            hipLaunchKernelGGL(kern, dim3(grid), dim3(block), lds, st, a);
            return hipGetLastError();
        }
    }
    if constexpr (MODE != GLP_FINAL_ROWS && LOG_E == 5) {
        // the PLAIN instantiation whenever no optional feature is asked for (see ntt_kernels.cuh).  Radix-32 work-items only: on the
        // radix-16 kernels the plain form measured SLOWER (strip 0.83 vs 0.705 ms at 128 x 2^20 under the max-ILP scheduler: gpurun_out r3d)
        if (glp_ntt_args_plain(a))
            return launch_kern<MODE, INV, LOG_E, true>(grid, block, lds, st, a);
    }
    return launch_kern<MODE, INV, LOG_E, false>(grid, block, lds, st, a);
}

#define GLP_CAT2(a, b) a##b
#define GLP_CAT(a, b) GLP_CAT2(a, b)
template <int LOG_E>
static hipError_t launch_mode(int mode, int inv, unsigned grid, unsigned block, size_t lds, hipStream_t st, const GlpNttPassArgs* a) {
    switch (mode * 2 + (inv ? 1 : 0)) {
        case GLP_STRIP * 2 + 0: return launch_one<GLP_STRIP, false, LOG_E>(grid, block, lds, st, *a);
        case GLP_STRIP * 2 + 1: return launch_one<GLP_STRIP, true, LOG_E>(grid, block, lds, st, *a);
        case GLP_FINAL_T * 2 + 0: return launch_one<GLP_FINAL_T, false, LOG_E>(grid, block, lds, st, *a);
        case GLP_FINAL_T * 2 + 1: return launch_one<GLP_FINAL_T, true, LOG_E>(grid, block, lds, st, *a);
        case GLP_FINAL_ROWS * 2 + 0: return launch_one<GLP_FINAL_ROWS, false, LOG_E>(grid, block, lds, st, *a);
        case GLP_FINAL_ROWS * 2 + 1: return launch_one<GLP_FINAL_ROWS, true, LOG_E>(grid, block, lds, st, *a);
    }
    return hipErrorInvalidValue;
}

#if defined(GLP_INST_E6)
// radix-64 work-items (2^11 / 2^12 tiles in two register steps): PLAIN forms with a compile-time tile width only — 128 VGPRs of data leave no
// room for the optional features' address arithmetic or for one LDS address register per element
template <int MODE, bool INV, int CTC>
static hipError_t launch_e6(unsigned grid, unsigned block, size_t lds, hipStream_t st, const GlpNttPassArgs& a) {
    auto kern = glp_ntt_pass_kernel<GLP_INST_LOG_R, MODE, INV, 6, true, CTC>;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(block), lds, st, a);
    return hipGetLastError();
}
extern "C" hipError_t GLP_CAT(GLP_CAT(glp_launch_ntt_pass_lr, GLP_INST_LOG_R), _e6)(int mode, int inv, unsigned grid, unsigned block, size_t lds,
                                                                                hipStream_t st, const GlpNttPassArgs* a) {
    if (!glp_ntt_args_plain(*a) || (a->log_c != 2 && a->log_c != 3)) return hipErrorInvalidValue;
    switch ((mode * 2 + (inv ? 1 : 0)) * 2 + (int)(a->log_c - 2)) {
        case (GLP_STRIP * 2 + 0) * 2 + 0: return launch_e6<GLP_STRIP, false, 2>(grid, block, lds, st, *a);
        case (GLP_STRIP * 2 + 0) * 2 + 1: return launch_e6<GLP_STRIP, false, 3>(grid, block, lds, st, *a);
        case (GLP_STRIP * 2 + 1) * 2 + 0: return launch_e6<GLP_STRIP, true, 2>(grid, block, lds, st, *a);
        case (GLP_STRIP * 2 + 1) * 2 + 1: return launch_e6<GLP_STRIP, true, 3>(grid, block, lds, st, *a);
        case (GLP_FINAL_T * 2 + 0) * 2 + 0: return launch_e6<GLP_FINAL_T, false, 2>(grid, block, lds, st, *a);
        case (GLP_FINAL_T * 2 + 0) * 2 + 1: return launch_e6<GLP_FINAL_T, false, 3>(grid, block, lds, st, *a);
        case (GLP_FINAL_T * 2 + 1) * 2 + 0: return launch_e6<GLP_FINAL_T, true, 2>(grid, block, lds, st, *a);
        case (GLP_FINAL_T * 2 + 1) * 2 + 1: return launch_e6<GLP_FINAL_T, true, 3>(grid, block, lds, st, *a);
    }
    return hipErrorInvalidValue;
}
#elif defined(GLP_INST_E5)
// the radix-32 work-items (32 elements each, split LDS exchange) are a translation unit of their own: they are compiled with the default
// scheduler (max-ILP interleaving costs them registers they do not have: 128 VGPRs at four waves per SIMD) — csrc/Makefile
extern "C" hipError_t GLP_CAT(GLP_CAT(glp_launch_ntt_pass_lr, GLP_INST_LOG_R), _e5)(int mode, int inv, unsigned grid, unsigned block, size_t lds,
                                                                                hipStream_t st, const GlpNttPassArgs* a) {
    return launch_mode<5>(mode, inv, grid, block, lds, st, a);
}
#else
#if GLP_INST_LOG_R >= 9
extern "C" hipError_t GLP_CAT(GLP_CAT(glp_launch_ntt_pass_lr, GLP_INST_LOG_R), _e5)(int mode, int inv, unsigned grid, unsigned block, size_t lds,
                                                                                hipStream_t st, const GlpNttPassArgs* a);
#endif
#if GLP_INST_LOG_R >= 11
extern "C" hipError_t GLP_CAT(GLP_CAT(glp_launch_ntt_pass_lr, GLP_INST_LOG_R), _e6)(int mode, int inv, unsigned grid, unsigned block, size_t lds,
                                                                                hipStream_t st, const GlpNttPassArgs* a);
#endif
extern "C" hipError_t GLP_CAT(glp_launch_ntt_pass_lr, GLP_INST_LOG_R)(int mode, int inv, int log_e, unsigned grid, unsigned block,
                                                                   size_t lds, hipStream_t st, const GlpNttPassArgs* a) {
    if (log_e == 4) return launch_mode<4>(mode, inv, grid, block, lds, st, a);
#if GLP_INST_LOG_R == 10
    if (log_e == 2) return launch_mode<2>(mode, inv, grid, block, lds, st, a);
    if (log_e == 3) return launch_mode<3>(mode, inv, grid, block, lds, st, a);   // 8-element work-items: twice the waves for latency-bound single transforms
#endif
#if GLP_INST_LOG_R >= 9
    if (log_e == 5) return GLP_CAT(GLP_CAT(glp_launch_ntt_pass_lr, GLP_INST_LOG_R), _e5)(mode, inv, grid, block, lds, st, a);
#endif
#if GLP_INST_LOG_R >= 11
    if (log_e == 6) return GLP_CAT(GLP_CAT(glp_launch_ntt_pass_lr, GLP_INST_LOG_R), _e6)(mode, inv, grid, block, lds, st, a);
#endif
    return hipErrorInvalidValue;
}
#endif
