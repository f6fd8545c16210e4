// ntt_inst.hip — explicit instantiations of glp_ntt_pass_kernel for ONE tile size
// (compiled once per GLP_INST_LOG_R by the Makefile so the instantiations build in parallel).
#include <hip/hip_runtime.h>
#include "ntt_kernels.cuh"
#include "ntt_launch.h"

#ifndef GLP_INST_LOG_R
#error "compile with -DGLP_INST_LOG_R=<6..12>"
#endif

template <int MODE, bool INV>
static hipError_t launch_one(unsigned grid, unsigned block, size_t lds, hipStream_t st, const GlpNttPassArgs& a) {
    auto kern = glp_ntt_pass_kernel<GLP_INST_LOG_R, MODE, INV>;
    static bool attr_done = false;   // one ctx per process per GPU: no concurrent first call
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(block), lds, st, a);
    return hipGetLastError();
}

#define GLP_CAT2(a, b) a##b
#define GLP_CAT(a, b) GLP_CAT2(a, b)
extern "C" hipError_t GLP_CAT(glp_launch_ntt_pass_lr, GLP_INST_LOG_R)(int mode, int inv, unsigned grid, unsigned block,
                                                                   size_t lds, hipStream_t st, const GlpNttPassArgs* a) {
    switch (mode * 2 + (inv ? 1 : 0)) {
        case GLP_STRIP * 2 + 0: return launch_one<GLP_STRIP, false>(grid, block, lds, st, *a);
        case GLP_STRIP * 2 + 1: return launch_one<GLP_STRIP, true>(grid, block, lds, st, *a);
        case GLP_FINAL_T * 2 + 0: return launch_one<GLP_FINAL_T, false>(grid, block, lds, st, *a);
        case GLP_FINAL_T * 2 + 1: return launch_one<GLP_FINAL_T, true>(grid, block, lds, st, *a);
        case GLP_FINAL_ROWS * 2 + 0: return launch_one<GLP_FINAL_ROWS, false>(grid, block, lds, st, *a);
        case GLP_FINAL_ROWS * 2 + 1: return launch_one<GLP_FINAL_ROWS, true>(grid, block, lds, st, *a);
    }
    return hipErrorInvalidValue;
}
