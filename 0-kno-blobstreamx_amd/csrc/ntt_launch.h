// ntt_launch.h — per-tile-size launchers defined in ntt_inst.hip (one object per LOG_R).
#pragma once
#include <hip/hip_runtime.h>
struct GlpNttPassArgs;
#define GLP_DECL_LAUNCH(n) \
    extern "C" hipError_t glp_launch_ntt_pass_lr##n(int mode, int inv, int log_e, unsigned grid, unsigned block, size_t lds, \
                                                   hipStream_t st, const GlpNttPassArgs* a);
GLP_DECL_LAUNCH(6) GLP_DECL_LAUNCH(7) GLP_DECL_LAUNCH(8) GLP_DECL_LAUNCH(9)
GLP_DECL_LAUNCH(10) GLP_DECL_LAUNCH(11) GLP_DECL_LAUNCH(12)
#undef GLP_DECL_LAUNCH
static inline hipError_t glp_launch_ntt_pass(int log_r, int mode, int inv, int log_e, unsigned grid, unsigned block, size_t lds,
                                             hipStream_t st, const GlpNttPassArgs* a) {
    switch (log_r) {
        case 6: return glp_launch_ntt_pass_lr6(mode, inv, log_e, grid, block, lds, st, a);
        case 7: return glp_launch_ntt_pass_lr7(mode, inv, log_e, grid, block, lds, st, a);
        case 8: return glp_launch_ntt_pass_lr8(mode, inv, log_e, grid, block, lds, st, a);
        case 9: return glp_launch_ntt_pass_lr9(mode, inv, log_e, grid, block, lds, st, a);
        case 10: return glp_launch_ntt_pass_lr10(mode, inv, log_e, grid, block, lds, st, a);
        case 11: return glp_launch_ntt_pass_lr11(mode, inv, log_e, grid, block, lds, st, a);
        case 12: return glp_launch_ntt_pass_lr12(mode, inv, log_e, grid, block, lds, st, a);
    }
    return hipErrorInvalidValue;
}
