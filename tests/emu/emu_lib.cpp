// tests/emu/emu_lib.cpp — TEST INFRASTRUCTURE.  Runs the product's kernel bodies and host
// planning code on the CPU through hip_emu.h so that index arithmetic, LDS sizing and
// barrier placement are checked (under ASan) before a kernel is ever launched on a GPU.
// Loaded by tests/test_emu_*.py via ctypes; never part of the product.
#include "hip_emu.h"
#include <map>
#include <vector>
#include "../../0-kno-blobstreamx_amd/csrc/ntt_exec.h"

namespace {
struct EmuBackend {
    std::map<int, std::vector<u64>> lo, hi;
    const u64* table_lo(int log_N, int inv) { ensure(log_N, inv); return lo[log_N * 2 + inv].data(); }
    const u64* table_hi(int log_N, int inv) {
        ensure(log_N, inv);
        auto& v = hi[log_N * 2 + inv];
        return v.empty() ? nullptr : v.data();
    }
    void ensure(int log_N, int inv) {
        int key = log_N * 2 + inv;
        if (lo.count(key)) return;
        lo[key].resize(glp_table_lo_len(log_N));
        hi[key].resize(glp_table_hi_len(log_N));
        glp_fill_table(log_N, inv, lo[key].data(), hi[key].data());
    }
    template <int LR>
    void run(const GlpPass& ps, int inv, unsigned grid, unsigned block, size_t lds, const GlpNttPassArgs& a) {
        auto go = [&](auto mode_, auto inv_) {
            constexpr int MODE = decltype(mode_)::value;
            constexpr bool INV = decltype(inv_)::value != 0;
            glp_emu_launch(grid, block, lds, [&] { glp_ntt_pass_kernel<LR, MODE, INV>(a); });
        };
        int key = ps.mode * 2 + (inv ? 1 : 0);
        switch (key) {
            case 0: go(glp_ic<GLP_STRIP>{}, glp_ic<0>{}); break;
            case 1: go(glp_ic<GLP_STRIP>{}, glp_ic<1>{}); break;
            case 2: go(glp_ic<GLP_FINAL_T>{}, glp_ic<0>{}); break;
            case 3: go(glp_ic<GLP_FINAL_T>{}, glp_ic<1>{}); break;
            case 4: go(glp_ic<GLP_FINAL_ROWS>{}, glp_ic<0>{}); break;
            case 5: go(glp_ic<GLP_FINAL_ROWS>{}, glp_ic<1>{}); break;
        }
    }
    int launch_pass(const GlpPass& ps, int inv, unsigned long long grid, unsigned block, size_t lds, const GlpNttPassArgs& a) {
        switch (ps.log_r) {
            case 6: run<6>(ps, inv, (unsigned)grid, block, lds, a); break;
            case 7: run<7>(ps, inv, (unsigned)grid, block, lds, a); break;
            case 8: run<8>(ps, inv, (unsigned)grid, block, lds, a); break;
            case 9: run<9>(ps, inv, (unsigned)grid, block, lds, a); break;
            case 10: run<10>(ps, inv, (unsigned)grid, block, lds, a); break;
            case 11: run<11>(ps, inv, (unsigned)grid, block, lds, a); break;
            case 12: run<12>(ps, inv, (unsigned)grid, block, lds, a); break;
            default: return -10;
        }
        return 0;
    }
    int launch_small(const u64* src, u64* dst, u64 ss, u64 ds, u32 log_n, u32 batch, const u64* tw, u64 scale, u32 rev) {
        unsigned block = 64, grid = (batch + block - 1) / block;
        glp_emu_launch(grid, block, 0, [&] { glp_ntt_small_kernel<0>(src, dst, ss, ds, log_n, batch, tw, scale, rev); });
        return 0;
    }
};
}  // namespace

extern "C" int emu_ntt(const u64* src, u64* dst, u64 src_stride, u64 dst_stride, int log_n, unsigned batch,
                       int inverse, int rev, const char* plan_override) {
    EmuBackend be;
    GlpPlan pl;
    if (log_n >= GLP_MIN_LOG_R) {
        int rc = glp_make_plan(log_n, rev, src == dst, plan_override, &pl);
        if (rc) return rc;
    }
    std::vector<u64> scratch;
    if (log_n >= GLP_MIN_LOG_R && pl.needs_scratch) scratch.resize((size_t)batch << log_n);
    GlpNttCall c{src, dst, scratch.empty() ? nullptr : scratch.data(), src_stride, dst_stride, batch, log_n, inverse, rev};
    return glp_exec_ntt(be, &pl, c);
}

// field helpers exposed for direct testing of the product's arithmetic on the host
extern "C" u64 emu_gl_add(u64 a, u64 b) { return gl_add(a, b); }
extern "C" u64 emu_gl_sub(u64 a, u64 b) { return gl_sub(a, b); }
extern "C" u64 emu_gl_mul(u64 a, u64 b) { return gl_mul(a, b); }
extern "C" u64 emu_gl_reduce128(u64 hi, u64 lo) { return gl_reduce128(hi, lo); }
extern "C" u64 emu_gl_mul_pow2(u64 x, int s) {
    u64 r = 0;
    bool found = false;
    glp_static_for<0, 192>([&](auto i_) {
        constexpr int i = decltype(i_)::value;
        if (i == s) { r = gl_mul_pow2<i>(x); found = true; }
    });
    return found ? r : ~0ull;
}
