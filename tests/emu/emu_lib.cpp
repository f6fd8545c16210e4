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
            if (ps.log_e == 5) {
                if constexpr (LR == 9 || LR == 10) glp_emu_launch(grid, block, lds, [&] { glp_ntt_pass_kernel<LR, MODE, INV, 5>(a); });
            } else {
                glp_emu_launch(grid, block, lds, [&] { glp_ntt_pass_kernel<LR, MODE, INV, 4>(a); });
            }
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
        int rc = glp_make_plan(log_n, rev, src == dst, plan_override, &pl, batch);
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

// ---- hash kernels under emulation -------------------------------------------------------
#include "../../0-kno-blobstreamx_amd/csrc/hash_kernels.cuh"

extern "C" int emu_poseidon_permute(u64* states, u64 n, const u64* consts384, int small) {
    GlpPoseidonConsts k{consts384, consts384 + 360, consts384 + 372};
    unsigned block = 64, grid = (unsigned)((n + block - 1) / block);
    if (small) glp_emu_launch(grid, block, 0, [&] { glp_poseidon_permute_kernel<true>(states, n, k); });
    else glp_emu_launch(grid, block, 0, [&] { glp_poseidon_permute_kernel<false>(states, n, k); });
    return 0;
}

extern "C" int emu_merkle(const u64* src, u64 stride, int poly_major, u32 leaf_len, u32 log_leaves, u32 cap_h, u64* digests,
                          const u64* consts384, int small) {
    GlpPoseidonConsts k{consts384, consts384 + 360, consts384 + 372};
    const u64 nl = 1ull << log_leaves;
    unsigned block = 64, grid = (unsigned)((nl + block - 1) / block);
    auto leaves = [&](auto sm_, auto pm_) {
        constexpr bool SM = decltype(sm_)::value != 0, PM = decltype(pm_)::value != 0;
        glp_emu_launch(grid, block, 0, [&] { glp_hash_leaves_kernel<SM, PM>(src, stride, leaf_len, nl, digests, k); });
    };
    if (small) { if (poly_major) leaves(glp_ic<1>{}, glp_ic<1>{}); else leaves(glp_ic<1>{}, glp_ic<0>{}); }
    else { if (poly_major) leaves(glp_ic<0>{}, glp_ic<1>{}); else leaves(glp_ic<0>{}, glp_ic<0>{}); }
    u64* prev = digests;
    u64 cnt = nl;
    for (u32 lvl = log_leaves; lvl > cap_h; lvl--) {
        u64* cur = prev + 4 * cnt;
        cnt >>= 1;
        unsigned g = (unsigned)((cnt + block - 1) / block);
        if (small) glp_emu_launch(g, block, 0, [&] { glp_merkle_level_kernel<true>(prev, cur, cnt, k); });
        else glp_emu_launch(g, block, 0, [&] { glp_merkle_level_kernel<false>(prev, cur, cnt, k); });
        prev = cur;
    }
    return 0;
}

extern "C" int emu_fri_fold2(const u64* evals, u64* out, u32 log_n, u64 shift, const u64* beta) {
    std::vector<u64> lo(glp_table_lo_len(log_n)), hi(glp_table_hi_len(log_n) ? glp_table_hi_len(log_n) : 1);
    glp_fill_table(log_n, 1, lo.data(), hi.data());
    const u64* hip = glp_table_hi_len(log_n) ? hi.data() : nullptr;
    const u64 half = 1ull << (log_n - 1);
    unsigned block = 64, grid = (unsigned)((half + block - 1) / block);
    u64 half_inv = gl_inv(2), cc = gl_inv(gl_mul(2, shift));
    gl_ext2 b{beta[0], beta[1]};
    glp_emu_launch(grid, block, 0, [&] { glp_fri_fold2_kernel<0>(evals, out, log_n, half_inv, cc, b, lo.data(), hip); });
    return 0;
}

extern "C" int emu_sha256_trace(const uint8_t* blocks, u64 n_msgs, u32 bpm, u32* digests, u32* trace, const u32* k256) {
    unsigned block = 64, grid = (unsigned)((n_msgs + block - 1) / block);
    glp_emu_launch(grid, block, 0, [&] { glp_sha256_trace_kernel<0>(blocks, n_msgs, bpm, digests, trace, k256); });
    return 0;
}
extern "C" int emu_sha512_trace(const uint8_t* blocks, u64 n_msgs, u32 bpm, u64* digests, u64* trace, const u64* k512) {
    unsigned block = 64, grid = (unsigned)((n_msgs + block - 1) / block);
    glp_emu_launch(grid, block, 0, [&] { glp_sha512_trace_kernel<0>(blocks, n_msgs, bpm, digests, trace, k512); });
    return 0;
}
