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
    std::map<int, std::vector<u64>> full;
    const u64* table_full(int log_N, int log_m, int inv) {
        const int key = (log_N * 64 + log_m) * 2 + inv;
        if (!full.count(key)) {
            ensure(log_N, inv);
            auto& v = full[key];
            v.resize((size_t)1 << log_N);
            const u64* lo_p = lo[log_N * 2 + inv].data();
            const u64* hi_p = hi[log_N * 2 + inv].empty() ? nullptr : hi[log_N * 2 + inv].data();
            u64* out = v.data();
            glp_emu_launch(4, 64, 0, [&] { glp_build_full_tw_kernel<0>(out, (u32)log_N, (u32)log_m, lo_p, hi_p); });
        }
        return full[key].data();
    }
    std::map<int, std::vector<u64>> cos_row, cos_col;
    int coset_tables(int log_n, int rb, u64 shift, int log_r, int log_m, const u64** row, const u64** col) {
        const int key = (int)cos_row.size();
        cos_row[key].resize((size_t)1 << (rb + log_r));
        if (col) cos_col[key].resize((size_t)1 << (rb + log_m));
        glp_fill_coset_tables(log_n, rb, shift, log_r, log_m, cos_row[key].data(), col ? cos_col[key].data() : nullptr);
        *row = cos_row[key].data();
        if (col) *col = cos_col[key].data();
        return 0;
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
            // the PLAIN instantiation under the condition the host launcher uses (ntt_inst.hip)
            const bool plain = MODE != GLP_FINAL_ROWS && ps.log_e == 5 && glp_ntt_args_plain(a);
            auto both = [&](auto e_) {
                constexpr int LE = decltype(e_)::value;
                if constexpr (MODE != GLP_FINAL_ROWS && LE == 5) {
                    if (plain) { glp_emu_launch(grid, block, lds, [&] { glp_ntt_pass_kernel<LR, MODE, INV, LE, true>(a); }); return; }
                }
                glp_emu_launch(grid, block, lds, [&] { glp_ntt_pass_kernel<LR, MODE, INV, LE, false>(a); });
            };
            if (ps.log_e == 6) {
                // radix-64 work-items: PLAIN, compile-time tile width (what ntt_inst.hip instantiates)
                if constexpr (LR >= 11 && MODE != GLP_FINAL_ROWS) {
                    if (!glp_ntt_args_plain(a) || (a.log_c != 2 && a.log_c != 3)) { failed = true; return; }
                    if (a.log_c == 2) glp_emu_launch(grid, block, lds, [&] { glp_ntt_pass_kernel<LR, MODE, INV, 6, true, 2>(a); });
                    else glp_emu_launch(grid, block, lds, [&] { glp_ntt_pass_kernel<LR, MODE, INV, 6, true, 3>(a); });
                } else failed = true;
            } else if (ps.log_e == 5) {
                if constexpr (LR >= 9) both(glp_ic<5>{});
            } else if (ps.log_e == 3) {
                if constexpr (LR == 10) both(glp_ic<3>{}); else failed = true;
            } else if (ps.log_e == 2) {
                if constexpr (LR == 10) both(glp_ic<2>{}); else failed = true;
            } else {
                both(glp_ic<4>{});
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
    bool failed = false;
    int launch_pass(const GlpPass& ps, int inv, unsigned long long grid, unsigned block, size_t lds, const GlpNttPassArgs& a) {
        failed = false;
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
        return failed ? -11 : 0;
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

// coset LDE with bit-reversed output: coeffs [batch][n] -> out [batch][n << rb]
extern "C" int emu_lde_coset_bitrev(const u64* coeffs, u64* out, int log_n, int rb, unsigned batch, u64 shift, const char* plan_override) {
    EmuBackend be;
    GlpPlan pl;
    int rc = glp_make_plan(log_n, 1, 0, plan_override, &pl, (unsigned long long)batch << rb);
    if (rc) return rc;
    GlpNttCall c{coeffs, out, nullptr, 1ull << log_n, 1ull << (log_n + rb), batch << rb, log_n, 0, 1};
    c.coset_log = (u32)rb;
    c.coset_shift = shift;
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
#include "../../0-kno-blobstreamx_amd/csrc/poseidon_precomp.h"

// small != 0: fast MDS path; small == 2: also the grouped partial rounds
static GlpPoseidonConsts emu_consts(const u64* c384, int small, std::vector<u32>& cf, std::vector<u64>& cs) {
    GlpPoseidonConsts k{c384, c384 + 360, c384 + 372, nullptr, nullptr};
    if (small == 2 && glp_poseidon_group_tables(c384, cf, cs)) { k.pg_coef = cf.data(); k.pg_cst = cs.data(); }
    return k;
}

extern "C" int emu_poseidon_grouped_available(const u64* consts384) {
    std::vector<u32> cf; std::vector<u64> cs;
    return glp_poseidon_group_tables(consts384, cf, cs) ? 1 : 0;
}
extern "C" int emu_poseidon_permute(u64* states, u64 n, const u64* consts384, int small) {
    std::vector<u32> cf; std::vector<u64> cs;
    GlpPoseidonConsts k = emu_consts(consts384, small, cf, cs);
    unsigned block = 64, grid = (unsigned)((n + block - 1) / block);
    if (small) glp_emu_launch(grid, block, 0, [&] { glp_poseidon_permute_kernel<true>(states, n, k); });
    else glp_emu_launch(grid, block, 0, [&] { glp_poseidon_permute_kernel<false>(states, n, k); });
    return 0;
}

extern "C" int emu_merkle(const u64* src, u64 stride, int poly_major, u32 leaf_len, u32 log_leaves, u32 cap_h, u64* digests,
                          const u64* consts384, int small, int coop_max) {
    std::vector<u32> cf; std::vector<u64> cs;
    GlpPoseidonConsts k = emu_consts(consts384, small, cf, cs);
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
    // coop_max: levels of at most this many nodes use the lane-cooperative kernels (0 = never), as the product's merkle_impl does
    for (u32 lvl = log_leaves; lvl > cap_h; lvl--) {
        u64* cur = prev + 4 * cnt;
        const u64 out = cnt >> 1;
        if (coop_max && out <= GLP_COOP_TOP_NODES && out <= (u64)coop_max) {
            const u32 n_levels = lvl - cap_h;
            if (small) glp_emu_launch(1, 1024, 0, [&] { glp_merkle_top_coop_kernel<true>(prev, cnt, n_levels, k); });
            else glp_emu_launch(1, 1024, 0, [&] { glp_merkle_top_coop_kernel<false>(prev, cnt, n_levels, k); });
            break;
        }
        cnt = out;
        if (coop_max && cnt <= (u64)coop_max) {
            unsigned g = (unsigned)((cnt * 16 + 255) / 256);
            if (small) glp_emu_launch(g, 256, 0, [&] { glp_merkle_level_coop_kernel<true>(prev, cur, cnt, k); });
            else glp_emu_launch(g, 256, 0, [&] { glp_merkle_level_coop_kernel<false>(prev, cur, cnt, k); });
        } else {
            unsigned g = (unsigned)((cnt + block - 1) / block);
            if (small) glp_emu_launch(g, block, 0, [&] { glp_merkle_level_kernel<true>(prev, cur, cnt, k); });
            else glp_emu_launch(g, block, 0, [&] { glp_merkle_level_kernel<false>(prev, cur, cnt, k); });
        }
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

// ---- FRI kernels + challenger under emulation ---------------------------------------------
#include "../../0-kno-blobstreamx_amd/csrc/fri_kernels.cuh"
#include "../../0-kno-blobstreamx_amd/csrc/challenger.h"

extern "C" int emu_challenger(const u64* consts384, int small, const u64* script, u64 n_script, u64* out) {
    // script: sequence of ops: (0, value) = observe value ; (1, _) = emit one challenge
    glp_challenger ch;
    memset(ch.state, 0, sizeof(ch.state));
    ch.n_in = ch.n_out = 0;
    ch.consts.assign(consts384, consts384 + 384);
    ch.small_mds = small != 0;
    u64 k = 0;
    for (u64 i = 0; i < n_script; i++) {
        if (script[2 * i] == 0) ch.observe(script[2 * i + 1]);
        else out[k++] = ch.challenge();
    }
    return (int)k;
}

extern "C" int emu_eval_at_ext(const u64* coeffs, u64 stride, u32 log_n, u32 n_polys, const u64* z, u64* out) {
    const u64 n = 1ull << log_n, nhi = (n + 255) / 256;
    std::vector<u64> lo(512), hi(2 * nhi), zp(2 * n);
    gl_ext2 zz{z[0], z[1]}, t{1, 0};
    for (int j = 0; j < 256; j++) { lo[2 * j] = t.a; lo[2 * j + 1] = t.b; t = gl_ext_mul(t, zz); }
    gl_ext2 z256 = t; t = gl_ext2{1, 0};
    for (u64 j = 0; j < nhi; j++) { hi[2 * j] = t.a; hi[2 * j + 1] = t.b; t = gl_ext_mul(t, z256); }
    glp_emu_launch((unsigned)((n + 255) / 256), 256, 0, [&] { glp_ext_powers_kernel<0>(zp.data(), n, lo.data(), hi.data()); });
    const u32 n_chunks = (u32)((n + GLP_EVAL_CHUNK - 1) / GLP_EVAL_CHUNK);
    std::vector<u64> part((size_t)n_polys * n_chunks * 2);
    glp_emu_launch(n_polys * n_chunks, 256, 0, [&] { glp_eval_ext_kernel<0>(coeffs, stride, n, n_chunks, zp.data(), part.data()); });
    for (u32 p = 0; p < n_polys; p++) {
        u64 a = 0, b = 0;
        for (u32 k = 0; k < n_chunks; k++) { a = gl_add(a, part[2 * ((size_t)p * n_chunks + k)]); b = gl_add(b, part[2 * ((size_t)p * n_chunks + k) + 1]); }
        out[2 * p] = a; out[2 * p + 1] = b;
    }
    return 0;
}

extern "C" int emu_fri_combine(const u64* lde, u32 n_polys, u32 log_N, const u64* alpha_pow, const u64* Y, const u64* z, u64 shift,
                               u64* acc, int first, int finish) {
    std::vector<u64> lo(glp_table_lo_len(log_N)), hi(glp_table_hi_len(log_N) ? glp_table_hi_len(log_N) : 1);
    glp_fill_table(log_N, 0, lo.data(), hi.data());
    GlpCombineArgs a;
    a.lde = lde; a.poly_stride = 1ull << log_N; a.n_polys = n_polys; a.alpha_pow = alpha_pow; a.acc = acc; a.log_N = log_N;
    a.first = first; a.finish = finish; a.Y = gl_ext2{Y[0], Y[1]}; a.z = gl_ext2{z[0], z[1]}; a.shift = shift;
    a.w_lo = lo.data(); a.w_hi = glp_table_hi_len(log_N) ? hi.data() : nullptr;
    const u64 threads = (1ull << log_N) / 4;
    glp_emu_launch((unsigned)((threads + 63) / 64), 64, 0, [&] { glp_fri_combine_kernel<0>(a); });
    return 0;
}

extern "C" int emu_pow(const u64* seed, u64 base, u64 count, u32 pow_bits, const u64* consts384, int small, unsigned long long* found) {
    std::vector<u32> cf; std::vector<u64> cs;
    GlpPoseidonConsts k = emu_consts(consts384, small, cf, cs);
    *found = ~0ull;
    if (small) glp_emu_launch((unsigned)((count + 63) / 64), 64, 0, [&] { glp_pow_kernel<true>(seed, base, count, pow_bits, found, k); });
    else glp_emu_launch((unsigned)((count + 63) / 64), 64, 0, [&] { glp_pow_kernel<false>(seed, base, count, pow_bits, found, k); });
    return 0;
}

// ---- PLONK kernels (rows a6, a7) under emulation ------------------------------------------
#include "../../0-kno-blobstreamx_amd/csrc/plonk_kernels.cuh"

extern "C" int emu_plonk_zs(const u64* wires, const u64* sigmas, u32 log_n, u32 W, const u64* beta, const u64* gamma, u64* zs) {
    const u64 n = 1ull << log_n;
    const u32 M = W / GLP_PLONK_CHUNK;
    std::vector<u64> lo(glp_table_lo_len(log_n)), hi(glp_table_hi_len(log_n) ? glp_table_hi_len(log_n) : 1), ks(W);
    glp_fill_table(log_n, 0, lo.data(), hi.data());
    { u64 t = 1; for (u32 j = 0; j < W; j++) { ks[j] = t; t = gl_mul(t, 7); } }
    std::vector<u64> qv((size_t)GLP_PLONK_NCHAL * M * n), rr((size_t)GLP_PLONK_NCHAL * n);
    GlpPermArgs pa;
    pa.wires = wires; pa.sigmas = sigmas; pa.ks = ks.data(); pa.log_n = log_n; pa.W = W;
    for (int t = 0; t < GLP_PLONK_NCHAL; t++) { pa.beta[t] = beta[t]; pa.gamma[t] = gamma[t]; }
    pa.w_lo = lo.data(); pa.w_hi = glp_table_hi_len(log_n) ? hi.data() : nullptr; pa.qv = qv.data(); pa.rr = rr.data();
    glp_emu_launch((unsigned)((n + 63) / 64), 64, 0, [&] { glp_perm_quotients_kernel<0>(pa); });
    const u32 nb = (u32)((n + GLP_SCAN_BLOCK - 1) / GLP_SCAN_BLOCK);
    std::vector<u64> bprod((size_t)GLP_PLONK_NCHAL * nb);
    // 2-D grids: emulate blockIdx.y by an outer loop
    for (u32 t = 0; t < GLP_PLONK_NCHAL; t++) {
        glp_emu_launch(nb, 256, 0, [&] { blockIdx.y = t; gridDim.y = GLP_PLONK_NCHAL; glp_scan_reduce_kernel<0>(rr.data(), n, bprod.data()); });
    }
    glp_emu_launch(GLP_PLONK_NCHAL, GLP_SCAN_TOP, 0, [&] { glp_scan_blocks_kernel<0>(bprod.data(), nb); });
    for (u32 t = 0; t < GLP_PLONK_NCHAL; t++) {
        glp_emu_launch(nb, 256, 0, [&] { blockIdx.y = t; gridDim.y = GLP_PLONK_NCHAL; glp_scan_apply_kernel<0>(rr.data(), qv.data(), n, M, bprod.data(), zs); });
    }
    return 0;
}

extern "C" int emu_plonk_quotient(const u64* consts, const u64* sigmas, const u64* wires, const u64* zs, const u64* pi, u32 log_n, u32 rb, u32 W,
                                  u32 R, u32 flags, const u64* pos_consts, const u64* beta, const u64* gamma, const u64* alpha, u64* out) {
    const u32 log_N = log_n + rb, M = R / GLP_PLONK_CHUNK;
    const u32 n_con = 2 + 3 * M + ((flags & GLP_CIRCUIT_POSEIDON_GATE) ? GLP_POS_GATE_CONSTRAINTS : 0) +
                      ((flags & GLP_CIRCUIT_SHA_GATES) ? GLP_SHA_GATE_CONSTRAINTS : 0);
    const u64 n = 1ull << log_n, N = 1ull << log_N;
    std::vector<u64> lo(glp_table_lo_len(log_N)), hi(glp_table_hi_len(log_N) ? glp_table_hi_len(log_N) : 1), ks(R), inv(N);
    glp_fill_table(log_N, 0, lo.data(), hi.data());
    const u64* hip = glp_table_hi_len(log_N) ? hi.data() : nullptr;
    { u64 t = 1; for (u32 j = 0; j < R; j++) { ks[j] = t; t = gl_mul(t, 7); } }
    glp_emu_launch((unsigned)((N / 4 + 63) / 64), 64, 0, [&] { glp_inv_xm1_kernel<0>(inv.data(), log_N, 7, lo.data(), hip); });
    std::vector<u64> apow((size_t)GLP_PLONK_NCHAL * n_con);
    for (int t = 0; t < GLP_PLONK_NCHAL; t++) { u64 x = 1; for (u32 k = 0; k < n_con; k++) { apow[(size_t)t * n_con + k] = x; x = gl_mul(x, alpha[t]); } }
    GlpQuotientArgs qa;
    qa.consts = consts; qa.sigmas = sigmas; qa.wires = wires; qa.zs = zs; qa.pi = pi; qa.ks = ks.data();
    qa.q_ext = (flags & GLP_CIRCUIT_EXT_GATE) ? consts + (u64)(glp_plonk_n_const(flags) - 1) * N : nullptr;
    qa.log_n = log_n; qa.rate_bits = rb; qa.W = W; qa.R = R; qa.n_con = n_con;
    for (int t = 0; t < GLP_PLONK_NCHAL; t++) { qa.beta[t] = beta[t]; qa.gamma[t] = gamma[t]; }
    qa.alpha_pow = apow.data(); qa.pos_consts = pos_consts; qa.w_lo = lo.data(); qa.w_hi = hip; qa.shift = 7;
    const u64 sn = gl_pow(7, n), wr = gl_root_of_unity(rb);   // 7 here is the coset SHIFT (the multiplicative generator), not the two-adic generator
    for (u32 k = 0; k < (1u << rb); k++) qa.zh_inv[k] = gl_inv(gl_sub(gl_mul(sn, gl_pow(wr, k)), 1));
    qa.n_inv = gl_inv(n % GL_P); qa.inv_xm1 = inv.data(); qa.out = out;
    if (flags & GLP_CIRCUIT_POSEIDON_GATE) {
        // the product picks the small-integer MDS form of the row constraints when the constants allow it: run that one, and hold it against
        // the generic form (the gate's definition) point for point
        bool small = true;
        unsigned __int128 sum = 0;
        u64 maxdiag = 0;
        for (int i = 0; i < 12; i++) {
            const u64 cv = pos_consts[360 + i], dv = pos_consts[372 + i];
            if (cv >> 24 || dv >> 24) small = false;
            sum += cv;
            if (dv > maxdiag) maxdiag = dv;
        }
        if (sum + maxdiag >= ((unsigned __int128)1 << 24)) small = false;
        glp_emu_launch((unsigned)((N + 63) / 64), 64, 0, [&] { glp_quotient_kernel<true>(qa); });
        if (small) {
            std::vector<u64> generic(out, out + (size_t)GLP_PLONK_NCHAL * N);
            glp_emu_launch((unsigned)((N + 63) / 64), 64, 0, [&] { glp_quotient_kernel<true, true>(qa); });
            for (size_t k = 0; k < generic.size(); k++) if (generic[k] != out[k]) return 2;
        }
    } else glp_emu_launch((unsigned)((N + 63) / 64), 64, 0, [&] { glp_quotient_kernel<false>(qa); });
    if (flags & GLP_CIRCUIT_SHA_GATES)      // consts then has GLP_PLONK_NCONST_SHA rows
        glp_emu_launch((unsigned)((N + 63) / 64), 64, 0, [&] { glp_quotient_sha_kernel<0>(qa, n_con - GLP_SHA_GATE_CONSTRAINTS); });
    return 0;
}

// SHA-row witness through the product's kernel: the bit wires of the listed rows from their routed words
extern "C" int emu_sha_gate_fill_rows(u64* wires, u32 log_n, const u32* rows, const u32* kinds, u32 n_rows) {
    glp_emu_launch((n_rows + 63) / 64, 64, 0, [&] { glp_sha_gate_fill_kernel<0>(wires, 1ull << log_n, rows, kinds, n_rows); });
    return 0;
}

// Poseidon-row witness through the product's kernel: wires [W][n] in place, for the listed rows
extern "C" int emu_poseidon_gate_fill_rows(u64* wires, u32 log_n, const u32* rows, u32 n_rows, const u64* pos_consts) {
    // the definition first, then (small MDS) the fast form the product launches: both must write the same wires
    const u64 n = 1ull << log_n;
    bool small = true;
    unsigned __int128 sum = 0;
    u64 maxdiag = 0;
    for (int i = 0; i < 12; i++) {
        const u64 cv = pos_consts[360 + i], dv = pos_consts[372 + i];
        if (cv >> 24 || dv >> 24) small = false;
        sum += cv;
        if (dv > maxdiag) maxdiag = dv;
    }
    if (sum + maxdiag >= ((unsigned __int128)1 << 24)) small = false;
    u32 W = 0;
    for (u32 k = 0; k < n_rows; k++) (void)rows[k];
    W = GLP_POS_GATE_WIRES;                              // the rows' Poseidon wires are all this function touches
    std::vector<u64> before((size_t)W * n);
    for (size_t i = 0; i < before.size(); i++) before[i] = wires[i];
    glp_emu_launch((n_rows + 63) / 64, 64, 0, [&] { glp_poseidon_gate_fill_kernel<0>(wires, n, rows, n_rows, pos_consts); });
    if (small) {
        std::vector<u64> generic((size_t)W * n);
        for (size_t i = 0; i < generic.size(); i++) { generic[i] = wires[i]; wires[i] = before[i]; }
        glp_emu_launch((n_rows + 63) / 64, 64, 0, [&] { glp_poseidon_gate_fill_kernel<1>(wires, n, rows, n_rows, pos_consts); });
        for (size_t i = 0; i < generic.size(); i++) if (generic[i] != wires[i]) return 2;
    }
    return 0;
}

extern "C" int emu_bitrev_scale(const u64* in, u64* out, u32 log_n, u32 batch, u64 s) {
    const u64 n = 1ull << log_n;
    glp_emu_launch((unsigned)((((u64)batch << log_n) + 63) / 64), 64, 0, [&] { glp_bitrev_permute_kernel<0>(in, out, log_n, batch); });
    std::vector<u64> lo(4096), hi(n > 4096 ? (n >> 12) : 1);
    u64 t = 1;
    for (u32 k = 0; k < 4096; k++) { lo[k] = t; t = gl_mul(t, s); }
    const u64 s4096 = t; t = 1;
    for (size_t k = 0; k < hi.size(); k++) { hi[k] = t; t = gl_mul(t, s4096); }
    glp_emu_launch(8, 64, 0, [&] { glp_scale_pow_kernel<0>(out, log_n, batch, lo.data(), n > 4096 ? hi.data() : nullptr); });
    return 0;
}

extern "C" int emu_tm_merkle_root(const uint8_t* leaves, u32 leaf_len, u64 n, const u32* k256, uint8_t* root32) {
    std::vector<u32> a(n * 8 + 8), b(((n + 1) / 2) * 8 + 8);
    glp_emu_launch((unsigned)((n + 63) / 64), 64, 0, [&] { glp_tm_leaf_kernel<0>(leaves, leaf_len, n, a.data(), k256); });
    u64 cnt = n;
    u32 *src = a.data(), *dst = b.data();
    while (cnt > 1) {
        const u64 nout = (cnt + 1) / 2;
        glp_emu_launch((unsigned)((nout + 63) / 64), 64, 0, [&] { glp_tm_inner_kernel<0>(src, cnt, dst, k256); });
        cnt = nout;
        std::swap(src, dst);
    }
    for (int k = 0; k < 8; k++) { root32[4*k] = src[k] >> 24; root32[4*k+1] = src[k] >> 16; root32[4*k+2] = src[k] >> 8; root32[4*k+3] = src[k]; }
    return 0;
}

extern "C" int emu_tm_merkle_root_var(const uint8_t* data, const u64* offsets, u64 n, const u32* k256, uint8_t* root32) {
    std::vector<u32> a(n * 8 + 8), b(((n + 1) / 2) * 8 + 8);
    glp_emu_launch((unsigned)((n + 63) / 64), 64, 0, [&] { glp_tm_leaf_var_kernel<0>(data, offsets, n, a.data(), k256); });
    u64 cnt = n;
    u32 *src = a.data(), *dst = b.data();
    while (cnt > 1) {
        const u64 nout = (cnt + 1) / 2;
        glp_emu_launch((unsigned)((nout + 63) / 64), 64, 0, [&] { glp_tm_inner_kernel<0>(src, cnt, dst, k256); });
        cnt = nout;
        std::swap(src, dst);
    }
    for (int k = 0; k < 8; k++) { root32[4*k] = src[k] >> 24; root32[4*k+1] = src[k] >> 16; root32[4*k+2] = src[k] >> 8; root32[4*k+3] = src[k]; }
    return 0;
}

// ---- Ed25519 witness kernel (row a10) under emulation -----------------------------------------
#include "../../0-kno-blobstreamx_amd/csrc/ed25519_kernels.cuh"
extern "C" int emu_ed25519_witness(const uint8_t* pubs, const uint8_t* sigs, const uint8_t* msgs, u32 msg_stride, const u32* lens, u64 n,
                                   const u64* k512, u64* out) {
    glp_emu_launch((unsigned)((n + 63) / 64), 64, 0, [&] { glp_ed25519_witness_kernel<0>(pubs, sigs, msgs, msg_stride, lens, n, k512, out); });
    return 0;
}
