// plonk_kernels.cuh — permutation-argument products (Z and partial products) and quotient
// evaluation for the build-defined circuit of DESIGN.md §3.6 (SURVEY.md §8a rows a6, a7;
// upstream names recalled, unverified: plonky2::plonk::prover::compute_partial_products_and_z_polys,
// compute_quotient_polys / vanishing_poly::eval_vanishing_poly_base_batch — reference
// file:line NONE, the mount is empty; the gate set and constraint order here are build-defined).
//
// Circuit: n = 2^k rows, W wire columns of which the first R are routed (R % 8 == 0); gates: plonk_gates.h
// (arithmetic/constant on every group of 4 routed wires, public inputs on wire 0, Poseidon rows).
// Copy constraints via the plonky2-style permutation argument over the routed wires
// with chunks of 8 wires and num_challenges = 2:
//   num_j = w_j + beta*k_j*x + gamma,   den_j = w_j + beta*sigma_j + gamma
//   Z(g x) = Z(x) * prod_c q_c(x),  q_c = prod_{j in chunk c} num_j / den_j,
//   partial products pi_c = Z * q_0 ... q_c  (c < M-1),  M = R/8 chunks.
// Plain HIP C++ without AMD builtins (tests/emu runs these bodies on the CPU).
#pragma once
#include "gl_field.cuh"
#include "plonk_gates.h"

#define GLP_PLONK_CHUNK 8
#define GLP_PLONK_NCHAL 2

// ---- K6a: per-row chunk quotients ------------------------------------------------------
// wires: [W][n], sigmas: [R][n] values on the trace domain (natural order); GlpPermArgs::W = R, the routed count.  ks[j] = k_j.
// qv: [NCHAL][M][n] chunk quotients, rr: [NCHAL][n] row ratios prod_c q_c.
// xs = table of w_n^i (two-level, forward) to get x = w^i.
struct GlpPermArgs {
    const u64* wires; const u64* sigmas; const u64* ks;
    u32 log_n; u32 W;
    u64 beta[GLP_PLONK_NCHAL], gamma[GLP_PLONK_NCHAL];
    const u64* w_lo; const u64* w_hi;
    u64* qv; u64* rr;
};
// One field inversion per (row, challenge): the M chunk denominators are inverted together
// (prefix products parked in qv, which is overwritten by the quotients on the way back).
template <int UNUSED = 0>
__global__ void __launch_bounds__(256) glp_perm_quotients_kernel(GlpPermArgs a) {
    const u64 n = 1ull << a.log_n;
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u32 M = a.W / GLP_PLONK_CHUNK;
    u64 x = a.w_lo[i & 4095u];
    if (a.w_hi) x = gl_mul(x, a.w_hi[i >> 12]);
    for (u32 t = 0; t < GLP_PLONK_NCHAL; t++) {
        const u64 bx = gl_mul(a.beta[t], x);
        auto chunk = [&](u32 c, u64& num, u64& den) {
            num = 1; den = 1;
            glp_static_for<0, GLP_PLONK_CHUNK>([&](auto j_) {
                constexpr int jj = decltype(j_)::value;
                const u32 j = c * GLP_PLONK_CHUNK + jj;
                const u64 wg = gl_add(a.wires[(u64)j * n + i], a.gamma[t]);
                num = gl_mul(num, gl_add(wg, gl_mul(bx, a.ks[j])));
                den = gl_mul(den, gl_add(wg, gl_mul(a.beta[t], a.sigmas[(u64)j * n + i])));
            });
        };
        u64 run = 1;                                   // prefix products of the denominators
        for (u32 c = 0; c < M; c++) {
            u64 num, den;
            chunk(c, num, den);
            run = gl_mul(run, den);
            a.qv[((u64)t * M + c) * n + i] = run;
        }
        u64 inv = gl_inv(run);                          // != 0 except with negligible probability
        u64 ratio = 1;
        for (u32 c = M; c-- > 0;) {
            u64 num, den;
            chunk(c, num, den);
            const u64 before = c ? a.qv[((u64)t * M + c - 1) * n + i] : 1ull;
            const u64 q = gl_mul(num, gl_mul(inv, before));   // num_c / den_c
            inv = gl_mul(inv, den);
            a.qv[((u64)t * M + c) * n + i] = q;
            ratio = gl_mul(ratio, q);
        }
        a.rr[(u64)t * n + i] = ratio;
    }
}

// ---- K6b: exclusive prefix product over rows ------------------------------------------------
// Three launches: (1) every block reduces its 1024 rows to one product; (2) one block per challenge scans
// the block products; (3) every block rescans its rows with its prefix.
#define GLP_SCAN_BLOCK 1024u
template <int UNUSED = 0>
__global__ void __launch_bounds__(256) glp_scan_reduce_kernel(const u64* __restrict__ rr, u64 n, u64* __restrict__ block_prod) {
    __shared__ u64 sh[256];
    const u64 base = (u64)blockIdx.x * GLP_SCAN_BLOCK;     // blockIdx.y = challenge
    const u64* src = rr + (u64)blockIdx.y * n;
    u64 p = 1;
    for (u32 k = 0; k < GLP_SCAN_BLOCK / 256; k++) {
        const u64 idx = base + (u64)threadIdx.x * (GLP_SCAN_BLOCK / 256) + k;
        if (idx < n) p = gl_mul(p, src[idx]);
    }
    sh[threadIdx.x] = p;
    __syncthreads();
    for (u32 s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) sh[threadIdx.x] = gl_mul(sh[threadIdx.x], sh[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) block_prod[(u64)blockIdx.y * gridDim.x + blockIdx.x] = sh[0];
}
// one block per challenge: exclusive scan of the nb block products in place.  Each work-item owns a contiguous run of
// ceil(nb / GLP_SCAN_TOP) products (serial inside the run), the run totals are scanned through LDS (Hillis-Steele).
#define GLP_SCAN_TOP 256u
template <int UNUSED = 0>
__global__ void __launch_bounds__(GLP_SCAN_TOP) glp_scan_blocks_kernel(u64* __restrict__ block_prod, u32 nb) {
    __shared__ u64 sh[GLP_SCAN_TOP];
    u64* bp = block_prod + (u64)blockIdx.x * nb;
    const u32 per = (nb + GLP_SCAN_TOP - 1) / GLP_SCAN_TOP;
    const u32 b0 = threadIdx.x * per;
    u64 mine = 1;
    for (u32 k = 0; k < per; k++) if (b0 + k < nb) mine = gl_mul(mine, bp[b0 + k]);
    sh[threadIdx.x] = mine;
    __syncthreads();
    for (u32 off = 1; off < GLP_SCAN_TOP; off <<= 1) {
        const u64 add = (threadIdx.x >= off) ? sh[threadIdx.x - off] : 1;
        __syncthreads();
        sh[threadIdx.x] = gl_mul(sh[threadIdx.x], add);
        __syncthreads();
    }
    u64 run = threadIdx.x ? sh[threadIdx.x - 1] : 1;
    for (u32 k = 0; k < per; k++)
        if (b0 + k < nb) { const u64 v = bp[b0 + k]; bp[b0 + k] = run; run = gl_mul(run, v); }
}
// Z[t][i] = prefix(block) * prod_{i' in block, i' < i} rr[t][i'];  each thread handles 4 rows.
// Also writes the partial products pi_c = Z * q_0..q_c for c < M-1 into zs[t][1+c][i].
// zs layout: [NCHAL][M][n]  (poly 0 = Z, poly 1+c = pi_c).
template <int UNUSED = 0>
__global__ void __launch_bounds__(256) glp_scan_apply_kernel(const u64* __restrict__ rr, const u64* __restrict__ qv, u64 n, u32 M,
                                                             const u64* __restrict__ block_prefix, u64* __restrict__ zs) {
    __shared__ u64 sh[256];
    const u32 t = blockIdx.y;
    const u64 base = (u64)blockIdx.x * GLP_SCAN_BLOCK + (u64)threadIdx.x * 4;
    const u64* src = rr + (u64)t * n;
    u64 v[4], p = 1;
    for (u32 k = 0; k < 4; k++) { v[k] = (base + k < n) ? src[base + k] : 1; }
    const u64 mine = gl_mul(gl_mul(v[0], v[1]), gl_mul(v[2], v[3]));
    // inclusive scan of the 256 thread products (Hillis-Steele in LDS)
    sh[threadIdx.x] = mine;
    __syncthreads();
    for (u32 off = 1; off < 256; off <<= 1) {
        u64 add = (threadIdx.x >= off) ? sh[threadIdx.x - off] : 1;
        __syncthreads();
        sh[threadIdx.x] = gl_mul(sh[threadIdx.x], add);
        __syncthreads();
    }
    const u64 excl = (threadIdx.x == 0) ? 1 : sh[threadIdx.x - 1];
    p = gl_mul(block_prefix[(u64)t * gridDim.x + blockIdx.x], excl);
    for (u32 k = 0; k < 4; k++) {
        const u64 i = base + k;
        if (i < n) {
            zs[((u64)t * M + 0) * n + i] = p;
            u64 run = p;
            for (u32 c = 0; c + 1 < M; c++) {
                run = gl_mul(run, qv[((u64)t * M + c) * n + i]);
                zs[((u64)t * M + 1 + c) * n + i] = run;
            }
        }
        p = gl_mul(p, v[k]);
    }
}

// ---- K7: quotient evaluation on the LDE domain ------------------------------------------------
// All inputs are LDE values, polynomial-major [.][N], bit-reversed index order, N = n << rate_bits.
// consts: [GLP_PLONK_NCONST][N] (plonk_gates.h); sigmas: [R][N]; wires: [W][N]; zs: [NCHAL*M][N], M = R/8;
// pi: [N] the public-input polynomial (null when the circuit has none).
// Constraint order (index into alpha powers), per challenge t:
//   0            L_1(x) * (Z(x) - 1)
//   1            q_pi * wire_0 - PI(x)
//   2 + 3c       prev_c * prod(num) - next_c * prod(den)            chunk c of 8 routed wires
//   3+3c, 4+3c   the two arithmetic gates of chunk c (wires 8c..8c+3 and 8c+4..8c+7)
//   2+3M+k       q_pos * (constraint k of the Poseidon row), k < 123     (POS circuits only)
// out[t][i] = (sum alpha_t^idx * constraint_idx) / (x^n - 1).
struct GlpQuotientArgs {
    const u64* consts; const u64* sigmas; const u64* wires; const u64* zs; const u64* pi; const u64* ks;
    const u64* q_ext;              // [N] the extension-arithmetic selector (GLP_CIRCUIT_EXT_GATE circuits), else null
    u32 log_n; u32 rate_bits; u32 W; u32 R; u32 n_con;
    u64 beta[GLP_PLONK_NCHAL], gamma[GLP_PLONK_NCHAL];
    const u64* alpha_pow;          // [NCHAL][n_con]
    const u64* pos_consts;         // POS: rc[360], circ[12], diag[12] on the device
    const u64* w_lo; const u64* w_hi;   // forward table of w_N
    u64 shift;                     // coset shift of the LDE domain
    u64 zh_inv[64];                // 1 / (x^n - 1) for the 2^rate_bits values x^n takes, indexed by (natural index) mod 2^rate_bits
    u64 n_inv;                     // 1 / n
    const u64* inv_xm1;            // [N] 1 / (x_i - 1), bit-reversed order (built once per domain)
    u64* out;                      // [NCHAL][N]
};
// The 123 Poseidon-row constraints in the base field with the permutation kernels' arithmetic (hash_kernels.cuh): S-boxes on arbitrary u64
// representatives, the MDS layer on its small-integer accumulators (SMALL: every entry < 2^24 — 12 x 12 multiply-adds per round cost ~1.4 VALU each
// instead of a full field multiplication and addition, the difference between 95k and 6k VALU per LDE point), values canonicalised only where
// a constraint is emitted.  The same walk, in the same order, as glp_poseidon_gate_constraints<GlpGateBase> (which stays the definition: the
// verifier uses it in the extension field, K7 uses it when the MDS is not small, and the parity tests compare both paths with the restatement).
template <bool SMALL, class WireFn, class EmitFn>
GL_HD void glp_poseidon_gate_constraints_fast(WireFn&& wire, const u64* rc, const u64* circ, const u64* diag, EmitFn&& emit) {
    u64 s[12];
    {
        const u64 sw = wire(GLP_POS_SWAP_WIRE);
        emit(gl_sub(gl_mul(sw, sw), sw));
        u64 in[12];
        for (int i = 0; i < 12; i++) in[i] = wire(i);
        for (int i = 0; i < 4; i++) {
            const u64 d = wire(GLP_POS_DELTA0 + i);
            emit(gl_sub(d, gl_mul(sw, gl_sub(in[4 + i], in[i]))));
            in[i] = gl_add(in[i], d);
            in[4 + i] = gl_sub(in[4 + i], d);
        }
        for (int i = 0; i < 12; i++) s[i] = gl_add(in[i], rc[i]);
    }
    int rnd = 0, aw = GLP_POS_ADVICE0;
    for (int r = 0; r < GLP_POS_FULL_HALF; r++, rnd++) {
        if (r > 0) {
            for (int i = 0; i < 12; i++) { const u64 v = wire(aw + i); emit(gl_sub(v, gl_canon(s[i]))); s[i] = v; }
            aw += 12;
        }
        for (int i = 0; i < 12; i++) s[i] = glp_sbox7(s[i]);
        glp_mds_layer<SMALL>(s, circ, diag, rc + (rnd + 1) * 12);
    }
    for (int r = 0; r < GLP_POS_PARTIAL; r++, rnd++) {
        const u64 p = wire(aw++);
        emit(gl_sub(p, gl_canon(s[0])));
        s[0] = glp_sbox7(p);
        glp_mds_layer<SMALL>(s, circ, diag, rc + (rnd + 1) * 12);
    }
    for (int r = 0; r < GLP_POS_FULL_HALF; r++, rnd++) {
        for (int i = 0; i < 12; i++) { const u64 v = wire(aw + i); emit(gl_sub(v, gl_canon(s[i]))); s[i] = glp_sbox7(v); }
        aw += 12;
        glp_mds_layer<SMALL>(s, circ, diag, rnd + 1 < GLP_POS_ROUNDS ? rc + (rnd + 1) * 12 : nullptr);
    }
    for (int i = 0; i < 12; i++) emit(gl_sub(wire(12 + i), gl_canon(s[i])));
}

template <bool POS, bool SMALL_MDS = false>
__global__ void __launch_bounds__(256) glp_quotient_kernel(GlpQuotientArgs a) {
    const u32 log_N = a.log_n + a.rate_bits;
    const u64 N = 1ull << log_N;
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const u32 M = a.R / GLP_PLONK_CHUNK;
    u64 e = 0;                                   // natural index of this point
    for (u32 b = 0; b < log_N; b++) e |= ((i >> b) & 1ull) << (log_N - 1 - b);
    u64 x = a.w_lo[e & 4095u];
    if (a.w_hi) x = gl_mul(x, a.w_hi[e >> 12]);
    x = gl_mul(x, a.shift);
    // the "next row" point g*x is natural index e + 2^rate_bits (mod N) -> its bit-reversed position
    const u64 en = (e + (1ull << a.rate_bits)) & (N - 1);
    u64 inext = 0;
    for (u32 b = 0; b < log_N; b++) inext |= ((en >> b) & 1ull) << (log_N - 1 - b);
    const u64 zhi = a.zh_inv[e & ((1ull << a.rate_bits) - 1)];
    // L_1(x) * (Z - 1) / (x^n - 1) = (Z - 1) / (n (x - 1)): the vanishing factor cancels
    const u64 l1_over_zh = gl_mul(a.n_inv, a.inv_xm1[i]);
    const u64 q = a.consts[i], c0 = a.consts[N + i], c1 = a.consts[2 * N + i], c2 = a.consts[3 * N + i], q_pi = a.consts[4 * N + i];
    const u64 qx = a.q_ext ? a.q_ext[i] : 0ull;
    u64 acc[GLP_PLONK_NCHAL], prev[GLP_PLONK_NCHAL], bx[GLP_PLONK_NCHAL];
    // public inputs (alpha^1)
    const u64 pi_con = gl_sub(gl_mul(q_pi, a.wires[i]), a.pi ? a.pi[i] : 0ull);
    for (u32 t = 0; t < GLP_PLONK_NCHAL; t++) {
        const u64 z = a.zs[((u64)t * M) * N + i];
        acc[t] = gl_mul(a.alpha_pow[(u64)t * a.n_con + 1], pi_con);
        prev[t] = z;
        bx[t] = gl_mul(a.beta[t], x);
    }
    for (u32 c = 0; c < M; c++) {
        u64 w[GLP_PLONK_CHUNK], sg[GLP_PLONK_CHUNK], kk[GLP_PLONK_CHUNK];
        glp_static_for<0, GLP_PLONK_CHUNK>([&](auto j_) {
            constexpr int jj = decltype(j_)::value;
            const u64 j = (u64)c * GLP_PLONK_CHUNK + jj;
            w[jj] = a.wires[j * N + i];
            sg[jj] = a.sigmas[j * N + i];
            kk[jj] = a.ks[j];
        });
        // the two arithmetic gates of this chunk (shared by both challenges up to alpha)
        u64 g0 = gl_mul(q, glp_arith_gate<GlpGateBase>(c0, c1, c2, w[0], w[1], w[2], w[3]));
        u64 g1 = gl_mul(q, glp_arith_gate<GlpGateBase>(c0, c1, c2, w[4], w[5], w[6], w[7]));
        if (a.q_ext) {                           // wave-uniform: the chunk as one extension multiply-add, in the same two slots
            u64 e0, e1;
            glp_ext_gate<GlpGateBase>(w, e0, e1);
            g0 = gl_add(g0, gl_mul(qx, e0));
            g1 = gl_add(g1, gl_mul(qx, e1));
        }
        for (u32 t = 0; t < GLP_PLONK_NCHAL; t++) {
            u64 num = 1, den = 1;
            glp_static_for<0, GLP_PLONK_CHUNK>([&](auto j_) {
                constexpr int jj = decltype(j_)::value;
                const u64 wg = gl_add(w[jj], a.gamma[t]);
                num = gl_mul(num, gl_add(wg, gl_mul(bx[t], kk[jj])));
                den = gl_mul(den, gl_add(wg, gl_mul(a.beta[t], sg[jj])));
            });
            const u64 next = (c + 1 < M) ? a.zs[((u64)t * M + 1 + c) * N + i] : a.zs[((u64)t * M) * N + inext];
            const u64 perm = gl_sub(gl_mul(prev[t], num), gl_mul(next, den));
            const u64* ap = a.alpha_pow + (u64)t * a.n_con + 2 + 3 * c;
            acc[t] = gl_add(acc[t], gl_mul(ap[0], perm));
            acc[t] = gl_add(acc[t], gl_mul(ap[1], g0));
            acc[t] = gl_add(acc[t], gl_mul(ap[2], g1));
            prev[t] = next;
        }
    }
    if constexpr (POS) {
        // one Poseidon row per point: the 123 constraints, each weighted by its alpha power, then the selector once
        const u64 q_pos = a.consts[5 * N + i];
        u64 pacc[GLP_PLONK_NCHAL];
        for (u32 t = 0; t < GLP_PLONK_NCHAL; t++) pacc[t] = 0;
        u32 k = 0;
        const u64* ap0 = a.alpha_pow + 2 + 3 * M;
        auto wire = [&](int j) -> u64 { return a.wires[(u64)j * N + i]; };
        auto emit = [&](u64 con) {
            for (u32 t = 0; t < GLP_PLONK_NCHAL; t++) pacc[t] = gl_add(pacc[t], gl_mul(ap0[(u64)t * a.n_con + k], con));
            k++;
        };
        if constexpr (SMALL_MDS) glp_poseidon_gate_constraints_fast<true>(wire, a.pos_consts, a.pos_consts + 360, a.pos_consts + 372, emit);
        else glp_poseidon_gate_constraints<GlpGateBase>(wire, a.pos_consts, a.pos_consts + 360, a.pos_consts + 372, emit);
        for (u32 t = 0; t < GLP_PLONK_NCHAL; t++) acc[t] = gl_add(acc[t], gl_mul(q_pos, pacc[t]));
    }
    for (u32 t = 0; t < GLP_PLONK_NCHAL; t++) {
        const u64 z = a.zs[((u64)t * M) * N + i];
        a.out[(u64)t * N + i] = gl_add(gl_mul(acc[t], zhi), gl_mul(l1_over_zh, gl_sub(z, 1)));   // alpha^0 = 1
    }
}

// ---- K7s: the SHA-row block of the constraint sum (GLP_CIRCUIT_SHA_GATES circuits), added to what K7 wrote ----------------
// consts then has GLP_PLONK_NCONST_SHA rows; the block's alpha powers start at first_con = 2 + 3M (+ 118 with Poseidon rows).
// A kernel of its own: the 64 bit wires it keeps live (X and Y groups, for the rotations) would push K7 into spills, and circuits
// without SHA rows do not pay for it.
template <int UNUSED = 0>
__global__ void __launch_bounds__(256) glp_quotient_sha_kernel(GlpQuotientArgs a, u32 first_con) {
    const u32 log_N = a.log_n + a.rate_bits;
    const u64 N = 1ull << log_N;
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    u64 e = 0;
    for (u32 b = 0; b < log_N; b++) e |= ((i >> b) & 1ull) << (log_N - 1 - b);
    const u64 zhi = a.zh_inv[e & ((1ull << a.rate_bits) - 1)];
    const u64 q[4] = {a.consts[6 * N + i], a.consts[7 * N + i], a.consts[8 * N + i], a.consts[9 * N + i]};
    const u64 c2 = a.consts[3 * N + i];
    u64 acc[GLP_PLONK_NCHAL];
    for (u32 t = 0; t < GLP_PLONK_NCHAL; t++) acc[t] = 0;
    u32 k = 0;
    const u64* ap0 = a.alpha_pow + first_con;
    glp_sha_gate_constraints<GlpGateBase>([&](int j) -> u64 { return a.wires[(u64)j * N + i]; }, q, c2, [&](u64 con) {
        for (u32 t = 0; t < GLP_PLONK_NCHAL; t++) acc[t] = gl_add(acc[t], gl_mul(ap0[(u64)t * a.n_con + k], con));
        k++;
    });
    for (u32 t = 0; t < GLP_PLONK_NCHAL; t++) a.out[(u64)t * N + i] = gl_add(a.out[(u64)t * N + i], gl_mul(acc[t], zhi));
}

// ---- SHA-row witness: for each listed row of the listed kind, the bit wires 12..143 from the routed words 0..11 (plonk_gates.h) ----
template <int UNUSED = 0>
__global__ void __launch_bounds__(64) glp_sha_gate_fill_kernel(u64* __restrict__ wires, u64 n, const u32* __restrict__ rows, const u32* __restrict__ kinds,
                                                              u32 n_rows) {
    const u32 k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_rows) return;
    const u64 row = rows[k];
    if (row >= n || kinds[k] > 3) return;
    u64 r[12], bits[132];
    for (int j = 0; j < 12; j++) r[j] = wires[(u64)j * n + row];
    glp_sha_gate_fill((int)kinds[k], r, bits);
    for (int j = 0; j < 132; j++) wires[(u64)(12 + j) * n + row] = bits[j];
}

// ---- Poseidon-row witness: for each listed row, wires 12..134 from wires 0..11 and the swap bit in wire 24 (plonk_gates.h) ----
// wires: [W][n] values on the trace domain; rows: n_rows row indices; consts: rc[360], circ[12], diag[12] (device)
// SMALL_MDS: the permutation kernels' arithmetic (S-boxes on arbitrary representatives, small-integer MDS accumulators), every stored value
// canonicalised — the same values as glp_poseidon_gate_fill (the definition, used when the MDS is not small; tests compare both with the
// restatement), at a sixth of the instructions (117k -> ~20k VALU per row).
template <bool SMALL_MDS>
GL_HD void glp_poseidon_gate_fill_fast(const u64 (&in_)[12], u64 swap, const u64* rc, const u64* circ, const u64* diag, u64 (&out)[GLP_POS_GATE_WIRES - 12]) {
    u64 in[12], s[12];
    for (int i = 0; i < 12; i++) in[i] = in_[i];
    out[12] = swap;
    for (int i = 0; i < 4; i++) {
        const u64 d = gl_mul(swap, gl_sub(in[4 + i], in[i]));
        out[GLP_POS_DELTA0 - 12 + i] = d;
        in[i] = gl_add(in[i], d);
        in[4 + i] = gl_sub(in[4 + i], d);
    }
    for (int i = 0; i < 12; i++) s[i] = gl_add(in[i], rc[i]);
    int rnd = 0, aw = GLP_POS_ADVICE0 - 12;
    for (int r = 0; r < GLP_POS_FULL_HALF; r++, rnd++) {
        if (r > 0) { for (int i = 0; i < 12; i++) { s[i] = gl_canon(s[i]); out[aw + i] = s[i]; } aw += 12; }
        for (int i = 0; i < 12; i++) s[i] = glp_sbox7(s[i]);
        glp_mds_layer<SMALL_MDS>(s, circ, diag, rc + (rnd + 1) * 12);
    }
    for (int r = 0; r < GLP_POS_PARTIAL; r++, rnd++) {
        s[0] = gl_canon(s[0]);
        out[aw++] = s[0];
        s[0] = glp_sbox7(s[0]);
        glp_mds_layer<SMALL_MDS>(s, circ, diag, rc + (rnd + 1) * 12);
    }
    for (int r = 0; r < GLP_POS_FULL_HALF; r++, rnd++) {
        for (int i = 0; i < 12; i++) { s[i] = gl_canon(s[i]); out[aw + i] = s[i]; s[i] = glp_sbox7(s[i]); }
        aw += 12;
        glp_mds_layer<SMALL_MDS>(s, circ, diag, rnd + 1 < GLP_POS_ROUNDS ? rc + (rnd + 1) * 12 : nullptr);
    }
    for (int i = 0; i < 12; i++) out[i] = gl_canon(s[i]);
}

template <int SMALL_MDS = 0>
__global__ void __launch_bounds__(64) glp_poseidon_gate_fill_kernel(u64* __restrict__ wires, u64 n, const u32* __restrict__ rows, u32 n_rows,
                                                                   const u64* __restrict__ consts) {
    const u32 k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_rows) return;
    const u64 row = rows[k];
    if (row >= n) return;
    u64 in[12], out[GLP_POS_GATE_WIRES - 12];
    for (int j = 0; j < 12; j++) in[j] = wires[(u64)j * n + row];
    const u64 swap = wires[(u64)GLP_POS_SWAP_WIRE * n + row];
    if constexpr (SMALL_MDS != 0) glp_poseidon_gate_fill_fast<true>(in, swap, consts, consts + 360, consts + 372, out);
    else glp_poseidon_gate_fill(in, swap, consts, consts + 360, consts + 372, out);
    for (int j = 0; j < GLP_POS_GATE_WIRES - 12; j++) wires[(u64)(12 + j) * n + row] = out[j];
}

// inv[i] = 1 / (x_i - 1) over the LDE domain (bit-reversed order), 4 points per work-item share
// one inversion.  Built once per (log_N, shift).
template <int UNUSED = 0>
__global__ void __launch_bounds__(256) glp_inv_xm1_kernel(u64* __restrict__ inv, u32 log_N, u64 shift, const u64* __restrict__ w_lo,
                                                          const u64* __restrict__ w_hi) {
    const u64 N = 1ull << log_N;
    const u64 i0 = ((u64)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i0 >= N) return;
    u64 d[4];
    glp_static_for<0, 4>([&](auto k_) {
        constexpr int k = decltype(k_)::value;
        const u64 i = i0 + k;
        u64 e = 0;
        for (u32 b = 0; b < log_N; b++) e |= ((i >> b) & 1ull) << (log_N - 1 - b);
        u64 x = w_lo[e & 4095u];
        if (w_hi) x = gl_mul(x, w_hi[e >> 12]);
        d[k] = gl_sub(gl_mul(x, shift), 1);
    });
    const u64 p01 = gl_mul(d[0], d[1]), p012 = gl_mul(p01, d[2]);
    u64 t = gl_inv(gl_mul(p012, d[3]));
    inv[i0 + 3] = gl_mul(t, p012); t = gl_mul(t, d[3]);
    inv[i0 + 2] = gl_mul(t, p01); t = gl_mul(t, d[2]);
    inv[i0 + 1] = gl_mul(t, d[0]);
    inv[i0 + 0] = gl_mul(t, d[1]);
}

// out[rev(i)] = in[i] per polynomial (bit-reversed <-> natural order)
template <int UNUSED = 0>
__global__ void __launch_bounds__(256) glp_bitrev_permute_kernel(const u64* __restrict__ in, u64* __restrict__ out, u32 log_n,
                                                                 u32 batch) {
    const u64 n = 1ull << log_n;
    const u64 g = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= ((u64)batch << log_n)) return;
    const u64 b = g >> log_n, i = g & (n - 1);
    u64 r = 0;
    for (u32 k = 0; k < log_n; k++) r |= ((i >> k) & 1ull) << (log_n - 1 - k);
    out[b * n + r] = in[g];
}

// data[b][j] *= s^j  (s^j = lo[j & 4095] * hi[j >> 12])
template <int UNUSED = 0>
__global__ void __launch_bounds__(256) glp_scale_pow_kernel(u64* __restrict__ data, u32 log_n, u32 batch, const u64* __restrict__ lo,
                                                            const u64* __restrict__ hi) {
    const u64 n = 1ull << log_n;
    const u64 total = (u64)batch << log_n;
    for (u64 g = (u64)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (u64)gridDim.x * blockDim.x) {
        const u64 j = g & (n - 1);
        u64 s = lo[j & 4095u];
        if (hi) s = gl_mul(s, hi[j >> 12]);
        data[g] = gl_mul(data[g], s);
    }
}
