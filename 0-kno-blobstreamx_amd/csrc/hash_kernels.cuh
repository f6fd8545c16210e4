// hash_kernels.cuh — Poseidon-Goldilocks permutation (width 12, x^7, 4 + 22 + 4 rounds),
// overwrite-mode sponge (rate 8), 2-to-1 compression, Merkle levels; arity-2 FRI fold;
// SHA-256 / SHA-512 compression with per-round trace.   SURVEY.md §8a rows a4, a8, a9.
// Upstream names (recalled, unverified; reference file:line NONE — the mount is empty):
// plonky2::hash::poseidon::Poseidon::poseidon, hashing::hash_n_to_hash_no_pad,
// PoseidonHash::two_to_one, merkle_tree::MerkleTree::new, fri::prover (fold), curta SHA chips.
//
// Round constants and the MDS vectors are INJECTED (glp_set_poseidon_constants): the
// library ships none, so digests are "self-consistent, not plonky2-compatible" until the
// real table is supplied.
//
// One permutation per work-item, the 12-word state in 24 VGPRs; constants are read with
// wave-uniform addresses (scalar loads).  When every MDS entry and their sum is < 2^24 (true
// for any small-integer MDS such as plonky2's), a row of the MDS layer is two 64-bit
// accumulators of 32x32 products (lo halves and hi halves of the state) and one cheap fix-up,
// instead of twelve full field multiplications; the next round's constants ride along in
// the accumulators.
//
// Plain HIP C++ without AMD builtins: tests/emu runs these bodies on the CPU.
#pragma once
#include "gl_field.cuh"

#define GLP_POS_WIDTH 12
#define GLP_POS_RATE 8
#define GLP_POS_FULL_HALF 4
#define GLP_POS_PARTIAL 22
#define GLP_POS_ROUNDS 30

struct GlpPoseidonConsts {
    const u64* rc;     // [30][12]
    const u64* circ;   // [12]
    const u64* diag;   // [12]
    // grouped partial rounds (nullptr = plain): per group of 3 rounds GLP_PG_COEF u32 + GLP_PG_CST u64
    const u32* pg_coef;
    const u64* pg_cst;
};
#define GLP_PG_K 3                 // partial rounds per group
#define GLP_PG_GROUPS 7            // 7 x 3 = 21 of the 22 partial rounds; the last one runs plain
#define GLP_PG_COEF (12 + 13 + 12 * 14)
#define GLP_PG_CST (1 + 1 + 12)

#define glp_hfor glp_static_for

// a scheduling fence for the device compiler (no instruction is emitted): keeps it from hoisting the scalar loads of LATER
// dot-product rows above the current row's arithmetic — hoisted all at once they need > 100 SGPRs and spill into VGPR lanes
// (v_writelane / v_readlane: 20 % of a partial-round group's VALU instructions before this fence; with it none, VGPRs 69 -> 62,
// 2^20 x 80 proof 86.7 -> 83.3 ms, wires commitment 46.0 -> 43.0 ms in a same-box A/B: profiles/r02_ab_sched_fence.txt)
#if defined(__HIP_DEVICE_COMPILE__) && !defined(GLP_EMU)
#define GLP_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define GLP_SCHED_FENCE() ((void)0)
#endif

// x^7.  Inside a permutation values are kept as arbitrary u64 representatives (products and MDS
// rows skip the ">= p" check); glp_poseidon_permute canonicalises its 12 outputs once.
GL_HD u64 glp_sbox7(u64 x) {
    u64 x2 = gl_mul_nc(x, x), x3 = gl_mul_nc(x2, x), x4 = gl_mul_nc(x2, x2);
    return gl_mul_nc(x3, x4);
}

// s <- MDS * s (+ rc_next)  with  row r = sum_i s[(i + r) % 12] * circ[i] + s[r] * diag[r].
// rc_next (nullable) = the NEXT round's constants, folded into the accumulators so that the
// constant layer costs no VALU work of its own.
// SMALL (every entry and their sum < 2^24): a row is two 64-bit accumulators of 32x32 products,
//   al = sum lo_i c_i + rc_lo,  ah = sum hi_i c_i + rc_hi   (both < 2^57),
// and  al + ah*2^32 = al + (ah mod 2^32)*2^32 + (ah >> 32)*(2^32 - 1)  (mod p)  overflows 2^64 at
// most once, so one fused carry fix-up (gl_fold_small) finishes the row instead of a generic
// 128-bit reduction.
template <bool SMALL>
GL_HD void glp_mds_layer(u64 (&s)[12], const u64* __restrict__ circ, const u64* __restrict__ diag, const u64* __restrict__ rc_next) {
    u64 out[12];
    if constexpr (SMALL) {
        u32 lo[12], hi[12], c[12], dg[12];
        glp_hfor<0, 12>([&](auto i_) {
            constexpr int i = decltype(i_)::value;
            lo[i] = (u32)s[i]; hi[i] = (u32)(s[i] >> 32);
            c[i] = (u32)circ[i]; dg[i] = (u32)diag[i];
        });
        glp_hfor<0, 12>([&](auto r_) {
            constexpr int r = decltype(r_)::value;
            const u64 k = rc_next ? rc_next[r] : 0ull;
            u64 al = (u64)lo[r] * dg[r] + (u32)k, ah = (u64)hi[r] * dg[r] + (k >> 32);
            glp_hfor<0, 12>([&](auto i_) {
                constexpr int i = decltype(i_)::value;
                al += (u64)lo[(i + r) % 12] * c[i];
                ah += (u64)hi[(i + r) % 12] * c[i];
            });
            out[r] = gl_fold_small(al, ah);             // representative in [0, 2^64), canonicalised at the end
        });
    } else {
        glp_hfor<0, 12>([&](auto i_) { constexpr int i = decltype(i_)::value; s[i] = gl_canon(s[i]); });   // S-box outputs are not canonical
        glp_hfor<0, 12>([&](auto r_) {
            constexpr int r = decltype(r_)::value;
            u64 acc = gl_mul(s[r], diag[r]);
            glp_hfor<0, 12>([&](auto i_) {
                constexpr int i = decltype(i_)::value;
                acc = gl_add(acc, gl_mul(s[(i + r) % 12], circ[i]));
            });
            out[r] = rc_next ? gl_add(acc, rc_next[r]) : acc;
        });
    }
    glp_hfor<0, 12>([&](auto i_) { constexpr int i = decltype(i_)::value; s[i] = out[i]; });
}

// sum_i v_i * coef_i + cst (mod p) for small integer coefficients (each < 2^24, sum < 2^26): the
// lo and hi 32-bit halves of the v_i accumulate separately in 64 bits (no carries inside the sum),
// and one fused fix-up reduces  al + ah * 2^32  (see glp_mds_layer).
template <int N>
GL_HD u64 glp_dot_small(const u32 (&lo)[N], const u32 (&hi)[N], const u32* __restrict__ coef, u64 cst) {
    u64 al = (u32)cst, ah = cst >> 32;
    glp_hfor<0, N>([&](auto i_) {
        constexpr int i = decltype(i_)::value;
        al += (u64)lo[i] * coef[i];
        ah += (u64)hi[i] * coef[i];
    });
    return gl_fold_small(al, ah);                   // representative in [0, 2^64), not canonicalised
}

// Three consecutive partial rounds at once.  Only lane 0 is non-linear, so the state before
// round r0+j is an affine-linear function of (the 11 untouched lanes at r0, the S-box outputs
// f_0..f_{j-1}): the host multiplies the small-integer MDS out ahead of time (entries of M^3
// stay < 2^24) and the kernel evaluates two 12/13-term dot products for the next S-box inputs
// and one 12 x 14 product for the state after the group: 193 multiply-adds and 14 fix-ups
// instead of 432 and 36.  Identical results to three plain rounds (integer identities mod p).
GL_HD void glp_partial_group(u64 (&s)[12], const u32* __restrict__ cf, const u64* __restrict__ cs) {
    u32 lo[14], hi[14];                         // [0..10] = lanes 1..11 at r0, [11..13] = f_0..f_2
    glp_hfor<0, 11>([&](auto i_) { constexpr int i = decltype(i_)::value; lo[i] = (u32)s[i + 1]; hi[i] = (u32)(s[i + 1] >> 32); });
    const u64 f0 = glp_sbox7(s[0]);
    lo[11] = (u32)f0; hi[11] = (u32)(f0 >> 32);
    u32 l12[12], h12[12];
    glp_hfor<0, 12>([&](auto i_) { constexpr int i = decltype(i_)::value; l12[i] = lo[i]; h12[i] = hi[i]; });
    const u64 f1 = glp_sbox7(glp_dot_small<12>(l12, h12, cf, cs[0]));
    lo[12] = (u32)f1; hi[12] = (u32)(f1 >> 32);
    u32 l13[13], h13[13];
    glp_hfor<0, 13>([&](auto i_) { constexpr int i = decltype(i_)::value; l13[i] = lo[i]; h13[i] = hi[i]; });
    const u64 f2 = glp_sbox7(glp_dot_small<13>(l13, h13, cf + 12, cs[1]));
    lo[13] = (u32)f2; hi[13] = (u32)(f2 >> 32);
    glp_hfor<0, 12>([&](auto r_) {
        constexpr int r = decltype(r_)::value;
        GLP_SCHED_FENCE();
        s[r] = glp_dot_small<14>(lo, hi, cf + 25 + 14 * r, cs[2 + r]);
    });
    GLP_SCHED_FENCE();
}

// 4 full, 22 partial, 4 full rounds; round = add constants, x^7 (all lanes / lane 0), MDS.
// Written as: constants of round 0, then per round {S-box, MDS + constants of the next round}.
template <bool SMALL>
GL_HD void glp_poseidon_permute(u64 (&s)[12], const GlpPoseidonConsts& k) {
    glp_hfor<0, 12>([&](auto i_) { constexpr int i = decltype(i_)::value; s[i] = gl_add(s[i], k.rc[i]); });
    int rnd = 0;
    for (int r = 0; r < GLP_POS_FULL_HALF; r++, rnd++) {
        glp_hfor<0, 12>([&](auto i_) { constexpr int i = decltype(i_)::value; s[i] = glp_sbox7(s[i]); });
        glp_mds_layer<SMALL>(s, k.circ, k.diag, k.rc + (rnd + 1) * 12);
    }
    int r = 0;
    if constexpr (SMALL) {
        if (k.pg_coef) {
            for (int g = 0; g < GLP_PG_GROUPS; g++, r += GLP_PG_K, rnd += GLP_PG_K)
                glp_partial_group(s, k.pg_coef + g * GLP_PG_COEF, k.pg_cst + g * GLP_PG_CST);
        }
    }
    for (; r < GLP_POS_PARTIAL; r++, rnd++) {
        s[0] = glp_sbox7(s[0]);
        glp_mds_layer<SMALL>(s, k.circ, k.diag, k.rc + (rnd + 1) * 12);
    }
    for (int r = 0; r + 1 < GLP_POS_FULL_HALF; r++, rnd++) {
        glp_hfor<0, 12>([&](auto i_) { constexpr int i = decltype(i_)::value; s[i] = glp_sbox7(s[i]); });
        glp_mds_layer<SMALL>(s, k.circ, k.diag, k.rc + (rnd + 1) * 12);
    }
    // the last round has no "next constants": peeled so that the null test is a compile-time fact, not 12 branches per round
    glp_hfor<0, 12>([&](auto i_) { constexpr int i = decltype(i_)::value; s[i] = glp_sbox7(s[i]); });
    glp_mds_layer<SMALL>(s, k.circ, k.diag, nullptr);
    glp_hfor<0, 12>([&](auto i_) { constexpr int i = decltype(i_)::value; s[i] = gl_canon(s[i]); });
}

// n independent permutations, in place, states [n][12]
template <bool SMALL>
__global__ void __launch_bounds__(256) glp_poseidon_permute_kernel(u64* states, u64 n, GlpPoseidonConsts k) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u64 s[12];
    glp_hfor<0, 12>([&](auto j_) { constexpr int j = decltype(j_)::value; s[j] = states[i * 12 + j]; });
    glp_poseidon_permute<SMALL>(s, k);
    glp_hfor<0, 12>([&](auto j_) { constexpr int j = decltype(j_)::value; states[i * 12 + j] = s[j]; });
}

// Leaf digests.  Leaf i has leaf_len elements:
//   POLY_MAJOR: element j of leaf i = src[j * stride + i]   (columns of the LDE; coalesced over i)
//   else      : element j of leaf i = src[i * stride + j]   (leaf-major rows)
// leaf_len <= 4: the digest is the zero-padded leaf itself (hash_or_noop); otherwise the
// overwrite-mode sponge with rate 8.
template <bool SMALL, bool POLY_MAJOR>
__global__ void __launch_bounds__(256) glp_hash_leaves_kernel(const u64* __restrict__ src, u64 stride, u32 leaf_len,
                                                              u64 n_leaves, u64* __restrict__ digests, GlpPoseidonConsts k) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_leaves) return;
    auto at = [&](u32 j) -> u64 { return POLY_MAJOR ? src[(u64)j * stride + i] : src[i * stride + j]; };
    u64 s[12];
    glp_hfor<0, 12>([&](auto j_) { constexpr int j = decltype(j_)::value; s[j] = 0; });
    if (leaf_len <= 4) {
        glp_hfor<0, 4>([&](auto j_) { constexpr int j = decltype(j_)::value; if ((u32)j < leaf_len) s[j] = at(j); });
    } else {
        u32 off = 0;
        for (; off + GLP_POS_RATE <= leaf_len; off += GLP_POS_RATE) {
            glp_hfor<0, 8>([&](auto j_) { constexpr int j = decltype(j_)::value; s[j] = at(off + j); });
            glp_poseidon_permute<SMALL>(s, k);
        }
        if (off < leaf_len) {
            glp_hfor<0, 8>([&](auto j_) { constexpr int j = decltype(j_)::value; if (off + j < leaf_len) s[j] = at(off + j); });
            glp_poseidon_permute<SMALL>(s, k);
        }
    }
    glp_hfor<0, 4>([&](auto j_) { constexpr int j = decltype(j_)::value; digests[i * 4 + j] = s[j]; });
}

// one Merkle level: cur[i] = two_to_one(prev[2i], prev[2i+1]), digests are 4 u64
template <bool SMALL>
__global__ void __launch_bounds__(256) glp_merkle_level_kernel(const u64* __restrict__ prev, u64* __restrict__ cur, u64 count,
                                                               GlpPoseidonConsts k) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    u64 s[12];
    glp_hfor<0, 8>([&](auto j_) { constexpr int j = decltype(j_)::value; s[j] = prev[i * 8 + j]; });
    glp_hfor<8, 12>([&](auto j_) { constexpr int j = decltype(j_)::value; s[j] = 0; });
    glp_poseidon_permute<SMALL>(s, k);
    glp_hfor<0, 4>([&](auto j_) { constexpr int j = decltype(j_)::value; cur[i * 4 + j] = s[j]; });
}

// ---- lane-cooperative permutation: 16 lanes per permutation, 12 of them holding one state word each --------------------
// A Merkle level too small to fill the chip is, one permutation per lane, a ~40 us dependency chain run by a handful of
// waves (and a proof walks ~7 trees x 10 such levels).  Spreading ONE permutation over 12 lanes cuts the chain ~4x — the S-box
// of a full round runs on 12 lanes at once, each lane computes its own MDS row — at ~4x the instructions per permutation,
// a good trade only while most of the chip is idle (levels of <= GLP_COOP_MAX_NODES nodes).  State words cross lanes with
// wave shuffles (__shfl = ds_bpermute_b32 pairs).  Every lane of the wave must call it (idle lanes included): r = lane & 15,
// r >= 12 idle; lane_base = first lane of this 16-lane group inside its wave.  x = the lane's state word (canonical).
template <bool SMALL>
__device__ __forceinline__ u64 glp_poseidon_permute_coop(u64 x, u32 r, u32 lane_base, const GlpPoseidonConsts& k) {
    const u32 ri = r < 12u ? r : 0u;
    x = gl_add(x, k.rc[ri]);
    const u32 dg = (u32)k.diag[ri];
    for (int rnd = 0; rnd < GLP_POS_ROUNDS; rnd++) {
        const bool full = rnd < GLP_POS_FULL_HALF || rnd >= GLP_POS_FULL_HALF + GLP_POS_PARTIAL;
        const u64 sx = glp_sbox7(x);
        x = (full || r == 0u) ? sx : x;
        const u64 kn = (rnd + 1 < GLP_POS_ROUNDS) ? k.rc[(rnd + 1) * 12 + ri] : 0ull;
        if constexpr (SMALL) {
            u64 al = (u64)(u32)x * dg + (u32)kn, ah = (u64)(u32)(x >> 32) * dg + (kn >> 32);
            glp_hfor<0, 12>([&](auto i_) {
                constexpr int i = decltype(i_)::value;
                u32 j = ri + (u32)i;
                if (j >= 12u) j -= 12u;
                const u64 v = (u64)__shfl((unsigned long long)x, (int)(lane_base + j));
                al += (u64)(u32)v * (u32)k.circ[i];
                ah += (u64)(u32)(v >> 32) * (u32)k.circ[i];
            });
            x = gl_fold_small(al, ah);
        } else {
            x = gl_canon(x);
            u64 acc = gl_mul(x, k.diag[ri]);
            glp_hfor<0, 12>([&](auto i_) {
                constexpr int i = decltype(i_)::value;
                u32 j = ri + (u32)i;
                if (j >= 12u) j -= 12u;
                const u64 v = (u64)__shfl((unsigned long long)x, (int)(lane_base + j));
                acc = gl_add(acc, gl_mul(v, k.circ[i]));
            });
            x = gl_add(acc, kn);
        }
    }
    return gl_canon(x);
}

// one Merkle level, 16 lanes per node (small levels: see above)
template <bool SMALL>
__global__ void __launch_bounds__(256) glp_merkle_level_coop_kernel(const u64* __restrict__ prev, u64* __restrict__ cur, u64 count,
                                                                    GlpPoseidonConsts k) {
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const u64 node = t >> 4;
    const u32 r = threadIdx.x & 15u, lane_base = (threadIdx.x & 63u) & ~15u;
    const bool active = node < count;
    u64 x = 0;
    if (active && r < 8u) x = prev[node * 8 + r];
    if (((t >> 6) << 2) < count) x = glp_poseidon_permute_coop<SMALL>(x, r, lane_base, k);      // whole waves past the last node skip (wave-uniform)
    if (active && r < 4u) cur[node * 4 + r] = x;
}

// the top of a tree in ONE launch: starting from a level of count_prev <= 128 nodes at `prev`, n_levels further levels
// (each stored right after its predecessor, the layout of glp_merkle), one workgroup, levels handed on through LDS
#define GLP_COOP_TOP_NODES 64u
template <bool SMALL>
__global__ void __launch_bounds__(1024) glp_merkle_top_coop_kernel(u64* __restrict__ prev, u64 count_prev, u32 n_levels, GlpPoseidonConsts k) {
    __shared__ u64 buf[2][GLP_COOP_TOP_NODES * 4];
    const u32 node = threadIdx.x >> 4, r = threadIdx.x & 15u, lane_base = (threadIdx.x & 63u) & ~15u;
    u64 cnt = count_prev;
    for (u32 l = 0; l < n_levels; l++) {
        u64* cur = prev + 4 * cnt;
        cnt >>= 1;
        const bool active = node < cnt;
        u64 x = 0;
        if (active && r < 8u) x = l ? buf[(l - 1) & 1][node * 8 + r] : prev[(u64)node * 8 + r];
        // a wave (4 nodes) with no node left at this level sits the permutation out — the shuffles are wave-local, so only whole waves may —
        // instead of sharing its SIMD's issue slots with the waves that still have work: the 16 waves of the workgroup sit 4 to a SIMD, and
        // the upper levels need 8, 4, 2, 1, 1 of them (206 -> ~60 us per tree top)
        if ((u64)((threadIdx.x >> 6) << 2) < cnt) x = glp_poseidon_permute_coop<SMALL>(x, r, lane_base, k);
        if (active && r < 4u) { cur[(u64)node * 4 + r] = x; buf[l & 1][node * 4 + r] = x; }
        __syncthreads();
        prev = cur;
    }
}

// ---- FRI arity-2 fold (row a8) -------------------------------------------------------------
// evals [n][2] (extension elements) in bit-reversed order over shift*<w_n>; pair i =
// (f(x), f(-x)) with x = shift * w^{rev(i)} (rev over log_n - 1 bits).
//   out[i] = (f(x)+f(-x))/2 + beta * (f(x)-f(-x)) / (2x)
// half_inv = 1/2, c = 1/(2*shift); iw_lo/iw_hi = two-level table of w_n^{-e}.
template <int UNUSED = 0>
__global__ void __launch_bounds__(256) glp_fri_fold2_kernel(const u64* __restrict__ evals, u64* __restrict__ out, u32 log_n,
                                                            u64 half_inv, u64 c, gl_ext2 beta, const u64* __restrict__ iw_lo,
                                                            const u64* __restrict__ iw_hi) {
    const u64 half = 1ull << (log_n - 1);
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= half) return;
    // bit reversal of i over (log_n - 1) bits, portable form (also runs under tests/emu)
    u64 e = 0;
    for (u32 b = 0; b + 1 < log_n; b++) e |= ((i >> b) & 1ull) << (log_n - 2 - b);
    u64 xi = iw_lo[e & 4095u];
    if (iw_hi) xi = gl_mul(xi, iw_hi[e >> 12]);
    xi = gl_mul(xi, c);                                   // 1 / (2x)
    const gl_ext2 f0{evals[4 * i], evals[4 * i + 1]}, f1{evals[4 * i + 2], evals[4 * i + 3]};
    const gl_ext2 sum = gl_ext_scale(gl_ext_add(f0, f1), half_inv);
    const gl_ext2 dif = gl_ext_scale(gl_ext_sub(f0, f1), xi);
    const gl_ext2 r = gl_ext_add(sum, gl_ext_mul(beta, dif));
    out[2 * i] = r.a;
    out[2 * i + 1] = r.b;
}

// ---- SHA-2 witness traces (row a9) ---------------------------------------------------------
GL_HD u32 glp_ror32(u32 x, int r) { return (x >> r) | (x << (32 - r)); }
GL_HD u64 glp_ror64(u64 x, int r) { return (x >> r) | (x << (64 - r)); }

// One work-item per message.  blocks: [n_msgs][blocks_per_msg][64] bytes, already padded.
// digests [n_msgs][8] u32; trace (optional) [n_msgs][blocks][576]: w[64] then 64 x (a..h).
// k256: the 64 round constants (device table, uniform reads).
template <int UNUSED = 0>
__global__ void __launch_bounds__(256) glp_sha256_trace_kernel(const uint8_t* __restrict__ blocks, u64 n_msgs, u32 bpm,
                                                               u32* __restrict__ digests, u32* __restrict__ trace,
                                                               const u32* __restrict__ k256) {
    const u64 m = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= n_msgs) return;
    u32 h[8] = {0x6a09e667u, 0xbb67ae85u, 0x3c6ef372u, 0xa54ff53au, 0x510e527fu, 0x9b05688cu, 0x1f83d9abu, 0x5be0cd19u};
    for (u32 b = 0; b < bpm; b++) {
        const uint8_t* p = blocks + (m * bpm + b) * 64;
        u32* tr = trace ? trace + (m * bpm + b) * 576 : nullptr;
        u32 w[16];
        for (int i = 0; i < 16; i++) w[i] = ((u32)p[4 * i] << 24) | ((u32)p[4 * i + 1] << 16) | ((u32)p[4 * i + 2] << 8) | p[4 * i + 3];
        u32 s0 = h[0], s1 = h[1], s2 = h[2], s3 = h[3], s4 = h[4], s5 = h[5], s6 = h[6], s7 = h[7];
        #pragma unroll
        for (int i = 0; i < 64; i++) {
            u32 wi;
            if (i < 16) wi = w[i];
            else {
                const u32 w15 = w[(i - 15) & 15], w2 = w[(i - 2) & 15];
                const u32 g0 = glp_ror32(w15, 7) ^ glp_ror32(w15, 18) ^ (w15 >> 3);
                const u32 g1 = glp_ror32(w2, 17) ^ glp_ror32(w2, 19) ^ (w2 >> 10);
                wi = w[i & 15] + g0 + w[(i - 7) & 15] + g1;
                w[i & 15] = wi;
            }
            if (tr) tr[i] = wi;
            const u32 S1 = glp_ror32(s4, 6) ^ glp_ror32(s4, 11) ^ glp_ror32(s4, 25);
            const u32 ch = (s4 & s5) ^ (~s4 & s6);
            const u32 t1 = s7 + S1 + ch + k256[i] + wi;
            const u32 S0 = glp_ror32(s0, 2) ^ glp_ror32(s0, 13) ^ glp_ror32(s0, 22);
            const u32 mj = (s0 & s1) ^ (s0 & s2) ^ (s1 & s2);
            const u32 t2 = S0 + mj;
            s7 = s6; s6 = s5; s5 = s4; s4 = s3 + t1; s3 = s2; s2 = s1; s1 = s0; s0 = t1 + t2;
            if (tr) {
                u32* q = tr + 64 + 8 * i;
                q[0] = s0; q[1] = s1; q[2] = s2; q[3] = s3; q[4] = s4; q[5] = s5; q[6] = s6; q[7] = s7;
            }
        }
        h[0] += s0; h[1] += s1; h[2] += s2; h[3] += s3; h[4] += s4; h[5] += s5; h[6] += s6; h[7] += s7;
    }
    for (int i = 0; i < 8; i++) digests[m * 8 + i] = h[i];
}

// 128-byte blocks; digests [n_msgs][8] u64; trace [n_msgs][blocks][720]: w[80] then 80 x (a..h)
template <int UNUSED = 0>
__global__ void __launch_bounds__(256) glp_sha512_trace_kernel(const uint8_t* __restrict__ blocks, u64 n_msgs, u32 bpm,
                                                               u64* __restrict__ digests, u64* __restrict__ trace,
                                                               const u64* __restrict__ k512) {
    const u64 m = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= n_msgs) return;
    u64 h[8] = {0x6a09e667f3bcc908ull, 0xbb67ae8584caa73bull, 0x3c6ef372fe94f82bull, 0xa54ff53a5f1d36f1ull,
                0x510e527fade682d1ull, 0x9b05688c2b3e6c1full, 0x1f83d9abfb41bd6bull, 0x5be0cd19137e2179ull};
    for (u32 b = 0; b < bpm; b++) {
        const uint8_t* p = blocks + (m * bpm + b) * 128;
        u64* tr = trace ? trace + (m * bpm + b) * 720 : nullptr;
        u64 w[16];
        for (int i = 0; i < 16; i++) {
            u64 v = 0;
            for (int j = 0; j < 8; j++) v = (v << 8) | p[8 * i + j];
            w[i] = v;
        }
        u64 s0 = h[0], s1 = h[1], s2 = h[2], s3 = h[3], s4 = h[4], s5 = h[5], s6 = h[6], s7 = h[7];
        #pragma unroll
        for (int i = 0; i < 80; i++) {
            u64 wi;
            if (i < 16) wi = w[i];
            else {
                const u64 w15 = w[(i - 15) & 15], w2 = w[(i - 2) & 15];
                const u64 g0 = glp_ror64(w15, 1) ^ glp_ror64(w15, 8) ^ (w15 >> 7);
                const u64 g1 = glp_ror64(w2, 19) ^ glp_ror64(w2, 61) ^ (w2 >> 6);
                wi = w[i & 15] + g0 + w[(i - 7) & 15] + g1;
                w[i & 15] = wi;
            }
            if (tr) tr[i] = wi;
            const u64 S1 = glp_ror64(s4, 14) ^ glp_ror64(s4, 18) ^ glp_ror64(s4, 41);
            const u64 ch = (s4 & s5) ^ (~s4 & s6);
            const u64 t1 = s7 + S1 + ch + k512[i] + wi;
            const u64 S0 = glp_ror64(s0, 28) ^ glp_ror64(s0, 34) ^ glp_ror64(s0, 39);
            const u64 mj = (s0 & s1) ^ (s0 & s2) ^ (s1 & s2);
            const u64 t2 = S0 + mj;
            s7 = s6; s6 = s5; s5 = s4; s4 = s3 + t1; s3 = s2; s2 = s1; s1 = s0; s0 = t1 + t2;
            if (tr) {
                u64* q = tr + 80 + 8 * i;
                q[0] = s0; q[1] = s1; q[2] = s2; q[3] = s3; q[4] = s4; q[5] = s5; q[6] = s6; q[7] = s7;
            }
        }
        h[0] += s0; h[1] += s1; h[2] += s2; h[3] += s3; h[4] += s4; h[5] += s5; h[6] += s6; h[7] += s7;
    }
    for (int i = 0; i < 8; i++) digests[m * 8 + i] = h[i];
}

// ---- Tendermint "simple" Merkle tree over SHA-256 (RFC 6962 domain separation) ----------------
// The header / validator-set hashing of the light-client circuits (BASELINE configs[0]).
// One SHA-256 compression core shared by the two kernels below.
GL_HD void glp_sha256_compress_words(u32 (&h)[8], const u32 (&m)[16], const u32* __restrict__ k256) {
    u32 w[16];
    #pragma unroll
    for (int i = 0; i < 16; i++) w[i] = m[i];
    u32 s0 = h[0], s1 = h[1], s2 = h[2], s3 = h[3], s4 = h[4], s5 = h[5], s6 = h[6], s7 = h[7];
    #pragma unroll
    for (int i = 0; i < 64; i++) {
        u32 wi;
        if (i < 16) wi = w[i];
        else {
            const u32 w15 = w[(i - 15) & 15], w2 = w[(i - 2) & 15];
            const u32 g0 = glp_ror32(w15, 7) ^ glp_ror32(w15, 18) ^ (w15 >> 3);
            const u32 g1 = glp_ror32(w2, 17) ^ glp_ror32(w2, 19) ^ (w2 >> 10);
            wi = w[i & 15] + g0 + w[(i - 7) & 15] + g1;
            w[i & 15] = wi;
        }
        const u32 S1 = glp_ror32(s4, 6) ^ glp_ror32(s4, 11) ^ glp_ror32(s4, 25);
        const u32 ch = (s4 & s5) ^ (~s4 & s6);
        const u32 t1 = s7 + S1 + ch + k256[i] + wi;
        const u32 S0 = glp_ror32(s0, 2) ^ glp_ror32(s0, 13) ^ glp_ror32(s0, 22);
        const u32 mj = (s0 & s1) ^ (s0 & s2) ^ (s1 & s2);
        const u32 t2 = S0 + mj;
        s7 = s6; s6 = s5; s5 = s4; s4 = s3 + t1; s3 = s2; s2 = s1; s1 = s0; s0 = t1 + t2;
    }
    h[0] += s0; h[1] += s1; h[2] += s2; h[3] += s3; h[4] += s4; h[5] += s5; h[6] += s6; h[7] += s7;
}

// leaf hashes: out[i] = SHA256(0x00 || leaf_i), leaves of fixed leaf_len <= 118 bytes (two blocks max).
// digests are 8 big-endian words stored as u32.
template <int UNUSED = 0>
__global__ void __launch_bounds__(256) glp_tm_leaf_kernel(const uint8_t* __restrict__ leaves, u32 leaf_len, u64 n, u32* __restrict__ out,
                                                          const u32* __restrict__ k256) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint8_t* p = leaves + i * leaf_len;
    const u32 total = leaf_len + 1;                       // with the 0x00 prefix
    const u32 nblk = (total + 9 + 63) / 64;
    u32 h[8] = {0x6a09e667u, 0xbb67ae85u, 0x3c6ef372u, 0xa54ff53au, 0x510e527fu, 0x9b05688cu, 0x1f83d9abu, 0x5be0cd19u};
    for (u32 b = 0; b < nblk; b++) {
        u32 m[16];
        for (int wd = 0; wd < 16; wd++) {
            u32 v = 0;
            for (int k = 0; k < 4; k++) {
                const u32 pos = b * 64 + wd * 4 + k;      // byte position in the padded message
                u32 byte = 0;
                if (pos == 0) byte = 0x00;
                else if (pos < total) byte = p[pos - 1];
                else if (pos == total) byte = 0x80;
                else if (pos >= nblk * 64 - 8) { const u64 bits = (u64)total * 8; byte = (u32)(bits >> (8 * (nblk * 64 - 1 - pos))) & 0xff; }
                v = (v << 8) | byte;
            }
            m[wd] = v;
        }
        glp_sha256_compress_words(h, m, k256);
    }
    for (int k = 0; k < 8; k++) out[i * 8 + k] = h[k];
}

// the same for leaves of DIFFERENT lengths (protobuf-encoded validators, header fields): leaf i is
// data[offsets[i] .. offsets[i+1]), any length
template <int UNUSED = 0>
__global__ void __launch_bounds__(256) glp_tm_leaf_var_kernel(const uint8_t* __restrict__ data, const u64* __restrict__ offsets, u64 n,
                                                              u32* __restrict__ out, const u32* __restrict__ k256) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint8_t* p = data + offsets[i];
    const u64 total = offsets[i + 1] - offsets[i] + 1;     // with the 0x00 prefix
    const u64 nblk = (total + 9 + 63) / 64;
    u32 h[8] = {0x6a09e667u, 0xbb67ae85u, 0x3c6ef372u, 0xa54ff53au, 0x510e527fu, 0x9b05688cu, 0x1f83d9abu, 0x5be0cd19u};
    for (u64 b = 0; b < nblk; b++) {
        u32 m[16];
        for (int wd = 0; wd < 16; wd++) {
            u32 v = 0;
            for (int k = 0; k < 4; k++) {
                const u64 pos = b * 64 + wd * 4 + k;      // byte position in the padded message
                u32 byte = 0;
                if (pos == 0) byte = 0x00;
                else if (pos < total) byte = p[pos - 1];
                else if (pos == total) byte = 0x80;
                else if (pos >= nblk * 64 - 8) { const u64 bits = total * 8; byte = (u32)(bits >> (8 * (nblk * 64 - 1 - pos))) & 0xff; }
                v = (v << 8) | byte;
            }
            m[wd] = v;
        }
        glp_sha256_compress_words(h, m, k256);
    }
    for (int k = 0; k < 8; k++) out[i * 8 + k] = h[k];
}

// one level: out[i] = SHA256(0x01 || in[2i] || in[2i+1]) for i < n/2; an odd last node is promoted.
// The 65-byte message is two blocks.
template <int UNUSED = 0>
__global__ void __launch_bounds__(256) glp_tm_inner_kernel(const u32* __restrict__ in, u64 n, u32* __restrict__ out,
                                                           const u32* __restrict__ k256) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const u64 n_out = (n + 1) / 2;
    if (i >= n_out) return;
    if (2 * i + 1 >= n) {                                  // promoted
        for (int k = 0; k < 8; k++) out[i * 8 + k] = in[2 * i * 8 + k];
        return;
    }
    u32 d[16];
    for (int k = 0; k < 16; k++) d[k] = in[2 * i * 8 + k];   // left || right, big-endian words
    u32 h[8] = {0x6a09e667u, 0xbb67ae85u, 0x3c6ef372u, 0xa54ff53au, 0x510e527fu, 0x9b05688cu, 0x1f83d9abu, 0x5be0cd19u};
    u32 m[16];
    // block 0: 0x01 then the first 63 bytes of (left||right): every word is shifted by one byte
    m[0] = (0x01u << 24) | (d[0] >> 8);
    for (int k = 1; k < 16; k++) m[k] = (d[k - 1] << 24) | (d[k] >> 8);
    glp_sha256_compress_words(h, m, k256);
    // block 1: the last byte of right, 0x80, zeros, length = 65*8 = 520 bits
    for (int k = 0; k < 16; k++) m[k] = 0;
    m[0] = (d[15] << 24) | (0x80u << 16);
    m[15] = 520;
    glp_sha256_compress_words(h, m, k256);
    for (int k = 0; k < 8; k++) out[i * 8 + k] = h[k];
}
