// ntt_kernels.cuh — Goldilocks radix-2^k NTT pass kernels for gfx950 (SURVEY.md §8a row a2;
// upstream names recalled as plonky2_field::fft::{fft, ifft, coset_fft, lde} — reference
// file:line: NONE, /root/reference holds no source).
//
// One transform of size n = R_1 * R_2 * ... * R_P is P launches ("passes") of ONE kernel
// template.  Pass i computes, for every other index, a size-R_i DFT along the axis whose
// stride is m_i = n / (R_1...R_i), then multiplies by the inter-pass twiddle
// w_{R_i*m_i}^{j' * k_i} (decimation in frequency: big strides first, twiddle after).
//
// A workgroup owns a tile of R rows x C columns (C = 2^log_c adjacent "other" indices), 16
// elements per thread, NT = R*C/16 threads:
//   STRIP      rows are m apart in memory, the C columns are contiguous (C*8-byte segments);
//              results go back to the same tile positions (or the same positions of dst).
//   FINAL_T    last pass of a natural-order transform: the C "columns" are C whole
//              contiguous rows of src; X[k] is written to its natural position, which
//              is a C-wide contiguous segment per k (the digit-reversal transpose is paid
//              here, in C*8-byte segments, and nowhere else).
//   FINAL_ROWS last/only pass when outputs stay in their row (single-pass transforms,
//              or bit-reversed output): results are restaged through LDS and written as
//              whole contiguous rows.
// Inside the tile the size-R DFT is ceil(log2 R / 4) register steps of radix <= 16, with an
// LDS exchange ([R][C+1] u64, +1 pad = conflict-free for both access patterns) between
// steps.  Radix-16 butterflies use only shift twiddles (w_16 = 2^156); one table twiddle
// per element per step boundary.
//
// This file is plain HIP C++ with no AMD builtins so that tests/emu can run the very same
// kernel bodies on the CPU under ASan (test infrastructure; the product never does).
#pragma once
#include "gl_field.cuh"
#include "ntt_plan.h"

constexpr int glp_bitrev_c(int v, int bits) {
    int r = 0;
    for (int i = 0; i < bits; i++) r |= ((v >> i) & 1) << (bits - 1 - i);
    return r;
}
// exponent e with w_m^i = 2^e (m = 2^k <= 64): w_64 = 2^GLP_W64_LOG2 (39 under the default generator-7 roots; gl_field.cuh).
constexpr int glp_tw_exp(int m, int i, bool inv) {
    int e = (GLP_W64_LOG2 * (64 / m) * i) % 192;
    return inv ? (192 - e) % 192 : e;
}

// In-register DIF DFT of size 2^Q on x[OFF .. OFF+2^Q); X[k] lands in x[OFF + bitrev_Q(k)].
template <int Q, bool INV, int OFF, int NX>
GL_HD void glp_dft_inreg(u64 (&x)[NX]) {
    glp_static_for<0, Q>([&](auto s_) {
        constexpr int s = decltype(s_)::value;
        constexpr int half = 1 << (Q - 1 - s);
        constexpr int m = 2 * half;
        glp_static_for<0, (1 << Q) / 2>([&](auto t_) {
            constexpr int t = decltype(t_)::value;
            constexpr int blk = t / half, i = t % half;
            constexpr int i0 = OFF + blk * m + i, i1 = i0 + half;
            constexpr int e = glp_tw_exp(m, i, INV);
            u64 u = x[i0], v = x[i1];
            x[i0] = gl_add(u, v);
            if constexpr (gl_pow2_neg(e)) x[i1] = gl_mul_pow2_mag<e>(gl_sub(v, u));
            else x[i1] = gl_mul_pow2_mag<e>(gl_sub(u, v));
        });
    });
}


struct GlpNttPassArgs {
    const u64* src;
    u64* dst;
    u64 src_poly_stride;   // elements between consecutive polynomials in src
    u64 dst_poly_stride;
    const u64* tw_tile;    // w_R^e, e < R          (inverse powers when INV)
    const u64* tw_lo;      // w_N^e, e < min(N,4096), N = R*m   (STRIP only, m > 1)
    const u64* tw_hi;      // w_N^(4096 e), e < N/4096          (only when N > 4096)
    u64 scale;             // multiply every output by this (1 = none); inverse: n^-1
    u32 log_n;             // transform size
    u32 log_m;             // STRIP: log2 of the axis stride m
    u32 log_c;             // log2 of columns per tile
    u32 batch;             // number of polynomials
    u32 rev;               // bit-reversed output order (STRIP: row placement; FINAL_ROWS: index)
    u32 nprev;             // FINAL_T: number of earlier passes, and their log2 radices
    u32 log_rprev[3];
    u32 xcd_group_log;     // STRIP with C*8 < 128 B: log2 of strips sharing one 128-B line (0 = no remap)
    const u64* tw_full;    // STRIP, optional: per-element inter-pass twiddles w_N^{j'k} at [k*m + j'] (batched sizes)
    u32 poly_minor;        // STRIP: consecutive workgroups walk the polynomials of one tile position first
    // Low-degree extension by cosets (bit-reversed output): `batch` counts VIRTUAL polynomials
    // v = (p << coset_log) | k — source polynomial p evaluated on the coset shift * w_N^k * <w_n> by a
    // size-n transform of c_j * s_k^j.  Bit reversal of the N-point index puts that coset in the
    // contiguous block bitrev(k) of destination row p, so nothing is zero-padded and no pass touches
    // more than n points per (p, k).
    u32 coset_log;         // log2 of the blow-up (0 = plain transform)
    u32 src_coset;         // src already has the destination's coset-blocked layout (every pass but the first)
    const u64* in_row;     // first pass: s_k^(row * m) at [(k << LOG_R) | row], multiplied into the loaded element
    const u64* in_col;     // first pass when STRIP: s_k^c at [(k << log_m) | c], folded into the inter-pass twiddle
};

// the arguments ask for none of the optional features: the PLAIN instantiation serves them (host launcher and the emulation decide alike)
static inline bool glp_ntt_args_plain(const GlpNttPassArgs& a) {
    return !a.coset_log && !a.src_coset && !a.in_row && !a.in_col && !a.tw_full && !a.poly_minor && !a.rev;
}

// LOG_E = log2 of the elements held per work-item (4: radix <= 16 steps, 5: radix <= 32 steps, 6: radix <= 64 steps on 2^11 / 2^12 tiles;
// 2 and 3: small work-items for latency-bound single transforms of 2^20 — more waves per CU, more register steps)
template <int LOG_R, int LOG_E>
struct GlpSteps {
    static constexpr int S = (LOG_R + LOG_E - 1) / LOG_E;                       // register steps
    static constexpr int q(int t) { return t == 0 ? LOG_R - LOG_E * (S - 1) : LOG_E; }  // small radix first
    static constexpr int log_sigma(int t) { return LOG_E * (S - 1 - t); }           // stride of digit t
    static constexpr int low_bits(int t) { return t == 0 ? 0 : q(0) + LOG_E * (t - 1); }
};

// natural index k of the value sitting at tile row `pos` after all steps (digit reversal)
template <int LOG_R, int LOG_E>
GL_HD u32 glp_digit_reverse(u32 pos) {
    using ST = GlpSteps<LOG_R, LOG_E>;
    u32 k = 0;
    glp_static_for<0, ST::S>([&](auto t_) {
        constexpr int t = decltype(t_)::value;
        u32 digit = (pos >> ST::log_sigma(t)) & ((1u << ST::q(t)) - 1u);
        k |= digit << ST::low_bits(t);
    });
    return k;
}

GL_HD u32 glp_bitrev32(u32 v, u32 bits) {
    if (bits == 0) return 0;
#if defined(__HIP_DEVICE_COMPILE__)
    return __brev(v) >> (32 - bits);
#else
    u32 r = 0;
    for (u32 i = 0; i < bits; i++) r |= ((v >> i) & 1u) << (bits - 1 - i);
    return r;
#endif
}

#if defined(GLP_EMU)
#define GLP_DYN_LDS(name) u64* name = reinterpret_cast<u64*>(glp_emu_dyn_lds())
#else
#define GLP_DYN_LDS(name) extern __shared__ __attribute__((aligned(16))) u64 name[]
#endif

// SPLIT exchange (radix-32 work-items, every mode but FINAL_ROWS): the tile lives in LDS as 32-bit HALVES — the low words of all elements are
// exchanged first, then the high words through the same [R][C+1] u32 array.  The LDS footprint per workgroup halves, so a CU holds four
// workgroups of 32-element work-items instead of two: the radix-32 steps (one tile-twiddle layer and one exchange per 2^10 tile instead of
// two) run at the occupancy the radix-16 kernel has (round 2 measured them at HALF of it: 0.97 vs 0.71 ms).
template <int LOG_E, int MODE> struct GlpSplit { static constexpr bool value = (LOG_E >= 5 && MODE != GLP_FINAL_ROWS); };

// PLAIN: a natural-order transform without the optional features — no coset blocks (LDE), no input scale, no per-element inter-pass table, no
// bit-reversed placement, no polynomial-minor tile order.  Those are runtime switches of the general kernel; compiled out, their address
// arithmetic and live ranges leave the hot kernel (the radix-32 work-items have no registers to spare for them).  The host launches the PLAIN
// instantiation whenever the arguments allow it (ntt_inst.hip); results are identical by construction.
// Launch bounds.  Radix-16 work-items: up to 1024 threads, ~122 VGPRs (four waves per SIMD, what two 72 KiB tiles per CU allow anyway).
// Radix-32 work-items hold 64 VGPRs of data: the strip pass runs 512-thread workgroups at four waves per SIMD (128 VGPRs, a handful of
// spills); the FINAL_T pass runs 256-thread workgroups (C = 8) at THREE waves per SIMD — 142 VGPRs, no spills, three 37 KiB tiles per CU —
// which measured 0.50 instead of 0.56 ms at 128 x 2^20 (same-box A/B, profiles/r03_ntt_e5_probe.jsonl).  The planner keeps FINAL_T radix-32
// tiles at <= 256 threads (ntt_plan.h).
template <int LOG_E, int MODE, int CT_LOG_C = -1> struct GlpBounds {
    // (a FINAL_T tile with a compile-time width needs 106 VGPRs: it may run 512 threads at four waves per SIMD)
    // radix-64 work-items (2^11 = 32 * 64 and 2^12 = 64 * 64 tiles: two register steps, ONE exchange): 128 VGPRs of data, two waves per SIMD
    static constexpr int threads = LOG_E == 6 ? 512 : ((LOG_E == 5 && MODE == GLP_FINAL_T) ? 256 : 1024);
    static constexpr int waves = LOG_E == 6 ? 2 : (LOG_E == 5 ? (MODE == GLP_FINAL_T ? 3 : 4) : 1);
};
// CT_LOG_C >= 0: the tile width is a compile-time constant (the host launches such an instantiation only when a.log_c equals it): every LDS
// address becomes base + immediate offset — the runtime-width kernel keeps one address VGPR per element and side of the exchange (64 of
// them on radix-32 work-items: that, not the data, is what spilled).
template <int LOG_R, int MODE, bool INV, int LOG_E = 4, bool PLAIN = false, int CT_LOG_C = -1>
__global__ void __launch_bounds__((GlpBounds<LOG_E, MODE, CT_LOG_C>::threads), (GlpBounds<LOG_E, MODE, CT_LOG_C>::waves)) glp_ntt_pass_kernel(GlpNttPassArgs a) {
    static_assert(!PLAIN || MODE != GLP_FINAL_ROWS, "FINAL_ROWS has no plain form");
    using ST = GlpSteps<LOG_R, LOG_E>;
    const u32 a_coset_log = PLAIN ? 0u : a.coset_log;
    const u32 a_src_coset = PLAIN ? 0u : a.src_coset;
    const u64* const a_in_row = PLAIN ? nullptr : a.in_row;
    const u64* const a_in_col = PLAIN ? nullptr : a.in_col;
    const u64* const a_tw_full = PLAIN ? nullptr : a.tw_full;
    const u32 a_poly_minor = PLAIN ? 0u : a.poly_minor;
    const u32 a_rev = PLAIN ? 0u : a.rev;
    constexpr u32 R = 1u << LOG_R;
    constexpr u32 E = 1u << LOG_E;
    constexpr bool SPLIT = GlpSplit<LOG_E, MODE>::value;
    GLP_DYN_LDS(lds);
    u32* const lds32 = reinterpret_cast<u32*>(lds);

    const u32 log_c = CT_LOG_C >= 0 ? (u32)CT_LOG_C : a.log_c;
    const u32 C = 1u << log_c;
    const u32 NT = (R << log_c) >> LOG_E;    // == blockDim.x (checked on the host)
    const u32 tid = threadIdx.x;
    const u32 ldA = C + 1;                   // layout A: [row][C+1]
    u64 tile = blockIdx.x;
    if constexpr (MODE == GLP_STRIP || MODE == GLP_FINAL_T) {
        // (FINAL_T, round 3: its tiles WRITE C * 8-byte segments, so tiles t and t + 1 write the two halves of the same 128-B lines; on the same
        // XCD the halves can merge in its L2 before they leave for HBM)
        // Strips narrower than a 128-B line share lines with their neighbours.  Workgroups b and
        // b+8 are observed to land on the same XCD (private L2), so give the 2^g line-sharing
        // strips the ids b, b+8, ...: the second toucher then hits in L2 instead of HBM.
        // Bijective on [0, grid) because the host only enables it when grid % (8 << g) == 0;
        // placement affects speed only, never results.
        const u32 g = a.xcd_group_log;
        if (g) {
            const u64 per = 8ull << g;
            const u64 base = tile & ~(per - 1);
            const u32 rem = (u32)(tile & (per - 1));
            tile = base + ((u64)(rem & 7u) << g) + (rem >> 3);
        }
    }

    // element offset of (virtual) polynomial v in a buffer of row stride `stride`
    auto poly_off = [&](u64 v, u64 stride, bool blocked) -> u64 {
        if (!a_coset_log) return v * stride;
        const u32 kc = (u32)v & ((1u << a_coset_log) - 1u);
        return (v >> a_coset_log) * stride + (blocked ? ((u64)glp_bitrev32(kc, a_coset_log) << a.log_n) : 0ull);
    };
    const u32 coset_mask = (1u << a_coset_log) - 1u;

    // ---- tile geometry -------------------------------------------------------------
    // STRIP: tile -> (poly, hi, lo0); element (row, col) at  hi*R*m + row*m + lo0 + col
    u64 sbase = 0, dbase = 0;
    u32 lo0 = 0, kcos = 0;                   // STRIP: coset of this tile's (virtual) polynomial
    // FINAL_*: tile -> first of C global rows
    u64 row_first = 0;
    const u32 log_rows = a.log_n - LOG_R;    // rows per polynomial (FINAL modes)
    if constexpr (MODE == GLP_STRIP) {
        const u32 log_tpp = a.log_n - LOG_R - log_c;              // tiles per polynomial
        u64 poly;
        u32 t;
        if (a_poly_minor) {
            // (tile position, polynomial) with the polynomial varying fastest, keeping the 2^g
            // line-sharing strips adjacent: the per-element twiddle tile is then reused by
            // `batch` consecutive workgroups out of L2
            const u32 g = a.xcd_group_log;
            const u64 rest = tile >> g;
            poly = rest % a.batch;
            t = (u32)(((rest / a.batch) << g) | (tile & ((1ull << g) - 1)));
        } else {
            poly = tile >> log_tpp;
            t = (u32)(tile & ((1ull << log_tpp) - 1));
        }
        const u32 log_mc = a.log_m - log_c;
        const u32 hi = t >> log_mc;
        lo0 = (t & ((1u << log_mc) - 1u)) << log_c;
        const u64 off = ((u64)hi << (LOG_R + a.log_m)) + lo0;
        sbase = poly_off(poly, a.src_poly_stride, a_src_coset != 0) + off;
        dbase = poly_off(poly, a.dst_poly_stride, true) + off;
        kcos = (u32)poly & coset_mask;
    } else {
        row_first = tile << log_c;
    }
    const u64 total_rows = (u64)a.batch << log_rows;

    u64 x[E];

    // geometry of unit g of this work-item in register step T: (column, first tile row, index below the step's digit)
    auto unit_geom = [&](auto T_, u32 g, u32& col, u32& row0, u32& o_lo) {
        constexpr int T = decltype(T_)::value;
        constexpr int q = ST::q(T);
        constexpr int lsg = ST::log_sigma(T);
        const u32 u = g * NT + tid;
        u32 o;
        if (T == 0 && MODE != GLP_STRIP) {      // lanes along the contiguous NTT axis
            o = u & ((R >> q) - 1u);
            col = u >> (LOG_R - q);
        } else {                                // lanes along the C contiguous columns
            col = u & (C - 1u);
            o = u >> log_c;
        }
        o_lo = o & ((1u << lsg) - 1u);
        row0 = ((o >> lsg) << (q + lsg)) | o_lo;
    };

    glp_static_for<0, ST::S>([&](auto t_) {
        constexpr int t = decltype(t_)::value;
        constexpr int q = ST::q(t);
        constexpr u32 r = 1u << q;
        constexpr int lsg = ST::log_sigma(t);
        constexpr u32 G = E / r;             // units per thread in this step
        constexpr bool first = (t == 0), last = (t == ST::S - 1);

        // ---- decode + load ---------------------------------------------------------
        u32 colv[G], row0v[G], olov[G];
        glp_static_for<0, (int)G>([&](auto g_) {
            constexpr int g = decltype(g_)::value;
            u32 col, row0, o_lo;
            unit_geom(t_, (u32)g, col, row0, o_lo);
            colv[g] = col; row0v[g] = row0; olov[g] = o_lo;
            if constexpr (first) {
                if constexpr (MODE == GLP_STRIP) {
                    // one per-lane base pointer, then wave-uniform multiples of the row stride (the per-element form
                    // `(row0 + (d << lsg)) << log_m` cost four VALU instructions per load: or, move, 64-bit shift, 64-bit add)
                    const u64* const ps = a.src + (sbase + col + ((u64)row0 << a.log_m));
                    const u64 st = 1ull << (lsg + a.log_m);
                    glp_static_for<0, (int)r>([&](auto d_) {
                        constexpr int d = decltype(d_)::value;
                        x[g * r + d] = ps[(u64)d * st];
                    });
                    if (a_in_row) {
                        // coset LDE: the input scale.  All r table words are loaded before the first product: written per element
                        // (load, load, wait, multiply) the compiler serialised every pair behind an s_waitcnt vmcnt(0)
                        u64 sc[r];
                        glp_static_for<0, (int)r>([&](auto d_) {
                            constexpr int d = decltype(d_)::value;
                            sc[d] = a_in_row[(kcos << LOG_R) + row0 + ((u32)d << lsg)];
                        });
                        glp_static_for<0, (int)r>([&](auto d_) {
                            constexpr int d = decltype(d_)::value;
                            x[g * r + d] = gl_mul(x[g * r + d], sc[d]);
                        });
                    }
                } else {
                    u64 grow = row_first + col;      // global row handled by this column
                    bool active = grow < total_rows;
                    u64 poly = grow >> log_rows;
                    u32 rr = (u32)(grow & ((1ull << log_rows) - 1));
                    if constexpr (MODE == GLP_FINAL_T) {
                        // rr is kappa = k_1 + R_1 k_2 + ...; source row = k_1*(Q/R_1) + k_2*(Q/R_1R_2) ...
                        u32 rem = rr, sh = log_rows, rho = 0;
                        for (u32 i = 0; i < a.nprev; i++) {
                            u32 lr = a.log_rprev[i];
                            sh -= lr;
                            rho |= (rem & ((1u << lr) - 1u)) << sh;
                            rem >>= lr;
                        }
                        rr = rho;
                    }
                    const u64 p = poly_off(poly, a.src_poly_stride, a_src_coset != 0) + ((u64)rr << LOG_R) + row0;
                    const u32 kc = (u32)poly & coset_mask;
                    glp_static_for<0, (int)r>([&](auto d_) {
                        constexpr int d = decltype(d_)::value;
                        x[g * r + d] = active ? a.src[p + ((u32)d << lsg)] : 0ull;
                    });
                    if (a_in_row) {                                                  // single-pass sizes: m = 1
                        u64 sc[r];
                        glp_static_for<0, (int)r>([&](auto d_) {
                            constexpr int d = decltype(d_)::value;
                            sc[d] = a_in_row[(kc << LOG_R) + row0 + ((u32)d << lsg)];
                        });
                        glp_static_for<0, (int)r>([&](auto d_) {
                            constexpr int d = decltype(d_)::value;
                            x[g * r + d] = gl_mul(x[g * r + d], sc[d]);
                        });
                    }
                }
            } else if constexpr (!SPLIT) {
                const u32 base = row0 * ldA + col;
                glp_static_for<0, (int)r>([&](auto d_) {
                    constexpr int d = decltype(d_)::value;
                    x[g * r + d] = lds[base + ((u32)d << lsg) * ldA];
                });
            }                                   // SPLIT: x was filled by the previous step's exchange
        });

        if constexpr (last && MODE == GLP_FINAL_ROWS && !first) __syncthreads();  // A fully read before B is written

        // ---- butterflies -----------------------------------------------------------
        glp_static_for<0, (int)G>([&](auto g_) {
            constexpr int g = decltype(g_)::value;
            glp_dft_inreg<q, INV, g * (int)r>(x);
        });

        // ---- twiddle + store ---------------------------------------------------------
        glp_static_for<0, (int)G>([&](auto g_) {
            constexpr int g = decltype(g_)::value;
            const u32 col = colv[g], row0 = row0v[g];
            if constexpr (!last && SPLIT) {
                // the twiddled values stay in registers (digit order) until the two half exchanges below
                const u32 o_lo = olov[g];
                u64 v[r];
                glp_static_for<0, (int)r>([&](auto d_) {
                    constexpr int d = decltype(d_)::value;
                    v[d] = x[g * r + glp_bitrev_c(d, q)];
                    if constexpr (d != 0) v[d] = gl_mul(v[d], a.tw_tile[(o_lo * (u32)d) << ST::low_bits(t)]);
                });
                glp_static_for<0, (int)r>([&](auto d_) { x[g * r + decltype(d_)::value] = v[decltype(d_)::value]; });
            } else if constexpr (!last) {
                // X[k_t = d] *= w_{r*sigma}^{o_lo * d}  ==  tw_tile[(o_lo*d) << low_bits(t)]
                const u32 o_lo = olov[g];
                const u32 base = row0 * ldA + col;
                glp_static_for<0, (int)r>([&](auto d_) {
                    constexpr int d = decltype(d_)::value;
                    u64 v = x[g * r + glp_bitrev_c(d, q)];
                    if constexpr (d != 0) v = gl_mul(v, a.tw_tile[(o_lo * (u32)d) << ST::low_bits(t)]);
                    lds[base + ((u32)d << lsg) * ldA] = v;
                });
            } else {
                // sigma == 1 in the last step: the 2^q outputs of this work-item are the natural indices
                // k = k0 + d * (R/r), so every address below is a base plus a wave-uniform multiple of d.
                const u32 k0 = glp_digit_reverse<LOG_R, LOG_E>(row0);
                constexpr int KS = LOG_R - q;                               // log2 of the k stride
                if constexpr (MODE == GLP_STRIP) {
                    // inter-pass twiddle X[k] *= w_N^{j' k}, j' = lo0 + col: a geometric progression in d
                    // (two table look-ups + a running product instead of 16 pairs of gathered loads)
                    u64 tw = 1, ratio = 1;
                    const u64* const twf = a_tw_full;
                    if (!twf) {
                        const u32 log_N = LOG_R + a.log_m;
                        const u64 jq = (u64)(lo0 + col);
                        const u64 e0 = jq * k0;                                  // < N <= 2^32
                        const u64 e1 = jq << KS;                                 // j' * R/r
                        if (log_N <= 12) {
                            tw = a.tw_lo[e0];
                            ratio = a.tw_lo[e1];
                        } else {
                            tw = gl_mul(a.tw_lo[e0 & 4095u], a.tw_hi[e0 >> 12]);
                            ratio = gl_mul(a.tw_lo[e1 & 4095u], a.tw_hi[e1 >> 12]);
                        }
                        // the column's share s_k^(lo0 + col) of the input scale commutes with the row transform
                        if (a_in_col) tw = gl_mul(tw, a_in_col[((u64)kcos << a.log_m) + lo0 + col]);
                    }
                    // natural order: row k; bit-reversed: row bitrev(k) = bitrev(k0) + bitrev_q(d)
                    const u32 orow0 = a_rev ? glp_bitrev32(k0, LOG_R) : k0;
                    const u64 p0 = dbase + ((u64)orow0 << a.log_m) + col;
                    const u32 sh = a_rev ? a.log_m : a.log_m + KS;
                    glp_static_for<0, (int)r>([&](auto d_) {
                        constexpr int d = decltype(d_)::value;
                        u64 v = x[g * r + glp_bitrev_c(d, q)];
                        if (twf) {
                            v = gl_mul(v, twf[((u64)(k0 + ((u32)d << KS)) << a.log_m) + lo0 + col]);
                        } else {
                            v = gl_mul(v, tw);
                            if constexpr (d + 1 < (int)r) tw = gl_mul(tw, ratio);
                        }
                        const u64 dm = a_rev ? (u64)glp_bitrev_c(d, q) : (u64)d;
                        a.dst[p0 + (dm << sh)] = v;
                    });
                } else if constexpr (MODE == GLP_FINAL_T) {
                    const u64 grow = row_first + col;                        // = poly*Q + kappa
                    if (grow < total_rows) {
                        const u64 poly = grow >> log_rows;
                        const u64 kappa = grow & ((1ull << log_rows) - 1);
                        const u64 p0 = poly_off(poly, a.dst_poly_stride, true) + ((u64)k0 << log_rows) + kappa;
                        const u32 sh = KS + log_rows;
                        glp_static_for<0, (int)r>([&](auto d_) {
                            constexpr int d = decltype(d_)::value;
                            u64 v = x[g * r + glp_bitrev_c(d, q)];
                            if (a.scale != 1) v = gl_mul(v, a.scale);
                            a.dst[p0 + ((u64)d << sh)] = v;
                        });
                    }
                } else {
                    const u32 kb = col * (R + 1u) + (a_rev ? glp_bitrev32(k0, LOG_R) : k0);   // layout B: [col][R+1]
                    const u32 sh = a_rev ? 0u : (u32)KS;
                    glp_static_for<0, (int)r>([&](auto d_) {
                        constexpr int d = decltype(d_)::value;
                        u64 v = x[g * r + glp_bitrev_c(d, q)];
                        if (a.scale != 1) v = gl_mul(v, a.scale);
                        const u32 dm = a_rev ? (u32)glp_bitrev_c(d, q) : (u32)d;
                        lds[kb + (dm << sh)] = v;
                    });
                }
            }
        });
        if constexpr (!last && SPLIT) {
            // exchange into the NEXT step's geometry, 32 bits at a time: write the low words at this step's positions, read them at the next
            // step's; then the same for the high words.  A work-item reads (next step) and later writes (end of next step) the same positions,
            // so only the three barriers inside the exchange are needed.
            constexpr int q2 = ST::q(t + 1);
            constexpr u32 r2 = 1u << q2;
            constexpr int lsg2 = ST::log_sigma(t + 1);
            constexpr u32 G2 = E / r2;
            u32 nb[G2];
            glp_static_for<0, (int)G2>([&](auto g_) {
                constexpr int g = decltype(g_)::value;
                u32 col, row0, o_lo;
                unit_geom(glp_ic<t + 1>{}, (u32)g, col, row0, o_lo);
                nb[g] = row0 * ldA + col;
            });
            u32 lo[E];
            glp_static_for<0, (int)G>([&](auto g_) {
                constexpr int g = decltype(g_)::value;
                const u32 base = row0v[g] * ldA + colv[g];
                glp_static_for<0, (int)r>([&](auto d_) {
                    constexpr int d = decltype(d_)::value;
                    lds32[base + ((u32)d << lsg) * ldA] = (u32)x[g * r + d];
                });
            });
            __syncthreads();
            glp_static_for<0, (int)G2>([&](auto g_) {
                constexpr int g = decltype(g_)::value;
                glp_static_for<0, (int)r2>([&](auto d_) {
                    constexpr int d = decltype(d_)::value;
                    lo[g * r2 + d] = lds32[nb[g] + ((u32)d << lsg2) * ldA];
                });
            });
            __syncthreads();
            glp_static_for<0, (int)G>([&](auto g_) {
                constexpr int g = decltype(g_)::value;
                const u32 base = row0v[g] * ldA + colv[g];
                glp_static_for<0, (int)r>([&](auto d_) {
                    constexpr int d = decltype(d_)::value;
                    lds32[base + ((u32)d << lsg) * ldA] = (u32)(x[g * r + d] >> 32);
                });
            });
            __syncthreads();
            glp_static_for<0, (int)G2>([&](auto g_) {
                constexpr int g = decltype(g_)::value;
                glp_static_for<0, (int)r2>([&](auto d_) {
                    constexpr int d = decltype(d_)::value;
                    x[g * r2 + d] = gl_make64(lo[g * r2 + d], lds32[nb[g] + ((u32)d << lsg2) * ldA]);
                });
            });
        } else if constexpr (!last || MODE == GLP_FINAL_ROWS) __syncthreads();
    });

    if constexpr (MODE == GLP_FINAL_ROWS) {
        // coalesced write of C whole rows from layout B
        #pragma unroll 4
        for (u32 i = tid; i < (R << log_c); i += NT) {
            const u32 col = i >> LOG_R, kk = i & (R - 1u);
            const u64 grow = row_first + col;
            if (grow < total_rows) {
                const u64 poly = grow >> log_rows;
                const u64 rr = grow & ((1ull << log_rows) - 1);
                a.dst[poly_off(poly, a.dst_poly_stride, true) + (rr << LOG_R) + kk] = lds[col * (R + 1u) + kk];
            }
        }
    }
}

// Transforms smaller than one tile (n < 64): one work-item per polynomial, O(n^2) with the
// w_n^e table.  Sizes this small only occur at the tail of FRI and in tests.
template <int UNUSED = 0>
__global__ void __launch_bounds__(256) glp_ntt_small_kernel(const u64* src, u64* dst, u64 src_poly_stride,
                                                            u64 dst_poly_stride, u32 log_n, u32 batch,
                                                            const u64* tw, u64 scale, u32 rev) {
    const u32 b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    const u32 n = 1u << log_n;
    u64 in[32];
    for (u32 j = 0; j < n; j++) in[j] = src[(u64)b * src_poly_stride + j];
    for (u32 k = 0; k < n; k++) {
        u64 acc = 0;
        for (u32 j = 0; j < n; j++) acc = gl_add(acc, gl_mul(in[j], tw[(j * k) & (n - 1u)]));
        if (scale != 1) acc = gl_mul(acc, scale);
        const u32 ko = rev ? glp_bitrev32(k, log_n) : k;
        dst[(u64)b * dst_poly_stride + ko] = acc;
    }
}

// full[k*m + j'] = w_N^{j' * k},  N = R * m, k < R, j' < m  (built once per (N, m, direction))
template <int UNUSED = 0>
__global__ void __launch_bounds__(256) glp_build_full_tw_kernel(u64* __restrict__ full, u32 log_N, u32 log_m, const u64* __restrict__ tw_lo,
                                                                const u64* __restrict__ tw_hi) {
    const u64 N = 1ull << log_N;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (u64)gridDim.x * blockDim.x) {
        const u64 k = i >> log_m, j = i & ((1ull << log_m) - 1);
        const u64 e = k * j;                                 // < N
        u64 w = tw_lo[e & 4095u];
        if (log_N > 12) w = gl_mul(w, tw_hi[e >> 12]);
        full[i] = w;
    }
}
