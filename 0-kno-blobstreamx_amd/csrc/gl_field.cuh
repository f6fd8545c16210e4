// gl_field.cuh — Goldilocks field arithmetic, p = 2^64 - 2^32 + 1 (SURVEY.md §8a row a1;
// upstream name recalled as plonky2_field::GoldilocksField — reference file:line: NONE,
// /root/reference holds no source).
//
// All functions take and return CANONICAL values (< p) unless the name says otherwise.
// The 64x64->128 product is four 32x32 multiplies on CDNA4 (v_mad_u64_u32 / v_mul_hi_u32);
// the reduction uses 2^64 = 2^32 - 1 and 2^96 = -1 (mod p).
//
// 2 is an element of order 192 (2^96 = -1), and with the generator-7 roots of unity
// w_64 = 2^39: every twiddle of a radix-<=64 butterfly is a power of two, so in-register
// sub-transforms multiply by shifts (gl_mul_pow2<S>) instead of full products.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define GL_HD __host__ __device__ __forceinline__
#else
#define GL_HD inline __attribute__((always_inline))
#endif

typedef uint64_t u64;
typedef uint32_t u32;

#define GL_P 0xFFFFFFFF00000001ULL
#define GL_EPS 0xFFFFFFFFULL  // 2^64 mod p = 2^32 - 1

// compile-time unrolled loop: f(glp_ic<I>{}) for I in [I0, N)
template <int V> struct glp_ic { static constexpr int value = V; };
template <int I, int N, class F>
GL_HD void glp_static_for(F&& f) {
    if constexpr (I < N) {
        f(glp_ic<I>{});
        glp_static_for<I + 1, N>(f);
    }
}

// Cost model on gfx950 (profiles/r01_ubench_valu2.txt): plain 32-bit VOP2 ops issue in ~2.3
// cycles per wave64; everything 64-bit, VOP3, carry- or mask-producing (v_lshl_add_u64,
// v_mad_u64_u32, v_cmp_*, v_cndmask, v_add_co/v_addc) in ~4.2.  Hence "+ (cond ? eps : 0)"
// (one v_cndmask + one 64-bit add) rather than computing both candidates and selecting
// (one more 64-bit add and a second v_cndmask).
GL_HD u64 gl_add(u64 a, u64 b) {
    const u64 s = a + b;
    const bool over = (s < a) | (s >= GL_P);     // true sum >= p
    return s + (over ? GL_EPS : 0ULL);            // - p  (mod 2^64)
}
GL_HD u64 gl_sub(u64 a, u64 b) {
    u64 d = a - b;
    return (a < b) ? d - GL_EPS : d;  // d + p (mod 2^64)
}
GL_HD u64 gl_neg(u64 a) { return a ? GL_P - a : 0; }
GL_HD u64 gl_canon(u64 a) { return a >= GL_P ? a - GL_P : a; }

// r (+ 2^64 if carry) mod p, for a sum whose true value is < 2^64 + p
GL_HD u64 gl_fold_carry(u64 r, bool carry) { return r + ((carry | (r >= GL_P)) ? GL_EPS : 0ULL); }

// (hi*2^64 + lo) mod p, any hi, lo:  lo - hi_hi + hi_lo*(2^32 - 1)
GL_HD u64 gl_reduce128(u64 hi, u64 lo) {
    const u32 h0 = (u32)hi, h1 = (u32)(hi >> 32);
    u64 t0;
    const bool bor = __builtin_sub_overflow(lo, (u64)h1, &t0);
    t0 = bor ? t0 + GL_P : t0;                    // -2^64 = +p - 2^64 ... (mod 2^64): t0 - eps
    const u64 t1 = ((u64)h0 << 32) - h0;          // h0 * (2^32 - 1)  <= 2^64 - 2^33 + 1
    u64 r;
    const bool c = __builtin_add_overflow(t0, t1, &r);
    return gl_fold_carry(r, c);                   // carry: r + eps < p (r < t1); else one conditional subtract
}

GL_HD u64 gl_mul(u64 a, u64 b) {
    // 64x64 -> 128 from four 32x32 products chained through v_mad_u64_u32 addends
    const u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
    const u64 p0 = (u64)a0 * b0;
    const u64 p1 = (u64)a0 * b1 + (p0 >> 32);
    const u64 p2 = (u64)a1 * b0 + (u32)p1;
    const u64 p3 = (u64)a1 * b1 + (p1 >> 32) + (p2 >> 32);
    return gl_reduce128(p3, (p2 << 32) | (u32)p0);
}
GL_HD u64 gl_sqr(u64 a) { return gl_mul(a, a); }

// "nc" = not canonicalised: inputs may be ANY u64 representative, the result is a correct
// representative in [0, 2^64) that may be >= p.  Saves the (r >= p) compare of every product in
// long multiplication chains whose end result is canonicalised once (Poseidon S-boxes).
GL_HD u64 gl_mul_nc(u64 a, u64 b) {
    const u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
    const u64 p0 = (u64)a0 * b0;
    const u64 p1 = (u64)a0 * b1 + (p0 >> 32);
    const u64 p2 = (u64)a1 * b0 + (u32)p1;
    const u64 hi = (u64)a1 * b1 + (p1 >> 32) + (p2 >> 32), lo = (p2 << 32) | (u32)p0;
    const u32 h0 = (u32)hi, h1 = (u32)(hi >> 32);
    u64 t0;
    const bool bor = __builtin_sub_overflow(lo, (u64)h1, &t0);
    t0 = bor ? t0 + GL_P : t0;
    const u64 t1 = ((u64)h0 << 32) - h0;
    u64 r;
    const bool c = __builtin_add_overflow(t0, t1, &r);
    return r + (c ? GL_EPS : 0ULL);               // carry: r < t1 <= 2^64 - 2^33 + 1, no second overflow
}

// ---- multiplication by powers of two (the twiddles of every radix <= 64 butterfly) ------
// x * 2^T, 0 < T < 32:  (x << T) + (x >> (64-T)) * (2^32 - 1)
template <int T>
GL_HD u64 gl_shl_small(u64 x) {
    static_assert(T > 0 && T < 32, "");
    const u64 lo = x << T;
    const u32 w2 = (u32)(x >> (64 - T));
    const u64 t1 = ((u64)w2 << 32) - w2;
    u64 r;
    const bool c = __builtin_add_overflow(lo, t1, &r);
    return gl_fold_carry(r, c);
}
// x * 2^-K, 0 < K <= 32, branch-free and already canonical (Montgomery-style exact division):
// m = -x mod 2^K makes x + m*p divisible by 2^K (p = 1 mod 2^32), and
// (x + m*p) / 2^K = (x + m) / 2^K + (m << (32-K)) * (2^32 - 1)  <  p.
template <int K>
GL_HD u64 gl_shr_small(u64 x) {
    static_assert(K > 0 && K <= 32, "");
    const u32 m = (K == 32) ? (0u - (u32)x) : ((0u - (u32)x) & ((1u << (K & 31)) - 1u));
    const u64 a = (x + m) >> (K & 63);
    const u32 mm = (K == 32) ? m : (m << ((32 - K) & 31));
    return a + (((u64)mm << 32) - mm);
}
// x * 2^32 = -x1 + (x0 + x1) * 2^32  ... as reduce128(hi = x >> 32, lo = x << 32)
GL_HD u64 gl_shl32(u64 x) { return gl_reduce128(x >> 32, x << 32); }

// |x * 2^S| up to sign, S in [0,192): returns v with  x * 2^S = (gl_pow2_neg(S) ? -v : v).
// Every case is one or two of the primitives above; exponents 32 < e < 96 go through the
// inverse shifts using 2^96 = -1  (2^e = -2^-(96-e)).
constexpr bool gl_pow2_neg(int S) {
    const int e = S % 96;
    const bool flip = (e > 32);          // cases routed through -2^-(96-e)
    return ((S / 96) & 1) != (flip ? 1 : 0);
}
template <int S>
GL_HD u64 gl_mul_pow2_mag(u64 x) {
    static_assert(S >= 0 && S < 192, "shift out of range");
    constexpr int e = S % 96;
    if constexpr (e == 0) return x;
    else if constexpr (e < 32) return gl_shl_small<e>(x);
    else if constexpr (e == 32) return gl_shl32(x);
    else if constexpr (e < 64) return gl_shr_small<32>(gl_shr_small<64 - e>(x));   // 2^-(96-e), 96-e in (32,64)
    else return gl_shr_small<96 - e>(x);                                           // 96-e in (0,32]
}
// x * 2^S mod p for a compile-time S in [0, 192).
template <int S>
GL_HD u64 gl_mul_pow2(u64 x) {
    const u64 v = gl_mul_pow2_mag<S>(x);
    if constexpr (gl_pow2_neg(S)) return gl_neg(v);
    else return v;
}

// runtime exponent version (host-side table building, slow paths)
GL_HD u64 gl_pow(u64 a, u64 e) {
    u64 r = 1;
    while (e) {
        if (e & 1) r = gl_mul(r, a);
        a = gl_mul(a, a);
        e >>= 1;
    }
    return r;
}
GL_HD u64 gl_inv(u64 a) { return gl_pow(a, GL_P - 2); }
// primitive 2^k-th root of unity (k <= 32): 7^((p-1)/2^k)
GL_HD u64 gl_root_of_unity(unsigned k) { return gl_pow(7, (GL_P - 1) >> k); }

// ---- quadratic extension F_p[X]/(X^2 - 7) (row a8; W = 7 recalled, unpinned) ----
struct gl_ext2 { u64 a, b; };
GL_HD gl_ext2 gl_ext_add(gl_ext2 x, gl_ext2 y) { return {gl_add(x.a, y.a), gl_add(x.b, y.b)}; }
GL_HD gl_ext2 gl_ext_sub(gl_ext2 x, gl_ext2 y) { return {gl_sub(x.a, y.a), gl_sub(x.b, y.b)}; }
GL_HD gl_ext2 gl_ext_mul(gl_ext2 x, gl_ext2 y) {
    u64 bb = gl_mul(x.b, y.b);
    u64 w = gl_add(gl_add(gl_add(bb, bb), gl_add(bb, bb)), gl_add(gl_add(bb, bb), bb));  // 7*bb
    return {gl_add(gl_mul(x.a, y.a), w), gl_add(gl_mul(x.a, y.b), gl_mul(x.b, y.a))};
}
GL_HD gl_ext2 gl_ext_scale(gl_ext2 x, u64 s) { return {gl_mul(x.a, s), gl_mul(x.b, s)}; }
