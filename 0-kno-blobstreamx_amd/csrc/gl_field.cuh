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

GL_HD u64 gl_add(u64 a, u64 b) {
    u64 s = a + b;
    u64 t = s + GL_EPS;  // s - p (mod 2^64)
    return (s < a || s >= GL_P) ? t : s;
}
GL_HD u64 gl_sub(u64 a, u64 b) {
    u64 d = a - b;
    return (a < b) ? d - GL_EPS : d;  // d + p (mod 2^64)
}
GL_HD u64 gl_neg(u64 a) { return a ? GL_P - a : 0; }
GL_HD u64 gl_canon(u64 a) { return a >= GL_P ? a - GL_P : a; }

// (hi*2^64 + lo) mod p, any hi, lo.
GL_HD u64 gl_reduce128(u64 hi, u64 lo) {
    u64 hh = hi >> 32, hl = hi & GL_EPS;
    u64 t0 = lo - hh;
    if (lo < hh) t0 -= GL_EPS;       // borrow: -2^64 = -(2^32-1)
    u64 t1 = (hl << 32) - hl;        // hl * (2^32 - 1)
    u64 r = t0 + t1;
    if (r < t1) r += GL_EPS;         // carry: +2^64 = +(2^32-1)
    return gl_canon(r);
}

GL_HD u64 gl_mulhi64(u64 a, u64 b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul64hi(a, b);
#else
    return (u64)(((unsigned __int128)a * b) >> 64);
#endif
}
GL_HD u64 gl_mul(u64 a, u64 b) { return gl_reduce128(gl_mulhi64(a, b), a * b); }
GL_HD u64 gl_sqr(u64 a) { return gl_mul(a, a); }

// x * 2^S mod p for a compile-time S in [0, 192).
template <int S>
GL_HD u64 gl_mul_pow2(u64 x) {
    static_assert(S >= 0 && S < 192, "shift out of range");
    if constexpr (S == 0) {
        return x;
    } else if constexpr (S >= 96) {
        return gl_neg(gl_mul_pow2<S - 96>(x));  // 2^96 = -1
    } else if constexpr (S <= 64) {
        u64 hi = x >> (64 - S);
        u64 lo = (S == 64) ? 0ULL : (x << (S & 63));
        return gl_reduce128(hi, lo);
    } else {
        // 64 < S < 96: x*2^S = lo'*2^64 + hi'*2^128 with (hi',lo') = x << (S-64);
        // 2^128 = -2^32, and hi' < 2^32 so hi'<<32 <= p-1 is canonical.
        u64 lo2 = x << (S - 64);
        u64 hi2 = x >> (128 - S);
        return gl_sub(gl_reduce128(lo2, 0), hi2 << 32);
    }
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
