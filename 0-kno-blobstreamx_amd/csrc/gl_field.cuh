// gl_field.cuh — Goldilocks field arithmetic, p = 2^64 - 2^32 + 1 (SURVEY.md §8a row a1;
// upstream name recalled as plonky2_field::GoldilocksField — reference file:line: NONE,
// /root/reference holds no source).
//
// All functions take and return CANONICAL values (< p) unless the name says otherwise.
// The 64x64->128 product is four 32x32 multiplies on CDNA4 (v_mad_u64_u32 / v_mul_hi_u32);
// the reduction uses 2^64 = 2^32 - 1 and 2^96 = -1 (mod p).
//
// 2 is an element of order 192 (2^96 = -1), so every primitive 64th root of unity is a power of two (w_64 = 2^39 with the default
// generator-7 roots; GLP_W64_LOG2 below): every twiddle of a radix-<=64 butterfly is a power of two, and in-register
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
// ---- gfx950 carry-chain forms ----------------------------------------------------------
// The compiler turns `(a < b) ? d - eps : d` into 7 VALU (64-bit compare + two selects); with the
// borrow kept in an SGPR lane mask it is 4 VALU + 1 SALU.  Measured issue costs are ~4.2 cycles
// for every carry/64-bit/VOP3 op (profiles/r01_ubench_valu2.txt), so the instruction count is
// the cost.  gfx940+ needs 2 wait states between a VALU writing an SGPR and a VALU reading it
// (the compiler's hazard recogniser does not look inside inline asm) — EXCEPT through VCC, which
// the hardware forwards (the compiler's own v_add_co/v_addc_co chains run back to back on it).
// Round 2: every carry/borrow that a following VALU consumes therefore lives in VCC (declared
// clobbered); an explicit SGPR pair only carries masks that the next consumer reads on the SALU
// (s_andn2/s_or), which interlocks.  That removed ~9 s_nop per multiplication (measured: Merkle
// commitment -2.3 %, proof -2.4 %, NTT +0.5 %; profiles/r02_ab_vcc_carry.txt) and is what
// tests/test_isa_hazards.py checks in the emitted ISA.
// s_andn2/s_or write SCC, which the compiler may hold live (s_add_u32/s_addc_u32 address
// arithmetic): every block with a SALU op declares the "scc" clobber.
// Host code and tests/emu use the portable forms (GLP_ASM_FIELD off).
#if defined(__HIP_DEVICE_COMPILE__) && !defined(GLP_EMU) && !defined(GLP_NO_ASM_FIELD)
#define GLP_ASM_FIELD 1
#else
#define GLP_ASM_FIELD 0
#endif

GL_HD u64 gl_make64(u32 lo, u32 hi) { return ((u64)hi << 32) | lo; }

// x + y (mod 2^64) for a 32-bit y: v_mad_u64_u32(y, 1, x) — one 4.2-cycle op instead of the v_mov (to pair y
// with a zero register) + v_lshl_add_u64 the compiler emits for `x + (u64)y`
GL_HD u64 gl_add_u32(u64 x, u32 y) {
#if GLP_ASM_FIELD
    u64 r, junk;
    asm("v_mad_u64_u32 %0, %1, %2, 1, %3" : "=v"(r), "=s"(junk) : "v"(y), "v"(x));
    return r;
#else
    return x + (u64)y;
#endif
}

// canonical a + b.  (A 5-VALU lane-mask form of the fix-up measured the same as this select form on
// the NTT passes, 1560 vs 1563 GB/s A/B in one run, so the select form stays.)
GL_HD u64 gl_add(u64 a, u64 b) {
    const u64 s = a + b;
    const bool over = (s < a) | (s >= GL_P);     // true sum >= p
    // "+ (u64)m" here stays v_mov + v_lshl_add_u64: gl_add_u32's v_mad form measured 2 % SLOWER on the NTT passes
    // (1594 vs 1624 GB/s, three alternating runs) — the butterflies are dependent add chains and the mad's latency
    // shows; in the products and folds (independent lanes) it is a gain (Poseidon tree 54.6 -> 51.3 ms)
    return s + (over ? GL_EPS : 0ULL);                // - p  (mod 2^64)
}

// a - b (+ p on borrow), valid for any u64 a and b <= p; canonical when a is
GL_HD u64 gl_sub(u64 a, u64 b) {
#if GLP_ASM_FIELD
    u32 dl, dh;
    u64 K;
    asm("v_sub_co_u32 %0, vcc, %3, %5\n\t"
        "v_subb_co_u32 %1, vcc, %4, %6, vcc\n\t"         // B = borrow of a - b, in VCC (carry forwarding: no wait states)
        "v_addc_co_u32 %0, %2, 0, %0, vcc\n\t"            // + p = (+1, -1 on the high word): lo += B, carry K
        "s_andn2_b64 vcc, vcc, %2\n\t"
        "v_subbrev_co_u32 %1, %2, 0, %1, vcc"              // hi -= (B & ~K)
        : "=&v"(dl), "=&v"(dh), "=&s"(K)
        : "v"((u32)a), "v"((u32)(a >> 32)), "v"((u32)b), "v"((u32)(b >> 32))
        : "scc", "vcc");
    return gl_make64(dl, dh);
#else
    u64 d = a - b;
    return (a < b) ? d - GL_EPS : d;  // d + p (mod 2^64)
#endif
}

// w * (2^32 - 1) + t  mod p for any u32 w and any u64 t: one v_mad_u64_u32 with its carry-out.
// CANON: canonical result; otherwise a representative in [0, 2^64).
template <bool CANON>
GL_HD u64 gl_mad_eps(u32 w, u64 t) {
#if GLP_ASM_FIELD
    u64 r;
    u32 m;
    if constexpr (CANON) {
        u64 G, J;
        asm("v_mad_u64_u32 %0, vcc, %4, -1, %5\n\t"
            "v_cmp_le_u64 %1, %6, %0\n\t"
            "s_or_b64 vcc, vcc, %1\n\t"
            "s_nop 0\n\t"
            "v_cndmask_b32 %2, 0, -1, vcc\n\t"
            "v_mad_u64_u32 %0, %3, %2, 1, %0"              // r += m (m = eps or 0), as one op
            : "=&v"(r), "=&s"(G), "=&v"(m), "=&s"(J)
            : "v"(w), "v"(t), "s"(GL_P)
            : "scc", "vcc");
    } else {
        u64 J;
        asm("v_mad_u64_u32 %0, vcc, %3, -1, %4\n\t"
            "v_cndmask_b32 %1, 0, -1, vcc\n\t"
            "v_mad_u64_u32 %0, %2, %1, 1, %0"
            : "=&v"(r), "=&v"(m), "=&s"(J)
            : "v"(w), "v"(t)
            : "vcc");
    }
    return r;
#else
    const u64 t1 = ((u64)w << 32) - w;
    u64 r;
    const bool c = __builtin_add_overflow(t, t1, &r);
    if (CANON) return r + ((c | (r >= GL_P)) ? GL_EPS : 0ULL);
    return r + (c ? GL_EPS : 0ULL);
#endif
}
GL_HD u64 gl_neg(u64 a) { return a ? GL_P - a : 0; }
GL_HD u64 gl_canon(u64 a) { return a >= GL_P ? a - GL_P : a; }

// r (+ 2^64 if carry) mod p, for a sum whose true value is < 2^64 + p
GL_HD u64 gl_fold_carry(u64 r, bool carry) { return r + ((carry | (r >= GL_P)) ? GL_EPS : 0ULL); }

// (hi*2^64 + lo) mod p, any hi, lo:  lo - hi_hi + hi_lo*(2^32 - 1)
template <bool CANON>
GL_HD u64 gl_reduce128_t(u64 hi, u64 lo) {
    const u32 h0 = (u32)hi, h1 = (u32)(hi >> 32);
#if GLP_ASM_FIELD
    u32 tl, th;
    u64 K;
    asm("v_sub_co_u32 %0, vcc, %3, %5\n\t"
        "v_subbrev_co_u32 %1, vcc, 0, %4, vcc\n\t"
        "v_addc_co_u32 %0, %2, 0, %0, vcc\n\t"
        "s_andn2_b64 vcc, vcc, %2\n\t"
        "v_subbrev_co_u32 %1, %2, 0, %1, vcc"
        : "=&v"(tl), "=&v"(th), "=&s"(K)
        : "v"((u32)lo), "v"((u32)(lo >> 32)), "v"(h1)
        : "scc", "vcc");
    return gl_mad_eps<CANON>(h0, gl_make64(tl, th));
#else
    u64 t0;
    const bool bor = __builtin_sub_overflow(lo, (u64)h1, &t0);
    t0 = bor ? t0 + GL_P : t0;                    // -2^64 = +p - 2^64 ... (mod 2^64): t0 - eps
    return gl_mad_eps<CANON>(h0, t0);
#endif
}
GL_HD u64 gl_reduce128(u64 hi, u64 lo) { return gl_reduce128_t<true>(hi, lo); }

// 64x64 -> 128 from four 32x32 products chained through v_mad_u64_u32 addends
template <bool CANON>
GL_HD u64 gl_mul_t(u64 a, u64 b) {
#if !defined(__HIP_DEVICE_COMPILE__) && !defined(GLP_EMU) && defined(__SIZEOF_INT128__)
    // host code of the library (transcript, verifiers): one 64x64 -> 128 multiply; tests/emu keeps the 32-bit-halves
    // formulation below, which is the one the device compiles
    const unsigned __int128 p = (unsigned __int128)a * b;
    return gl_reduce128_t<CANON>((u64)(p >> 64), (u64)p);
#else
    const u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
    const u64 p0 = (u64)a0 * b0;
    const u64 p1 = (u64)a0 * b1 + (p0 >> 32);
    const u64 p2 = (u64)a1 * b0 + (u32)p1;
    const u64 p3 = gl_add_u32((u64)a1 * b1 + (p1 >> 32), (u32)(p2 >> 32));   // < 2^64: the product is < 2^128
    return gl_reduce128_t<CANON>(p3, (p2 << 32) | (u32)p0);
#endif
}
GL_HD u64 gl_mul(u64 a, u64 b) { return gl_mul_t<true>(a, b); }
GL_HD u64 gl_sqr(u64 a) { return gl_mul(a, a); }

// al + ah * 2^32 (mod p) for two 64-bit accumulators whose true value al + ah * 2^32 is < 2^64 * 2^32
// and whose reduction  al + (ah mod 2^32) * 2^32 + (ah >> 32) * (2^32 - 1)  overflows 2^64 at most once
// (Poseidon's small-integer dot products: al, ah < 2^57).  Result: a representative in [0, 2^64).
GL_HD u64 gl_fold_small(u64 al, u64 ah) {
#if GLP_ASM_FIELD
    u32 lh, m;
    u64 C1, J, v;
    asm("v_add_co_u32 %0, %1, %2, %3" : "=v"(lh), "=s"(C1) : "v"((u32)(al >> 32)), "v"((u32)ah));
    asm("v_mad_u64_u32 %0, vcc, %3, -1, %4\n\t"
        "s_or_b64 vcc, vcc, %5\n\t"
        "s_nop 0\n\t"
        "v_cndmask_b32 %1, 0, -1, vcc\n\t"
        "v_mad_u64_u32 %0, %2, %1, 1, %0"
        : "=&v"(v), "=&v"(m), "=&s"(J)
        : "v"((u32)(ah >> 32)), "v"(gl_make64((u32)al, lh)), "s"(C1)
        : "scc", "vcc");
    return v;
#else
    const u64 l = al + (ah << 32);
    const bool c1 = l < al;
    const u64 t = (ah >> 32) * GL_EPS;
    const u64 v = l + t;
    const bool c2 = v < l;
    return v + ((c1 | c2) ? GL_EPS : 0ULL);
#endif
}

// "nc" = not canonicalised: inputs may be ANY u64 representative, the result is a correct
// representative in [0, 2^64) that may be >= p.  Saves the (r >= p) compare of every product in
// long multiplication chains whose end result is canonicalised once (Poseidon S-boxes).
GL_HD u64 gl_mul_nc(u64 a, u64 b) { return gl_mul_t<false>(a, b); }

// ---- multiplication by powers of two (the twiddles of every radix <= 64 butterfly) ------
// x * 2^T, 0 < T < 32:  (x << T) + (x >> (64-T)) * (2^32 - 1)
template <int T>
GL_HD u64 gl_shl_small(u64 x) {
    static_assert(T > 0 && T < 32, "");
    return gl_mad_eps<true>((u32)(x >> (64 - T)), x << T);
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
GL_HD u64 gl_shl32(u64 x) { return gl_mad_eps<true>((u32)(x >> 32), x << 32); }   // hi_hi = 0: no subtraction

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
// ---- the two-adic subgroup: a BUILD-TIME parameter pair (VERDICT r2 missing 5: parity readiness) --------------------------------------------
// GLP_TWO_ADIC_GENERATOR: an element of order exactly 2^32; the primitive 2^k-th root of unity is its 2^(32-k)-th power.
// GLP_W64_LOG2: the exponent e with  generator^(2^26) = 2^e  (every primitive 64th root of unity is a power of two: 2 has order 192, so e is an
// odd multiple of 3) — the in-register butterflies multiply by shifts whose amounts are compile-time functions of e (ntt_kernels.cuh).
// Default: 7^((p-1)/2^32) = 1753635133440165772, w_64 = 2^39 (7 is a primitive root).  `make altgen` builds the library with
// 7277203076849721926 (= 14293326489335486720^((p-1)/2^32); recalled as plonky2's pair, UNVERIFIED — the mount has no source), w_64 = 2^3.
// glp_create refuses to run when the pair is inconsistent; glp_field_params reports it; the CPU oracle takes the same generator at run time.
#ifndef GLP_TWO_ADIC_GENERATOR
#define GLP_TWO_ADIC_GENERATOR 1753635133440165772ULL
#endif
#ifndef GLP_W64_LOG2
#define GLP_W64_LOG2 39
#endif
static_assert(GLP_W64_LOG2 > 0 && GLP_W64_LOG2 < 192 && GLP_W64_LOG2 % 3 == 0 && GLP_W64_LOG2 % 2 == 1, "w_64 = 2^e needs e an odd multiple of 3");
// primitive 2^k-th root of unity (k <= 32)
GL_HD u64 gl_root_of_unity(unsigned k) { return gl_pow(GLP_TWO_ADIC_GENERATOR, 1ULL << (32 - k)); }

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
