// nnf25519.h — witness arithmetic for the NON-NATIVE field F_q, q = 2^255 - 19 (Curve25519's base field), as the in-circuit Ed25519
// gadgets use it (0-kno-blobstreamx_amd/ed25519_circuit.py; upstream name recalled: curta's Ed25519
// chip — reference file:line NONE, the mount is empty).  Host C++ only: this is the straight-line witness evaluator's op 14 (verify.hip).
//
// Representation: 11 limbs of 24 bits, little endian (264 bits).  A product a * b = k * q + r is shown IN THE CIRCUIT over the integers, column by
// column of the limb convolution, with carries:
//     col_t = sum_{i+j=t} a_i b_j + 19 k_t - 2^15 k_{t-10} - r_t,      col_t + c_{t-1} = c_t * 2^24   (t = 0..21, c_{-1} = 0, c_21 = 0)
// (q = 2^255 - 19 and 255 = 10 * 24 + 15: k * 2^255 puts k_j * 2^15 into column j + 10).  The prover supplies r (11 limbs, canonical: r < q),
// k (12 limbs) and the carries c_0..c_20 (signed, stored mod p); the circuit range-checks them and evaluates the columns with arithmetic gates.
// This header computes exactly those 44 values from the operands' limb values.  Operand limbs may be "loose" (up to 2^27): the builder tracks
// the bounds and refuses products whose columns could leave the carry range.
#pragma once
#include <stdint.h>

#define GLP_NNF_LIMBS 11
#define GLP_NNF_BITS 24
#define GLP_NNF_KLIMBS 12
#define GLP_NNF_COLS 22
#define GLP_NNF_OUT (GLP_NNF_LIMBS + GLP_NNF_KLIMBS + GLP_NNF_COLS - 1)   // r, k, c_0..c_20 = 44 values

namespace glp_nnf {
typedef unsigned __int128 u128;
typedef __int128 i128;
static const uint64_t GOLDILOCKS_P = 0xFFFFFFFF00000001ULL;

// little-endian multi-word unsigned integers, 10 x 64 bits (640 bits: products of two 272-bit values fit)
struct Big { uint64_t w[10]; };
static inline Big big_zero() { Big z; for (int i = 0; i < 10; i++) z.w[i] = 0; return z; }
static inline void big_add_shifted(Big& z, u128 v, unsigned bit) {          // z += v << bit
    unsigned word = bit >> 6, sh = bit & 63;
    uint64_t parts[3] = {(uint64_t)v, (uint64_t)(v >> 64), 0};
    if (sh) { parts[2] = parts[1] >> (64 - sh); parts[1] = (parts[1] << sh) | (parts[0] >> (64 - sh)); parts[0] <<= sh; }
    unsigned carry = 0;
    for (int i = 0; i < 3 || carry; i++) {
        if (word + i >= 10) break;
        u128 s = (u128)z.w[word + i] + (i < 3 ? parts[i] : 0) + carry;
        z.w[word + i] = (uint64_t)s;
        carry = (unsigned)(s >> 64);
    }
}
static inline bool big_is_zero_from(const Big& z, unsigned bit) {           // z >> bit == 0
    unsigned word = bit >> 6, sh = bit & 63;
    if (word >= 10) return true;
    if (z.w[word] >> sh) return false;
    for (unsigned i = word + 1; i < 10; i++) if (z.w[i]) return false;
    return true;
}
static inline Big big_shr(const Big& z, unsigned bit) {
    Big r = big_zero();
    unsigned word = bit >> 6, sh = bit & 63;
    for (unsigned i = 0; i + word < 10; i++) {
        uint64_t lo = z.w[i + word] >> sh;
        uint64_t hi = (sh && i + word + 1 < 10) ? (z.w[i + word + 1] << (64 - sh)) : 0;
        r.w[i] = lo | hi;
    }
    return r;
}
static inline Big big_low(const Big& z, unsigned bits) {                    // z mod 2^bits
    Big r = z;
    unsigned word = bits >> 6, sh = bits & 63;
    if (word < 10) { if (sh) r.w[word] &= ((1ULL << sh) - 1); else r.w[word] = 0; for (unsigned i = word + 1; i < 10; i++) r.w[i] = 0; }
    return r;
}
static inline Big big_add(const Big& a, const Big& b) {
    Big r; unsigned carry = 0;
    for (int i = 0; i < 10; i++) { u128 s = (u128)a.w[i] + b.w[i] + carry; r.w[i] = (uint64_t)s; carry = (unsigned)(s >> 64); }
    return r;
}
static inline Big big_mul_small(const Big& a, uint64_t m) {
    Big r; uint64_t carry = 0;
    for (int i = 0; i < 10; i++) { u128 s = (u128)a.w[i] * m + carry; r.w[i] = (uint64_t)s; carry = (uint64_t)(s >> 64); }
    return r;
}
static inline int big_cmp(const Big& a, const Big& b) {
    for (int i = 9; i >= 0; i--) { if (a.w[i] != b.w[i]) return a.w[i] < b.w[i] ? -1 : 1; }
    return 0;
}
static inline Big big_sub(const Big& a, const Big& b) {                     // a >= b
    Big r; unsigned borrow = 0;
    for (int i = 0; i < 10; i++) { u128 d = (u128)a.w[i] - b.w[i] - borrow; r.w[i] = (uint64_t)d; borrow = (unsigned)((d >> 64) & 1); }
    return r;
}
static inline uint64_t big_limb(const Big& z, unsigned idx) {               // 24-bit limb idx
    unsigned bit = idx * GLP_NNF_BITS, word = bit >> 6, sh = bit & 63;
    if (word >= 10) return 0;
    uint64_t v = z.w[word] >> sh;
    if (sh > 64 - GLP_NNF_BITS && word + 1 < 10) v |= z.w[word + 1] << (64 - sh);
    return v & ((1ULL << GLP_NNF_BITS) - 1);
}

// out[0..11) = r limbs, out[11..23) = k limbs, out[23..44) = carries c_0..c_20 as Goldilocks elements (negative c -> p - |c|).
// a, b: limb values (any u64 below 2^28; the circuit's own bound tracking is stricter).  Returns false when an operand limb is out of range or
// the carries leave +-2^62 (cannot happen for in-range operands).
static inline bool mul_hints(const uint64_t* a, const uint64_t* b, uint64_t* out) {
    for (int i = 0; i < GLP_NNF_LIMBS; i++) if ((a[i] >> 28) || (b[i] >> 28)) return false;
    u128 ab[GLP_NNF_COLS];
    for (int t = 0; t < GLP_NNF_COLS; t++) ab[t] = 0;
    for (int i = 0; i < GLP_NNF_LIMBS; i++)
        for (int j = 0; j < GLP_NNF_LIMBS; j++) ab[i + j] += (u128)a[i] * b[j];
    Big P = big_zero();
    for (int t = 0; t < 2 * GLP_NNF_LIMBS - 1; t++) big_add_shifted(P, ab[t], (unsigned)(t * GLP_NNF_BITS));
    // P = k * q + r with q = 2^255 - 19:  P = Hi * 2^255 + Lo = Hi * q + (19 Hi + Lo), repeated until the remainder is below 2^255
    Big k = big_zero(), rem = P;
    while (!big_is_zero_from(rem, 255)) {
        const Big hi = big_shr(rem, 255);
        k = big_add(k, hi);
        rem = big_add(big_low(rem, 255), big_mul_small(hi, 19));
    }
    Big q = big_zero();
    q.w[0] = 0xFFFFFFFFFFFFFFEDULL; q.w[1] = 0xFFFFFFFFFFFFFFFFULL; q.w[2] = 0xFFFFFFFFFFFFFFFFULL; q.w[3] = 0x7FFFFFFFFFFFFFFFULL;
    if (big_cmp(rem, q) >= 0) { rem = big_sub(rem, q); Big one = big_zero(); one.w[0] = 1; k = big_add(k, one); }
    uint64_t r[GLP_NNF_LIMBS], kk[GLP_NNF_KLIMBS];
    for (int i = 0; i < GLP_NNF_LIMBS; i++) r[i] = big_limb(rem, (unsigned)i);
    for (int i = 0; i < GLP_NNF_KLIMBS; i++) kk[i] = big_limb(k, (unsigned)i);
    if (!big_is_zero_from(k, GLP_NNF_KLIMBS * GLP_NNF_BITS)) return false;
    for (int i = 0; i < GLP_NNF_LIMBS; i++) out[i] = r[i];
    for (int i = 0; i < GLP_NNF_KLIMBS; i++) out[GLP_NNF_LIMBS + i] = kk[i];
    i128 carry = 0;
    for (int t = 0; t < GLP_NNF_COLS; t++) {
        i128 col = (i128)ab[t] + carry;
        if (t < GLP_NNF_KLIMBS) col += (i128)19 * (i128)kk[t];
        if (t >= 10 && t - 10 < GLP_NNF_KLIMBS) col -= ((i128)kk[t - 10]) << 15;
        if (t < GLP_NNF_LIMBS) col -= (i128)r[t];
        if (col & (((i128)1 << GLP_NNF_BITS) - 1)) return false;           // the columns of an exact identity divide by 2^24
        carry = col >> GLP_NNF_BITS;                                        // arithmetic shift: exact
        if (carry > ((i128)1 << 62) || carry < -((i128)1 << 62)) return false;
        if (t < GLP_NNF_COLS - 1) out[GLP_NNF_LIMBS + GLP_NNF_KLIMBS + t] = carry >= 0 ? (uint64_t)carry : GOLDILOCKS_P - (uint64_t)(-carry);
    }
    return carry == 0;
}
}  // namespace glp_nnf
