// plonk_gates.h — the gate set of the build-defined circuit, written ONCE for the two places that
// evaluate it: the quotient kernel (K7: base field, at every point of the LDE domain) and the native verifier
// (extension field, at zeta).  Upstream names (recalled, unverified; reference file:line NONE — the mount is empty):
// plonky2::gates::{arithmetic_base::ArithmeticGate, constant::ConstantGate, public_input::PublicInputGate,
// poseidon::PoseidonGate}.  The gates here have the same JOBS, not the same wire layouts.
//
// Constant columns (per row): [0] q_arith  [1] c0  [2] c1  [3] c2  [4] q_pi  [5] q_pos
//   arithmetic  every group of 4 routed wires (x, y, z, w):   q_arith * (c0*x*y + c1*z + c2 - w) = 0
//               (c0 = c1 = 0 makes it the constant gate  w = c2)
//   public input  row i < n_public:                            q_pi * wire_0 - PI(x) = 0,  PI = the polynomial that is
//               public_input[i] on row i < n_public and 0 on every other row (the verifier evaluates it itself)
//   Poseidon    a q_pos row carries one width-12 permutation with an optional SWAP of its first two 4-word blocks (the Merkle
//               path step: "is this node the right child?"):  wires 0..11 = input, 12..23 = output, 24 = swap bit s, 25..130 = the
//               S-box inputs of every round after the first (3 x 12 full, 22 partial, 4 x 12 full), 131..134 = delta_i =
//               s * (in[4+i] - in[i]); the permuted state is (in[0..4) + delta, in[4..8) - delta, in[8..12)).  Every
//               constraint has degree <= 7 in the wires (8 with the selector): 123 constraints
//                 s (s - 1)                                                                       (1)
//                 delta_i - s (in[4+i] - in[i])                                   i = 0..3      (4)
//                 a_r[i]  - (state before the S-box of full round r)[i]          r = 1..3      (36)
//                 p_r     - (lane 0 before the S-box of partial round r)          r = 0..21     (22)
//                 b_r[i]  - (state before the S-box of final full round r)[i]     r = 0..3      (48)
//                 out[i]  - (state after the last MDS layer)[i]                                 (12)
//               with the ctx's injected round constants and MDS (the same permutation as glp_poseidon_permute).
#pragma once
#include "gl_field.cuh"
#include "hash_kernels.cuh"

#define GLP_PLONK_NCONST 6
#define GLP_POS_GATE_WIRES 135
#define GLP_POS_GATE_CONSTRAINTS 123
#define GLP_POS_SWAP_WIRE 24
#define GLP_POS_ADVICE0 25            // first S-box-input wire
#define GLP_POS_DELTA0 131
#define GLP_CIRCUIT_POSEIDON_GATE 1u

// ---- field policies ------------------------------------------------------------------------------------------------
// base field, canonical values in and out of every operation (K7, and the witness filler)
struct GlpGateBase {
    typedef u64 F;
    static GL_HD F zero() { return 0; }
    static GL_HD F add(F a, F b) { return gl_add(a, b); }
    static GL_HD F sub(F a, F b) { return gl_sub(a, b); }
    static GL_HD F mul(F a, F b) { return gl_mul(a, b); }
    static GL_HD F scale(F a, u64 k) { return gl_mul(a, k); }
    static GL_HD F addc(F a, u64 k) { return gl_add(a, k); }
};
// quadratic extension (the verifier, at zeta)
struct GlpGateExt {
    typedef gl_ext2 F;
    static GL_HD F zero() { return gl_ext2{0, 0}; }
    static GL_HD F add(F a, F b) { return gl_ext_add(a, b); }
    static GL_HD F sub(F a, F b) { return gl_ext_sub(a, b); }
    static GL_HD F mul(F a, F b) { return gl_ext_mul(a, b); }
    static GL_HD F scale(F a, u64 k) { return gl_ext_scale(a, k); }
    static GL_HD F addc(F a, u64 k) { return gl_ext2{gl_add(a.a, k), a.b}; }
};

template <class O>
GL_HD typename O::F glp_gate_sbox7(typename O::F x) {
    const typename O::F x2 = O::mul(x, x), x3 = O::mul(x2, x), x4 = O::mul(x2, x2);
    return O::mul(x3, x4);
}
// s <- MDS * s + rc_next   (row r = sum_i s[(i + r) % 12] * circ[i] + s[r] * diag[r]; rc_next may be null)
template <class O>
GL_HD void glp_gate_mds(typename O::F (&s)[12], const u64* circ, const u64* diag, const u64* rc_next) {
    typename O::F out[12];
    for (int r = 0; r < 12; r++) {
        typename O::F acc = O::scale(s[r], diag[r]);
        for (int i = 0; i < 12; i++) acc = O::add(acc, O::scale(s[(i + r) % 12], circ[i]));
        out[r] = rc_next ? O::addc(acc, rc_next[r]) : acc;
    }
    for (int r = 0; r < 12; r++) s[r] = out[r];
}

// The 123 constraints of one Poseidon row, in order.  wire(j) -> F gives wire j of the row (j < GLP_POS_GATE_WIRES);
// emit(F) receives each constraint value (zero on a correctly filled row).
// consts: rc [30][12], circ [12], diag [12] (the arguments of glp_set_poseidon_constants).
template <class O, class WireFn, class EmitFn>
GL_HD void glp_poseidon_gate_constraints(WireFn&& wire, const u64* rc, const u64* circ, const u64* diag, EmitFn&& emit) {
    typedef typename O::F F;
    F s[12];
    {   // the conditional swap: 5 constraints, then the state that is permuted
        const F sw = wire(GLP_POS_SWAP_WIRE);
        emit(O::sub(O::mul(sw, sw), sw));
        F in[12];
        for (int i = 0; i < 12; i++) in[i] = wire(i);
        for (int i = 0; i < 4; i++) {
            const F d = wire(GLP_POS_DELTA0 + i);
            emit(O::sub(d, O::mul(sw, O::sub(in[4 + i], in[i]))));
            in[i] = O::add(in[i], d);
            in[4 + i] = O::sub(in[4 + i], d);
        }
        for (int i = 0; i < 12; i++) s[i] = O::addc(in[i], rc[i]);        // S-box inputs of round 0: state + constants
    }
    int rnd = 0, aw = GLP_POS_ADVICE0;
    for (int r = 0; r < GLP_POS_FULL_HALF; r++, rnd++) {
        if (r > 0) {
            for (int i = 0; i < 12; i++) { const F a = wire(aw + i); emit(O::sub(a, s[i])); s[i] = a; }
            aw += 12;
        }
        for (int i = 0; i < 12; i++) s[i] = glp_gate_sbox7<O>(s[i]);
        glp_gate_mds<O>(s, circ, diag, rc + (rnd + 1) * 12);
    }
    for (int r = 0; r < GLP_POS_PARTIAL; r++, rnd++) {
        const F p = wire(aw++);
        emit(O::sub(p, s[0]));
        s[0] = glp_gate_sbox7<O>(p);
        glp_gate_mds<O>(s, circ, diag, rc + (rnd + 1) * 12);
    }
    for (int r = 0; r < GLP_POS_FULL_HALF; r++, rnd++) {
        for (int i = 0; i < 12; i++) { const F b = wire(aw + i); emit(O::sub(b, s[i])); s[i] = glp_gate_sbox7<O>(b); }
        aw += 12;
        glp_gate_mds<O>(s, circ, diag, rnd + 1 < GLP_POS_ROUNDS ? rc + (rnd + 1) * 12 : nullptr);
    }
    for (int i = 0; i < 12; i++) emit(O::sub(wire(12 + i), s[i]));
}

// Witness of one Poseidon row: given the 12 inputs and the swap bit, every other wire — out[j] = wire 12 + j for j < 12 (the output), out[12] is
// wire 24 (the swap bit itself, rewritten unchanged), out[13 .. 118] the S-box inputs in the gate's wire order (wires 25..130), out[119..122] the
// deltas (wires 131..134).  Base field, canonical.  The same walk as the constraints, storing instead of comparing — so a row filled by this function
// satisfies them by construction (for a boolean swap bit), and with swap = 0 the output equals glp_poseidon_permute of the input.
GL_HD void glp_poseidon_gate_fill(const u64 (&in_)[12], u64 swap, const u64* rc, const u64* circ, const u64* diag, u64 (&out)[GLP_POS_GATE_WIRES - 12]) {
    typedef GlpGateBase O;
    u64 in[12], s[12];
    for (int i = 0; i < 12; i++) in[i] = in_[i];
    out[12] = swap;
    for (int i = 0; i < 4; i++) {
        const u64 d = O::mul(swap, O::sub(in[4 + i], in[i]));
        out[GLP_POS_DELTA0 - 12 + i] = d;
        in[i] = O::add(in[i], d);
        in[4 + i] = O::sub(in[4 + i], d);
    }
    for (int i = 0; i < 12; i++) s[i] = O::addc(in[i], rc[i]);
    int rnd = 0, aw = GLP_POS_ADVICE0 - 12;                                   // out[] index of wire 25
    for (int r = 0; r < GLP_POS_FULL_HALF; r++, rnd++) {
        if (r > 0) { for (int i = 0; i < 12; i++) out[aw + i] = s[i]; aw += 12; }
        for (int i = 0; i < 12; i++) s[i] = glp_gate_sbox7<O>(s[i]);
        glp_gate_mds<O>(s, circ, diag, rc + (rnd + 1) * 12);
    }
    for (int r = 0; r < GLP_POS_PARTIAL; r++, rnd++) {
        out[aw++] = s[0];
        s[0] = glp_gate_sbox7<O>(s[0]);
        glp_gate_mds<O>(s, circ, diag, rc + (rnd + 1) * 12);
    }
    for (int r = 0; r < GLP_POS_FULL_HALF; r++, rnd++) {
        for (int i = 0; i < 12; i++) { out[aw + i] = s[i]; s[i] = glp_gate_sbox7<O>(s[i]); }
        aw += 12;
        glp_gate_mds<O>(s, circ, diag, rnd + 1 < GLP_POS_ROUNDS ? rc + (rnd + 1) * 12 : nullptr);
    }
    for (int i = 0; i < 12; i++) out[i] = s[i];
}

// arithmetic gate on one group of 4 wires, without the selector:  c0*x*y + c1*z + c2 - w
template <class O>
GL_HD typename O::F glp_arith_gate(typename O::F c0, typename O::F c1, typename O::F c2, typename O::F x, typename O::F y, typename O::F z,
                                   typename O::F w) {
    return O::sub(O::add(O::add(O::mul(c0, O::mul(x, y)), O::mul(c1, z)), c2), w);
}

// ---- SHA-256 rows (GLP_CIRCUIT_SHA_GATES) ------------------------------------------------------------------------------------
// Four row kinds that make one SHA-256 compression 64 + 64 + 48 (+ 2) rows instead of ~66k arithmetic gates (the JOB
// upstream gives to a separate STARK — curta's SHA chip — done here as custom gates of the same proof system).  Words travel between
// rows as routed 32-bit values (copy constraints); every row decomposes the words it needs into boolean wires of its own:
//   wires 0..11   routed words (per kind, below)          wires 12..43 X bits   44..75 Y bits   76..107 Z bits   108..139 N bits
//   wires 140..143 carry bits                              (bit i of a group = wire base + i, least significant first)
// Four more constant columns select the kind: [6] q_she [7] q_sha [8] q_shw [9] q_add (at most one of q_arith/q_pos/these is 1 on a row).
//   E  (q_she): 0 e 1 f 2 g 3 h 4 d 5 w 6 T1 7 e_new;  X,Y,Z,N = bits of e,f,g,e_new;  c2 of the row = K_t
//               T1 = h + Sigma1(e) + Ch(e,f,g) + K_t + w  (not reduced: < 5 * 2^32),   e_new + 2^32*carry = d + T1
//   A  (q_sha): 0 a 1 b 2 c 3 T1 4 a_new;  X,Y,Z,N = bits of a,b,c,a_new;   a_new + 2^32*carry = T1 + Sigma0(a) + Maj(a,b,c)
//   W  (q_shw): 0 w16 1 w15 2 w7 3 w2 4 w_new;  X,Y,N = bits of w15,w2,w_new (Z = 0);  w_new + 2^32*carry = w16 + sigma0(w15) + w7 + sigma1(w2)
//   ADD(q_add): four additions mod 2^32 — 3k x_k, 3k+1 y_k, 3k+2 s_k;  group k = bits of s_k;  s_k + 2^32*carry_k = x_k + y_k
//               (with y = 0 and a copy constraint s = x it is the 32-bit range check of x)
// Constraints (GLP_SHA_GATE_CONSTRAINTS = 140; degree <= 4 with the selector):
//   0..131   (q_she+q_sha+q_shw+q_add) * b(b-1) for the 132 bit wires 12..143
//   132..139 eight slots shared by the kinds:  slot s = sum_kind q_kind * (the kind's s-th equation)   — see glp_sha_gate_constraints
// Inputs that no row decomposes (h, d, w of E; w16, w7 of W; T1 of A; x, y of ADD) must be range-checked words where they come from
// (outputs of other rows are; free inputs go through an ADD-row range check first).
#define GLP_PLONK_NCONST_SHA 10
#define GLP_CIRCUIT_SHA_GATES 2u
#define GLP_SHA_GATE_WIRES 144
#define GLP_SHA_GATE_CONSTRAINTS 140
#define GLP_SHA_ROW_E 0
#define GLP_SHA_ROW_A 1
#define GLP_SHA_ROW_W 2
#define GLP_SHA_ROW_ADD 3
// ---- extension-arithmetic rows (GLP_CIRCUIT_EXT_GATE) ------------------------------------------------------------------------------
// One more selector column, q_ext (the LAST constant column: index 6, or 10 next to the SHA selectors).  On a q_ext row every chunk of 8 routed
// wires (x0, x1, y0, y1, z0, z1, w0, w1) is one multiply-add in F_p[X]/(X^2 - 7):  w = x * y + z, i.e.
//     e0 = x0*y0 + 7*x1*y1 + z0 - w0,      e1 = x0*y1 + x1*y0 + z1 - w1
// The two values share the chunk's two arithmetic-gate slots of the alpha order (slot = q_arith * gate + q_ext * e: at most one of the selectors
// is set on a row), so the constraint count does not change.  What an in-circuit verifier mostly does is extension arithmetic: four arithmetic
// gates (16 wires) per product become one chunk (8 wires).
#define GLP_CIRCUIT_EXT_GATE 4u
GL_HD int glp_plonk_n_const(u32 flags) {
    return GLP_PLONK_NCONST + ((flags & GLP_CIRCUIT_SHA_GATES) ? 4 : 0) + ((flags & GLP_CIRCUIT_EXT_GATE) ? 1 : 0);
}
template <class O>
GL_HD void glp_ext_gate(const typename O::F (&w)[8], typename O::F& e0, typename O::F& e1) {
    e0 = O::sub(O::add(O::add(O::mul(w[0], w[2]), O::scale(O::mul(w[1], w[3]), 7)), w[4]), w[6]);
    e1 = O::sub(O::add(O::add(O::mul(w[0], w[3]), O::mul(w[1], w[2])), w[5]), w[7]);
}

// wire(j) -> F: wire j of the row; q[4] = (q_she, q_sha, q_shw, q_add) at the point; c2 = the row's c2 column; emit receives the 140
// selector-weighted constraint values in order.
template <class O, class WireFn, class EmitFn>
GL_HD void glp_sha_gate_constraints(WireFn&& wire, const typename O::F (&q)[4], typename O::F c2, EmitFn&& emit) {
    typedef typename O::F F;
    const F q_any = O::add(O::add(q[0], q[1]), O::add(q[2], q[3]));
    auto boolc = [&](F b) { emit(O::mul(q_any, O::sub(O::mul(b, b), b))); };
    auto xr = [&](F a, F b) { return O::sub(O::add(a, b), O::scale(O::mul(a, b), 2)); };          // a xor b for booleans
    F x[32], y[32];
    F px = O::zero(), py = O::zero(), pz = O::zero(), pn = O::zero();
    for (int i = 0; i < 32; i++) { x[i] = wire(12 + i); boolc(x[i]); px = O::add(px, O::scale(x[i], 1ull << i)); }
    for (int i = 0; i < 32; i++) { y[i] = wire(44 + i); boolc(y[i]); py = O::add(py, O::scale(y[i], 1ull << i)); }
    F s_ch = O::zero(), s_maj = O::zero();                   // Ch(x,y,z) and Maj(x,y,z) as words
    for (int i = 0; i < 32; i++) {
        const F z = wire(76 + i);
        boolc(z);
        pz = O::add(pz, O::scale(z, 1ull << i));
        const F ch = O::add(z, O::mul(x[i], O::sub(y[i], z)));
        const F xy = O::mul(x[i], y[i]);
        const F maj = O::add(xy, O::mul(z, O::sub(O::add(x[i], y[i]), O::scale(xy, 2))));
        s_ch = O::add(s_ch, O::scale(ch, 1ull << i));
        s_maj = O::add(s_maj, O::scale(maj, 1ull << i));
    }
    for (int i = 0; i < 32; i++) { const F nb = wire(108 + i); boolc(nb); pn = O::add(pn, O::scale(nb, 1ull << i)); }
    F cw[4];
    for (int k = 0; k < 4; k++) { cw[k] = wire(140 + k); boolc(cw[k]); }
    F S1 = O::zero(), S0 = O::zero(), s0 = O::zero(), s1 = O::zero();   // Sigma1(x), Sigma0(x), sigma0(x), sigma1(y)
    for (int i = 0; i < 32; i++) {
        const u64 w = 1ull << i;
        S1 = O::add(S1, O::scale(xr(xr(x[(i + 6) & 31], x[(i + 11) & 31]), x[(i + 25) & 31]), w));
        S0 = O::add(S0, O::scale(xr(xr(x[(i + 2) & 31], x[(i + 13) & 31]), x[(i + 22) & 31]), w));
        F a = xr(x[(i + 7) & 31], x[(i + 18) & 31]);
        if (i + 3 < 32) a = xr(a, x[i + 3]);
        s0 = O::add(s0, O::scale(a, w));
        F b = xr(y[(i + 17) & 31], y[(i + 19) & 31]);
        if (i + 10 < 32) b = xr(b, y[i + 10]);
        s1 = O::add(s1, O::scale(b, w));
    }
    const u64 two32 = 1ull << 32;
    const F w0 = wire(0), w1 = wire(1), w2 = wire(2), w3 = wire(3), w4 = wire(4), w5 = wire(5), w6 = wire(6), w7 = wire(7), w8 = wire(8),
            w9 = wire(9), w10 = wire(10), w11 = wire(11);
    const F car2 = O::add(cw[0], O::scale(cw[1], 2)), car3 = O::add(car2, O::scale(cw[2], 4));
    auto mix = [&](F e, F a, F w, F d) { return O::add(O::add(O::mul(q[0], e), O::mul(q[1], a)), O::add(O::mul(q[2], w), O::mul(q[3], d))); };
    const F zero = O::zero();
    // slot 0..3: the packed bit groups are the routed words they decompose
    emit(mix(O::sub(px, w0), O::sub(px, w0), O::sub(px, w1), O::sub(px, w2)));
    emit(mix(O::sub(py, w1), O::sub(py, w1), O::sub(py, w3), O::sub(py, w5)));
    emit(mix(O::sub(pz, w2), O::sub(pz, w2), zero, O::sub(pz, w8)));
    emit(mix(O::sub(pn, w7), O::sub(pn, w4), O::sub(pn, w4), O::sub(pn, w11)));
    // slot 4: E: T1;  A: a_new;  W: w_new;  ADD: sum 0
    const F e4 = O::sub(w6, O::add(O::add(O::add(w3, S1), O::add(s_ch, c2)), w5));
    const F a4 = O::sub(O::add(w4, O::scale(car3, two32)), O::add(O::add(w3, S0), s_maj));
    const F ww4 = O::sub(O::add(w4, O::scale(car2, two32)), O::add(O::add(w0, s0), O::add(w2, s1)));
    const F d4 = O::sub(O::add(w2, O::scale(cw[0], two32)), O::add(w0, w1));
    emit(mix(e4, a4, ww4, d4));
    // slot 5: E: e_new;  ADD: sum 1.   slots 6, 7: ADD sums 2, 3
    const F e5 = O::sub(O::add(w7, O::scale(car3, two32)), O::add(w4, w6));
    const F d5 = O::sub(O::add(w5, O::scale(cw[1], two32)), O::add(w3, w4));
    emit(mix(e5, zero, zero, d5));
    emit(O::mul(q[3], O::sub(O::add(w8, O::scale(cw[2], two32)), O::add(w6, w7))));
    emit(O::mul(q[3], O::sub(O::add(w11, O::scale(cw[3], two32)), O::add(w9, w10))));
}

// SHA-256 word functions on the host / in the filler (words in the low 32 bits of a u64)
GL_HD u64 glp_sha_rotr(u64 x, int r) { return ((x >> r) | (x << (32 - r))) & 0xFFFFFFFFull; }
GL_HD u64 glp_sha_S1(u64 e) { return glp_sha_rotr(e, 6) ^ glp_sha_rotr(e, 11) ^ glp_sha_rotr(e, 25); }
GL_HD u64 glp_sha_S0(u64 a) { return glp_sha_rotr(a, 2) ^ glp_sha_rotr(a, 13) ^ glp_sha_rotr(a, 22); }
GL_HD u64 glp_sha_s0(u64 w) { return glp_sha_rotr(w, 7) ^ glp_sha_rotr(w, 18) ^ (w >> 3); }
GL_HD u64 glp_sha_s1(u64 w) { return glp_sha_rotr(w, 17) ^ glp_sha_rotr(w, 19) ^ (w >> 10); }
GL_HD u64 glp_sha_ch(u64 e, u64 f, u64 g) { return (e & f) ^ (~e & g & 0xFFFFFFFFull); }
GL_HD u64 glp_sha_maj(u64 a, u64 b, u64 c) { return (a & b) ^ (a & c) ^ (b & c); }

// Witness of one SHA row: the 132 bit wires (12..143) from the routed words r[0..12) of the row (already holding the outputs: the
// witness program computed them).  Words that are not 32-bit values give bits that do not satisfy the row — the proof then fails.
GL_HD void glp_sha_gate_fill(int kind, const u64 (&r)[12], u64 (&bits)[132]) {
    u64 g[4] = {0, 0, 0, 0}, carry = 0;
    const u64 m = 0xFFFFFFFFull;
    switch (kind) {
        case GLP_SHA_ROW_E: g[0] = r[0]; g[1] = r[1]; g[2] = r[2]; g[3] = r[7]; carry = (r[4] + r[6] - r[7]) >> 32; break;
        case GLP_SHA_ROW_A: g[0] = r[0]; g[1] = r[1]; g[2] = r[2]; g[3] = r[4];
                            carry = (r[3] + glp_sha_S0(r[0] & m) + glp_sha_maj(r[0] & m, r[1] & m, r[2] & m) - r[4]) >> 32; break;
        case GLP_SHA_ROW_W: g[0] = r[1]; g[1] = r[3]; g[3] = r[4];
                            carry = (r[0] + glp_sha_s0(r[1] & m) + r[2] + glp_sha_s1(r[3] & m) - r[4]) >> 32; break;
        default:            g[0] = r[2]; g[1] = r[5]; g[2] = r[8]; g[3] = r[11];
                            carry = (((r[0] + r[1] - r[2]) >> 32) & 1) | ((((r[3] + r[4] - r[5]) >> 32) & 1) << 1) | ((((r[6] + r[7] - r[8]) >> 32) & 1) << 2) |
                                    ((((r[9] + r[10] - r[11]) >> 32) & 1) << 3);
                            break;
    }
    for (int k = 0; k < 4; k++)
        for (int i = 0; i < 32; i++) bits[32 * k + i] = (g[k] >> i) & 1ull;
    for (int k = 0; k < 4; k++) bits[128 + k] = (carry >> k) & 1ull;
}
