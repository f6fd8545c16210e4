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
//   Poseidon    a q_pos row carries one width-12 permutation:  wires 0..11 = input, 12..23 = output, 24..129 = the
//               S-box inputs of every round after the first (3 x 12 full, 22 partial, 4 x 12 full), so that every
//               constraint has degree 7 in the wires (8 with the selector): 118 constraints
//                 a_r[i]  - (state before the S-box of full round r)[i]          r = 1..3      (36)
//                 p_r     - (lane 0 before the S-box of partial round r)          r = 0..21     (22)
//                 b_r[i]  - (state before the S-box of final full round r)[i]     r = 0..3      (48)
//                 out[i]  - (state after the last MDS layer)[i]                                 (12)
//               with the ctx's injected round constants and MDS (the same permutation as glp_poseidon_permute).
#pragma once
#include "gl_field.cuh"
#include "hash_kernels.cuh"

#define GLP_PLONK_NCONST 6
#define GLP_POS_GATE_WIRES 130
#define GLP_POS_GATE_CONSTRAINTS 118
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

// The 118 constraints of one Poseidon row, in order.  wire(j) -> F gives wire j of the row (j < GLP_POS_GATE_WIRES);
// emit(F) receives each constraint value (zero on a correctly filled row).
// consts: rc [30][12], circ [12], diag [12] (the arguments of glp_set_poseidon_constants).
template <class O, class WireFn, class EmitFn>
GL_HD void glp_poseidon_gate_constraints(WireFn&& wire, const u64* rc, const u64* circ, const u64* diag, EmitFn&& emit) {
    typedef typename O::F F;
    F s[12];
    for (int i = 0; i < 12; i++) s[i] = O::addc(wire(i), rc[i]);          // S-box inputs of round 0: input + constants
    int rnd = 0, aw = 24;
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

// Witness of one Poseidon row: given the 12 inputs, every other wire (out[0..118) = wires 12..129: output, then the
// S-box inputs in the gate's wire order).  Base field, canonical.  The same walk as the constraints, storing instead of
// comparing — so a row filled by this function satisfies them by construction, and the output equals
// glp_poseidon_permute of the input (tests compare both with the oracle's permutation).
GL_HD void glp_poseidon_gate_fill(const u64 (&in)[12], const u64* rc, const u64* circ, const u64* diag, u64 (&out)[GLP_POS_GATE_WIRES - 12]) {
    typedef GlpGateBase O;
    u64 s[12];
    for (int i = 0; i < 12; i++) s[i] = O::addc(in[i], rc[i]);
    int rnd = 0, aw = 12;                                                     // out[] index of wire 24
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
