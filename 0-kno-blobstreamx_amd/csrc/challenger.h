// challenger.h — the Fiat-Shamir transcript (upstream name recalled:
// plonky2::iop::challenger::Challenger — reference file:line NONE).  Host-side C++: a proof
// needs a few hundred permutations of transcript, which is not kernel work.  Build-defined.
#pragma once
#include <vector>
#include "hash_kernels.cuh"

// ---------------------------------------------------------------------------------------
// Challenger: Poseidon duplex sponge, width 12 / rate 8, overwrite mode.
//   observe(x): invalidates pending outputs, buffers x; a full buffer (8) is absorbed at once.
//   challenge(): absorbs whatever is buffered (or squeezes again when outputs ran out), then
//   pops outputs from the END of state[0..8).
// ---------------------------------------------------------------------------------------
struct glp_challenger {
    u64 state[12];
    u64 in[8]; int n_in;
    u64 out[8]; int n_out;
    std::vector<u64> consts;
    bool small_mds;
    void permute() {
        GlpPoseidonConsts k{consts.data(), consts.data() + 360, consts.data() + 372, nullptr, nullptr};   // plain rounds on the host
        if (small_mds) glp_poseidon_permute<true>(state, k);
        else glp_poseidon_permute<false>(state, k);
    }
    void duplex() {
        for (int i = 0; i < n_in; i++) state[i] = in[i];
        n_in = 0;
        permute();
        for (int i = 0; i < 8; i++) out[i] = state[i];
        n_out = 8;
    }
    void observe(u64 x) {
        n_out = 0;
        in[n_in++] = x;
        if (n_in == 8) duplex();
    }
    void observe_ext(gl_ext2 x) { observe(x.a); observe(x.b); }
    u64 challenge() {
        if (n_in > 0 || n_out == 0) duplex();
        return out[--n_out];
    }
    gl_ext2 ext_challenge() { const u64 a = challenge(); const u64 b = challenge(); return {a, b}; }
};

