// hash_state.h — per-context hashing state shared by hash.hip and fri.hip (internal).
#pragma once
#include <vector>
#include "glp_ctx.h"
#include "hash_kernels.cuh"

struct glp_hash_state {
    u64* d_consts = nullptr;          // rc[360] | circ[12] | diag[12] on the device
    std::vector<u64> h_consts;        // same, host copy (the Fiat-Shamir challenger runs on the host)
    u32* d_pg_coef = nullptr;         // grouped partial-round tables (poseidon_precomp.h), or null
    u64* d_pg_cst = nullptr;
    std::vector<u32> h_pg_coef;
    std::vector<u64> h_pg_cst;
    bool have_consts = false;
    bool small_mds = false;
    u32* d_k256 = nullptr;
    u64* d_k512 = nullptr;
};
glp_hash_state* glp_hash_get(glp_ctx* c);
static inline GlpPoseidonConsts glp_dev_consts(const glp_hash_state* h) {
    return GlpPoseidonConsts{h->d_consts, h->d_consts + 360, h->d_consts + 372, h->d_pg_coef, h->d_pg_cst};
}
static inline GlpPoseidonConsts glp_host_consts(const glp_hash_state* h) {
    return GlpPoseidonConsts{h->h_consts.data(), h->h_consts.data() + 360, h->h_consts.data() + 372,
                             h->h_pg_coef.empty() ? nullptr : h->h_pg_coef.data(), h->h_pg_cst.empty() ? nullptr : h->h_pg_cst.data()};
}
