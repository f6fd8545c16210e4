// hash.hip — Poseidon-Goldilocks permutation, sponge, Merkle tree.
#include <hip/hip_runtime.h>
#include "glp_ctx.h"

struct glp_hash_state {
    u64* d_rc = nullptr;
};

void glp_hash_destroy(glp_ctx* c) {
    if (!c || !c->hash) return;
    if (c->hash->d_rc) hipFree(c->hash->d_rc);
    delete c->hash;
    c->hash = nullptr;
}
