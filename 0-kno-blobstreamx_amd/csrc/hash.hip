// hash.hip — C-ABI entry points for Poseidon / Merkle / FRI fold / SHA-2 witness traces
// (include/glprover.h).  Kernels: hash_kernels.cuh.
#include <hip/hip_runtime.h>
#include <string.h>
#include <vector>
#include "glp_ctx.h"
#include "hash_kernels.cuh"

#include "hash_state.h"
#include "poseidon_precomp.h"
#include "ed25519_kernels.cuh"

int glp_ntt_table(glp_ctx* c, int log_N, int inv, const u64** lo, const u64** hi);   // glprover.hip

static const u32 K256[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01,
    0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc,
    0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147,
    0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
    0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08,
    0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
    0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
static const u64 K512[80] = {
    0x428a2f98d728ae22ull, 0x7137449123ef65cdull, 0xb5c0fbcfec4d3b2full, 0xe9b5dba58189dbbcull, 0x3956c25bf348b538ull,
    0x59f111f1b605d019ull, 0x923f82a4af194f9bull, 0xab1c5ed5da6d8118ull, 0xd807aa98a3030242ull, 0x12835b0145706fbeull,
    0x243185be4ee4b28cull, 0x550c7dc3d5ffb4e2ull, 0x72be5d74f27b896full, 0x80deb1fe3b1696b1ull, 0x9bdc06a725c71235ull,
    0xc19bf174cf692694ull, 0xe49b69c19ef14ad2ull, 0xefbe4786384f25e3ull, 0x0fc19dc68b8cd5b5ull, 0x240ca1cc77ac9c65ull,
    0x2de92c6f592b0275ull, 0x4a7484aa6ea6e483ull, 0x5cb0a9dcbd41fbd4ull, 0x76f988da831153b5ull, 0x983e5152ee66dfabull,
    0xa831c66d2db43210ull, 0xb00327c898fb213full, 0xbf597fc7beef0ee4ull, 0xc6e00bf33da88fc2ull, 0xd5a79147930aa725ull,
    0x06ca6351e003826full, 0x142929670a0e6e70ull, 0x27b70a8546d22ffcull, 0x2e1b21385c26c926ull, 0x4d2c6dfc5ac42aedull,
    0x53380d139d95b3dfull, 0x650a73548baf63deull, 0x766a0abb3c77b2a8ull, 0x81c2c92e47edaee6ull, 0x92722c851482353bull,
    0xa2bfe8a14cf10364ull, 0xa81a664bbc423001ull, 0xc24b8b70d0f89791ull, 0xc76c51a30654be30ull, 0xd192e819d6ef5218ull,
    0xd69906245565a910ull, 0xf40e35855771202aull, 0x106aa07032bbd1b8ull, 0x19a4c116b8d2d0c8ull, 0x1e376c085141ab53ull,
    0x2748774cdf8eeb99ull, 0x34b0bcb5e19b48a8ull, 0x391c0cb3c5c95a63ull, 0x4ed8aa4ae3418acbull, 0x5b9cca4f7763e373ull,
    0x682e6ff3d6b2b8a3ull, 0x748f82ee5defb2fcull, 0x78a5636f43172f60ull, 0x84c87814a1f0ab72ull, 0x8cc702081a6439ecull,
    0x90befffa23631e28ull, 0xa4506cebde82bde9ull, 0xbef9a3f7b2c67915ull, 0xc67178f2e372532bull, 0xca273eceea26619cull,
    0xd186b8c721c0c207ull, 0xeada7dd6cde0eb1eull, 0xf57d4f7fee6ed178ull, 0x06f067aa72176fbaull, 0x0a637dc5a2c898a6ull,
    0x113f9804bef90daeull, 0x1b710b35131c471bull, 0x28db77f523047d84ull, 0x32caab7b40c72493ull, 0x3c9ebe0a15c9bebcull,
    0x431d67c49c100d4cull, 0x4cc5d4becb3e42b6ull, 0x597f299cfc657e2aull, 0x5fcb6fab3ad6faecull, 0x6c44198c4a475817ull};

glp_hash_state* glp_hash_get(glp_ctx* c) {
    if (!c->hash) c->hash = new glp_hash_state();
    return c->hash;
}
static glp_hash_state* hs(glp_ctx* c) { return glp_hash_get(c); }

void glp_hash_destroy(glp_ctx* c) {
    if (!c || !c->hash) return;
    if (c->hash->d_consts) hipFree(c->hash->d_consts);
    if (c->hash->d_k256) hipFree(c->hash->d_k256);
    if (c->hash->d_k512) hipFree(c->hash->d_k512);
    if (c->hash->d_pg_coef) hipFree(c->hash->d_pg_coef);
    if (c->hash->d_pg_cst) hipFree(c->hash->d_pg_cst);
    delete c->hash;
    c->hash = nullptr;
}

static GlpPoseidonConsts consts_of(glp_hash_state* h) { return glp_dev_consts(h); }

extern "C" int glp_set_poseidon_constants(glp_ctx* c, const uint64_t* rc, size_t n_rc, const uint64_t* circ, const uint64_t* diag) {
    if (!c) return GLP_E_INVALID;
    GLP_BIND(c);
    if (!rc || !circ || !diag || n_rc != 360) { glp_set_err(c, "glp_set_poseidon_constants: need 360 round constants, 12 + 12 MDS entries"); return GLP_E_INVALID; }
    std::vector<u64> all(384);
    for (int i = 0; i < 360; i++) { if (rc[i] >= GL_P) { glp_set_err(c, "round constant %d not canonical", i); return GLP_E_INVALID; } all[i] = rc[i]; }
    unsigned __int128 sum = 0;
    u64 maxdiag = 0;
    bool small = true;
    for (int i = 0; i < 12; i++) {
        if (circ[i] >= GL_P || diag[i] >= GL_P) { glp_set_err(c, "MDS entry %d not canonical", i); return GLP_E_INVALID; }
        all[360 + i] = circ[i]; all[372 + i] = diag[i];
        sum += circ[i];
        if (diag[i] > maxdiag) maxdiag = diag[i];
        if (circ[i] >> 24 || diag[i] >> 24) small = false;
    }
    if (sum + maxdiag >= ((unsigned __int128)1 << 24)) small = false;   // bound the fast MDS path relies on (hash_kernels.cuh)
    glp_hash_state* h = hs(c);
    GLP_HIPCHK(c, hipStreamSynchronize(c->stream));
    if (!h->d_consts) GLP_HIPCHK(c, hipMalloc((void**)&h->d_consts, 384 * 8));
    GLP_HIPCHK(c, hipMemcpy(h->d_consts, all.data(), 384 * 8, hipMemcpyHostToDevice));
    h->h_consts = all;
    h->have_consts = true;
    h->small_mds = small;
    // grouped partial rounds: only for small-integer MDS whose cubes stay small
    if (h->d_pg_coef) { hipFree(h->d_pg_coef); h->d_pg_coef = nullptr; }
    if (h->d_pg_cst) { hipFree(h->d_pg_cst); h->d_pg_cst = nullptr; }
    h->h_pg_coef.clear(); h->h_pg_cst.clear();
    if (small && glp_poseidon_group_tables(all.data(), h->h_pg_coef, h->h_pg_cst)) {
        GLP_HIPCHK(c, hipMalloc((void**)&h->d_pg_coef, h->h_pg_coef.size() * 4));
        GLP_HIPCHK(c, hipMalloc((void**)&h->d_pg_cst, h->h_pg_cst.size() * 8));
        GLP_HIPCHK(c, hipMemcpy(h->d_pg_coef, h->h_pg_coef.data(), h->h_pg_coef.size() * 4, hipMemcpyHostToDevice));
        GLP_HIPCHK(c, hipMemcpy(h->d_pg_cst, h->h_pg_cst.data(), h->h_pg_cst.size() * 8, hipMemcpyHostToDevice));
    } else {
        h->h_pg_coef.clear(); h->h_pg_cst.clear();
    }
    return GLP_OK;
}

static int need_consts(glp_ctx* c) {
    if (!c->hash || !c->hash->have_consts) { glp_set_err(c, "Poseidon constants not set (glp_set_poseidon_constants)"); return GLP_E_STATE; }
    return GLP_OK;
}

extern "C" int glp_poseidon_permute(glp_ctx* c, uint64_t* d_states, uint64_t n) {
    if (!c) return GLP_E_INVALID;
    GLP_BIND(c);
    if (!d_states && n) { glp_set_err(c, "glp_poseidon_permute: null states"); return GLP_E_INVALID; }
    int rc = need_consts(c);
    if (rc) return rc;
    if (n == 0) return GLP_OK;
    const u64 blocks = (n + 255) / 256;
    if (blocks > 0x7fffffffull) return GLP_E_UNSUPPORTED;
    glp_hash_state* h = c->hash;
    if (h->small_mds) hipLaunchKernelGGL(glp_poseidon_permute_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, c->stream, d_states, n, consts_of(h));
    else hipLaunchKernelGGL(glp_poseidon_permute_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, c->stream, d_states, n, consts_of(h));
    GLP_HIPCHK(c, hipGetLastError());
    return GLP_OK;
}

static int merkle_impl(glp_ctx* c, const u64* src, u64 stride, bool poly_major, u32 leaf_len, u32 log_leaves, u32 cap_h,
                       u64* digests, u64* h_cap) {
    if (!c) return GLP_E_INVALID;
    GLP_BIND(c);
    if (!src || !digests || log_leaves > 40 || cap_h > log_leaves || leaf_len == 0) { glp_set_err(c, "glp_merkle: bad argument"); return GLP_E_INVALID; }
    int rc = need_consts(c);
    if (rc) return rc;
    glp_hash_state* h = c->hash;
    const GlpPoseidonConsts k = consts_of(h);
    const u64 nl = 1ull << log_leaves;
    u64 blocks = (nl + 255) / 256;
    if (blocks > 0x7fffffffull) return GLP_E_UNSUPPORTED;
    const dim3 g((unsigned)blocks), b(256);
    if (h->small_mds) {
        if (poly_major) hipLaunchKernelGGL((glp_hash_leaves_kernel<true, true>), g, b, 0, c->stream, src, stride, leaf_len, nl, digests, k);
        else hipLaunchKernelGGL((glp_hash_leaves_kernel<true, false>), g, b, 0, c->stream, src, stride, leaf_len, nl, digests, k);
    } else {
        if (poly_major) hipLaunchKernelGGL((glp_hash_leaves_kernel<false, true>), g, b, 0, c->stream, src, stride, leaf_len, nl, digests, k);
        else hipLaunchKernelGGL((glp_hash_leaves_kernel<false, false>), g, b, 0, c->stream, src, stride, leaf_len, nl, digests, k);
    }
    GLP_HIPCHK(c, hipGetLastError());
    u64* prev = digests;
    u64 cnt = nl;
    // levels of <= coop_max nodes leave most of the chip idle and are latency-bound: there a permutation is spread over 12 lanes
    // (glp_poseidon_permute_coop), and the last levels (<= 64 nodes) run in one launch.  GLP_COOP_MAX_NODES=0 turns both off.
    static const u64 coop_max = [] { const char* e = getenv("GLP_COOP_MAX_NODES"); return e ? (u64)atoll(e) : (u64)16384; }();
    for (u32 lvl = log_leaves; lvl > cap_h; lvl--) {
        u64* cur = prev + 4 * cnt;
        const u64 out = cnt >> 1;
        if (coop_max && out <= GLP_COOP_TOP_NODES && out <= coop_max) {          // the rest of the tree in one launch
            const u32 n_levels = lvl - cap_h;
            if (h->small_mds) hipLaunchKernelGGL(glp_merkle_top_coop_kernel<true>, dim3(1), dim3(1024), 0, c->stream, prev, cnt, n_levels, k);
            else hipLaunchKernelGGL(glp_merkle_top_coop_kernel<false>, dim3(1), dim3(1024), 0, c->stream, prev, cnt, n_levels, k);
            GLP_HIPCHK(c, hipGetLastError());
            for (u32 l = 0; l < n_levels; l++) { prev += 4 * cnt; cnt >>= 1; }
            break;
        }
        cnt = out;
        if (coop_max && cnt <= coop_max) {
            blocks = (cnt * 16 + 255) / 256;
            if (h->small_mds) hipLaunchKernelGGL(glp_merkle_level_coop_kernel<true>, dim3((unsigned)blocks), b, 0, c->stream, prev, cur, cnt, k);
            else hipLaunchKernelGGL(glp_merkle_level_coop_kernel<false>, dim3((unsigned)blocks), b, 0, c->stream, prev, cur, cnt, k);
        } else {
            blocks = (cnt + 255) / 256;
            if (h->small_mds) hipLaunchKernelGGL(glp_merkle_level_kernel<true>, dim3((unsigned)blocks), b, 0, c->stream, prev, cur, cnt, k);
            else hipLaunchKernelGGL(glp_merkle_level_kernel<false>, dim3((unsigned)blocks), b, 0, c->stream, prev, cur, cnt, k);
        }
        GLP_HIPCHK(c, hipGetLastError());
        prev = cur;
    }
    if (h_cap) {
        GLP_HIPCHK(c, hipMemcpyAsync(h_cap, prev, (size_t)32 << cap_h, hipMemcpyDeviceToHost, c->stream));
        GLP_HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    return GLP_OK;
}

extern "C" int glp_merkle(glp_ctx* c, const uint64_t* d_leaves, uint32_t leaf_len, uint32_t log_leaves, uint32_t cap_h,
                          uint64_t* d_digests, uint64_t* h_cap) {
    return merkle_impl(c, d_leaves, leaf_len, false, leaf_len, log_leaves, cap_h, d_digests, h_cap);
}
extern "C" int glp_merkle_from_polys(glp_ctx* c, const uint64_t* d_polys, uint64_t poly_stride, uint32_t leaf_len,
                                     uint32_t log_leaves, uint32_t cap_h, uint64_t* d_digests, uint64_t* h_cap) {
    if (c && log_leaves <= 40 && poly_stride < (1ull << log_leaves)) { glp_set_err(c, "glp_merkle_from_polys: stride < leaves"); return GLP_E_INVALID; }
    return merkle_impl(c, d_polys, poly_stride, true, leaf_len, log_leaves, cap_h, d_digests, h_cap);
}

extern "C" int glp_fri_fold2(glp_ctx* c, const uint64_t* d_evals, uint64_t* d_out, uint32_t log_n, uint64_t shift,
                             const uint64_t* h_beta) {
    if (!c) return GLP_E_INVALID;
    GLP_BIND(c);
    if (!d_evals || !d_out || !h_beta || log_n == 0 || log_n > 32 || shift == 0 || shift >= GL_P || h_beta[0] >= GL_P || h_beta[1] >= GL_P) {
        glp_set_err(c, "glp_fri_fold2: bad argument");
        return GLP_E_INVALID;
    }
    const u64* lo = nullptr; const u64* hi = nullptr;
    int rc = glp_ntt_table(c, (int)log_n, 1, &lo, &hi);
    if (rc) return rc;
    const u64 half = 1ull << (log_n - 1);
    const u64 half_inv = gl_inv(2), cc = gl_inv(gl_mul(2, shift));
    hipLaunchKernelGGL(glp_fri_fold2_kernel<0>, dim3((unsigned)((half + 255) / 256)), dim3(256), 0, c->stream, d_evals, d_out, log_n,
                       half_inv, cc, gl_ext2{h_beta[0], h_beta[1]}, lo, hi);
    GLP_HIPCHK(c, hipGetLastError());
    return GLP_OK;
}

static int ensure_sha_tables(glp_ctx* c) {
    glp_hash_state* h = hs(c);
    if (h->d_k256) return GLP_OK;
    GLP_HIPCHK(c, hipMalloc((void**)&h->d_k256, sizeof(K256)));
    GLP_HIPCHK(c, hipMemcpy(h->d_k256, K256, sizeof(K256), hipMemcpyHostToDevice));
    GLP_HIPCHK(c, hipMalloc((void**)&h->d_k512, sizeof(K512)));
    GLP_HIPCHK(c, hipMemcpy(h->d_k512, K512, sizeof(K512), hipMemcpyHostToDevice));
    return GLP_OK;
}

extern "C" int glp_sha256_trace(glp_ctx* c, const uint8_t* d_blocks, uint64_t n_msgs, uint32_t bpm, uint32_t* d_digests,
                                uint32_t* d_trace) {
    if (!c) return GLP_E_INVALID;
    GLP_BIND(c);
    if ((!d_blocks || !d_digests) && n_msgs) { glp_set_err(c, "glp_sha256_trace: null buffer"); return GLP_E_INVALID; }
    if (n_msgs == 0) return GLP_OK;
    if (bpm == 0 || (n_msgs + 63) / 64 > 0x7fffffffull) { glp_set_err(c, "glp_sha256_trace: bad size"); return GLP_E_INVALID; }
    int rc = ensure_sha_tables(c);
    if (rc) return rc;
    hipLaunchKernelGGL(glp_sha256_trace_kernel<0>, dim3((unsigned)((n_msgs + 63) / 64)), dim3(64), 0, c->stream, d_blocks, n_msgs, bpm,
                       d_digests, d_trace, c->hash->d_k256);
    GLP_HIPCHK(c, hipGetLastError());
    return GLP_OK;
}

extern "C" int glp_sha512_trace(glp_ctx* c, const uint8_t* d_blocks, uint64_t n_msgs, uint32_t bpm, uint64_t* d_digests,
                                uint64_t* d_trace) {
    if (!c) return GLP_E_INVALID;
    GLP_BIND(c);
    if ((!d_blocks || !d_digests) && n_msgs) { glp_set_err(c, "glp_sha512_trace: null buffer"); return GLP_E_INVALID; }
    if (n_msgs == 0) return GLP_OK;
    if (bpm == 0 || (n_msgs + 63) / 64 > 0x7fffffffull) { glp_set_err(c, "glp_sha512_trace: bad size"); return GLP_E_INVALID; }
    int rc = ensure_sha_tables(c);
    if (rc) return rc;
    hipLaunchKernelGGL(glp_sha512_trace_kernel<0>, dim3((unsigned)((n_msgs + 63) / 64)), dim3(64), 0, c->stream, d_blocks, n_msgs, bpm,
                       d_digests, d_trace, c->hash->d_k512);
    GLP_HIPCHK(c, hipGetLastError());
    return GLP_OK;
}

// Tendermint simple Merkle root of n fixed-size leaves (RFC 6962: 0x00/0x01 prefixes, split at the
// largest power of two < n == pair adjacent nodes level by level, promoting an odd last node).
static int tm_merkle_root_impl(glp_ctx* c, const uint8_t* d_leaves, uint32_t leaf_len, const uint64_t* d_offsets, uint64_t n, uint8_t* h_root32);

extern "C" int glp_tm_merkle_root(glp_ctx* c, const uint8_t* d_leaves, uint32_t leaf_len, uint64_t n, uint8_t* h_root32) {
    if (!c) return GLP_E_INVALID;
    GLP_BIND(c);
    if (!h_root32 || (!d_leaves && n) || leaf_len == 0 || leaf_len > 118 || n > (1ull << 31)) { glp_set_err(c, "glp_tm_merkle_root: bad argument"); return GLP_E_INVALID; }
    return tm_merkle_root_impl(c, d_leaves, leaf_len, nullptr, n, h_root32);
}
// leaves of different lengths: leaf i = d_data[d_offsets[i] .. d_offsets[i+1]) (n + 1 non-decreasing offsets on the device)
extern "C" int glp_tm_merkle_root_var(glp_ctx* c, const uint8_t* d_data, uint64_t data_len, const uint64_t* d_offsets, uint64_t n,
                                      uint8_t* h_root32) {
    if (!c) return GLP_E_INVALID;
    GLP_BIND(c);
    if (!h_root32 || ((!d_data || !d_offsets) && n) || n > (1ull << 31)) { glp_set_err(c, "glp_tm_merkle_root_var: bad argument"); return GLP_E_INVALID; }
    if (n) {
        // the kernel trusts the offsets: check them here (n + 1 words, tiny next to the hashing) — non-decreasing and inside the data
        std::vector<u64> offs(n + 1);
        GLP_HIPCHK(c, hipMemcpyAsync(offs.data(), d_offsets, (n + 1) * 8, hipMemcpyDeviceToHost, c->stream));
        GLP_HIPCHK(c, hipStreamSynchronize(c->stream));
        for (u64 i = 0; i < n; i++)
            if (offs[i] > offs[i + 1]) { glp_set_err(c, "glp_tm_merkle_root_var: offsets decrease at leaf %llu", (unsigned long long)i); return GLP_E_INVALID; }
        if (offs[n] > data_len) { glp_set_err(c, "glp_tm_merkle_root_var: offsets reach %llu, past data_len %llu", (unsigned long long)offs[n], (unsigned long long)data_len); return GLP_E_INVALID; }
    }
    return tm_merkle_root_impl(c, d_data, 0, d_offsets, n, h_root32);
}

static int tm_merkle_root_impl(glp_ctx* c, const uint8_t* d_leaves, uint32_t leaf_len, const uint64_t* d_offsets, uint64_t n, uint8_t* h_root32) {
    if (n == 0) {   // SHA256("")
        static const uint8_t e[32] = {0xe3,0xb0,0xc4,0x42,0x98,0xfc,0x1c,0x14,0x9a,0xfb,0xf4,0xc8,0x99,0x6f,0xb9,0x24,0x27,0xae,0x41,0xe4,0x64,0x9b,0x93,0x4c,0xa4,0x95,0x99,0x1b,0x78,0x52,0xb8,0x55};
        memcpy(h_root32, e, 32);
        return GLP_OK;
    }
    int rc = ensure_sha_tables(c);
    if (rc) return rc;
    u32 *a = nullptr, *b = nullptr;
    GLP_HIPCHK(c, hipMalloc((void**)&a, n * 32));
    hipError_t e2 = hipMalloc((void**)&b, ((n + 1) / 2) * 32);
    if (e2 != hipSuccess) { hipFree(a); glp_set_err(c, "glp_tm_merkle_root: alloc"); return GLP_E_NOMEM; }
    if (d_offsets) hipLaunchKernelGGL(glp_tm_leaf_var_kernel<0>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, d_leaves, d_offsets, n, a, c->hash->d_k256);
    else hipLaunchKernelGGL(glp_tm_leaf_kernel<0>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, d_leaves, leaf_len, n, a, c->hash->d_k256);
    u64 cnt = n;
    u32 *src = a, *dst = b;
    while (cnt > 1) {
        const u64 nout = (cnt + 1) / 2;
        hipLaunchKernelGGL(glp_tm_inner_kernel<0>, dim3((unsigned)((nout + 255) / 256)), dim3(256), 0, c->stream, src, cnt, dst, c->hash->d_k256);
        cnt = nout;
        u32* t = src; src = dst; dst = t;
    }
    u32 hw[8];
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(hw, src, 32, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    hipFree(a); hipFree(b);
    if (e != hipSuccess) { glp_set_err(c, "glp_tm_merkle_root: %s", hipGetErrorString(e)); return GLP_E_HIP; }
    for (int k = 0; k < 8; k++) { h_root32[4*k] = hw[k] >> 24; h_root32[4*k+1] = hw[k] >> 16; h_root32[4*k+2] = hw[k] >> 8; h_root32[4*k+3] = hw[k]; }
    return GLP_OK;
}

// Ed25519 verification witness for a batch of signatures (row a10): pubs [n][32], sigs [n][64],
// msgs [n][msg_stride] with lens[n] (all device); out [n][GLP_ED25519_RECORD_WORDS] u64.
extern "C" int glp_ed25519_witness(glp_ctx* c, const uint8_t* d_pubs, const uint8_t* d_sigs, const uint8_t* d_msgs, uint32_t msg_stride,
                                   const uint32_t* d_lens, uint64_t n, uint64_t* d_out) {
    if (!c) return GLP_E_INVALID;
    GLP_BIND(c);
    if ((!d_pubs || !d_sigs || !d_msgs || !d_lens || !d_out) && n) { glp_set_err(c, "glp_ed25519_witness: null buffer"); return GLP_E_INVALID; }
    if (n == 0) return GLP_OK;
    if (msg_stride == 0 || n > (1ull << 30)) { glp_set_err(c, "glp_ed25519_witness: bad size"); return GLP_E_INVALID; }
    int rc = ensure_sha_tables(c);
    if (rc) return rc;
    hipLaunchKernelGGL(glp_ed25519_witness_kernel<0>, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, c->stream, d_pubs, d_sigs, d_msgs, msg_stride,
                       d_lens, n, c->hash->d_k512, d_out);
    GLP_HIPCHK(c, hipGetLastError());
    return GLP_OK;
}
