// tests/cpp/host_mapreduce.cpp — the MapReduce path of SURVEY.md §8e driven through the C ABI ONLY, by a compiled host (g++, no hipcc, no
// Python, no torch): what a Rust host would do on each rank.  Circuit setup -> leaf proofs with public inputs -> RCCL communicator owned by
// the ctx -> one all-gather of the packed leaf records -> native verification of every gathered leaf against (circuit key, its public inputs)
// -> verdict all-reduce -> leaf digests.  One rank here (one GPU per box); with N ranks every process runs the same calls with its own
// rank / device and the id bytes from rank 0.  Prints "OK" on success.  Run by tests/test_gpu_abi_host.py.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>
#include "glprover.h"

#define CHECK(x) do { int rc__ = (x); if (rc__ != GLP_OK) { std::printf("FAIL %s -> %d: %s\n", #x, rc__, ctx ? glp_last_error(ctx) : ""); return 1; } } while (0)

static const uint64_t P = 0xFFFFFFFF00000001ull;
static uint64_t splitmix(uint64_t& s) {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static uint64_t mulmod(uint64_t a, uint64_t b) { return (uint64_t)(((unsigned __int128)a * b) % P); }
static uint64_t powmod(uint64_t a, uint64_t e) { uint64_t r = 1; while (e) { if (e & 1) r = mulmod(r, a); a = mulmod(a, a); e >>= 1; } return r; }

int main() {
    glp_ctx* ctx = nullptr;
    CHECK(glp_create(&ctx, 0));
    uint64_t seed = 7;
    std::vector<uint64_t> rc(360), circ = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20}, diag(12, 0);
    diag[0] = 8;
    for (auto& v : rc) v = splitmix(seed) % P;
    CHECK(glp_set_poseidon_constants(ctx, rc.data(), rc.size(), circ.data(), diag.data()));

    // a leaf circuit: 2^10 rows, 16 routed wires, gate w = c0*x*y + c1*z + c2 on every group, no copy constraints, 3 public inputs
    const uint32_t log_n = 10, W = 16, n_pub = 3, n_leaves = 4, nq = 10, pw = 6;
    const size_t n = 1u << log_n;
    glp_circuit_shape sh;
    std::memset(&sh, 0, sizeof(sh));
    sh.log_n = log_n; sh.n_wires = W; sh.n_routed = W; sh.n_public = n_pub; sh.rate_bits = 3; sh.cap_height = 4;
    std::vector<uint64_t> consts(GLP_PLONK_NCONST * n, 0), sigma(W * n);
    for (size_t i = 0; i < n; i++) {
        consts[0 * n + i] = 1;
        consts[1 * n + i] = splitmix(seed) % P; consts[2 * n + i] = splitmix(seed) % P; consts[3 * n + i] = splitmix(seed) % P;
        consts[4 * n + i] = i < n_pub ? 1 : 0;
    }
    const uint64_t w_n = powmod(7, (P - 1) >> log_n);
    for (uint32_t j = 0; j < W; j++) { uint64_t kj = powmod(7, j), x = 1; for (size_t i = 0; i < n; i++) { sigma[j * n + i] = mulmod(kj, x); x = mulmod(x, w_n); } }
    uint64_t *d_consts = nullptr, *d_sigma = nullptr, *d_wires = nullptr;
    CHECK(glp_alloc(ctx, (void**)&d_consts, consts.size() * 8));
    CHECK(glp_alloc(ctx, (void**)&d_sigma, sigma.size() * 8));
    CHECK(glp_alloc(ctx, (void**)&d_wires, (size_t)W * n * 8));
    CHECK(glp_h2d(ctx, d_consts, consts.data(), consts.size() * 8));
    CHECK(glp_h2d(ctx, d_sigma, sigma.data(), sigma.size() * 8));
    glp_plonk_circuit* ck = nullptr;
    CHECK(glp_plonk_setup_ex(ctx, &sh, d_consts, d_sigma, &ck));
    size_t capw = 0;
    CHECK(glp_plonk_circuit_cap(ck, nullptr, &capw));
    std::vector<uint64_t> key(capw);
    CHECK(glp_plonk_circuit_cap(ck, key.data(), &capw));

    // Map: this rank's leaves (every leaf here: one rank), each with its own witness and public inputs
    std::vector<std::vector<uint8_t>> proofs(n_leaves);
    std::vector<std::vector<uint64_t>> publics(n_leaves);
    size_t max_len = 0;
    for (uint32_t l = 0; l < n_leaves; l++) {
        std::vector<uint64_t> wires((size_t)W * n);
        for (size_t i = 0; i < n; i++)
            for (uint32_t g = 0; g < W / 4; g++) {
                uint64_t x = splitmix(seed) % P, y = splitmix(seed) % P, z = splitmix(seed) % P;
                wires[(4 * g + 0) * n + i] = x; wires[(4 * g + 1) * n + i] = y; wires[(4 * g + 2) * n + i] = z;
                uint64_t w = (uint64_t)(((unsigned __int128)mulmod(consts[n + i], mulmod(x, y)) + mulmod(consts[2 * n + i], z) + consts[3 * n + i]) % P);
                wires[(4 * g + 3) * n + i] = w;
            }
        for (uint32_t i = 0; i < n_pub; i++) publics[l].push_back(wires[i]);          // wire 0 of row i
        CHECK(glp_h2d(ctx, d_wires, wires.data(), wires.size() * 8));
        uint8_t* pr = nullptr; size_t len = 0;
        CHECK(glp_plonk_prove_ex(ctx, ck, d_wires, publics[l].data(), nq, pw, &pr, &len));
        proofs[l].assign(pr, pr + len);
        glp_free_host(pr);
        if (len > max_len) max_len = len;
    }
    // exchange: records of (leaf index, length, zero-padded proof), one block per rank, ONE all-gather over the ctx's RCCL communicator
    uint8_t id[GLP_COMM_ID_BYTES];
    CHECK(glp_comm_unique_id(id));
    CHECK(glp_comm_init(ctx, id, 0, 1));
    const size_t rec = 16 + ((max_len + 63) & ~(size_t)63);
    std::vector<uint8_t> mine(n_leaves * rec, 0), all(n_leaves * rec, 0xEE);
    for (uint32_t l = 0; l < n_leaves; l++) {
        uint64_t hdr[2] = {l, proofs[l].size()};
        std::memcpy(&mine[l * rec], hdr, 16);
        std::memcpy(&mine[l * rec + 16], proofs[l].data(), proofs[l].size());
    }
    CHECK(glp_allgather_proofs(ctx, mine.data(), mine.size(), all.data()));
    // Reduce (native form): verify every gathered leaf against the circuit key AND its public inputs; digests for the aggregation tree
    uint64_t verdict = 1;
    for (uint32_t l = 0; l < n_leaves; l++) {
        uint64_t hdr[2];
        std::memcpy(hdr, &all[l * rec], 16);
        if (hdr[0] != l || hdr[1] != proofs[l].size()) { std::printf("FAIL: record %u garbled\n", l); return 1; }
        std::vector<uint64_t> words((hdr[1] + 7) / 8);
        std::memcpy(words.data(), &all[l * rec + 16], hdr[1]);
        const uint8_t* pb = (const uint8_t*)words.data();
        if (glp_plonk_verify_ex(ctx, pb, hdr[1], key.data(), key.size(), publics[l].data(), n_pub, nq, pw) != GLP_OK) verdict = 0;
        std::vector<uint64_t> other = publics[l];
        other[1] ^= 1;
        if (glp_plonk_verify_ex(ctx, pb, hdr[1], key.data(), key.size(), other.data(), n_pub, nq, pw) != GLP_E_REJECT) { std::printf("FAIL: wrong statement accepted\n"); return 1; }
        size_t np = n_pub; std::vector<uint64_t> got(n_pub);
        CHECK(glp_plonk_proof_public_inputs(pb, hdr[1], got.data(), &np));
        if (np != n_pub || got != publics[l]) { std::printf("FAIL: public inputs of leaf %u\n", l); return 1; }
        uint64_t dg[4];
        CHECK(glp_plonk_proof_digest(ctx, pb, hdr[1], dg));
        if (!(dg[0] | dg[1] | dg[2] | dg[3])) { std::printf("FAIL: empty digest\n"); return 1; }
    }
    CHECK(glp_allreduce_min_u64(ctx, &verdict, 1));
    if (verdict != 1) { std::printf("FAIL: a leaf proof did not verify\n"); return 1; }
    CHECK(glp_comm_destroy(ctx));
    glp_plonk_free(ck);
    CHECK(glp_free(ctx, d_consts)); CHECK(glp_free(ctx, d_sigma)); CHECK(glp_free(ctx, d_wires));
    glp_destroy(ctx);
    std::printf("OK\n");
    return 0;
}
