// tests/cpp/host_roundtrip.cpp — a compiled host program driving the hot path through the C ABI
// only (what a Rust/C++ prover would do): allocate, upload, forward + inverse NTT, coset LDE,
// Poseidon Merkle commitment, download, check the round trip.  Built with g++ (no hipcc) and run
// by tests/test_gpu_abi_host.py on the GPU box.  Prints "OK" on success.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>
#include "glprover.h"

#define CHECK(x) do { int rc__ = (x); if (rc__ != GLP_OK) { std::printf("FAIL %s -> %d: %s\n", #x, rc__, ctx ? glp_last_error(ctx) : ""); return 1; } } while (0)

static uint64_t splitmix(uint64_t& s) {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

int main() {
    const uint64_t P = 0xFFFFFFFF00000001ull;
    glp_ctx* ctx = nullptr;
    CHECK(glp_create(&ctx, 0));
    const uint32_t log_n = 16, batch = 5, rate_bits = 3, cap_h = 4;
    const size_t n = 1u << log_n, N = n << rate_bits;
    std::vector<uint64_t> x(batch * n), y(batch * n);
    uint64_t seed = 42;
    for (auto& v : x) v = splitmix(seed) % P;
    uint64_t *d = nullptr, *lde = nullptr, *dig = nullptr;
    CHECK(glp_alloc(ctx, (void**)&d, x.size() * 8));
    CHECK(glp_h2d(ctx, d, x.data(), x.size() * 8));
    CHECK(glp_ntt(ctx, d, log_n, batch, 0));
    CHECK(glp_d2h(ctx, y.data(), d, y.size() * 8));
    if (std::memcmp(x.data(), y.data(), x.size() * 8) == 0) { std::printf("FAIL: forward NTT left the data unchanged\n"); return 1; }
    CHECK(glp_ntt(ctx, d, log_n, batch, 1));
    CHECK(glp_d2h(ctx, y.data(), d, y.size() * 8));
    if (std::memcmp(x.data(), y.data(), x.size() * 8) != 0) { std::printf("FAIL: ifft(fft(x)) != x\n"); return 1; }
    // commitment: LDE (bit-reversed) + Merkle with injected toy constants
    std::vector<uint64_t> rc(360), circ = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20}, diag(12, 0);
    diag[0] = 8;
    for (auto& v : rc) v = splitmix(seed) % P;
    CHECK(glp_set_poseidon_constants(ctx, rc.data(), rc.size(), circ.data(), diag.data()));
    CHECK(glp_alloc(ctx, (void**)&lde, batch * N * 8));
    CHECK(glp_lde_coset(ctx, d, lde, log_n, rate_bits, batch, 7, GLP_NTT_BITREV));
    const size_t nd = 4 * ((2ull << (log_n + rate_bits)) - (1ull << cap_h));
    CHECK(glp_alloc(ctx, (void**)&dig, nd * 8));
    std::vector<uint64_t> cap(4u << cap_h), cap2(4u << cap_h);
    CHECK(glp_merkle_from_polys(ctx, lde, N, batch, log_n + rate_bits, cap_h, dig, cap.data()));
    CHECK(glp_merkle_from_polys(ctx, lde, N, batch, log_n + rate_bits, cap_h, dig, cap2.data()));
    if (cap != cap2) { std::printf("FAIL: Merkle cap not deterministic\n"); return 1; }
    uint64_t nz = 0;
    for (auto v : cap) nz |= v;
    if (!nz) { std::printf("FAIL: empty cap\n"); return 1; }
    // the MapReduce exchange behind the C ABI (RCCL): a one-rank communicator on this GPU — id, init, all-gather of a padded
    // block, verdict all-reduce, destroy.  (N > 1 ranks run the same calls, one process per GPU.)
    {
        uint8_t id[GLP_COMM_ID_BYTES];
        CHECK(glp_comm_unique_id(id));
        if (glp_allgather_proofs(ctx, id, 8, id) != GLP_E_STATE) { std::printf("FAIL: all-gather without a communicator accepted\n"); return 1; }
        CHECK(glp_comm_init(ctx, id, 0, 1));
        int rk = -1, nr = -1;
        CHECK(glp_comm_rank(ctx, &rk, &nr));
        if (rk != 0 || nr != 1) { std::printf("FAIL: comm rank %d of %d\n", rk, nr); return 1; }
        std::vector<uint8_t> mine(3 * (16 + 1000)), all(mine.size(), 0xEE);
        for (size_t i = 0; i < mine.size(); i++) mine[i] = (uint8_t)(splitmix(seed) >> 56);
        CHECK(glp_allgather_proofs(ctx, mine.data(), mine.size(), all.data()));
        if (mine != all) { std::printf("FAIL: one-rank all-gather changed the block\n"); return 1; }
        uint64_t verdict[2] = {1, 7};
        CHECK(glp_allreduce_min_u64(ctx, verdict, 2));
        if (verdict[0] != 1 || verdict[1] != 7) { std::printf("FAIL: one-rank all-reduce\n"); return 1; }
        if (glp_comm_init(ctx, id, 0, 1) != GLP_E_STATE) { std::printf("FAIL: second communicator on one ctx accepted\n"); return 1; }
        CHECK(glp_comm_destroy(ctx));
    }
    // error behaviour: bad arguments are reported, not fatal
    if (glp_ntt(ctx, d, 40, 1, 0) != GLP_E_INVALID) { std::printf("FAIL: log_n=40 accepted\n"); return 1; }
    CHECK(glp_free(ctx, d));
    CHECK(glp_free(ctx, lde));
    CHECK(glp_free(ctx, dig));
    glp_destroy(ctx);
    std::printf("OK\n");
    return 0;
}
