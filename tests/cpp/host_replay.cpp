// tests/cpp/host_replay.cpp — a circuit RECORDED by the Python builder and exported as raw arrays (recursion.WitnessProgram.export_raw), replayed by
// a compiled host through the C ABI only (g++, no hipcc, no Python, no torch): what a Rust prover process does at proving time.
//   commit the circuit (glp_plonk_setup_ex)  ->  witness: glp_witness_eval_mt on host threads, upload the variables, glp_gather_u64 with the
//   circuit's cell map, glp_poseidon_gate_fill_rows / glp_sha_gate_fill_rows  ->  glp_plonk_prove_ex  ->  glp_plonk_verify_ex
// usage: host_replay <dir> <poseidon constants: 384 u64 little-endian> [proof files...]   prints the key's first word, the public inputs and "OK".
// With proof files (and no inputs.bin) the recording is a VERIFIER circuit: its inputs are picked from the proofs by the recorded tags — a
// recursion node proved by a compiled host.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include <vector>
#include "glprover.h"

#define CHECK(x) do { int rc__ = (x); if (rc__ != GLP_OK) { std::printf("FAIL %s -> %d: %s\n", #x, rc__, ctx ? glp_last_error(ctx) : ""); return 1; } } while (0)

template <class T>
static std::vector<T> load(const std::string& dir, const std::string& name, size_t n) {
    std::vector<T> v(n);
    std::ifstream f(dir + "/" + name + ".bin", std::ios::binary);
    if (n && !f.read((char*)v.data(), (std::streamsize)(n * sizeof(T)))) { std::printf("FAIL: cannot read %s\n", name.c_str()); std::exit(1); }
    return v;
}

int main(int argc, char** argv) {
    if (argc < 3) { std::printf("usage: host_replay <dir> <poseidon_consts.bin>\n"); return 2; }
    const std::string dir = argv[1];
    std::map<std::string, uint64_t> meta, sizes;
    {
        std::ifstream f(dir + "/manifest.txt");
        std::string line;
        while (std::getline(f, line)) {
            std::istringstream ss(line);
            std::string k;
            ss >> k;
            if (k == "array") { std::string name; uint64_t n; ss >> name >> n; sizes[name] = n; }
            else { uint64_t v; ss >> v; meta[k] = v; }
        }
    }
    const uint32_t log_n = (uint32_t)meta["log_n"], W = (uint32_t)meta["n_wires"], R = (uint32_t)meta["n_routed"], n_pub = (uint32_t)meta["n_public"];
    const size_t n = (size_t)1 << log_n, n_values = meta["n_values"], n_inputs = meta["n_inputs"];
    auto consts = load<uint64_t>(dir, "consts", sizes["consts"]);
    auto sigma = load<uint64_t>(dir, "sigma", sizes["sigma"]);
    auto prog = load<uint64_t>(dir, "prog", sizes["prog"]);
    auto eq = load<uint64_t>(dir, "eq_pairs", sizes["eq_pairs"]);
    auto seg = load<uint64_t>(dir, "seg_bounds", sizes["seg_bounds"]);
    auto cell = load<uint32_t>(dir, "cell_index", sizes["cell_index"]);
    auto fixed = load<uint64_t>(dir, "fixed_values", sizes["fixed_values"]);
    auto pos_rows = load<uint32_t>(dir, "pos_rows", sizes["pos_rows"]);
    auto sha_rows = load<uint32_t>(dir, "sha_rows", sizes["sha_rows"]);
    auto sha_kinds = load<uint32_t>(dir, "sha_kinds", sizes["sha_kinds"]);
    auto pub_vars = load<uint64_t>(dir, "public_vars", sizes["public_vars"]);
    auto inputs = load<uint64_t>(dir, "inputs", sizes["inputs"]);
    if (inputs.empty() && sizes.count("input_tags")) {
        // a verifier circuit: the inputs are words of the proofs given on the command line (argv[3..]), picked by the recorded tags, after the
        // recorded facts about the other words (statement shape, the child circuit's key) have been checked
        auto tags = load<uint64_t>(dir, "input_tags", sizes["input_tags"]);
        auto wcc = load<uint64_t>(dir, "wc_const", sizes["wc_const"]);
        std::vector<std::vector<uint64_t>> proofs;
        for (int k = 3; k < argc; k++) {
            std::ifstream f(argv[k], std::ios::binary | std::ios::ate);
            const size_t bytes = (size_t)f.tellg();
            f.seekg(0);
            std::vector<uint64_t> w(bytes / 8);
            f.read((char*)w.data(), (std::streamsize)(w.size() * 8));
            proofs.push_back(std::move(w));
        }
        auto word = [&](uint64_t str, uint64_t pos, uint64_t& out) { if (str >= proofs.size() || pos >= proofs[str].size()) return false; out = proofs[str][pos]; return true; };
        for (size_t i = 0; i + 2 < wcc.size() + 1 && i < wcc.size(); i += 3) {
            uint64_t v;
            if (!word(wcc[i], wcc[i + 1], v) || v != wcc[i + 2]) { std::printf("FAIL: proof %llu is not what this circuit was built for (word %llu)\n", (unsigned long long)wcc[i], (unsigned long long)wcc[i + 1]); return 1; }
        }
        inputs.resize(tags.size() / 2);
        for (size_t i = 0; i < inputs.size(); i++)
            if (!word(tags[2 * i], tags[2 * i + 1], inputs[i])) { std::printf("FAIL: a proof is shorter than the circuit expects\n"); return 1; }
    }
    std::vector<uint64_t> pc(384);
    { std::ifstream f(argv[2], std::ios::binary); if (!f.read((char*)pc.data(), 384 * 8)) { std::printf("FAIL: constants\n"); return 1; } }
    if (inputs.size() != n_inputs || cell.size() != (size_t)W * n || sigma.size() != (size_t)R * n || consts.size() != meta["n_const"] * n) {
        std::printf("FAIL: manifest and arrays disagree\n");
        return 1;
    }

    glp_ctx* ctx = nullptr;
    CHECK(glp_create(&ctx, 0));
    CHECK(glp_set_poseidon_constants(ctx, pc.data(), 360, pc.data() + 360, pc.data() + 372));
    // ---- commit the circuit ----
    glp_circuit_shape sh;
    std::memset(&sh, 0, sizeof(sh));
    sh.log_n = log_n; sh.n_wires = W; sh.n_routed = R; sh.n_public = n_pub; sh.rate_bits = 3; sh.cap_height = (uint32_t)meta["cap_height"];
    sh.flags = (uint32_t)meta["flags"];
    uint64_t *d_consts = nullptr, *d_sigma = nullptr;
    CHECK(glp_alloc(ctx, (void**)&d_consts, consts.size() * 8));
    CHECK(glp_alloc(ctx, (void**)&d_sigma, sigma.size() * 8));
    CHECK(glp_h2d(ctx, d_consts, consts.data(), consts.size() * 8));
    CHECK(glp_h2d(ctx, d_sigma, sigma.data(), sigma.size() * 8));
    glp_plonk_circuit* ck = nullptr;
    CHECK(glp_plonk_setup_ex(ctx, &sh, d_consts, d_sigma, &ck));
    size_t capw = 0;
    CHECK(glp_plonk_circuit_cap(ck, nullptr, &capw));
    std::vector<uint64_t> key(capw);
    CHECK(glp_plonk_circuit_cap(ck, key.data(), &capw));
    // ---- witness: evaluate on host threads, place on the device ----
    std::vector<uint64_t> values(n_values + fixed.size(), 0);
    size_t bad = 0;
    const size_t n_seg = seg.size() > 1 ? seg.size() - 1 : 0;
    CHECK(glp_witness_eval_mt(pc.data(), pc.data() + 360, pc.data() + 372, prog.data(), prog.size(), inputs.data(), inputs.size(), values.data(),
                              n_values, eq.empty() ? nullptr : eq.data(), eq.size() / 2, &bad, n_seg ? seg.data() : nullptr, n_seg, 8));
    for (size_t i = 0; i < fixed.size(); i++) values[n_values + i] = fixed[i];
    uint64_t *d_vals = nullptr, *d_wires = nullptr;
    uint32_t *d_cell = nullptr, *d_rows = nullptr, *d_srows = nullptr, *d_kinds = nullptr;
    CHECK(glp_alloc(ctx, (void**)&d_vals, values.size() * 8));
    CHECK(glp_alloc(ctx, (void**)&d_wires, (size_t)W * n * 8));
    CHECK(glp_alloc(ctx, (void**)&d_cell, cell.size() * 4));
    CHECK(glp_h2d(ctx, d_vals, values.data(), values.size() * 8));
    CHECK(glp_h2d(ctx, d_cell, cell.data(), cell.size() * 4));
    CHECK(glp_gather_u64(ctx, d_wires, d_vals, values.size(), d_cell, (size_t)W * n));
    if (!pos_rows.empty()) {
        CHECK(glp_alloc(ctx, (void**)&d_rows, pos_rows.size() * 4));
        CHECK(glp_h2d(ctx, d_rows, pos_rows.data(), pos_rows.size() * 4));
        CHECK(glp_poseidon_gate_fill_rows(ctx, d_wires, log_n, W, d_rows, (uint32_t)pos_rows.size()));
    }
    if (!sha_rows.empty()) {
        CHECK(glp_alloc(ctx, (void**)&d_srows, sha_rows.size() * 4));
        CHECK(glp_alloc(ctx, (void**)&d_kinds, sha_kinds.size() * 4));
        CHECK(glp_h2d(ctx, d_srows, sha_rows.data(), sha_rows.size() * 4));
        CHECK(glp_h2d(ctx, d_kinds, sha_kinds.data(), sha_kinds.size() * 4));
        CHECK(glp_sha_gate_fill_rows(ctx, d_wires, log_n, W, d_srows, d_kinds, (uint32_t)sha_rows.size()));
    }
    CHECK(glp_sync(ctx));
    // ---- prove and verify ----
    std::vector<uint64_t> pub(n_pub);
    for (uint32_t i = 0; i < n_pub; i++) pub[i] = values[pub_vars[i]];
    uint8_t* proof = nullptr;
    size_t len = 0;
    CHECK(glp_plonk_prove_ex(ctx, ck, d_wires, pub.data(), 10, 6, &proof, &len));
    if (glp_plonk_verify_ex(ctx, proof, len, key.data(), key.size(), pub.data(), n_pub, 10, 6) != GLP_OK) { std::printf("FAIL: proof rejected: %s\n", glp_last_error(ctx)); return 1; }
    std::vector<uint64_t> other = pub;
    if (!other.empty()) {
        other[0] ^= 1;
        if (glp_plonk_verify_ex(ctx, proof, len, key.data(), key.size(), other.data(), n_pub, 10, 6) != GLP_E_REJECT) { std::printf("FAIL: wrong statement accepted\n"); return 1; }
    }
    // a witness that breaks a row: an input outside its range is refused by the evaluator
    if (!inputs.empty() && argc == 3) {
        std::vector<uint64_t> bad_in = inputs;
        bad_in[0] = 1ull << 40;
        std::vector<uint64_t> v2(n_values, 0);
        const int rc = glp_witness_eval_mt(pc.data(), pc.data() + 360, pc.data() + 372, prog.data(), prog.size(), bad_in.data(), bad_in.size(), v2.data(), n_values,
                                           eq.empty() ? nullptr : eq.data(), eq.size() / 2, &bad, n_seg ? seg.data() : nullptr, n_seg, 8);
        std::printf("out-of-range input -> %d\n", rc);
    }
    std::printf("key0 %llu\n", (unsigned long long)key[0]);
    std::printf("public");
    for (uint32_t i = 0; i < n_pub; i++) std::printf(" %llu", (unsigned long long)pub[i]);
    std::printf("\nproof_bytes %zu\n", len);
    glp_free_host(proof);
    glp_plonk_free(ck);
    glp_destroy(ctx);
    std::printf("OK\n");
    return 0;
}
