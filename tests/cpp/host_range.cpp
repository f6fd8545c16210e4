// tests/cpp/host_range.cpp — a whole MapReduce of proofs from a compiled host (g++, C ABI only: no hipcc, no Python, no torch): the per-rank loop of
// data_commitment_mr.DataCommitmentMapReduce.prove_range_distributed at nranks = 1, as a Rust prover process would run it from circuit artifacts.
//   Python is the offline circuit compiler: it records the leaf circuit and one node circuit per level and exports them as raw arrays
//   (recursion.WitnessProgram.export_raw).  This host then
//     Map      proves every leaf from its input vector (glp_witness_eval_mt on host threads -> upload -> glp_gather_u64 -> row fillers -> prove),
//     fold     level 1 locally: each node reads `fan_in` child proofs (inputs picked by the recorded tags after the recorded word checks),
//     exchange the level-1 node proofs through the ctx's RCCL communicator (glp_allgather_proofs; one rank here),
//     root     folds the gathered node proofs with the remaining levels' recordings,
//   verifies the root proof against the last recording's key and prints key0, the public inputs and OK; the root proof goes to <out file>.
// usage: host_range <poseidon consts: 384 u64> <queries> <pow bits> <fan_in> <leaf dir> <n leaves> <leaf inputs: n leaves * n_inputs u64>
//                   <out proof file> <node dir level 1> [<node dir level 2> ...]
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include <vector>
#include "glprover.h"

static glp_ctx* ctx = nullptr;
#define CHECK(x) do { int rc__ = (x); if (rc__ != GLP_OK) { std::printf("FAIL %s -> %d: %s\n", #x, rc__, ctx ? glp_last_error(ctx) : ""); std::exit(1); } } while (0)

template <class T>
static std::vector<T> load(const std::string& dir, const std::string& name, size_t n) {
    std::vector<T> v(n);
    std::ifstream f(dir + "/" + name + ".bin", std::ios::binary);
    if (n && !f.read((char*)v.data(), (std::streamsize)(n * sizeof(T)))) { std::printf("FAIL: cannot read %s/%s\n", dir.c_str(), name.c_str()); std::exit(1); }
    return v;
}

static std::vector<uint64_t> PC(384);
static uint32_t NQ = 0, PW = 0;

// one recorded circuit: arrays, committed circuit, resident device buffers
struct Recorded {
    std::map<std::string, uint64_t> meta, sizes;
    uint32_t log_n = 0, W = 0, R = 0, n_pub = 0;
    size_t n = 0, n_values = 0, n_inputs = 0;
    std::vector<uint64_t> prog, eq, seg, fixed, pub_vars, tags, wcc, key;
    std::vector<uint32_t> pos_rows, sha_rows;
    glp_plonk_circuit* ck = nullptr;
    uint64_t *d_vals = nullptr, *d_wires = nullptr;
    uint32_t *d_cell = nullptr, *d_rows = nullptr, *d_srows = nullptr, *d_kinds = nullptr;

    void open(const std::string& dir) {
        std::ifstream f(dir + "/manifest.txt");
        std::string line;
        while (std::getline(f, line)) {
            std::istringstream ss(line);
            std::string k;
            ss >> k;
            if (k == "array") { std::string name; uint64_t cnt; ss >> name >> cnt; sizes[name] = cnt; }
            else { uint64_t v; ss >> v; meta[k] = v; }
        }
        log_n = (uint32_t)meta["log_n"]; W = (uint32_t)meta["n_wires"]; R = (uint32_t)meta["n_routed"]; n_pub = (uint32_t)meta["n_public"];
        n = (size_t)1 << log_n; n_values = meta["n_values"]; n_inputs = meta["n_inputs"];
        auto consts = load<uint64_t>(dir, "consts", sizes["consts"]);
        auto sigma = load<uint64_t>(dir, "sigma", sizes["sigma"]);
        auto cell = load<uint32_t>(dir, "cell_index", sizes["cell_index"]);
        auto sha_kinds = load<uint32_t>(dir, "sha_kinds", sizes["sha_kinds"]);
        prog = load<uint64_t>(dir, "prog", sizes["prog"]);
        eq = load<uint64_t>(dir, "eq_pairs", sizes["eq_pairs"]);
        seg = load<uint64_t>(dir, "seg_bounds", sizes["seg_bounds"]);
        fixed = load<uint64_t>(dir, "fixed_values", sizes["fixed_values"]);
        pos_rows = load<uint32_t>(dir, "pos_rows", sizes["pos_rows"]);
        sha_rows = load<uint32_t>(dir, "sha_rows", sizes["sha_rows"]);
        pub_vars = load<uint64_t>(dir, "public_vars", sizes["public_vars"]);
        if (sizes.count("input_tags")) { tags = load<uint64_t>(dir, "input_tags", sizes["input_tags"]); wcc = load<uint64_t>(dir, "wc_const", sizes["wc_const"]); }
        if (cell.size() != (size_t)W * n || sigma.size() != (size_t)R * n || consts.size() != meta["n_const"] * n) { std::printf("FAIL: manifest and arrays disagree in %s\n", dir.c_str()); std::exit(1); }
        glp_circuit_shape sh;
        std::memset(&sh, 0, sizeof(sh));
        sh.log_n = log_n; sh.n_wires = W; sh.n_routed = R; sh.n_public = n_pub; sh.rate_bits = 3; sh.cap_height = (uint32_t)meta["cap_height"]; sh.flags = (uint32_t)meta["flags"];
        uint64_t *d_consts = nullptr, *d_sigma = nullptr;
        CHECK(glp_alloc(ctx, (void**)&d_consts, consts.size() * 8));
        CHECK(glp_alloc(ctx, (void**)&d_sigma, sigma.size() * 8));
        CHECK(glp_h2d(ctx, d_consts, consts.data(), consts.size() * 8));
        CHECK(glp_h2d(ctx, d_sigma, sigma.data(), sigma.size() * 8));
        CHECK(glp_plonk_setup_ex(ctx, &sh, d_consts, d_sigma, &ck));
        CHECK(glp_free(ctx, d_consts));
        CHECK(glp_free(ctx, d_sigma));
        size_t capw = 0;
        CHECK(glp_plonk_circuit_cap(ck, nullptr, &capw));
        key.resize(capw);
        CHECK(glp_plonk_circuit_cap(ck, key.data(), &capw));
        CHECK(glp_alloc(ctx, (void**)&d_vals, (n_values + fixed.size()) * 8));
        CHECK(glp_alloc(ctx, (void**)&d_wires, (size_t)W * n * 8));
        CHECK(glp_alloc(ctx, (void**)&d_cell, cell.size() * 4));
        CHECK(glp_h2d(ctx, d_cell, cell.data(), cell.size() * 4));
        if (!pos_rows.empty()) { CHECK(glp_alloc(ctx, (void**)&d_rows, pos_rows.size() * 4)); CHECK(glp_h2d(ctx, d_rows, pos_rows.data(), pos_rows.size() * 4)); }
        if (!sha_rows.empty()) {
            CHECK(glp_alloc(ctx, (void**)&d_srows, sha_rows.size() * 4));
            CHECK(glp_alloc(ctx, (void**)&d_kinds, sha_kinds.size() * 4));
            CHECK(glp_h2d(ctx, d_srows, sha_rows.data(), sha_rows.size() * 4));
            CHECK(glp_h2d(ctx, d_kinds, sha_kinds.data(), sha_kinds.size() * 4));
        }
    }

    // witness from an input vector, proof with its public inputs
    std::vector<uint8_t> prove(const uint64_t* inputs, std::vector<uint64_t>* pub_out) {
        std::vector<uint64_t> values(n_values + fixed.size(), 0);
        size_t bad = 0;
        const size_t n_seg = seg.size() > 1 ? seg.size() - 1 : 0;
        const int rc = glp_witness_eval_mt(PC.data(), PC.data() + 360, PC.data() + 372, prog.data(), prog.size(), inputs, n_inputs, values.data(), n_values,
                                           eq.empty() ? nullptr : eq.data(), eq.size() / 2, &bad, n_seg ? seg.data() : nullptr, n_seg, 8);
        if (rc != GLP_OK) { std::printf("FAIL: the inputs do not satisfy the circuit (witness evaluator -> %d)\n", rc); std::exit(1); }
        for (size_t i = 0; i < fixed.size(); i++) values[n_values + i] = fixed[i];
        CHECK(glp_h2d(ctx, d_vals, values.data(), values.size() * 8));
        CHECK(glp_gather_u64(ctx, d_wires, d_vals, values.size(), d_cell, (size_t)W * n));
        if (!pos_rows.empty()) CHECK(glp_poseidon_gate_fill_rows(ctx, d_wires, log_n, W, d_rows, (uint32_t)pos_rows.size()));
        if (!sha_rows.empty()) CHECK(glp_sha_gate_fill_rows(ctx, d_wires, log_n, W, d_srows, d_kinds, (uint32_t)sha_rows.size()));
        std::vector<uint64_t> pub(n_pub);
        for (uint32_t i = 0; i < n_pub; i++) pub[i] = values[pub_vars[i]];
        uint8_t* proof = nullptr;
        size_t len = 0;
        CHECK(glp_plonk_prove_ex(ctx, ck, d_wires, pub.data(), NQ, PW, &proof, &len));
        std::vector<uint8_t> out(proof, proof + len);
        glp_free_host(proof);
        if (pub_out) *pub_out = pub;
        return out;
    }

    // a verifier circuit: inputs = words of the child proofs, picked by the recorded tags after the recorded word checks
    std::vector<uint8_t> prove_children(const std::vector<std::vector<uint8_t>>& kids, std::vector<uint64_t>* pub_out) {
        auto word = [&](uint64_t k, uint64_t pos, uint64_t& out) {
            if (k >= kids.size() || (pos + 1) * 8 > kids[k].size()) return false;
            std::memcpy(&out, kids[k].data() + pos * 8, 8);
            return true;
        };
        for (size_t i = 0; i + 2 < wcc.size() + 1 && i < wcc.size(); i += 3) {
            uint64_t v;
            if (!word(wcc[i], wcc[i + 1], v) || v != wcc[i + 2]) { std::printf("FAIL: child %llu is not what this node circuit was built for\n", (unsigned long long)wcc[i]); std::exit(1); }
        }
        std::vector<uint64_t> inputs(tags.size() / 2);
        if (inputs.size() != n_inputs) { std::printf("FAIL: tags and n_inputs disagree\n"); std::exit(1); }
        for (size_t i = 0; i < inputs.size(); i++)
            if (!word(tags[2 * i], tags[2 * i + 1], inputs[i])) { std::printf("FAIL: a child proof is shorter than the node circuit expects\n"); std::exit(1); }
        return prove(inputs.data(), pub_out);
    }
};

int main(int argc, char** argv) {
    if (argc < 10) { std::printf("usage: host_range <consts> <queries> <pow bits> <fan_in> <leaf dir> <n leaves> <leaf inputs> <out proof> <node dir>...\n"); return 2; }
    { std::ifstream f(argv[1], std::ios::binary); if (!f.read((char*)PC.data(), 384 * 8)) { std::printf("FAIL: constants\n"); return 1; } }
    NQ = (uint32_t)std::atoi(argv[2]); PW = (uint32_t)std::atoi(argv[3]);
    const size_t fan = (size_t)std::atoi(argv[4]), n_leaves = (size_t)std::atoi(argv[6]);
    CHECK(glp_create(&ctx, 0));
    CHECK(glp_set_poseidon_constants(ctx, PC.data(), 360, PC.data() + 360, PC.data() + 372));
    // ---- Map ----
    Recorded leaf;
    leaf.open(argv[5]);
    std::vector<uint64_t> all_inputs(n_leaves * leaf.n_inputs);
    { std::ifstream f(argv[7], std::ios::binary); if (!f.read((char*)all_inputs.data(), (std::streamsize)(all_inputs.size() * 8))) { std::printf("FAIL: leaf inputs\n"); return 1; } }
    std::vector<std::vector<uint8_t>> cur;
    for (size_t l = 0; l < n_leaves; l++) cur.push_back(leaf.prove(all_inputs.data() + l * leaf.n_inputs, nullptr));
    std::printf("leaves %zu proof_bytes %zu\n", cur.size(), cur[0].size());
    // ---- fold, exchange after the first (local) level, fold on ----
    std::vector<uint64_t> pub, key;
    for (int a = 9; a < argc; a++) {
        if (cur.size() % fan && cur.size() > fan) { std::printf("FAIL: %zu proofs do not fold by %zu\n", cur.size(), fan); return 1; }
        Recorded node;
        node.open(argv[a]);
        const size_t take = cur.size() < fan ? cur.size() : fan;
        std::vector<std::vector<uint8_t>> next;
        for (size_t g = 0; g < cur.size(); g += take)
            next.push_back(node.prove_children(std::vector<std::vector<uint8_t>>(cur.begin() + (long)g, cur.begin() + (long)(g + take)), &pub));
        key = node.key;
        std::printf("level %d nodes %zu rows %zu\n", a - 8, next.size(), node.n);
        if (a == 9) {
            // the one exchange of the job: (length, zero-padded proof) records through the ctx's RCCL communicator (a single rank here)
            uint8_t id[GLP_COMM_ID_BYTES];
            CHECK(glp_comm_unique_id(id));
            CHECK(glp_comm_init(ctx, id, 0, 1));
            size_t max_len = 0;
            for (auto& p : next) if (p.size() > max_len) max_len = p.size();
            const size_t rec = 16 + ((max_len + 63) & ~(size_t)63);
            std::vector<uint8_t> mine(next.size() * rec, 0), all(next.size() * rec, 0xEE);
            for (size_t i = 0; i < next.size(); i++) {
                uint64_t hdr[2] = {i, next[i].size()};
                std::memcpy(&mine[i * rec], hdr, 16);
                std::memcpy(&mine[i * rec + 16], next[i].data(), next[i].size());
            }
            CHECK(glp_allgather_proofs(ctx, mine.data(), mine.size(), all.data()));
            for (size_t i = 0; i < next.size(); i++) {
                uint64_t hdr[2];
                std::memcpy(hdr, &all[i * rec], 16);
                if (hdr[0] != i || hdr[1] != next[i].size()) { std::printf("FAIL: gathered record %zu garbled\n", i); return 1; }
                next[i].assign(all.begin() + (long)(i * rec + 16), all.begin() + (long)(i * rec + 16 + hdr[1]));
            }
            CHECK(glp_comm_destroy(ctx));
            std::printf("exchanged %zu node proofs\n", next.size());
        }
        cur = std::move(next);
        // (the recordings of finished levels stay committed until exit: a few hundred MB at test sizes)
    }
    if (cur.size() != 1) { std::printf("FAIL: %zu proofs left after the last level\n", cur.size()); return 1; }
    if (glp_plonk_verify_ex(ctx, cur[0].data(), cur[0].size(), key.data(), key.size(), pub.data(), pub.size(), NQ, PW) != GLP_OK) { std::printf("FAIL: root rejected: %s\n", glp_last_error(ctx)); return 1; }
    std::vector<uint64_t> other = pub;
    other[0] ^= 1;
    if (glp_plonk_verify_ex(ctx, cur[0].data(), cur[0].size(), key.data(), key.size(), other.data(), other.size(), NQ, PW) != GLP_E_REJECT) { std::printf("FAIL: wrong statement accepted\n"); return 1; }
    { std::ofstream f(argv[8], std::ios::binary); f.write((const char*)cur[0].data(), (std::streamsize)cur[0].size()); }
    std::printf("key0 %llu\n", (unsigned long long)key[0]);
    std::printf("public");
    for (uint64_t v : pub) std::printf(" %llu", (unsigned long long)v);
    std::printf("\nroot_proof_bytes %zu\n", cur[0].size());
    glp_destroy(ctx);
    std::printf("OK\n");
    return 0;
}
