// plonk.hip — prover driver for the build-defined circuit of DESIGN.md §3.6 (SURVEY.md §8a rows
// a6, a7 and the prover skeleton around them; upstream name recalled, unverified:
// plonky2::plonk::prover::prove — reference file:line NONE, the mount is empty).
//
//   setup : constant columns (plonk_gates.h) and sigma columns -> ifft -> coset LDE -> Merkle  (batch 0)
//   prove : header + public inputs + preprocessed cap into the transcript ; wires -> batch 1 ; beta, gamma x2 ;
//           Z + partial products over the routed wires (K6) -> batch 2 ;
//           alpha x2 ; quotient on the LDE domain (K7) -> coefficients -> 8 chunks -> batch 3 ;
//           FRI opening proof of all four batches at zeta, and of batch 2 at g*zeta.
// Everything proportional to n runs in kernels; the host runs the transcript and sequencing.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>
#include <memory>
#include <vector>
#include "glp_ctx.h"
#include "hash_state.h"
#include "challenger.h"
#include "plonk_kernels.cuh"

int glp_ntt_table(glp_ctx* c, int log_N, int inv, const u64** lo, const u64** hi);
int glp_fri_prove_impl(glp_ctx* c, const glp_fri_config* cfg, const glp_fri_batch* batches, uint32_t n_batches, glp_challenger& ch,
                       std::vector<u64>& P);
uint8_t* glp_words_to_blob(const std::vector<u64>& P, size_t* len);

namespace {
typedef GlpPoolBuf DBuf;     // ctx pool blocks (glp_ctx.h)
struct Commit {                 // one PolynomialBatch kept on the device
    DBuf coeffs, lde, dig;
    std::vector<u64> cap;
    u32 n_polys = 0;
    explicit Commit(glp_ctx* c) : coeffs(c), lde(c), dig(c) {}
};
}  // namespace

struct glp_plonk_circuit {       // must be freed (glp_plonk_free) before its ctx is destroyed
    u32 log_n, W, R, n_public, flags, rate_bits, cap_h;
    u64 shift;
    DBuf sigma_vals;            // [R][n] values on the trace domain (K6 input)
    DBuf ks, inv_xm1;
    std::vector<u64> h_ks;
    Commit pre;                 // constants + sigmas
    explicit glp_plonk_circuit(glp_ctx* c) : sigma_vals(c), ks(c), inv_xm1(c), pre(c) {}
};

// values [n_polys][n] (device, consumed: turned into coefficients in place) -> Commit
static int commit_values(glp_ctx* c, u64* d_vals_owned, u32 n_polys, u32 log_n, u32 rb, u32 cap_h, Commit& out) {
    const u32 log_N = log_n + rb;
    const u64 N = 1ull << log_N;
    out.n_polys = n_polys;
    out.coeffs.adopt(d_vals_owned);
    int rc = glp_ntt(c, out.coeffs.u(), log_n, n_polys, 1);
    if (rc) return rc;
    GLP_HIPCHK(c, out.lde.alloc((size_t)n_polys * N * 8));
    rc = glp_lde_coset(c, out.coeffs.u(), out.lde.u(), log_n, rb, n_polys, 7, GLP_NTT_BITREV);
    if (rc) return rc;
    const u32 ch = cap_h < log_N ? cap_h : log_N;
    GLP_HIPCHK(c, out.dig.alloc(8 * 4 * ((2ull << log_N) - (1ull << ch))));
    out.cap.resize((size_t)4 << ch);
    return glp_merkle_from_polys(c, out.lde.u(), N, n_polys, log_N, ch, out.dig.u(), out.cap.data());
}

extern "C" int glp_plonk_setup_ex(glp_ctx* c, const glp_circuit_shape* sh, const uint64_t* d_const_vals, const uint64_t* d_sigma_vals,
                                  glp_plonk_circuit** out) {
    if (!c) return GLP_E_INVALID;
    GLP_BIND(c);
    if (!out || !sh || !d_const_vals || !d_sigma_vals) { glp_set_err(c, "glp_plonk_setup: null argument"); return GLP_E_INVALID; }
    const u32 log_n = sh->log_n, n_wires = sh->n_wires, R = sh->n_routed;
    // rate_bits is structural, not a tunable: the permutation constraint has degree 9 (Z times eight linear factors) and a
    // Poseidon row degree 8, so the quotient has degree < 8n — 8 chunks, evaluated on the 8n-point coset; a larger blow-up
    // would only make every commitment bigger.  W <= 160 = 20 chunks of wires: what the K7 register budget was sized for.
    if (log_n < 3 || log_n > 24 || n_wires == 0 || n_wires % 8 || n_wires > 160 || R == 0 || R % 8 || R > n_wires || sh->rate_bits != 3 ||
        sh->cap_height > 12 || sh->n_public > (1u << log_n) || (sh->flags & ~(GLP_CIRCUIT_POSEIDON_GATE | GLP_CIRCUIT_SHA_GATES | GLP_CIRCUIT_EXT_GATE))) {
        glp_set_err(c, "glp_plonk_setup: unsupported shape (W %% 8 == 0, W <= 160, routed %% 8 == 0, routed <= W, rate_bits == 3, n_public <= n)");
        return GLP_E_INVALID;
    }
    if ((sh->flags & GLP_CIRCUIT_POSEIDON_GATE) && (n_wires < GLP_POS_GATE_WIRES || R < 24)) {
        glp_set_err(c, "glp_plonk_setup: a Poseidon-gate circuit needs >= %d wires, >= 24 of them routed", GLP_POS_GATE_WIRES);
        return GLP_E_INVALID;
    }
    if ((sh->flags & GLP_CIRCUIT_SHA_GATES) && (n_wires < GLP_SHA_GATE_WIRES || R < 16)) {
        glp_set_err(c, "glp_plonk_setup: a SHA-row circuit needs >= %d wires, >= 16 of them routed", GLP_SHA_GATE_WIRES);
        return GLP_E_INVALID;
    }
    if (!c->hash || !c->hash->have_consts) { glp_set_err(c, "Poseidon constants not set"); return GLP_E_STATE; }
    const u32 n_const = (u32)glp_plonk_n_const(sh->flags);
    std::unique_ptr<glp_plonk_circuit> ck(new glp_plonk_circuit(c));
    ck->log_n = log_n; ck->W = n_wires; ck->R = R; ck->n_public = sh->n_public; ck->flags = sh->flags;
    ck->rate_bits = sh->rate_bits; ck->cap_h = sh->cap_height; ck->shift = 7;
    const u64 n = 1ull << log_n;
    const u32 log_N = log_n + ck->rate_bits;
    const u64 N = 1ull << log_N;
    // coset representatives k_j = 7^j
    ck->h_ks.resize(R);
    { u64 t = 1; for (u32 j = 0; j < R; j++) { ck->h_ks[j] = t; t = gl_mul(t, 7); } }
    GLP_HIPCHK(c, ck->ks.alloc(R * 8));
    GLP_HIPCHK(c, hipMemcpyAsync(ck->ks.p, ck->h_ks.data(), R * 8, hipMemcpyHostToDevice, c->stream));
    GLP_HIPCHK(c, ck->sigma_vals.alloc((size_t)R * n * 8));
    GLP_HIPCHK(c, hipMemcpyAsync(ck->sigma_vals.p, d_sigma_vals, (size_t)R * n * 8, hipMemcpyDeviceToDevice, c->stream));
    // batch 0 = [the constant columns (6, or 10 with SHA rows), sigma_0 .. sigma_{R-1}]
    GlpPoolBuf pre_vals(c);                      // RAII until commit_values adopts it: an error on the way does not strand the block in pool_live
    if (pre_vals.alloc((size_t)(n_const + R) * n * 8) != hipSuccess) return GLP_E_NOMEM;
    GLP_HIPCHK(c, hipMemcpyAsync(pre_vals.p, d_const_vals, (size_t)n_const * n * 8, hipMemcpyDeviceToDevice, c->stream));
    GLP_HIPCHK(c, hipMemcpyAsync(pre_vals.u() + (size_t)n_const * n, d_sigma_vals, (size_t)R * n * 8, hipMemcpyDeviceToDevice, c->stream));
    int rc = commit_values(c, (u64*)pre_vals.release(), n_const + R, log_n, ck->rate_bits, ck->cap_h, ck->pre);
    if (rc) return rc;
    // 1 / (x - 1) on the LDE domain
    const u64* w_lo = nullptr; const u64* w_hi = nullptr;
    rc = glp_ntt_table(c, (int)log_N, 0, &w_lo, &w_hi);
    if (rc) return rc;
    GLP_HIPCHK(c, ck->inv_xm1.alloc(N * 8));
    hipLaunchKernelGGL(glp_inv_xm1_kernel<0>, dim3((unsigned)((N / 4 + 255) / 256)), dim3(256), 0, c->stream, ck->inv_xm1.u(), log_N, ck->shift,
                       w_lo, w_hi);
    GLP_HIPCHK(c, hipGetLastError());
    GLP_HIPCHK(c, hipStreamSynchronize(c->stream));
    *out = ck.release();
    return GLP_OK;
}

// the round-1 entry point: every wire routed, no public inputs, constants (q, c0, c1) -> (q, c0, c1, 0, 0, 0)
extern "C" int glp_plonk_setup(glp_ctx* c, uint32_t log_n, uint32_t n_wires, const uint64_t* d_const_vals, const uint64_t* d_sigma_vals,
                               uint32_t rate_bits, uint32_t cap_height, glp_plonk_circuit** out) {
    if (!c) return GLP_E_INVALID;
    GLP_BIND(c);
    if (!d_const_vals || log_n > 24) { glp_set_err(c, "glp_plonk_setup: bad argument"); return GLP_E_INVALID; }
    const u64 n = 1ull << log_n;
    DBuf six(c);
    GLP_HIPCHK(c, six.alloc((size_t)GLP_PLONK_NCONST * n * 8));
    GLP_HIPCHK(c, hipMemsetAsync(six.p, 0, (size_t)GLP_PLONK_NCONST * n * 8, c->stream));
    GLP_HIPCHK(c, hipMemcpyAsync(six.p, d_const_vals, (size_t)3 * n * 8, hipMemcpyDeviceToDevice, c->stream));
    glp_circuit_shape sh;
    memset(&sh, 0, sizeof(sh));
    sh.log_n = log_n; sh.n_wires = n_wires; sh.n_routed = n_wires; sh.rate_bits = rate_bits; sh.cap_height = cap_height;
    return glp_plonk_setup_ex(c, &sh, six.u(), d_sigma_vals, out);
}

extern "C" void glp_plonk_free(glp_plonk_circuit* ck) { delete ck; }

// witness generation for Poseidon rows: wires 12..134 of every listed row from its wires 0..11 and its swap bit (plonk_gates.h)
extern "C" int glp_poseidon_gate_fill_rows(glp_ctx* c, uint64_t* d_wire_vals, uint32_t log_n, uint32_t n_wires, const uint32_t* d_rows,
                                           uint32_t n_rows) {
    if (!c) return GLP_E_INVALID;
    GLP_BIND(c);
    if (!d_wire_vals || (!d_rows && n_rows) || log_n > 24 || n_wires < GLP_POS_GATE_WIRES) { glp_set_err(c, "glp_poseidon_gate_fill_rows: bad argument"); return GLP_E_INVALID; }
    if (!c->hash || !c->hash->have_consts) { glp_set_err(c, "Poseidon constants not set"); return GLP_E_STATE; }
    if (n_rows == 0) return GLP_OK;
    if (c->hash->small_mds && !getenv("GLP_K7_GENERIC_MDS"))
        hipLaunchKernelGGL(glp_poseidon_gate_fill_kernel<1>, dim3((n_rows + 63) / 64), dim3(64), 0, c->stream, d_wire_vals, 1ull << log_n, d_rows, n_rows,
                           c->hash->d_consts);
    else
        hipLaunchKernelGGL(glp_poseidon_gate_fill_kernel<0>, dim3((n_rows + 63) / 64), dim3(64), 0, c->stream, d_wire_vals, 1ull << log_n, d_rows, n_rows,
                           c->hash->d_consts);
    GLP_HIPCHK(c, hipGetLastError());
    return GLP_OK;
}

// witness generation for SHA rows: the bit wires 12..143 of every listed row from its routed words 0..11 (plonk_gates.h)
extern "C" int glp_sha_gate_fill_rows(glp_ctx* c, uint64_t* d_wire_vals, uint32_t log_n, uint32_t n_wires, const uint32_t* d_rows,
                                      const uint32_t* d_kinds, uint32_t n_rows) {
    if (!c) return GLP_E_INVALID;
    GLP_BIND(c);
    if (!d_wire_vals || ((!d_rows || !d_kinds) && n_rows) || log_n > 24 || n_wires < GLP_SHA_GATE_WIRES) {
        glp_set_err(c, "glp_sha_gate_fill_rows: bad argument");
        return GLP_E_INVALID;
    }
    if (n_rows == 0) return GLP_OK;
    hipLaunchKernelGGL(glp_sha_gate_fill_kernel<0>, dim3((n_rows + 63) / 64), dim3(64), 0, c->stream, d_wire_vals, 1ull << log_n, d_rows, d_kinds, n_rows);
    GLP_HIPCHK(c, hipGetLastError());
    return GLP_OK;
}

extern "C" int glp_plonk_circuit_cap(glp_plonk_circuit* ck, uint64_t* h_cap, size_t* n_words) {
    if (!ck || !n_words || (!h_cap && *n_words)) return GLP_E_INVALID;
    const size_t need = ck->pre.cap.size(), k = *n_words < need ? *n_words : need;
    if (k) memcpy(h_cap, ck->pre.cap.data(), k * 8);
    *n_words = need;
    return GLP_OK;
}

// ---- K6: Z and partial products on the trace domain for given beta, gamma -> zs_vals [NCHAL*M][n] -------
static int perm_products(glp_ctx* c, glp_plonk_circuit* ck, const u64* d_wire_vals, const u64* beta, const u64* gamma, u64* zs_vals) {
    const u32 log_n = ck->log_n, W = ck->R, M = W / GLP_PLONK_CHUNK;      // W here = the ROUTED wires: rows 0..R-1 of d_wire_vals
    const u64 n = 1ull << log_n;
    const u64* wn_lo = nullptr; const u64* wn_hi = nullptr;
    int rc = glp_ntt_table(c, (int)log_n, 0, &wn_lo, &wn_hi);
    if (rc) return rc;
    DBuf qv(c), rr(c), bprod(c);
    GLP_HIPCHK(c, qv.alloc((size_t)GLP_PLONK_NCHAL * M * n * 8));
    GLP_HIPCHK(c, rr.alloc((size_t)GLP_PLONK_NCHAL * n * 8));
    GlpPermArgs pa;
    pa.wires = d_wire_vals; pa.sigmas = ck->sigma_vals.u(); pa.ks = ck->ks.u(); pa.log_n = log_n; pa.W = W;
    for (int t = 0; t < GLP_PLONK_NCHAL; t++) { pa.beta[t] = beta[t]; pa.gamma[t] = gamma[t]; }
    pa.w_lo = wn_lo; pa.w_hi = wn_hi; pa.qv = qv.u(); pa.rr = rr.u();
    hipLaunchKernelGGL(glp_perm_quotients_kernel<0>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, pa);
    GLP_HIPCHK(c, hipGetLastError());
    const u32 nb = (u32)((n + GLP_SCAN_BLOCK - 1) / GLP_SCAN_BLOCK);
    GLP_HIPCHK(c, bprod.alloc((size_t)GLP_PLONK_NCHAL * nb * 8));
    hipLaunchKernelGGL(glp_scan_reduce_kernel<0>, dim3(nb, GLP_PLONK_NCHAL), dim3(256), 0, c->stream, rr.u(), n, bprod.u());
    GLP_HIPCHK(c, hipGetLastError());
    hipLaunchKernelGGL(glp_scan_blocks_kernel<0>, dim3(GLP_PLONK_NCHAL), dim3(GLP_SCAN_TOP), 0, c->stream, bprod.u(), nb);
    GLP_HIPCHK(c, hipGetLastError());
    hipLaunchKernelGGL(glp_scan_apply_kernel<0>, dim3(nb, GLP_PLONK_NCHAL), dim3(256), 0, c->stream, rr.u(), qv.u(), n, M, bprod.u(), zs_vals);
    GLP_HIPCHK(c, hipGetLastError());
    return GLP_OK;
}

// ---- the public-input polynomial on the LDE domain: values (pi_i on row i < n_public, 0 elsewhere) -> ifft -> coset LDE ----
static int public_input_lde(glp_ctx* c, glp_plonk_circuit* ck, const u64* h_public, DBuf& pi_lde) {
    if (ck->n_public == 0) return GLP_OK;
    const u64 n = 1ull << ck->log_n;
    u64* v = (u64*)glp_pool_alloc(c, n * 8);
    if (!v) return GLP_E_NOMEM;
    DBuf co(c);
    co.adopt(v);
    GLP_HIPCHK(c, hipMemsetAsync(v, 0, n * 8, c->stream));
    GLP_HIPCHK(c, hipMemcpyAsync(v, h_public, (size_t)ck->n_public * 8, hipMemcpyHostToDevice, c->stream));
    GLP_HIPCHK(c, hipStreamSynchronize(c->stream));          // h_public is the caller's (pageable) memory
    int rc = glp_ntt(c, v, ck->log_n, 1, 1);
    if (rc) return rc;
    GLP_HIPCHK(c, pi_lde.alloc((n << ck->rate_bits) * 8));
    return glp_lde_coset(c, v, pi_lde.u(), ck->log_n, ck->rate_bits, 1, 7, GLP_NTT_BITREV);
}

static u32 n_constraints(const glp_plonk_circuit* ck) {
    return 2 + 3 * (ck->R / GLP_PLONK_CHUNK) + ((ck->flags & GLP_CIRCUIT_POSEIDON_GATE) ? GLP_POS_GATE_CONSTRAINTS : 0) +
           ((ck->flags & GLP_CIRCUIT_SHA_GATES) ? GLP_SHA_GATE_CONSTRAINTS : 0);
}

// ---- K7: quotient evaluations on the LDE domain (bit-reversed order) for given challenges -> quot_rev [NCHAL][N] ----
static int quotient_evals(glp_ctx* c, glp_plonk_circuit* ck, const u64* wires_lde, const u64* zs_lde, const u64* pi_lde, const u64* beta,
                          const u64* gamma, const u64* alpha, u64* quot_rev) {
    const u32 log_n = ck->log_n, rb = ck->rate_bits;
    const u32 log_N = log_n + rb;
    const u64 n = 1ull << log_n, N = 1ull << log_N;
    const u32 n_con = n_constraints(ck);
    std::vector<u64> apow((size_t)GLP_PLONK_NCHAL * n_con);
    for (int t = 0; t < GLP_PLONK_NCHAL; t++) { u64 x = 1; for (u32 k = 0; k < n_con; k++) { apow[(size_t)t * n_con + k] = x; x = gl_mul(x, alpha[t]); } }
    DBuf d_apow(c);
    GLP_HIPCHK(c, d_apow.alloc(apow.size() * 8));
    GLP_HIPCHK(c, hipMemcpyAsync(d_apow.p, apow.data(), apow.size() * 8, hipMemcpyHostToDevice, c->stream));
    const u64* wN_lo = nullptr; const u64* wN_hi = nullptr;
    int rc = glp_ntt_table(c, (int)log_N, 0, &wN_lo, &wN_hi);
    if (rc) return rc;
    GlpQuotientArgs qa;
    qa.consts = ck->pre.lde.u(); qa.sigmas = ck->pre.lde.u() + (u64)glp_plonk_n_const(ck->flags) * N;
    qa.q_ext = (ck->flags & GLP_CIRCUIT_EXT_GATE) ? ck->pre.lde.u() + (u64)(glp_plonk_n_const(ck->flags) - 1) * N : nullptr; qa.wires = wires_lde; qa.zs = zs_lde; qa.pi = pi_lde;
    qa.ks = ck->ks.u();
    qa.log_n = log_n; qa.rate_bits = rb; qa.W = ck->W; qa.R = ck->R; qa.n_con = n_con;
    for (int t = 0; t < GLP_PLONK_NCHAL; t++) { qa.beta[t] = beta[t]; qa.gamma[t] = gamma[t]; }
    qa.alpha_pow = d_apow.u(); qa.pos_consts = c->hash->d_consts; qa.w_lo = wN_lo; qa.w_hi = wN_hi; qa.shift = ck->shift;
    // x^n for natural index e: shift^n * w_N^(e n) = shift^n * w_{2^rb}^(e mod 2^rb)
    const u64 sn = gl_pow(ck->shift, n), wr = gl_root_of_unity(rb);
    for (u32 k = 0; k < (1u << rb); k++) qa.zh_inv[k] = gl_inv(gl_sub(gl_mul(sn, gl_pow(wr, k)), 1));
    qa.n_inv = gl_inv(n % GL_P);
    qa.inv_xm1 = ck->inv_xm1.u();
    qa.out = quot_rev;
    if ((ck->flags & GLP_CIRCUIT_POSEIDON_GATE) && c->hash->small_mds && !getenv("GLP_K7_GENERIC_MDS"))
        hipLaunchKernelGGL((glp_quotient_kernel<true, true>), dim3((unsigned)((N + 255) / 256)), dim3(256), 0, c->stream, qa);
    else if (ck->flags & GLP_CIRCUIT_POSEIDON_GATE)
        hipLaunchKernelGGL(glp_quotient_kernel<true>, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, c->stream, qa);
    else
        hipLaunchKernelGGL(glp_quotient_kernel<false>, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, c->stream, qa);
    GLP_HIPCHK(c, hipGetLastError());
    if (ck->flags & GLP_CIRCUIT_SHA_GATES) {
        const u32 first_con = n_con - GLP_SHA_GATE_CONSTRAINTS;
        hipLaunchKernelGGL(glp_quotient_sha_kernel<0>, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, c->stream, qa, first_con);
        GLP_HIPCHK(c, hipGetLastError());
    }
    GLP_HIPCHK(c, hipStreamSynchronize(c->stream));      // apow (host) and d_apow die with this frame
    return GLP_OK;
}

// values [n_polys][n] (device, consumed) -> coefficients + bit-reversed coset LDE, no Merkle tree (debug stage)
static int lde_values(glp_ctx* c, u64* d_vals_owned, u32 n_polys, u32 log_n, u32 rb, DBuf& coeffs, DBuf& lde) {
    coeffs.adopt(d_vals_owned);
    int rc = glp_ntt(c, coeffs.u(), log_n, n_polys, 1);
    if (rc) return rc;
    GLP_HIPCHK(c, lde.alloc(((size_t)n_polys << (log_n + rb)) * 8));
    return glp_lde_coset(c, coeffs.u(), lde.u(), log_n, rb, n_polys, 7, GLP_NTT_BITREV);
}

// Parity hook for rows a6 / a7: the prover's intermediate stages for CALLER-CHOSEN challenges, so that the HIP output can be
// compared with an independent restatement (tests/plonk_ref.py::ref_zs / ref_quotient) instead of only through accepted proofs.
extern "C" int glp_plonk_debug_stage(glp_ctx* c, glp_plonk_circuit* ck, const uint64_t* d_wire_vals, const uint64_t* h_public, int which,
                                     const uint64_t* h_challenges, uint64_t* d_out) {
    if (!c) return GLP_E_INVALID;
    GLP_BIND(c);
    if (!ck || !d_wire_vals || !h_challenges || !d_out || (which != GLP_DEBUG_ZS && which != GLP_DEBUG_QUOTIENT)) {
        glp_set_err(c, "glp_plonk_debug_stage: bad argument");
        return GLP_E_INVALID;
    }
    const u32 n_ch = which == GLP_DEBUG_ZS ? 2 * GLP_PLONK_NCHAL : 3 * GLP_PLONK_NCHAL;
    for (u32 i = 0; i < n_ch; i++) if (h_challenges[i] >= GL_P) { glp_set_err(c, "glp_plonk_debug_stage: challenge not canonical"); return GLP_E_INVALID; }
    const u64 *beta = h_challenges, *gamma = h_challenges + GLP_PLONK_NCHAL, *alpha = h_challenges + 2 * GLP_PLONK_NCHAL;
    const u32 log_n = ck->log_n, W = ck->W, rb = ck->rate_bits, M = ck->R / GLP_PLONK_CHUNK;
    const u64 n = 1ull << log_n;
    if (ck->n_public && !h_public) { glp_set_err(c, "glp_plonk_debug_stage: the circuit has public inputs"); return GLP_E_INVALID; }
    if (which == GLP_DEBUG_ZS) {
        int rc = perm_products(c, ck, d_wire_vals, beta, gamma, d_out);
        if (rc) return rc;
        GLP_HIPCHK(c, hipStreamSynchronize(c->stream));
        return GLP_OK;
    }
    u64* wv = (u64*)glp_pool_alloc(c, (size_t)W * n * 8);
    if (!wv) return GLP_E_NOMEM;
    DBuf wco(c), wlde(c), zco(c), zlde(c);
    GLP_HIPCHK(c, hipMemcpyAsync(wv, d_wire_vals, (size_t)W * n * 8, hipMemcpyDeviceToDevice, c->stream));
    int rc = lde_values(c, wv, W, log_n, rb, wco, wlde);
    if (rc) return rc;
    u64* zv = (u64*)glp_pool_alloc(c, (size_t)GLP_PLONK_NCHAL * M * n * 8);
    if (!zv) return GLP_E_NOMEM;
    zco.adopt(zv);
    rc = perm_products(c, ck, d_wire_vals, beta, gamma, zv);
    if (rc) return rc;
    zco.release();
    rc = lde_values(c, zv, GLP_PLONK_NCHAL * M, log_n, rb, zco, zlde);
    if (rc) return rc;
    DBuf pi_lde(c);
    rc = public_input_lde(c, ck, h_public, pi_lde);
    if (rc) return rc;
    return quotient_evals(c, ck, wlde.u(), zlde.u(), pi_lde.u(), beta, gamma, alpha, d_out);
}

extern "C" int glp_plonk_prove(glp_ctx* c, glp_plonk_circuit* ck, const uint64_t* d_wire_vals, uint32_t num_queries, uint32_t pow_bits,
                               uint8_t** proof_out, size_t* proof_len) {
    return glp_plonk_prove_ex(c, ck, d_wire_vals, nullptr, num_queries, pow_bits, proof_out, proof_len);
}

extern "C" int glp_plonk_prove_ex(glp_ctx* c, glp_plonk_circuit* ck, const uint64_t* d_wire_vals, const uint64_t* h_public, uint32_t num_queries,
                                  uint32_t pow_bits, uint8_t** proof_out, size_t* proof_len) {
    if (!c) return GLP_E_INVALID;
    GLP_BIND(c);
    if (!ck || !d_wire_vals || !proof_out || !proof_len || num_queries == 0 || num_queries > 256 || pow_bits > 32 || (ck->n_public && !h_public)) {
        glp_set_err(c, "glp_plonk_prove: bad argument");
        return GLP_E_INVALID;
    }
    for (u32 i = 0; i < ck->n_public; i++)
        if (h_public[i] >= GL_P) { glp_set_err(c, "glp_plonk_prove: public input %u not canonical", i); return GLP_E_INVALID; }
    *proof_out = nullptr; *proof_len = 0;
    if (!c->hash || !c->hash->have_consts) { glp_set_err(c, "Poseidon constants not set"); return GLP_E_STATE; }
    const u32 log_n = ck->log_n, W = ck->W, rb = ck->rate_bits, M = ck->R / GLP_PLONK_CHUNK;
    const u32 log_N = log_n + rb;
    const u64 n = 1ull << log_n, N = 1ull << log_N;

    glp_challenger ch;
    memset(ch.state, 0, sizeof(ch.state));
    ch.n_in = ch.n_out = 0;
    ch.consts = c->hash->h_consts;
    ch.small_mds = c->hash->small_mds;
    std::vector<u64> P;
    auto put = [&](u64 v) { P.push_back(v); ch.observe(v % GL_P); };
    // the statement first: shape, public inputs, the circuit's verifying key — all of it bound before the first challenge
    put(0x32304B4C504C4747ull /* "GGLPLK02" */); put(log_n); put(W); put(ck->R); put(rb); put(ck->cap_h); put(ck->n_public); put(ck->flags);
    for (u32 i = 0; i < ck->n_public; i++) put(h_public[i]);
    for (u64 v : ck->pre.cap) put(v);

    c->stages.clear(); c->stage_name.clear();
    glp_stage_mark(c, "commit_wires(ifft+lde+merkle)");
    // ---- wires ----------------------------------------------------------------------------
    Commit wires(c);
    {
        u64* wv = (u64*)glp_pool_alloc(c, (size_t)W * n * 8);
        if (!wv) return GLP_E_NOMEM;
        GLP_HIPCHK(c, hipMemcpyAsync(wv, d_wire_vals, (size_t)W * n * 8, hipMemcpyDeviceToDevice, c->stream));
        int rc = commit_values(c, wv, W, log_n, rb, ck->cap_h, wires);
        if (rc) return rc;
    }
    for (u64 v : wires.cap) put(v);
    u64 beta[GLP_PLONK_NCHAL], gamma[GLP_PLONK_NCHAL], alpha[GLP_PLONK_NCHAL];
    for (int t = 0; t < GLP_PLONK_NCHAL; t++) beta[t] = ch.challenge();
    for (int t = 0; t < GLP_PLONK_NCHAL; t++) gamma[t] = ch.challenge();

    glp_stage_mark(c, "perm_products(K6)");
    // ---- K6: Z and partial products on the trace domain --------------------------------------
    u64* zs_vals = (u64*)glp_pool_alloc(c, (size_t)GLP_PLONK_NCHAL * M * n * 8);
    if (!zs_vals) return GLP_E_NOMEM;
    Commit zs(c);
    zs.coeffs.adopt(zs_vals);       // owned from here on (released on any early return)
    int rc = perm_products(c, ck, d_wire_vals, beta, gamma, zs_vals);
    if (rc) return rc;
    zs.coeffs.release();
    glp_stage_mark(c, "commit_zs");
    rc = commit_values(c, zs_vals, GLP_PLONK_NCHAL * M, log_n, rb, ck->cap_h, zs);
    if (rc) return rc;
    for (u64 v : zs.cap) put(v);
    for (int t = 0; t < GLP_PLONK_NCHAL; t++) alpha[t] = ch.challenge();

    glp_stage_mark(c, "quotient(K7)+to_coeffs");
    // ---- K7: quotient on the LDE domain --------------------------------------------------------
    DBuf quot_rev(c);
    GLP_HIPCHK(c, quot_rev.alloc((size_t)GLP_PLONK_NCHAL * N * 8));
    DBuf pi_lde(c);
    rc = public_input_lde(c, ck, h_public, pi_lde);
    if (rc) return rc;
    rc = quotient_evals(c, ck, wires.lde.u(), zs.lde.u(), pi_lde.u(), beta, gamma, alpha, quot_rev.u());
    if (rc) return rc;
    pi_lde.reset();
    // evaluations (bit-reversed, coset) -> coefficients: un-bit-reverse, inverse NTT, unshift
    u64* quot_nat = (u64*)glp_pool_alloc(c, (size_t)GLP_PLONK_NCHAL * N * 8);
    if (!quot_nat) return GLP_E_NOMEM;
    Commit quot(c);
    quot.coeffs.adopt(quot_nat);
    hipLaunchKernelGGL(glp_bitrev_permute_kernel<0>, dim3((unsigned)(((u64)GLP_PLONK_NCHAL * N + 255) / 256)), dim3(256), 0, c->stream,
                       quot_rev.u(), quot_nat, log_N, GLP_PLONK_NCHAL);
    GLP_HIPCHK(c, hipGetLastError());
    rc = glp_ntt(c, quot_nat, log_N, GLP_PLONK_NCHAL, 1);
    if (rc) return rc;
    {
        const u64 sinv = gl_inv(ck->shift);
        std::vector<u64> lo(4096), hi(N > 4096 ? (N >> 12) : 1);
        u64 t = 1;
        for (u32 k = 0; k < 4096; k++) { lo[k] = t; t = gl_mul(t, sinv); }
        const u64 s4096 = t;
        t = 1;
        for (size_t k = 0; k < hi.size(); k++) { hi[k] = t; t = gl_mul(t, s4096); }
        DBuf dlo(c), dhi(c);
        GLP_HIPCHK(c, dlo.alloc(lo.size() * 8));
        GLP_HIPCHK(c, dhi.alloc(hi.size() * 8));
        GLP_HIPCHK(c, hipMemcpyAsync(dlo.p, lo.data(), lo.size() * 8, hipMemcpyHostToDevice, c->stream));
        GLP_HIPCHK(c, hipMemcpyAsync(dhi.p, hi.data(), hi.size() * 8, hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(glp_scale_pow_kernel<0>, dim3(4096), dim3(256), 0, c->stream, quot_nat, log_N, GLP_PLONK_NCHAL, dlo.u(),
                           N > 4096 ? dhi.u() : (const u64*)nullptr);
        GLP_HIPCHK(c, hipGetLastError());
        GLP_HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    glp_stage_mark(c, "commit_quotient(lde+merkle)");
    // each quotient has 8n coefficients = 8 chunks of n: [NCHAL][8][n] is already a dense
    // batch of 16 coefficient-form polynomials
    {
        const u32 nq = GLP_PLONK_NCHAL << rb;
        quot.n_polys = nq;
        GLP_HIPCHK(c, quot.lde.alloc((size_t)nq * N * 8));
        rc = glp_lde_coset(c, quot.coeffs.u(), quot.lde.u(), log_n, rb, nq, 7, GLP_NTT_BITREV);
        if (rc) return rc;
        const u32 chh = ck->cap_h < log_N ? ck->cap_h : log_N;
        GLP_HIPCHK(c, quot.dig.alloc(8 * 4 * ((2ull << log_N) - (1ull << chh))));
        quot.cap.resize((size_t)4 << chh);
        rc = glp_merkle_from_polys(c, quot.lde.u(), N, nq, log_N, chh, quot.dig.u(), quot.cap.data());
        if (rc) return rc;
    }
    for (u64 v : quot.cap) put(v);

    // ---- openings: everything at zeta, the Z batch also at g*zeta --------------------------------
    glp_fri_config fc;
    memset(&fc, 0, sizeof(fc));
    fc.log_n = log_n; fc.rate_bits = rb; fc.cap_height = ck->cap_h; fc.arity_bits = 4; fc.final_poly_bits = log_n < 5 ? log_n : 5;
    fc.num_queries = num_queries; fc.pow_bits = pow_bits; fc.shift = ck->shift;
    fc.n_points = 2; fc.point_mult[0] = 1; fc.point_mult[1] = gl_root_of_unity(log_n);
    glp_fri_batch fb[4];
    Commit* cs[4] = {&ck->pre, &wires, &zs, &quot};
    for (int b = 0; b < 4; b++) {
        fb[b].d_coeffs = cs[b]->coeffs.u(); fb[b].d_lde = cs[b]->lde.u(); fb[b].d_digests = cs[b]->dig.u();
        fb[b].h_cap = cs[b]->cap.data(); fb[b].n_polys = cs[b]->n_polys; fb[b].open_mask = (b == 2) ? 3u : 1u;
    }
    rc = glp_fri_prove_impl(c, &fc, fb, 4, ch, P);
    if (rc) return rc;
    glp_stage_mark(c, nullptr);
    *proof_out = glp_words_to_blob(P, proof_len);
    return *proof_out ? GLP_OK : GLP_E_NOMEM;
}
