// fri.hip — Fiat-Shamir challenger (host) and the FRI opening-proof prover (SURVEY.md §8a
// rows a5, a8, a12; upstream names recalled, unverified: plonky2::iop::challenger::Challenger,
// fri::prover::fri_proof, PolynomialBatch::prove_openings — reference file:line NONE).
// The protocol and the proof byte layout are BUILD-DEFINED: the transcript
// order, gate set and wire format of plonky2 cannot be restated without its source.  The
// proof is checked end to end by an independent verifier (tests/fri_verifier.py).
//
// Division of labour: everything proportional to the trace size runs in kernels
// (fri_kernels.cuh, hash_kernels.cuh, the NTT); the host runs the transcript (a few hundred
// Poseidon permutations), the 2^final_bits-point final interpolation and the serialisation.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>
#include <memory>
#include <vector>
#include "glp_ctx.h"
#include "hash_state.h"
#include "fri_kernels.cuh"

int glp_ntt_table(glp_ctx* c, int log_N, int inv, const u64** lo, const u64** hi);   // glprover.hip

#include "challenger.h"

static glp_challenger* challenger_new(glp_ctx* c) {
    if (!c->hash || !c->hash->have_consts) { glp_set_err(c, "Poseidon constants not set (glp_set_poseidon_constants)"); return nullptr; }
    glp_challenger* ch = new glp_challenger();
    memset(ch->state, 0, sizeof(ch->state));
    ch->n_in = ch->n_out = 0;
    ch->consts = c->hash->h_consts;
    ch->small_mds = c->hash->small_mds;
    return ch;
}

extern "C" int glp_challenger_new(glp_ctx* c, glp_challenger** out) {
    if (!c || !out) return GLP_E_INVALID;
    *out = challenger_new(c);
    return *out ? GLP_OK : GLP_E_STATE;
}
extern "C" void glp_challenger_free(glp_challenger* ch) { delete ch; }
extern "C" int glp_challenger_observe(glp_challenger* ch, const uint64_t* h_elems, size_t n) {
    if (!ch || (!h_elems && n)) return GLP_E_INVALID;
    for (size_t i = 0; i < n; i++) { if (h_elems[i] >= GL_P) return GLP_E_INVALID; ch->observe(h_elems[i]); }
    return GLP_OK;
}
extern "C" int glp_challenger_challenges(glp_challenger* ch, uint64_t* h_out, size_t n) {
    if (!ch || (!h_out && n)) return GLP_E_INVALID;
    for (size_t i = 0; i < n; i++) h_out[i] = ch->challenge();
    return GLP_OK;
}

extern "C" void glp_free_host(void* p) { free(p); }

// ---------------------------------------------------------------------------------------
// helpers
// ---------------------------------------------------------------------------------------
namespace {
typedef GlpPoolBuf DevBuf;   // temporaries come from the ctx pool (glp_ctx.h): no hipMalloc/hipFree per proof

// f_p(z) for every polynomial of one batch -> out[2*p], out[2*p+1]
int eval_batch_at(glp_ctx* c, const u64* d_coeffs, u64 stride, u32 log_n, u32 n_polys, const u64* d_zp, u64* h_out) {
    const u64 n = 1ull << log_n;
    const u32 n_chunks = (u32)((n + GLP_EVAL_CHUNK - 1) / GLP_EVAL_CHUNK);
    DevBuf part(c);
    GLP_HIPCHK(c, part.alloc((size_t)n_polys * n_chunks * 16));
    hipLaunchKernelGGL(glp_eval_ext_kernel<0>, dim3(n_polys * n_chunks), dim3(256), 0, c->stream, d_coeffs, stride, n, n_chunks, d_zp,
                       part.u());
    GLP_HIPCHK(c, hipGetLastError());
    std::vector<u64> hp((size_t)n_polys * n_chunks * 2);
    GLP_HIPCHK(c, hipMemcpyAsync(hp.data(), part.p, hp.size() * 8, hipMemcpyDeviceToHost, c->stream));
    GLP_HIPCHK(c, hipStreamSynchronize(c->stream));
    for (u32 p = 0; p < n_polys; p++) {
        u64 a = 0, b = 0;
        for (u32 k = 0; k < n_chunks; k++) { a = gl_add(a, hp[2 * ((size_t)p * n_chunks + k)]); b = gl_add(b, hp[2 * ((size_t)p * n_chunks + k) + 1]); }
        h_out[2 * p] = a; h_out[2 * p + 1] = b;
    }
    return GLP_OK;
}

// device table z^j, j < n
int build_zpowers(glp_ctx* c, gl_ext2 z, u32 log_n, DevBuf& zp) {
    const u64 n = 1ull << log_n;
    const u64 nhi = (n + 255) / 256;
    std::vector<u64> lo(512), hi(2 * nhi);
    gl_ext2 t{1, 0};
    for (int j = 0; j < 256; j++) { lo[2 * j] = t.a; lo[2 * j + 1] = t.b; t = gl_ext_mul(t, z); }
    const gl_ext2 z256 = t;
    t = gl_ext2{1, 0};
    for (u64 j = 0; j < nhi; j++) { hi[2 * j] = t.a; hi[2 * j + 1] = t.b; t = gl_ext_mul(t, z256); }
    DevBuf dlo(c), dhi(c);
    GLP_HIPCHK(c, dlo.alloc(lo.size() * 8));
    GLP_HIPCHK(c, dhi.alloc(hi.size() * 8));
    GLP_HIPCHK(c, hipMemcpyAsync(dlo.p, lo.data(), lo.size() * 8, hipMemcpyHostToDevice, c->stream));
    GLP_HIPCHK(c, hipMemcpyAsync(dhi.p, hi.data(), hi.size() * 8, hipMemcpyHostToDevice, c->stream));
    GLP_HIPCHK(c, zp.alloc(n * 16));
    u64 blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(glp_ext_powers_kernel<0>, dim3((unsigned)blocks), dim3(256), 0, c->stream, zp.u(), n, dlo.u(), dhi.u());
    GLP_HIPCHK(c, hipGetLastError());
    GLP_HIPCHK(c, hipStreamSynchronize(c->stream));   // dlo/dhi are freed on return
    return GLP_OK;
}

u64 digest_level_base(u64 n_leaves, u32 h) { return h == 0 ? 0 : 4 * (2 * n_leaves - (n_leaves >> (h - 1))); }

int gather(glp_ctx* c, const u64* d_src, const std::vector<u64>& offs, u64* h_out) {
    if (offs.empty()) return GLP_OK;
    DevBuf doffs(c), dout(c);
    GLP_HIPCHK(c, doffs.alloc(offs.size() * 8));
    GLP_HIPCHK(c, dout.alloc(offs.size() * 8));
    GLP_HIPCHK(c, hipMemcpyAsync(doffs.p, offs.data(), offs.size() * 8, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(glp_gather_kernel<0>, dim3((unsigned)((offs.size() + 255) / 256)), dim3(256), 0, c->stream, d_src, doffs.u(),
                       (u64)offs.size(), dout.u());
    GLP_HIPCHK(c, hipGetLastError());
    GLP_HIPCHK(c, hipMemcpyAsync(h_out, dout.p, offs.size() * 8, hipMemcpyDeviceToHost, c->stream));
    GLP_HIPCHK(c, hipStreamSynchronize(c->stream));
    return GLP_OK;
}
}  // namespace

extern "C" int glp_eval_at_ext(glp_ctx* c, const uint64_t* d_coeffs, uint64_t poly_stride, uint32_t log_n, uint32_t n_polys,
                               const uint64_t* h_z, uint64_t* h_out) {
    if (!c) return GLP_E_INVALID;
    GLP_BIND(c);
    if (!d_coeffs || !h_z || !h_out || log_n > 30 || poly_stride < (1ull << log_n) || h_z[0] >= GL_P || h_z[1] >= GL_P) {
        glp_set_err(c, "glp_eval_at_ext: bad argument");
        return GLP_E_INVALID;
    }
    if (n_polys == 0) return GLP_OK;
    DevBuf zp(c);
    int rc = build_zpowers(c, gl_ext2{h_z[0], h_z[1]}, log_n, zp);
    if (rc) return rc;
    return eval_batch_at(c, d_coeffs, poly_stride, log_n, n_polys, zp.u(), h_out);
}

extern "C" int glp_pow_grind(glp_ctx* c, const uint64_t* h_seed4, uint32_t pow_bits, uint64_t* h_nonce) {
    if (!c) return GLP_E_INVALID;
    GLP_BIND(c);
    if (!h_seed4 || !h_nonce || pow_bits > 40) { glp_set_err(c, "glp_pow_grind: bad argument"); return GLP_E_INVALID; }
    if (!c->hash || !c->hash->have_consts) { glp_set_err(c, "Poseidon constants not set"); return GLP_E_STATE; }
    if (pow_bits == 0) { *h_nonce = 0; return GLP_OK; }
    DevBuf seed(c), found(c);
    GLP_HIPCHK(c, seed.alloc(32));
    GLP_HIPCHK(c, found.alloc(8));
    GLP_HIPCHK(c, hipMemcpyAsync(seed.p, h_seed4, 32, hipMemcpyHostToDevice, c->stream));
    // expected work is 2^pow_bits tries: search windows of 4x that (>= 2^16, <= 2^22 nonces per launch)
    u64 window = 4ull << pow_bits;
    if (window < (1ull << 16)) window = 1ull << 16;
    if (window > (1ull << 22)) window = 1ull << 22;
    const GlpPoseidonConsts k = glp_dev_consts(c->hash);
    for (u64 base = 0; base < (1ull << 48); base += window) {
        unsigned long long init = ~0ull, got = 0;
        GLP_HIPCHK(c, hipMemcpyAsync(found.p, &init, 8, hipMemcpyHostToDevice, c->stream));
        if (c->hash->small_mds) hipLaunchKernelGGL(glp_pow_kernel<true>, dim3((unsigned)(window / 256)), dim3(256), 0, c->stream, seed.u(), base, window, pow_bits, (unsigned long long*)found.p, k);
        else hipLaunchKernelGGL(glp_pow_kernel<false>, dim3((unsigned)(window / 256)), dim3(256), 0, c->stream, seed.u(), base, window, pow_bits, (unsigned long long*)found.p, k);
        GLP_HIPCHK(c, hipGetLastError());
        GLP_HIPCHK(c, hipMemcpyAsync(&got, found.p, 8, hipMemcpyDeviceToHost, c->stream));
        GLP_HIPCHK(c, hipStreamSynchronize(c->stream));
        if (got != ~0ull) { *h_nonce = got; return GLP_OK; }
    }
    glp_set_err(c, "glp_pow_grind: no nonce found");
    return GLP_E_UNSUPPORTED;
}

// ---------------------------------------------------------------------------------------
// the prover
// ---------------------------------------------------------------------------------------
extern "C" int glp_merkle(glp_ctx* c, const uint64_t* d_leaves, uint32_t leaf_len, uint32_t log_leaves, uint32_t cap_h,
                          uint64_t* d_digests, uint64_t* h_cap);
extern "C" int glp_fri_fold2(glp_ctx* c, const uint64_t* d_evals, uint64_t* d_out, uint32_t log_n, uint64_t shift, const uint64_t* h_beta);
extern "C" int glp_field_op(glp_ctx* c, int op, const uint64_t* a, const uint64_t* b, uint64_t* out, uint64_t n);

int glp_fri_prove_impl(glp_ctx* c, const glp_fri_config* cfg, const glp_fri_batch* batches, uint32_t n_batches,
                       glp_challenger& ch, std::vector<u64>& P) {
    if (!cfg || !batches || n_batches == 0) { glp_set_err(c, "glp_fri_prove: null argument"); return GLP_E_INVALID; }
    const u32 log_n = cfg->log_n, rb = cfg->rate_bits, a = cfg->arity_bits, fb = cfg->final_poly_bits;
    const u32 log_N = log_n + rb;
    if (log_n < 2 || log_n > 26 || rb < 1 || rb > 6 || a < 1 || a > 5 || fb > log_n || cfg->num_queries == 0 || cfg->num_queries > 256 ||
        cfg->pow_bits > 32 || cfg->cap_height > 12 || cfg->shift == 0 || cfg->shift >= GL_P || cfg->n_points == 0 || cfg->n_points > 4) {
        glp_set_err(c, "glp_fri_prove: unsupported configuration");
        return GLP_E_INVALID;
    }
    const u32 NP = cfg->n_points;
    for (u32 p = 0; p < NP; p++) if (cfg->point_mult[p] == 0 || cfg->point_mult[p] >= GL_P) { glp_set_err(c, "glp_fri_prove: bad point multiplier"); return GLP_E_INVALID; }
    if (!c->hash || !c->hash->have_consts) { glp_set_err(c, "Poseidon constants not set (glp_set_poseidon_constants)"); return GLP_E_STATE; }
    const u64 n = 1ull << log_n, N = 1ull << log_N;
    const u32 cap0 = cfg->cap_height < log_N ? cfg->cap_height : log_N;
    u32 total_polys = 0;
    for (u32 b = 0; b < n_batches; b++) {
        if (!batches[b].d_coeffs || !batches[b].d_lde || !batches[b].d_digests || !batches[b].h_cap || batches[b].n_polys == 0 ||
            batches[b].open_mask == 0 || (batches[b].open_mask >> NP)) {
            glp_set_err(c, "glp_fri_prove: batch %u incomplete", b);
            return GLP_E_INVALID;
        }
        for (u32 p = 0; p < NP; p++) if ((batches[b].open_mask >> p) & 1u) total_polys += batches[b].n_polys;   // one opening per (point, poly)
    }
    {
        bool p0 = false;
        for (u32 b = 0; b < n_batches; b++) p0 = p0 || (batches[b].open_mask & 1u);
        if (!p0) { glp_set_err(c, "glp_fri_prove: no batch is opened at point 0"); return GLP_E_INVALID; }
    }
    const u32 L = (log_n > fb) ? (log_n - fb) / a : 0;      // committed fold layers
    const u32 final_bits = log_n - a * L;                   // degree bound of the final polynomial
    if (final_bits + rb > 12) { glp_set_err(c, "glp_fri_prove: final polynomial too large (2^%u points) for host interpolation", final_bits + rb); return GLP_E_UNSUPPORTED; }

    auto put = [&](u64 v) { P.push_back(v); };
    const size_t hdr0 = P.size();

    // header + caps
    put(0x32304952464C4747ull /* "GGLFRI02" little-endian tag */);
    put(log_n); put(rb); put(cap0); put(a); put(fb); put(cfg->num_queries); put(cfg->pow_bits); put(cfg->shift); put(n_batches);
    put(NP);
    for (u32 p = 0; p < NP; p++) put(cfg->point_mult[p]);
    for (u32 b = 0; b < n_batches; b++) { put(batches[b].n_polys); put(batches[b].open_mask); }
    for (size_t k = hdr0; k < P.size(); k++) ch.observe(P[k] % GL_P);   // bind the statement parameters
    for (u32 b = 0; b < n_batches; b++)
        for (u32 i = 0; i < (4u << cap0); i++) { const u64 v = batches[b].h_cap[i]; if (v >= GL_P) { glp_set_err(c, "cap not canonical"); return GLP_E_INVALID; } put(v); ch.observe(v); }

    glp_stage_mark(c, "fri:evaluate_openings");
    // opening points z_p = zeta * mult_p and the openings, in (point, batch, polynomial) order
    const gl_ext2 zeta = ch.ext_challenge();
    std::vector<u64> openings(2 * (size_t)total_polys);
    int rc;
    {
        size_t off = 0;
        for (u32 p = 0; p < NP; p++) {
            DevBuf zp(c);
            rc = build_zpowers(c, gl_ext_scale(zeta, cfg->point_mult[p]), log_n, zp);
            if (rc) return rc;
            for (u32 b = 0; b < n_batches; b++) {
                if (!((batches[b].open_mask >> p) & 1u)) continue;
                rc = eval_batch_at(c, batches[b].d_coeffs, n, log_n, batches[b].n_polys, zp.u(), openings.data() + off);
                if (rc) return rc;
                off += 2 * (size_t)batches[b].n_polys;
            }
        }
    }
    for (u64 v : openings) { put(v); ch.observe(v); }
    const gl_ext2 alpha = ch.ext_challenge();

    glp_stage_mark(c, "fri:combine");
    // alpha powers (one per opening) ; per point Y_p = sum alpha^k y_k over that point's openings
    std::vector<u64> apow(2 * (size_t)total_polys);
    {
        gl_ext2 t{1, 0};
        for (u32 k = 0; k < total_polys; k++) { apow[2 * k] = t.a; apow[2 * k + 1] = t.b; t = gl_ext_mul(t, alpha); }
    }
    DevBuf d_apow(c), d_code(c), d_tmp(c);
    GLP_HIPCHK(c, d_apow.alloc(apow.size() * 8));
    GLP_HIPCHK(c, hipMemcpyAsync(d_apow.p, apow.data(), apow.size() * 8, hipMemcpyHostToDevice, c->stream));
    GLP_HIPCHK(c, d_code.alloc(N * 16));
    if (NP > 1) GLP_HIPCHK(c, d_tmp.alloc(N * 16));
    const u64* w_lo = nullptr; const u64* w_hi = nullptr;
    rc = glp_ntt_table(c, (int)log_N, 0, &w_lo, &w_hi);
    if (rc) return rc;
    {
        u32 koff = 0;
        for (u32 p = 0; p < NP; p++) {
            u32 nb_here = 0, last_b = 0;
            for (u32 b = 0; b < n_batches; b++) if ((batches[b].open_mask >> p) & 1u) { nb_here++; last_b = b; }
            if (nb_here == 0) continue;
            gl_ext2 Y{0, 0};
            {
                u32 k = koff;
                for (u32 b = 0; b < n_batches; b++) {
                    if (!((batches[b].open_mask >> p) & 1u)) continue;
                    for (u32 q = 0; q < batches[b].n_polys; q++, k++)
                        Y = gl_ext_add(Y, gl_ext_mul(gl_ext2{apow[2 * k], apow[2 * k + 1]}, gl_ext2{openings[2 * k], openings[2 * k + 1]}));
                }
            }
            u64* target = (p == 0) ? d_code.u() : d_tmp.u();
            bool first = true;
            for (u32 b = 0; b < n_batches; b++) {
                if (!((batches[b].open_mask >> p) & 1u)) continue;
                GlpCombineArgs ca;
                ca.lde = batches[b].d_lde; ca.poly_stride = N; ca.n_polys = batches[b].n_polys;
                ca.alpha_pow = d_apow.u() + 2 * (size_t)koff;
                ca.acc = target; ca.log_N = log_N; ca.first = first; ca.finish = (b == last_b);
                ca.Y = Y; ca.z = gl_ext_scale(zeta, cfg->point_mult[p]); ca.shift = cfg->shift; ca.w_lo = w_lo; ca.w_hi = w_hi;
                hipLaunchKernelGGL(glp_fri_combine_kernel<0>, dim3((unsigned)((N / 4 + 255) / 256)), dim3(256), 0, c->stream, ca);
                GLP_HIPCHK(c, hipGetLastError());
                koff += batches[b].n_polys;
                first = false;
            }
            if (p > 0) {
                rc = glp_field_op(c, 0, d_code.u(), d_tmp.u(), d_code.u(), 2 * N);
                if (rc) return rc;
            }
        }
    }

    glp_stage_mark(c, "fri:fold_layers+merkle");
    // commit phase: L layers of arity 2^a
    struct Layer { DevBuf code, dig; u32 log_len; u32 cap_h; explicit Layer(glp_ctx* cx) : code(cx), dig(cx), log_len(0), cap_h(0) {} };
    std::vector<std::unique_ptr<Layer>> layers;
    DevBuf cur(c); cur.adopt(d_code.release());             // take ownership
    u32 log_len = log_N;
    u64 shift = cfg->shift;
    for (u32 l = 0; l < L; l++) {
        std::unique_ptr<Layer> ly(new Layer(c));
        ly->log_len = log_len;
        const u32 log_leaves = log_len - a;
        ly->cap_h = cfg->cap_height < log_leaves ? cfg->cap_height : log_leaves;
        GLP_HIPCHK(c, ly->dig.alloc(8 * 4 * ((2ull << log_leaves) - (1ull << ly->cap_h))));
        std::vector<u64> cap((size_t)4 << ly->cap_h);
        rc = glp_merkle(c, cur.u(), 2u << a, log_leaves, ly->cap_h, ly->dig.u(), cap.data());
        if (rc) return rc;
        for (u64 v : cap) { put(v); ch.observe(v); }
        gl_ext2 beta = ch.ext_challenge();
        DevBuf src(c); src.adopt(cur.release());
        const u64* in = (const u64*)src.p;
        std::vector<std::unique_ptr<DevBuf>> tmp;
        for (u32 f = 0; f < a; f++) {
            std::unique_ptr<DevBuf> o(new DevBuf(c));
            GLP_HIPCHK(c, o->alloc((size_t)16 << (log_len - 1)));
            const u64 hb[2] = {beta.a, beta.b};
            rc = glp_fri_fold2(c, in, o->u(), log_len, shift, hb);
            if (rc) return rc;
            in = o->u();
            tmp.push_back(std::move(o));
            beta = gl_ext_mul(beta, beta);
            shift = gl_mul(shift, shift);
            log_len--;
        }
        cur.adopt(tmp.back()->release());                   // keep the last; intermediates go back to the pool
        ly->code.adopt(src.release());
        layers.push_back(std::move(ly));
    }

    // final polynomial: the remaining codeword (2^(final_bits+rb) points, bit-reversed, on
    // shift * <w>) interpolated on the host; coefficients beyond 2^final_bits must vanish.
    {
        const u64 M = 1ull << log_len;
        std::vector<u64> hc(2 * M);
        GLP_HIPCHK(c, hipMemcpyAsync(hc.data(), cur.p, hc.size() * 8, hipMemcpyDeviceToHost, c->stream));
        GLP_HIPCHK(c, hipStreamSynchronize(c->stream));
        const u64 winv = gl_inv(gl_root_of_unity(log_len)), minv = gl_inv(M % GL_P), sinv = gl_inv(shift);
        std::vector<gl_ext2> nat(M);
        for (u64 i = 0; i < M; i++) {
            u64 r = 0;
            for (u32 bb = 0; bb < log_len; bb++) r |= ((i >> bb) & 1ull) << (log_len - 1 - bb);
            nat[r] = gl_ext2{hc[2 * i], hc[2 * i + 1]};
        }
        // inverse DFT (host radix-2, O(M log M): nat[] is in natural order, bit-reverse then DIT with
        // w^-1), scale by 1/M, then un-shift: co_j = c_j * shift^-j
        std::vector<gl_ext2> co(M);
        for (u64 i = 0; i < M; i++) {
            u64 r = 0;
            for (u32 bb = 0; bb < log_len; bb++) r |= ((i >> bb) & 1ull) << (log_len - 1 - bb);
            co[r] = nat[i];
        }
        for (u32 st = 1; st <= log_len; st++) {
            const u64 half = 1ull << (st - 1), m2 = half << 1;
            const u64 ws = gl_pow(winv, M >> st);
            for (u64 k = 0; k < M; k += m2) {
                u64 t = 1;
                for (u64 j = 0; j < half; j++) {
                    const gl_ext2 u = co[k + j], v = gl_ext_scale(co[k + j + half], t);
                    co[k + j] = gl_ext_add(u, v);
                    co[k + j + half] = gl_ext_sub(u, v);
                    t = gl_mul(t, ws);
                }
            }
        }
        {
            u64 sc = minv;
            for (u64 j = 0; j < M; j++) { co[j] = gl_ext_scale(co[j], sc); sc = gl_mul(sc, sinv); }
        }
        for (u64 j = (1ull << final_bits); j < M; j++)
            if (co[j].a || co[j].b) { glp_set_err(c, "glp_fri_prove: final codeword is not of degree < 2^%u (inconsistent batches?)", final_bits); return GLP_E_INVALID; }
        for (u64 j = 0; j < (1ull << final_bits); j++) { put(co[j].a); put(co[j].b); ch.observe_ext(co[j]); }
    }

    glp_stage_mark(c, "fri:proof_of_work");
    // proof of work
    {
        u64 seed[4];
        for (int i = 0; i < 4; i++) seed[i] = ch.challenge();
        u64 nonce = 0;
        rc = glp_pow_grind(c, seed, cfg->pow_bits, &nonce);
        if (rc) return rc;
        put(nonce);
        ch.observe(nonce % GL_P);
    }

    glp_stage_mark(c, "fri:queries");
    // query phase
    std::vector<u64> idx(cfg->num_queries);
    for (u32 q = 0; q < cfg->num_queries; q++) idx[q] = ch.challenge() & (N - 1);
    const u32 Q = cfg->num_queries;
    // initial trees: leaf values + paths
    std::vector<std::vector<u64>> leafv(n_batches), pathv(n_batches);
    const u32 path0 = log_N - cap0;
    for (u32 b = 0; b < n_batches; b++) {
        std::vector<u64> offs;
        for (u32 q = 0; q < Q; q++)
            for (u32 p = 0; p < batches[b].n_polys; p++) offs.push_back((u64)p * N + idx[q]);
        leafv[b].resize(offs.size());
        rc = gather(c, batches[b].d_lde, offs, leafv[b].data());
        if (rc) return rc;
        offs.clear();
        for (u32 q = 0; q < Q; q++)
            for (u32 h = 0; h < path0; h++) {
                const u64 sib = (idx[q] >> h) ^ 1ull;
                for (u32 j = 0; j < 4; j++) offs.push_back(digest_level_base(N, h) + 4 * sib + j);
            }
        pathv[b].resize(offs.size());
        rc = gather(c, batches[b].d_digests, offs, pathv[b].data());
        if (rc) return rc;
    }
    // fold layers: the 2^a-coset leaf containing the index, and its path
    std::vector<std::vector<u64>> lleaf(L), lpath(L);
    for (u32 l = 0; l < L; l++) {
        Layer& ly = *layers[l];
        const u32 log_leaves = ly.log_len - a;
        const u64 n_leaves = 1ull << log_leaves;
        std::vector<u64> offs;
        for (u32 q = 0; q < Q; q++) {
            const u64 leaf = idx[q] >> (a * (l + 1));
            for (u32 j = 0; j < (2u << a); j++) offs.push_back(leaf * (2u << a) + j);
        }
        lleaf[l].resize(offs.size());
        rc = gather(c, ly.code.u(), offs, lleaf[l].data());
        if (rc) return rc;
        offs.clear();
        for (u32 q = 0; q < Q; q++) {
            const u64 leaf = idx[q] >> (a * (l + 1));
            for (u32 h = 0; h < log_leaves - ly.cap_h; h++) {
                const u64 sib = (leaf >> h) ^ 1ull;
                for (u32 j = 0; j < 4; j++) offs.push_back(digest_level_base(n_leaves, h) + 4 * sib + j);
            }
        }
        lpath[l].resize(offs.size());
        rc = gather(c, ly.dig.u(), offs, lpath[l].data());
        if (rc) return rc;
    }
    for (u32 q = 0; q < Q; q++) {
        put(idx[q]);
        for (u32 b = 0; b < n_batches; b++) {
            const u32 np = batches[b].n_polys;
            for (u32 p = 0; p < np; p++) put(leafv[b][(size_t)q * np + p]);
            for (u32 j = 0; j < 4 * path0; j++) put(pathv[b][(size_t)q * 4 * path0 + j]);
        }
        for (u32 l = 0; l < L; l++) {
            const u32 ll = 2u << a;
            for (u32 j = 0; j < ll; j++) put(lleaf[l][(size_t)q * ll + j]);
            const u32 pl = 4 * ((layers[l]->log_len - a) - layers[l]->cap_h);
            for (u32 j = 0; j < pl; j++) put(lpath[l][(size_t)q * pl + j]);
        }
    }

    return GLP_OK;
}

uint8_t* glp_words_to_blob(const std::vector<u64>& P, size_t* len) {
    uint8_t* blob = (uint8_t*)malloc(P.size() * 8 + 8);
    if (!blob) return nullptr;
    memcpy(blob, P.data(), P.size() * 8);   // host is little-endian (x86-64): words are LE u64
    *len = P.size() * 8;
    return blob;
}

extern "C" int glp_fri_prove(glp_ctx* c, const glp_fri_config* cfg, const glp_fri_batch* batches, uint32_t n_batches,
                             uint8_t** proof_out, size_t* proof_len) {
    if (!c) return GLP_E_INVALID;
    GLP_BIND(c);
    if (!cfg || !batches || n_batches == 0 || !proof_out || !proof_len) { glp_set_err(c, "glp_fri_prove: null argument"); return GLP_E_INVALID; }
    *proof_out = nullptr; *proof_len = 0;
    glp_challenger* chp = challenger_new(c);
    if (!chp) return GLP_E_STATE;
    std::unique_ptr<glp_challenger> guard(chp);
    std::vector<u64> P;
    c->stages.clear(); c->stage_name.clear();
    int rc = glp_fri_prove_impl(c, cfg, batches, n_batches, *chp, P);
    if (rc) return rc;
    glp_stage_mark(c, nullptr);
    *proof_out = glp_words_to_blob(P, proof_len);
    return *proof_out ? GLP_OK : GLP_E_NOMEM;
}
