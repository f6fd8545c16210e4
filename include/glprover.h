/*
 * glprover.h — C ABI of the MI355X-native Goldilocks prover kernels (libglprover.so).
 *
 * This is the drop-in boundary SURVEY.md §8(b) defines.  The north_star asks for "Rust
 * calling the kernels through a thin extern-"C" FFI"; the reference mount
 * (/root/reference: `.gitignore:1`, `changelog.md:1-2`, nothing else) contains no FFI, no
 * Rust and no interface file, so for every entry point below the "reference interface it
 * replaces" is:  file:line NONE — absent from mount.  The upstream function each one
 * stands in for is given by NAME ONLY, recalled and unverified, so a
 * maintainer knows where the Rust binding of INTEGRATION.md would be called from.
 *
 * Conventions
 *   - plain C types only; every function returns 0 on success or a negative GLP_E* code
 *     and never aborts; glp_last_error(ctx) gives a ctx-owned message.
 *   - field elements are little-endian uint64, canonical (< p = 2^64 - 2^32 + 1), in and out.
 *   - pointers named d_* are device (HBM) pointers valid on the ctx's GPU; h_* are host.
 *   - one glp_ctx per GPU per process; a ctx is not thread-safe.  Work is enqueued on the
 *     ctx's HIP stream; functions without an _async suffix return after enqueueing and the
 *     results are ordered on that stream (call glp_sync before reading them on the host).
 *   - there is NO CPU fallback: without a usable gfx950 device glp_create fails, and nothing the prover
 *     computes has a host path.  The only host arithmetic in the library is what is host work by nature:
 *     the Fiat-Shamir transcript (a few hundred permutations with a serial dependency) and the VERIFIERS
 *     (glp_*_verify*), which a party without a GPU must be able to run.
 */
#ifndef GLPROVER_H
#define GLPROVER_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GLP_OK 0
#define GLP_E_INVALID -1     /* bad argument */
#define GLP_E_NODEVICE -2    /* no usable HIP device */
#define GLP_E_HIP -3         /* HIP runtime error (see glp_last_error) */
#define GLP_E_NOMEM -4
#define GLP_E_UNSUPPORTED -5
#define GLP_E_STATE -6       /* e.g. Poseidon constants not set */
#define GLP_E_REJECT -7      /* a verifier rejected the proof (reason: glp_last_error) */

#define GLP_NTT_INVERSE 1u   /* inverse transform, scaled by 1/n */
#define GLP_NTT_BITREV 2u    /* write outputs in bit-reversed index order */

typedef struct glp_ctx glp_ctx;

/* ---- context, memory, stream ------------------------------------------------------ */
int glp_create(glp_ctx** out, int device_id);
void glp_destroy(glp_ctx* ctx);
const char* glp_last_error(const glp_ctx* ctx);
const char* glp_version(void);
/* The two-adic subgroup this BUILD computes in (csrc/gl_field.cuh: GLP_TWO_ADIC_GENERATOR, an element of order 2^32, and GLP_W64_LOG2 with
 * generator^(2^26) = 2^GLP_W64_LOG2): every root of unity of every transform is a power of that generator.  Default 7^((p-1)/2^32), w_64 = 2^39;
 * `make -C csrc altgen` builds lib/libglprover_altgen.so on 7277203076849721926, w_64 = 2^3 (recalled as upstream's choice, unverified).
 * GLP_E_STATE when the compiled pair is inconsistent (glp_create then refuses too).  Either pointer may be NULL. */
int glp_field_params(uint64_t* two_adic_generator, uint32_t* w64_log2);
int glp_alloc(glp_ctx* ctx, void** d_ptr, size_t bytes);
int glp_free(glp_ctx* ctx, void* d_ptr);
int glp_h2d(glp_ctx* ctx, void* d_dst, const void* h_src, size_t bytes);  /* synchronous */
int glp_d2h(glp_ctx* ctx, void* h_dst, const void* d_src, size_t bytes);  /* synchronous */
int glp_sync(glp_ctx* ctx);
/* the prover drivers keep their temporaries in a ctx-owned pool (reused across proofs, no hipFree on
 * the hot path); this returns every cached block to the driver (GLP_POOL_CAP_MB caps the cache) */
int glp_trim_pool(glp_ctx* ctx);
/* A ctx may be driven by a host thread other than its creator (still one thread at a time per ctx); HIP's
 * current device is per thread, so such a thread calls this once before its first call on the ctx.
 * glp_plonk_setup / glp_plonk_prove / glp_fri_prove bind by themselves. */
int glp_bind_thread(glp_ctx* ctx);
/* adopt an external hipStream_t (e.g. torch's current stream); NULL restores the ctx's own.  The ctx's pool and NTT
 * scratch are reused in stream order, so the switch first WAITS for everything enqueued on the stream being left
 * (a host-side synchronise): work on the new stream never overlaps work still using those blocks on the old one. */
int glp_set_stream(glp_ctx* ctx, void* hip_stream);
/* HIP-event timer on the ctx's stream: start, enqueue work, stop -> elapsed milliseconds */
int glp_timer_start(glp_ctx* ctx);
int glp_timer_stop(glp_ctx* ctx, float* ms);

/* ---- field arithmetic (upstream name recalled: GoldilocksField) ------
 * element-wise on device arrays of n canonical elements: out[i] = a[i] (op) b[i].
 * op: 0 add, 1 sub, 2 mul, 3 a[i] * 2^(b[i] mod 192), 4 inverse of a[i] (0 -> 0).
 * ops 5..10 take ARBITRARY 64-bit words and expose the reduction primitives under the products
 * (results canonical): 5 (a*2^64 + b) mod p, 6 the same through the lazy form, 7 a*b for any
 * representatives, 8 (a>>7) + (b>>7)*2^32, 9/10 a + (b mod 2^32)*(2^32-1) lazy/canonical.
 * The prover never calls this; it exposes the exact device arithmetic of the kernels to
 * parity tests and to hosts that need a few field operations on resident data. */
int glp_field_op(glp_ctx* ctx, int op, const uint64_t* d_a, const uint64_t* d_b, uint64_t* d_out, uint64_t n);

/* ---- NTT / LDE (upstream names recalled: plonky2_field::fft::
 *      fft / ifft / coset_fft, PolynomialCoeffs::lde, fri::oracle::PolynomialBatch) ------ */
/* in place, natural order in and out:  X[k] = sum_j x[j] w_n^{jk},  w_n = 7^((p-1)/n) */
int glp_ntt(glp_ctx* ctx, uint64_t* d_io, uint32_t log_n, uint32_t batch, int inverse);
/* general form: src may equal dst; poly strides in elements (>= n); flags = GLP_NTT_* */
int glp_ntt_ex(glp_ctx* ctx, const uint64_t* d_src, uint64_t* d_dst, uint32_t log_n, uint32_t batch,
               uint64_t src_poly_stride, uint64_t dst_poly_stride, uint32_t flags);
/* coefficients [batch][n] -> evaluations [batch][n << rate_bits] on the coset shift*<w>:
 * zero-pad, scale coefficient j by shift^j, forward NTT.  flags: GLP_NTT_BITREV or 0 */
int glp_lde_coset(glp_ctx* ctx, const uint64_t* d_coeffs, uint64_t* d_out, uint32_t log_n, uint32_t rate_bits,
                  uint32_t batch, uint64_t shift, uint32_t flags);
/* [rows][cols] -> [cols][rows] (polynomial-major <-> leaf-major) */
int glp_transpose(glp_ctx* ctx, const uint64_t* d_in, uint64_t* d_out, uint64_t rows, uint64_t cols);
/* force the pass structure of subsequent NTTs of size 2^log_n ("r:c,r:c,..." log2 radix :
 * log2 columns per pass; NULL/"" = default heuristic).  Tuning/benchmark aid. */
int glp_ntt_set_plan(glp_ctx* ctx, uint32_t log_n, const char* plan);
/* describe the plan that would be used for this size and batch (for logs): NUL-terminated */
int glp_ntt_describe_plan(glp_ctx* ctx, uint32_t log_n, uint32_t batch, uint32_t flags, char* buf, size_t buf_len);
/* per-pass kernel times (ms) of the LAST glp_ntt*_ call when profiling is on; n_out <= 4 */
int glp_set_profiling(glp_ctx* ctx, int on);
int glp_last_pass_ms(glp_ctx* ctx, float* ms, int* n_out);
/* stage timing tree of the LAST glp_plonk_prove / glp_fri_prove while profiling is on (each stage
 * boundary then synchronises the stream): names are ';'-separated, *n_inout = capacity in, count out */
int glp_last_stage_ms(glp_ctx* ctx, char* names, size_t names_len, float* ms, int* n_inout);

/* ---- Poseidon / Merkle (row a4; upstream names recalled: plonky2::hash::poseidon,
 *      hashing::hash_n_to_hash_no_pad, PoseidonHash::two_to_one, merkle_tree::MerkleTree::new) */
/* rc: 360 round constants (30 rounds x 12), mds_circ: 12, mds_diag: 12.  Must be called
 * before any hashing entry point; the library ships no constants. */
int glp_set_poseidon_constants(glp_ctx* ctx, const uint64_t* h_rc, size_t n_rc, const uint64_t* h_mds_circ,
                               const uint64_t* h_mds_diag);
/* n_states independent width-12 permutations, in place, [n_states][12] */
int glp_poseidon_permute(glp_ctx* ctx, uint64_t* d_states, uint64_t n_states);
/* d_leaves: [2^log_leaves][leaf_len] (leaf-major).  d_digests receives every level from
 * the leaf digests down to the cap level: 4 * (2^(log_leaves+1) - 2^cap_h) u64.
 * h_cap (may be NULL) receives the 2^cap_h cap digests (4 u64 each) — synchronous if given. */
int glp_merkle(glp_ctx* ctx, const uint64_t* d_leaves, uint32_t leaf_len, uint32_t log_leaves, uint32_t cap_h,
               uint64_t* d_digests, uint64_t* h_cap);
/* same tree, but leaves given polynomial-major: d_polys [leaf_len][2^log_leaves]
 * (leaf i = column i), so the LDE output is hashed without a transpose pass */
int glp_merkle_from_polys(glp_ctx* ctx, const uint64_t* d_polys, uint64_t poly_stride, uint32_t leaf_len,
                          uint32_t log_leaves, uint32_t cap_h, uint64_t* d_digests, uint64_t* h_cap);

/* ---- FRI (row a8; upstream name recalled: plonky2::fri::prover) -------------------- */
/* arity-2 fold of extension-field evaluations in bit-reversed order over shift*<w_{log_n}>:
 * d_evals [n][2] -> d_out [n/2][2];  h_beta = 2 u64 */
int glp_fri_fold2(glp_ctx* ctx, const uint64_t* d_evals, uint64_t* d_out, uint32_t log_n, uint64_t shift,
                  const uint64_t* h_beta);

/* ---- Fiat-Shamir challenger (row a5; upstream name recalled: plonky2::iop::challenger) ----
 * Poseidon duplex sponge (width 12, rate 8, overwrite mode) run on the HOST with the injected
 * constants: observing invalidates pending outputs; a challenge absorbs whatever is buffered
 * and pops outputs from the end of state[0..8).  Build-defined (transcript order unpinned). */
typedef struct glp_challenger glp_challenger;
int glp_challenger_new(glp_ctx* ctx, glp_challenger** out);
void glp_challenger_free(glp_challenger* ch);
int glp_challenger_observe(glp_challenger* ch, const uint64_t* h_elems, size_t n);
int glp_challenger_challenges(glp_challenger* ch, uint64_t* h_out, size_t n);

/* ---- FRI opening proof (rows a8, a12; upstream names recalled: fri::prover::fri_proof,
 *      PolynomialBatch::prove_openings).  Protocol + byte layout are build-defined: DESIGN.md §3.5 */
/* f_p(z) for n_polys coefficient-form polynomials (z in the quadratic extension, h_z = 2 u64):
 * h_out receives n_polys pairs */
int glp_eval_at_ext(glp_ctx* ctx, const uint64_t* d_coeffs, uint64_t poly_stride, uint32_t log_n, uint32_t n_polys,
                    const uint64_t* h_z, uint64_t* h_out);
/* smallest nonce with the top pow_bits bits of Poseidon(seed[0..4], nonce, 0...)[0] clear */
int glp_pow_grind(glp_ctx* ctx, const uint64_t* h_seed4, uint32_t pow_bits, uint64_t* h_nonce);

typedef struct {
    uint32_t log_n;            /* every committed polynomial has 2^log_n coefficients */
    uint32_t rate_bits;        /* LDE blow-up used when the batches were committed */
    uint32_t cap_height;       /* Merkle cap height of the batches and (clamped) of the fold layers */
    uint32_t arity_bits;       /* fold arity 2^arity_bits per committed layer */
    uint32_t final_poly_bits;  /* stop folding when the degree bound is <= 2^final_poly_bits (+ remainder) */
    uint32_t num_queries;
    uint32_t pow_bits;
    uint64_t shift;            /* coset shift of the LDE domain (7) */
    uint32_t n_points;         /* 1..4 opening points z_p = zeta * point_mult[p] (zeta from the transcript) */
    uint64_t point_mult[4];    /* base-field multipliers; {1} for a single point, {1, w_n} to also open the next row */
} glp_fri_config;
typedef struct {               /* one committed batch, as produced by ifft -> glp_lde_coset(BITREV) -> glp_merkle_from_polys */
    const uint64_t* d_coeffs;  /* [n_polys][2^log_n], dense */
    const uint64_t* d_lde;     /* [n_polys][2^(log_n+rate_bits)], bit-reversed evaluation order */
    const uint64_t* d_digests; /* Merkle digests of that LDE (layout of glp_merkle) */
    const uint64_t* h_cap;     /* its cap: 4 << min(cap_height, log_n+rate_bits) words (host) */
    uint32_t n_polys;
    uint32_t open_mask;        /* bit p set: every polynomial of the batch is opened at point p */
} glp_fri_batch;
/* proves the openings of the batches at the transcript-derived point(s) selected by open_mask.
 * *proof is malloc'd by the library (little-endian u64 words), free with glp_free_host. */
int glp_fri_prove(glp_ctx* ctx, const glp_fri_config* cfg, const glp_fri_batch* batches, uint32_t n_batches,
                  uint8_t** proof, size_t* proof_len);
void glp_free_host(void* p);

/* ---- prover for the build-defined circuit (rows a6, a7 + driver; upstream names recalled:
 *      plonk::prover::prove, compute_partial_products_and_z_polys, compute_quotient_polys, gates::{arithmetic_base,
 *      constant, public_input, poseidon}).
 * Circuit: 2^log_n rows, n_wires columns of which the first n_routed take part in the permutation
 * argument (copy constraints through sigma; the others are advice wires).  Per-row constant columns, in this order:
 *   [0] q_arith [1] c0 [2] c1 [3] c2 [4] q_pi [5] q_pos          (GLP_PLONK_NCONST = 6)
 * Gates:
 *   arithmetic / constant   every group of 4 routed wires (x, y, z, w):  q_arith * (c0*x*y + c1*z + c2 - w) = 0
 *   public input            q_pi * wire_0 - PI(x) = 0: set q_pi = 1 on rows 0..n_public-1 — wire 0 of row i is public input i
 *   Poseidon (flag)         a q_pos row carries one permutation with the ctx's constants: wires 0..11 in, 12..23 out, 24 a swap bit s
 *                           (s = 1 exchanges in[0..4) and in[4..8) before the permutation: the Merkle-path step), 25..130 the S-box inputs of
 *                           the later rounds, 131..134 s * (in[4+i] - in[i]) (123 constraints of degree <= 7; fill wires 12..134 with
 *                           glp_poseidon_gate_fill_rows); set q_arith = 0 on such rows
 * d_const_vals: [6][n] row values; d_sigma_vals: [n_routed][n] with sigma_j(row i) = k_{j'} * w_n^{i'} for the cell (j', i')
 * that (j, i) maps to, k_j = 7^j.  rate_bits must be 3: the quotient has degree < 8n (permutation constraint degree 9,
 * Poseidon row degree 8), so 8 chunks on the 8n-point coset is exactly what fits; it is not a tunable. */
#define GLP_PLONK_NCONST 6
#define GLP_CIRCUIT_POSEIDON_GATE 1u
#define GLP_POS_GATE_WIRES 135
/*   SHA-256 rows (flag)     four more constant columns select the row kind — [6] q_she [7] q_sha [8] q_shw [9] q_add, so d_const_vals is
 *                           [GLP_PLONK_NCONST_SHA][n] for such a circuit — and one compression is 64 E rows (e-half of a round: T1 and the new e;
 *                           K_t in the row's c2), 64 A rows (the new a), 48 W rows (message schedule) and 2 ADD rows (the feed-forward; four
 *                           additions mod 2^32 per row, also the 32-bit range check).  Words are routed wires 0..11; wires 12..143 are the
 *                           row's own bit decompositions (fill them with glp_sha_gate_fill_rows); 140 constraints of degree <= 4; layouts
 *                           and equations: csrc/plonk_gates.h */
#define GLP_PLONK_NCONST_SHA 10
#define GLP_CIRCUIT_SHA_GATES 2u
/*   extension rows (flag)   one more constant column, q_ext, LAST (index 6, or 10 with the SHA selectors): on a q_ext row every chunk of 8 routed
 *                           wires (x0, x1, y0, y1, z0, z1, w0, w1) is w = x * y + z in F_p[X]/(X^2 - 7); no further constraints (the two
 *                           values share the chunk's arithmetic-gate slots).  d_const_vals then has glp n_const = 6 + 4 (SHA) + 1 rows */
#define GLP_CIRCUIT_EXT_GATE 4u
#define GLP_SHA_GATE_WIRES 144
#define GLP_SHA_ROW_E 0
#define GLP_SHA_ROW_A 1
#define GLP_SHA_ROW_W 2
#define GLP_SHA_ROW_ADD 3
typedef struct {
    uint32_t log_n;        /* 3..24 */
    uint32_t n_wires;      /* multiple of 8, <= 160 */
    uint32_t n_routed;     /* multiple of 8, 8..n_wires */
    uint32_t n_public;     /* <= 2^log_n */
    uint32_t rate_bits;    /* 3 */
    uint32_t cap_height;   /* <= 12 */
    uint32_t flags;        /* GLP_CIRCUIT_POSEIDON_GATE: needs n_wires >= GLP_POS_GATE_WIRES and n_routed >= 24;
                              GLP_CIRCUIT_SHA_GATES: n_wires >= GLP_SHA_GATE_WIRES, n_routed >= 16, ten constant columns */
} glp_circuit_shape;
typedef struct glp_plonk_circuit glp_plonk_circuit;
int glp_plonk_setup_ex(glp_ctx* ctx, const glp_circuit_shape* shape, const uint64_t* d_const_vals, const uint64_t* d_sigma_vals,
                       glp_plonk_circuit** out);
/* the round-1 form: every wire routed, no public inputs, no Poseidon rows, d_const_vals [3][n] = (q, c0, c1) */
int glp_plonk_setup(glp_ctx* ctx, uint32_t log_n, uint32_t n_wires, const uint64_t* d_const_vals, const uint64_t* d_sigma_vals,
                    uint32_t rate_bits, uint32_t cap_height, glp_plonk_circuit** out);
void glp_plonk_free(glp_plonk_circuit* ck);
/* d_wire_vals: [n_wires][n] witness values; h_public: n_public canonical words (NULL when the circuit has none).
 * Proof (little-endian u64 words): header (tag, log_n, n_wires, n_routed, rate_bits, cap_height, n_public, flags), the
 * public inputs, four caps, then the FRI opening proof (all batches at zeta, the Z batch also at w_n*zeta).  Header, public
 * inputs and the circuit's cap enter the transcript before the wires cap.  Free with glp_free_host. */
int glp_plonk_prove_ex(glp_ctx* ctx, glp_plonk_circuit* ck, const uint64_t* d_wire_vals, const uint64_t* h_public, uint32_t num_queries,
                       uint32_t pow_bits, uint8_t** proof, size_t* proof_len);
int glp_plonk_prove(glp_ctx* ctx, glp_plonk_circuit* ck, const uint64_t* d_wire_vals, uint32_t num_queries, uint32_t pow_bits,
                    uint8_t** proof, size_t* proof_len);
/* witness generation for Poseidon rows: for each of the n_rows row indices in d_rows (device, u32), wires 12..23 and 25..134 of that row
 * are computed from its wires 0..11 and its swap bit (wire 24), in place in d_wire_vals [n_wires][2^log_n].  Stream-ordered. */
int glp_poseidon_gate_fill_rows(glp_ctx* ctx, uint64_t* d_wire_vals, uint32_t log_n, uint32_t n_wires, const uint32_t* d_rows,
                                uint32_t n_rows);
/* witness generation for SHA rows: for row d_rows[k] of kind d_kinds[k] (GLP_SHA_ROW_*; both device, u32) the bit wires 12..143 are computed
 * from the routed words in wires 0..11 (inputs AND outputs of the row: the witness program computed the outputs), in place.  Stream-ordered. */
int glp_sha_gate_fill_rows(glp_ctx* ctx, uint64_t* d_wire_vals, uint32_t log_n, uint32_t n_wires, const uint32_t* d_rows,
                           const uint32_t* d_kinds, uint32_t n_rows);

/* Parity hook for rows a6 / a7: the prover's intermediate stages for CALLER-CHOSEN challenges, so the HIP output can be
 * compared directly with an independent restatement (tests/plonk_ref.py) and not only through accepted proofs.
 *   GLP_DEBUG_ZS:       h_challenges = beta[2], gamma[2];           d_out [2*M][n]  Z then the M-1 partial products per
 *                       challenge, values on the trace domain (M = n_routed / 8)
 *   GLP_DEBUG_QUOTIENT: h_challenges = beta[2], gamma[2], alpha[2]; d_out [2][8n]   quotient evaluations on the coset
 *                       7*<w_8n>, bit-reversed index order (before the division into chunks)
 * Synchronous.  The prover itself never calls it. */
#define GLP_DEBUG_ZS 0
#define GLP_DEBUG_QUOTIENT 1
int glp_plonk_debug_stage(glp_ctx* ctx, glp_plonk_circuit* ck, const uint64_t* d_wire_vals, const uint64_t* h_public, int which,
                          const uint64_t* h_challenges, uint64_t* d_out);

/* the circuit's preprocessed commitment (cap of the constants + sigmas batch) = its verifying key:
 * copies min(*n_words, needed) u64 to h_cap and stores the needed count in *n_words */
int glp_plonk_circuit_cap(glp_plonk_circuit* ck, uint64_t* h_cap, size_t* n_words);

/* ---- verification (rows a5, a8, a12 from the consuming side; the Reduce step of row a11 as far as this
 *      build goes; upstream names recalled: plonk::verifier::verify, fri::verifier::verify_fri_proof).
 * Host arithmetic with the ctx's Poseidon constants (a verification is a few thousand permutations with
 * a serial transcript).  proof: 8-byte aligned little-endian u64 words as written by the provers above.
 * Returns GLP_OK when the proof is accepted, GLP_E_REJECT when it is not (glp_last_error says why),
 * other codes for bad arguments.  min_queries / min_pow_bits: the security parameters the caller
 * requires (a proof declaring fewer is rejected). */
int glp_fri_verify(glp_ctx* ctx, const uint8_t* h_proof, size_t proof_len, uint32_t min_queries, uint32_t min_pow_bits);
/* A stand-alone FRI proof says "polynomials of degree < 2^log_n behind THESE caps take THESE values at THESE points"; all of that
 * comes from the proof, so OK/REJECT alone does not tell the caller which statement was proven.  The _ex form also requires a
 * minimum code rate (rate_bits >= min_rate_bits; rate_bits = 0 is rejected by every entry point: at rate 1 any claimed value
 * passes) and returns the statement.  The caller MUST compare it with the one it expects (log_n, the caps, the points, the
 * opened values): caps = words [caps_word_off, + n_batches*cap_words) of the proof, openings = n_openings (a, b) pairs from
 * openings_word_off in (point, batch, polynomial) order; point p is zeta * point_mult[p], zeta derived from the transcript. */
typedef struct {
    uint32_t log_n, rate_bits, cap_height, n_batches, n_points, num_queries, pow_bits, total_polys;
    uint64_t shift;
    uint64_t zeta[2];
    uint64_t point_mult[4];
    size_t caps_word_off, cap_words;
    size_t openings_word_off, n_openings;
    uint32_t n_polys[64], open_mask[64];
} glp_fri_statement;
int glp_fri_verify_ex(glp_ctx* ctx, const uint8_t* h_proof, size_t proof_len, uint32_t min_queries, uint32_t min_pow_bits,
                      uint32_t min_rate_bits, glp_fri_statement* statement /* may be NULL */);
/* h_circuit_cap (from glp_plonk_circuit_cap; may be NULL = do not bind to a circuit) */
int glp_plonk_verify(glp_ctx* ctx, const uint8_t* h_proof, size_t proof_len, const uint64_t* h_circuit_cap, size_t cap_words,
                     uint32_t min_queries, uint32_t min_pow_bits);
/* ... and to a statement: h_public (NULL = do not bind) must equal the proof's public inputs word for word.  A proof is about
 * (circuit, public inputs): a verifier that passes NULL for either accepts proofs of OTHER statements. */
int glp_plonk_verify_ex(glp_ctx* ctx, const uint8_t* h_proof, size_t proof_len, const uint64_t* h_circuit_cap, size_t cap_words,
                        const uint64_t* h_public, size_t n_public, uint32_t min_queries, uint32_t min_pow_bits);
/* the public inputs a proof carries: copies min(*n_words, n_public) words to h_out (may be NULL with *n_words = 0) and stores
 * n_public in *n_words.  No verification: GLP_E_INVALID when the bytes are not a circuit proof of this format. */
int glp_plonk_proof_public_inputs(const uint8_t* h_proof, size_t proof_len, uint64_t* h_out, size_t* n_words);
/* 4-word Poseidon digest of a circuit proof's statement and commitments: hash_no_pad(header || public inputs || the four caps) —
 * the leaf value of the Reduce step's aggregation tree (0-kno-blobstreamx_amd/recursion.py).  No verification. */
int glp_plonk_proof_digest(glp_ctx* ctx, const uint8_t* h_proof, size_t proof_len, uint64_t* h_out4);
/* Witness evaluator for circuits recorded by the host builder (0-kno-blobstreamx_amd/recursion.py::WitnessProgram): a straight-line program
 * (arithmetic gates, inputs, bit extractions, inverses, Poseidon permutations; encoding in csrc/verify.hip) computes every variable of the circuit
 * from its inputs in one forward pass, then the copy constraints between different variables (eq_pairs, 2 indices each) are checked:
 * GLP_E_REJECT + *first_bad when the witness does not satisfy the circuit.  Host arithmetic: the chain is sequential, like the transcript. */
int glp_witness_eval(const uint64_t* h_rc, const uint64_t* h_mds_circ, const uint64_t* h_mds_diag, const uint64_t* prog, size_t prog_words,
                     const uint64_t* inputs, size_t n_inputs, uint64_t* values, size_t n_values, const uint64_t* eq_pairs, size_t n_eq,
                     size_t* first_bad);
/* ... on several host threads: seg_bounds = n_seg + 1 ascending word offsets (op boundaries); the ops of [seg_bounds[k], seg_bounds[k+1]) are
 * n_seg mutually independent segments (each reads only what the prefix [0, seg_bounds[0]) or itself wrote — e.g. the verifier sub-circuits of
 * the proofs a recursion node checks), run on up to n_threads threads; the tail after seg_bounds[n_seg] runs last.  The independence claim is
 * checked while running: GLP_E_INVALID when a segment reads another segment's variable.  seg_bounds NULL / n_seg < 2 / n_threads < 2 = serial. */
int glp_witness_eval_mt(const uint64_t* h_rc, const uint64_t* h_mds_circ, const uint64_t* h_mds_diag, const uint64_t* prog, size_t prog_words,
                        const uint64_t* inputs, size_t n_inputs, uint64_t* values, size_t n_values, const uint64_t* eq_pairs, size_t n_eq,
                        size_t* first_bad, const uint64_t* seg_bounds, size_t n_seg, uint32_t n_threads);
/* Witness placement on the device: d_dst[i] = d_index[i] == 0xFFFFFFFF ? 0 : d_src[d_index[i]], i < n (d_index: uint32, every other entry < n_src
 * — checked on the device: an out-of-range entry writes 0 and the call returns GLP_E_INVALID).  With d_index = the circuit's cell -> variable map
 * (resident, built once per circuit) and d_src = the evaluated variables, this lays out the wire matrix without a host-side copy of it. */
int glp_gather_u64(glp_ctx* ctx, uint64_t* d_dst, const uint64_t* d_src, size_t n_src, const uint32_t* d_index, size_t n);
/* the Poseidon permutation on the host: n states of 12 canonical words, in place (same constants arguments as the host verifiers) */
int glp_poseidon_permute_host(const uint64_t* h_rc, const uint64_t* h_mds_circ, const uint64_t* h_mds_diag, uint64_t* states, size_t n);
int glp_plonk_proof_digest_host(const uint64_t* h_rc, const uint64_t* h_mds_circ, const uint64_t* h_mds_diag, const uint8_t* h_proof,
                                size_t proof_len, uint64_t* h_out4);

/* The same two verifiers for a host WITHOUT a GPU (a light client, CI): no ctx — the Poseidon constants are
 * passed explicitly (the arguments of glp_set_poseidon_constants: 360, 12, 12 words) and err (may be NULL)
 * receives the rejection reason.  GLP_E_INVALID for unusable arguments or non-canonical constants. */
int glp_fri_verify_host(const uint64_t* h_rc, const uint64_t* h_mds_circ, const uint64_t* h_mds_diag, const uint8_t* h_proof,
                        size_t proof_len, uint32_t min_queries, uint32_t min_pow_bits, char* err, size_t err_len);
int glp_fri_verify_host_ex(const uint64_t* h_rc, const uint64_t* h_mds_circ, const uint64_t* h_mds_diag, const uint8_t* h_proof,
                           size_t proof_len, uint32_t min_queries, uint32_t min_pow_bits, uint32_t min_rate_bits,
                           glp_fri_statement* statement, char* err, size_t err_len);
int glp_plonk_verify_host(const uint64_t* h_rc, const uint64_t* h_mds_circ, const uint64_t* h_mds_diag, const uint8_t* h_proof,
                          size_t proof_len, const uint64_t* h_circuit_cap, size_t cap_words, uint32_t min_queries,
                          uint32_t min_pow_bits, char* err, size_t err_len);
int glp_plonk_verify_host_ex(const uint64_t* h_rc, const uint64_t* h_mds_circ, const uint64_t* h_mds_diag, const uint8_t* h_proof,
                             size_t proof_len, const uint64_t* h_circuit_cap, size_t cap_words, const uint64_t* h_public, size_t n_public,
                             uint32_t min_queries, uint32_t min_pow_bits, char* err, size_t err_len);

/* ---- witness generation (rows a9; upstream names recalled: curta SHA-256/SHA-512 chips) */
/* n_msgs messages, each already padded to blocks_per_msg 64-byte blocks, [n_msgs][blocks*64].
 * d_digests: [n_msgs][8] u32 (big-endian words as u32).  d_trace (may be NULL):
 * [n_msgs][blocks][576] u32 = 64 schedule words then 64x8 round states. */
int glp_sha256_trace(glp_ctx* ctx, const uint8_t* d_blocks, uint64_t n_msgs, uint32_t blocks_per_msg,
                     uint32_t* d_digests, uint32_t* d_trace);
/* 128-byte blocks; digests [n_msgs][8] u64; trace [n_msgs][blocks][720] u64 */
int glp_sha512_trace(glp_ctx* ctx, const uint8_t* d_blocks, uint64_t n_msgs, uint32_t blocks_per_msg,
                     uint64_t* d_digests, uint64_t* d_trace);

/* Ed25519 verification witness (row a10; upstream name recalled: curta Ed25519 gadget), RFC 8032
 * cofactorless check [S]B = R + [k]A.  Per signature GLP_ED25519_RECORD_WORDS u64:
 * [0] valid, [1..4] k = SHA512(R||A||M) mod L, then affine Ax, Ay, Rx, Ry, P1x, P1y (= [S]B),
 * P2x, P2y (= [k]A), each 4 little-endian words.  The message bytes of signature i are
 * d_msgs[i*msg_stride .. + d_lens[i]); a row with d_lens[i] > msg_stride is invalid input: its record is all zero
 * (valid = 0) and nothing is read.  Stream-ordered (glp_sync before reading d_out). */
#define GLP_ED25519_RECORD_WORDS 37
int glp_ed25519_witness(glp_ctx* ctx, const uint8_t* d_pubs, const uint8_t* d_sigs, const uint8_t* d_msgs, uint32_t msg_stride,
                        const uint32_t* d_lens, uint64_t n, uint64_t* d_out);

/* Tendermint "simple" Merkle root (RFC 6962 prefixes) of n leaves of leaf_len (<= 118) bytes each:
 * the validator-set / header hashing of BASELINE configs[0].  Synchronous; h_root32 = 32 bytes. */
int glp_tm_merkle_root(glp_ctx* ctx, const uint8_t* d_leaves, uint32_t leaf_len, uint64_t n, uint8_t* h_root32);
/* the same tree over leaves of different lengths (protobuf-encoded validators, header fields):
 * leaf i = d_data[d_offsets[i] .. d_offsets[i+1]), n + 1 non-decreasing byte offsets (device), the last one <= data_len
 * (checked: GLP_E_INVALID otherwise, nothing is read out of range) */
int glp_tm_merkle_root_var(glp_ctx* ctx, const uint8_t* d_data, uint64_t data_len, const uint64_t* d_offsets, uint64_t n,
                           uint8_t* h_root32);

/* ---- Multi-GPU exchange (row a11 / SURVEY §8e, §8(b) glp_allgather_proofs; upstream name recalled: plonky2x mapreduce) ----
 * Leaf subproofs shard one per GPU, one process (and one ctx) per GPU.  The exchange is ONE RCCL all-gather of fixed-size
 * blocks over xGMI, on the ctx's stream.  Bootstrap as NCCL does: rank 0 makes the id and hands its GLP_COMM_ID_BYTES bytes
 * to the other ranks out of band (file, socket, environment, MPI, torch.distributed ...); then EVERY rank calls
 * glp_comm_init with the same id (collective).  One communicator per ctx; glp_destroy tears it down.
 * Block format used by the MapReduce host (0-kno-blobstreamx_amd/mapreduce.py::pack_leaves): per leaf a 16-byte header
 * (u64 leaf index, u64 payload length; index 2^64-1 = unused row) then the proof zero-padded to the agreed length. */
#define GLP_COMM_ID_BYTES 128
int glp_comm_unique_id(uint8_t* id_out /* GLP_COMM_ID_BYTES */);
int glp_comm_init(glp_ctx* ctx, const uint8_t* id /* GLP_COMM_ID_BYTES */, int rank, int nranks);
int glp_comm_rank(glp_ctx* ctx, int* rank, int* nranks);
int glp_comm_destroy(glp_ctx* ctx);       /* the communicator is gone afterwards either way; non-OK = it did not go cleanly */
/* Staging for the exchange lives with the communicator (allocated by glp_comm_init for blocks up to 4 MiB), so a collective entry point
 * cannot fail locally before it enters RCCL and strand its peers.  Ranks about to exchange LARGER blocks call glp_comm_reserve first (no
 * collective inside) and agree on the result, e.g. through glp_allreduce_min_u64.  Any non-OK return from glp_allgather_proofs /
 * glp_allreduce_min_u64 means the ranks may be out of step: abort the job on every rank. */
int glp_comm_reserve(glp_ctx* ctx, size_t padded_len);
/* every rank passes padded_len bytes (host); h_all receives nranks * padded_len bytes, rank r's block at r * padded_len */
int glp_allgather_proofs(glp_ctx* ctx, const uint8_t* h_mine, size_t padded_len, uint8_t* h_all);
/* element-wise minimum over the ranks, in place (the Reduce step's verdicts); n <= 4096 words */
int glp_allreduce_min_u64(glp_ctx* ctx, uint64_t* h_io, size_t n);

#ifdef __cplusplus
}
#endif
#endif /* GLPROVER_H */
