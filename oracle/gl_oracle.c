/*
 * oracle/gl_oracle.c — CPU restatement of the prover hot path (TEST INFRASTRUCTURE ONLY).
 *
 * STATUS: PARITY UNPINNED against the reference.  /root/reference holds only
 * `.gitignore:1` and `changelog.md:1-2`; there is no reference source, test, golden
 * vector or lockfile to follow or to check against.  This file
 * therefore restates the PUBLISHED definitions the north_star names:
 *   - Goldilocks field p = 2^64 - 2^32 + 1, multiplicative generator 7,
 *     2^k-th root of unity w_k = 7^((p-1)/2^k)              (rows a1/a2 of SURVEY §8a)
 *   - DFT  X[k] = sum_j x[j] w_n^{jk}, natural order in/out; inverse scales by 1/n
 *   - coset LDE: zero-pad coefficients to n*2^rate_bits, scale coeff j by shift^j, DFT
 *   - Poseidon permutation STRUCTURE (width 12, x^7, 4+22+4 rounds, circulant+diag MDS,
 *     overwrite-mode sponge rate 8, 2-to-1 compression), constants INJECTED by caller
 *   - Merkle tree with cap, FRI fold, SHA-256/512 (FIPS 180-4), Tendermint simple Merkle
 *     (RFC-6962 prefixes), Ed25519 verify (RFC 8032).
 * It is pinned by: tests/golden/ JSON fixtures produced by Python big-int / hashlib scripts
 * (tests/golden/gen_golden.py), FIPS 180-4 and RFC 8032 known answers, and the naive
 * O(n^2) DFT below.  It is NOT pinned by anything from plonky2/plonky2x/curta.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library.  The product (0-kno-blobstreamx_amd/) never links or calls it.
 *
 * Deliberately written differently from the HIP product code: all field arithmetic goes
 * through unsigned __int128 and the `%` operator, so an error in the product's hand
 * reduction cannot be mirrored here.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef unsigned __int128 u128;
#define P 0xFFFFFFFF00000001ULL

/* ------------------------------------------------------------------ field (row a1) */
uint64_t orc_add(uint64_t a, uint64_t b) { return (uint64_t)(((u128)a + b) % P); }
uint64_t orc_sub(uint64_t a, uint64_t b) { return (uint64_t)(((u128)a + P - (b % P)) % P); }
uint64_t orc_mul(uint64_t a, uint64_t b) { return (uint64_t)(((u128)a * b) % P); }
uint64_t orc_pow(uint64_t a, uint64_t e) {
    uint64_t r = 1; a %= P;
    while (e) { if (e & 1) r = orc_mul(r, a); a = orc_mul(a, a); e >>= 1; }
    return r;
}
uint64_t orc_inv(uint64_t a) { return orc_pow(a, P - 2); }
/* primitive 2^k-th root of unity, k <= 32: 7^((p-1)/2^k) */
/* the primitive 2^k-th root of unity: a power of the two-adic generator (an element of order 2^32).  Default 7^((p-1)/2^32); the product
 * library fixes its generator at build time (csrc/gl_field.cuh GLP_TWO_ADIC_GENERATOR) and the tests hand the same value to the oracle. */
static uint64_t TWO_ADIC_GEN = 0;
void orc_set_two_adic_generator(uint64_t g) { TWO_ADIC_GEN = g % P; }
uint64_t orc_two_adic_generator(void) { if (!TWO_ADIC_GEN) TWO_ADIC_GEN = orc_pow(7, (P - 1) >> 32); return TWO_ADIC_GEN; }
uint64_t orc_root(unsigned k) { return orc_pow(orc_two_adic_generator(), 1ULL << (32 - k)); }
int orc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* cap the OpenMP team (bench.py: the box's cgroup CPU quota, not its core count, is what the job may use) */
void orc_set_num_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* ------------------------------------------------------------------ NTT (row a2) */
/* definition, O(n^2): the pin for the fast transform below */
void orc_dft_naive(const uint64_t *x, uint64_t *out, unsigned log_n, int inverse) {
    uint64_t n = 1ULL << log_n, w = orc_root(log_n);
    if (inverse) w = orc_inv(w);
    uint64_t ninv = orc_inv(n % P);
    for (uint64_t k = 0; k < n; k++) {
        uint64_t wk = orc_pow(w, k), acc = 0, t = 1;
        for (uint64_t j = 0; j < n; j++) { acc = orc_add(acc, orc_mul(x[j], t)); t = orc_mul(t, wk); }
        out[k] = inverse ? orc_mul(acc, ninv) : acc;
    }
}

static void bitrev_permute(uint64_t *a, unsigned log_n) {
    uint64_t n = 1ULL << log_n;
    for (uint64_t i = 0; i < n; i++) {
        uint64_t r = 0;
        for (unsigned b = 0; b < log_n; b++) r |= ((i >> b) & 1) << (log_n - 1 - b);
        if (r > i) { uint64_t t = a[i]; a[i] = a[r]; a[r] = t; }
    }
}

/* textbook iterative radix-2 decimation-in-time, in place, natural in / natural out.
 * tw = table of w^i, i < n/2 (shared across a batch). */
static void ntt_one(uint64_t *a, unsigned log_n, const uint64_t *tw) {
    uint64_t n = 1ULL << log_n;
    bitrev_permute(a, log_n);
    for (unsigned s = 1; s <= log_n; s++) {
        uint64_t m = 1ULL << s, half = m >> 1, step = n >> s;
        for (uint64_t k = 0; k < n; k += m)
            for (uint64_t j = 0; j < half; j++) {
                uint64_t u = a[k + j], v = orc_mul(a[k + j + half], tw[j * step]);
                a[k + j] = orc_add(u, v);
                a[k + j + half] = orc_sub(u, v);
            }
    }
}

/* batch of `batch` transforms, each n contiguous u64; OpenMP over the batch. */
void orc_ntt(uint64_t *data, unsigned log_n, uint64_t batch, int inverse) {
    uint64_t n = 1ULL << log_n;
    if (log_n == 0) return;
    uint64_t w = orc_root(log_n);
    if (inverse) w = orc_inv(w);
    uint64_t *tw = (uint64_t *)malloc(sizeof(uint64_t) * (n / 2 ? n / 2 : 1));
    tw[0] = 1;
    for (uint64_t i = 1; i < n / 2; i++) tw[i] = orc_mul(tw[i - 1], w);
    uint64_t ninv = orc_inv(n % P);
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t b = 0; b < (int64_t)batch; b++) {
        uint64_t *a = data + (uint64_t)b * n;
        ntt_one(a, log_n, tw);
        if (inverse) for (uint64_t i = 0; i < n; i++) a[i] = orc_mul(a[i], ninv);
    }
    free(tw);
}

/* One large transform using all cores (for the batch=1 CPU baseline): same butterflies,
 * stage loops parallelised.  Result identical to orc_ntt(batch=1). */
void orc_ntt_par(uint64_t *a, unsigned log_n, int inverse) {
    uint64_t n = 1ULL << log_n;
    if (log_n == 0) return;
    uint64_t w = orc_root(log_n);
    if (inverse) w = orc_inv(w);
    uint64_t *tw = (uint64_t *)malloc(sizeof(uint64_t) * (n / 2 ? n / 2 : 1));
    tw[0] = 1;
    for (uint64_t i = 1; i < n / 2; i++) tw[i] = orc_mul(tw[i - 1], w);
    bitrev_permute(a, log_n);
    for (unsigned s = 1; s <= log_n; s++) {
        uint64_t half = 1ULL << (s - 1), step = n >> s;
#pragma omp parallel for schedule(static)
        for (int64_t t = 0; t < (int64_t)(n / 2); t++) {
            uint64_t j = (uint64_t)t & (half - 1), k = ((uint64_t)t >> (s - 1)) << s;
            uint64_t u = a[k + j], v = orc_mul(a[k + j + half], tw[j * step]);
            a[k + j] = orc_add(u, v);
            a[k + j + half] = orc_sub(u, v);
        }
    }
    if (inverse) {
        uint64_t ninv = orc_inv(n % P);
#pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < (int64_t)n; i++) a[i] = orc_mul(a[i], ninv);
    }
    free(tw);
}

void orc_bitrev_rows(uint64_t *data, unsigned log_n, uint64_t batch) {
    for (uint64_t b = 0; b < batch; b++) bitrev_permute(data + (b << log_n), log_n);
}

/* coset LDE (row a2/a3): coeffs [batch][n] -> values [batch][n << rate_bits], natural order:
 * out[k] = sum_j coeffs[j] * shift^j * w_N^{jk},  N = n << rate_bits */
void orc_lde_coset(const uint64_t *coeffs, uint64_t *out, unsigned log_n, unsigned rate_bits,
                   uint64_t batch, uint64_t shift) {
    uint64_t n = 1ULL << log_n, N = n << rate_bits;
    uint64_t *sp = (uint64_t *)malloc(sizeof(uint64_t) * n);
    sp[0] = 1;
    for (uint64_t j = 1; j < n; j++) sp[j] = orc_mul(sp[j - 1], shift);
    for (uint64_t b = 0; b < batch; b++) {
        uint64_t *o = out + b * N;
        for (uint64_t j = 0; j < n; j++) o[j] = orc_mul(coeffs[b * n + j] % P, sp[j]);
        memset(o + n, 0, sizeof(uint64_t) * (N - n));
    }
    free(sp);
    orc_ntt(out, log_n + rate_bits, batch, 0);
}

/* [rows][cols] -> [cols][rows] */
void orc_transpose(const uint64_t *in, uint64_t *out, uint64_t rows, uint64_t cols) {
    for (uint64_t r = 0; r < rows; r++)
        for (uint64_t c = 0; c < cols; c++) out[c * rows + r] = in[r * cols + c];
}

/* ------------------------------------------------------------------ Poseidon (row a4) */
/* constants are injected: rc[30*12], mds_circ[12], mds_diag[12]. */
static uint64_t PS_RC[360], PS_CIRC[12], PS_DIAG[12];
void orc_poseidon_set_constants(const uint64_t *rc, const uint64_t *circ, const uint64_t *diag) {
    for (int i = 0; i < 360; i++) PS_RC[i] = rc[i] % P;
    for (int i = 0; i < 12; i++) { PS_CIRC[i] = circ[i] % P; PS_DIAG[i] = diag[i] % P; }
}
static uint64_t sbox7(uint64_t x) {
    uint64_t x2 = orc_mul(x, x), x4 = orc_mul(x2, x2), x3 = orc_mul(x2, x);
    return orc_mul(x4, x3);
}
static void mds_layer(uint64_t *s) {
    uint64_t r[12];
    for (int row = 0; row < 12; row++) {
        uint64_t acc = 0;
        for (int i = 0; i < 12; i++) acc = orc_add(acc, orc_mul(s[(i + row) % 12], PS_CIRC[i]));
        acc = orc_add(acc, orc_mul(s[row], PS_DIAG[row]));
        r[row] = acc;
    }
    memcpy(s, r, sizeof r);
}
/* naive (unoptimised) permutation: 4 full, 22 partial, 4 full rounds; every round adds
 * all 12 round constants, applies x^7 (all lanes in a full round, lane 0 in a partial
 * round), then the MDS layer. */
void orc_poseidon_permute(uint64_t *s) {
    int rnd = 0;
    for (int phase = 0; phase < 3; phase++) {
        int cnt = phase == 1 ? 22 : 4;
        for (int r = 0; r < cnt; r++, rnd++) {
            for (int i = 0; i < 12; i++) s[i] = orc_add(s[i] % P, PS_RC[rnd * 12 + i]);
            if (phase == 1) s[0] = sbox7(s[0]);
            else for (int i = 0; i < 12; i++) s[i] = sbox7(s[i]);
            mds_layer(s);
        }
    }
}
/* overwrite-mode sponge, rate 8, 4-element digest */
void orc_hash_no_pad(const uint64_t *in, uint64_t len, uint64_t *out4) {
    uint64_t s[12] = {0};
    for (uint64_t off = 0; off < len; off += 8) {
        uint64_t c = len - off < 8 ? len - off : 8;
        for (uint64_t i = 0; i < c; i++) s[i] = in[off + i] % P;
        orc_poseidon_permute(s);
    }
    memcpy(out4, s, 4 * sizeof(uint64_t));
}
/* leaves of <= 4 elements are their own (zero-padded) digest */
void orc_hash_or_noop(const uint64_t *in, uint64_t len, uint64_t *out4) {
    if (len <= 4) { for (int i = 0; i < 4; i++) out4[i] = (uint64_t)i < len ? in[i] % P : 0; }
    else orc_hash_no_pad(in, len, out4);
}
void orc_two_to_one(const uint64_t *l, const uint64_t *r, uint64_t *out4) {
    uint64_t s[12] = {0};
    memcpy(s, l, 32); memcpy(s + 4, r, 32);
    orc_poseidon_permute(s);
    memcpy(out4, s, 32);
}
/* Merkle tree over 2^log_leaves leaves of leaf_len elements each ([leaf][leaf_len]).
 * digests: level 0 = leaf digests (2^log_leaves * 4), level 1 = parents, ... stored
 * level after level down to the cap level; cap = the 2^cap_h nodes of level
 * (log_leaves - cap_h), which are also the last level written to `digests`.
 * digests must hold 4 * (2^(log_leaves+1) - 2^cap_h) u64. */
void orc_merkle(const uint64_t *leaves, uint64_t leaf_len, unsigned log_leaves, unsigned cap_h,
                uint64_t *digests, uint64_t *cap) {
    uint64_t nl = 1ULL << log_leaves;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < (int64_t)nl; i++) orc_hash_or_noop(leaves + (uint64_t)i * leaf_len, leaf_len, digests + 4 * i);
    uint64_t *prev = digests, cnt = nl;
    for (unsigned lvl = log_leaves; lvl > cap_h; lvl--) {
        uint64_t *cur = prev + 4 * cnt;
        cnt >>= 1;
#pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < (int64_t)cnt; i++) orc_two_to_one(prev + 8 * i, prev + 8 * i + 4, cur + 4 * i);
        prev = cur;
    }
    memcpy(cap, prev, 4 * sizeof(uint64_t) * (1ULL << cap_h));
}

/* ------------------------------------------------------------------ quadratic extension + FRI fold (row a8) */
/* F_p[X]/(X^2 - 7): element (a0, a1) = a0 + a1*X */
void orc_ext_mul(const uint64_t *a, const uint64_t *b, uint64_t *o) {
    uint64_t c0 = orc_add(orc_mul(a[0], b[0]), orc_mul(7, orc_mul(a[1], b[1])));
    uint64_t c1 = orc_add(orc_mul(a[0], b[1]), orc_mul(a[1], b[0]));
    o[0] = c0; o[1] = c1;
}
/* arity-2 FRI fold of evaluations over the coset shift*<w_{log_n}>, given in
 * BIT-REVERSED order (so that x and -x are adjacent): for pair i,
 *   x_i = shift * w^{bitrev(2i)},  out[i] = (f(x)+f(-x))/2 + beta*(f(x)-f(-x))/(2x)
 * evals: [n][2] ext elements; out: [n/2][2]; result domain is shift^2 * <w^2> bit-reversed. */
void orc_fri_fold2(const uint64_t *evals, uint64_t *out, unsigned log_n, uint64_t shift,
                   const uint64_t *beta) {
    uint64_t n = 1ULL << log_n, w = orc_root(log_n), inv2 = orc_inv(2);
    for (uint64_t i = 0; i < n / 2; i++) {
        uint64_t idx = 2 * i, r = 0;
        for (unsigned b = 0; b < log_n; b++) r |= ((idx >> b) & 1) << (log_n - 1 - b);
        uint64_t x = orc_mul(shift, orc_pow(w, r));
        uint64_t xinv = orc_inv(x);
        const uint64_t *f0 = evals + 4 * i, *f1 = evals + 4 * i + 2;
        uint64_t s[2] = { orc_mul(orc_add(f0[0], f1[0]), inv2), orc_mul(orc_add(f0[1], f1[1]), inv2) };
        uint64_t d[2] = { orc_mul(orc_mul(orc_sub(f0[0], f1[0]), inv2), xinv),
                          orc_mul(orc_mul(orc_sub(f0[1], f1[1]), inv2), xinv) };
        uint64_t bd[2];
        orc_ext_mul(beta, d, bd);
        out[2 * i] = orc_add(s[0], bd[0]);
        out[2 * i + 1] = orc_add(s[1], bd[1]);
    }
}

/* ------------------------------------------------------------------ SHA-2 (row a9) */
static const uint32_t K256[64] = {
    0x428a2f98,0x71374491,0xb5c0fbcf,0xe9b5dba5,0x3956c25b,0x59f111f1,0x923f82a4,0xab1c5ed5,
    0xd807aa98,0x12835b01,0x243185be,0x550c7dc3,0x72be5d74,0x80deb1fe,0x9bdc06a7,0xc19bf174,
    0xe49b69c1,0xefbe4786,0x0fc19dc6,0x240ca1cc,0x2de92c6f,0x4a7484aa,0x5cb0a9dc,0x76f988da,
    0x983e5152,0xa831c66d,0xb00327c8,0xbf597fc7,0xc6e00bf3,0xd5a79147,0x06ca6351,0x14292967,
    0x27b70a85,0x2e1b2138,0x4d2c6dfc,0x53380d13,0x650a7354,0x766a0abb,0x81c2c92e,0x92722c85,
    0xa2bfe8a1,0xa81a664b,0xc24b8b70,0xc76c51a3,0xd192e819,0xd6990624,0xf40e3585,0x106aa070,
    0x19a4c116,0x1e376c08,0x2748774c,0x34b0bcb5,0x391c0cb3,0x4ed8aa4a,0x5b9cca4f,0x682e6ff3,
    0x748f82ee,0x78a5636f,0x84c87814,0x8cc70208,0x90befffa,0xa4506ceb,0xbef9a3f7,0xc67178f2};
static uint32_t ror32(uint32_t x, int r) { return (x >> r) | (x << (32 - r)); }
/* one compression; if trace != NULL writes 64 message-schedule words then 64*8 state words
 * (state AFTER each round, a..h) = 576 u32 per block (FIPS 180-4 §6.2.2). */
void orc_sha256_compress(uint32_t *h, const uint8_t *blk, uint32_t *trace) {
    uint32_t w[64], s[8];
    for (int i = 0; i < 16; i++) w[i] = (uint32_t)blk[4*i] << 24 | (uint32_t)blk[4*i+1] << 16 | (uint32_t)blk[4*i+2] << 8 | blk[4*i+3];
    for (int i = 16; i < 64; i++) {
        uint32_t s0 = ror32(w[i-15],7) ^ ror32(w[i-15],18) ^ (w[i-15] >> 3);
        uint32_t s1 = ror32(w[i-2],17) ^ ror32(w[i-2],19) ^ (w[i-2] >> 10);
        w[i] = w[i-16] + s0 + w[i-7] + s1;
    }
    memcpy(s, h, 32);
    for (int i = 0; i < 64; i++) {
        uint32_t S1 = ror32(s[4],6) ^ ror32(s[4],11) ^ ror32(s[4],25);
        uint32_t ch = (s[4] & s[5]) ^ (~s[4] & s[6]);
        uint32_t t1 = s[7] + S1 + ch + K256[i] + w[i];
        uint32_t S0 = ror32(s[0],2) ^ ror32(s[0],13) ^ ror32(s[0],22);
        uint32_t mj = (s[0] & s[1]) ^ (s[0] & s[2]) ^ (s[1] & s[2]);
        uint32_t t2 = S0 + mj;
        s[7] = s[6]; s[6] = s[5]; s[5] = s[4]; s[4] = s[3] + t1;
        s[3] = s[2]; s[2] = s[1]; s[1] = s[0]; s[0] = t1 + t2;
        if (trace) memcpy(trace + 64 + 8 * i, s, 32);
    }
    if (trace) memcpy(trace, w, 256);
    for (int i = 0; i < 8; i++) h[i] += s[i];
}
static const uint32_t H256[8] = {0x6a09e667,0xbb67ae85,0x3c6ef372,0xa54ff53a,0x510e527f,0x9b05688c,0x1f83d9ab,0x5be0cd19};
/* padded length in 64-byte blocks */
uint64_t orc_sha256_nblocks(uint64_t len) { return (len + 9 + 63) / 64; }
/* full hash; trace (optional) receives 576 u32 per block */
void orc_sha256(const uint8_t *msg, uint64_t len, uint8_t *out32, uint32_t *trace) {
    uint32_t h[8]; memcpy(h, H256, 32);
    uint64_t nb = orc_sha256_nblocks(len);
    uint8_t *buf = (uint8_t *)calloc(nb, 64);
    memcpy(buf, msg, len); buf[len] = 0x80;
    uint64_t bits = len * 8;
    for (int i = 0; i < 8; i++) buf[nb * 64 - 1 - i] = (uint8_t)(bits >> (8 * i));
    for (uint64_t b = 0; b < nb; b++) orc_sha256_compress(h, buf + 64 * b, trace ? trace + 576 * b : NULL);
    free(buf);
    for (int i = 0; i < 8; i++) { out32[4*i] = h[i] >> 24; out32[4*i+1] = h[i] >> 16; out32[4*i+2] = h[i] >> 8; out32[4*i+3] = h[i]; }
}

static const uint64_t K512[80] = {
    0x428a2f98d728ae22ULL,0x7137449123ef65cdULL,0xb5c0fbcfec4d3b2fULL,0xe9b5dba58189dbbcULL,0x3956c25bf348b538ULL,0x59f111f1b605d019ULL,0x923f82a4af194f9bULL,0xab1c5ed5da6d8118ULL,
    0xd807aa98a3030242ULL,0x12835b0145706fbeULL,0x243185be4ee4b28cULL,0x550c7dc3d5ffb4e2ULL,0x72be5d74f27b896fULL,0x80deb1fe3b1696b1ULL,0x9bdc06a725c71235ULL,0xc19bf174cf692694ULL,
    0xe49b69c19ef14ad2ULL,0xefbe4786384f25e3ULL,0x0fc19dc68b8cd5b5ULL,0x240ca1cc77ac9c65ULL,0x2de92c6f592b0275ULL,0x4a7484aa6ea6e483ULL,0x5cb0a9dcbd41fbd4ULL,0x76f988da831153b5ULL,
    0x983e5152ee66dfabULL,0xa831c66d2db43210ULL,0xb00327c898fb213fULL,0xbf597fc7beef0ee4ULL,0xc6e00bf33da88fc2ULL,0xd5a79147930aa725ULL,0x06ca6351e003826fULL,0x142929670a0e6e70ULL,
    0x27b70a8546d22ffcULL,0x2e1b21385c26c926ULL,0x4d2c6dfc5ac42aedULL,0x53380d139d95b3dfULL,0x650a73548baf63deULL,0x766a0abb3c77b2a8ULL,0x81c2c92e47edaee6ULL,0x92722c851482353bULL,
    0xa2bfe8a14cf10364ULL,0xa81a664bbc423001ULL,0xc24b8b70d0f89791ULL,0xc76c51a30654be30ULL,0xd192e819d6ef5218ULL,0xd69906245565a910ULL,0xf40e35855771202aULL,0x106aa07032bbd1b8ULL,
    0x19a4c116b8d2d0c8ULL,0x1e376c085141ab53ULL,0x2748774cdf8eeb99ULL,0x34b0bcb5e19b48a8ULL,0x391c0cb3c5c95a63ULL,0x4ed8aa4ae3418acbULL,0x5b9cca4f7763e373ULL,0x682e6ff3d6b2b8a3ULL,
    0x748f82ee5defb2fcULL,0x78a5636f43172f60ULL,0x84c87814a1f0ab72ULL,0x8cc702081a6439ecULL,0x90befffa23631e28ULL,0xa4506cebde82bde9ULL,0xbef9a3f7b2c67915ULL,0xc67178f2e372532bULL,
    0xca273eceea26619cULL,0xd186b8c721c0c207ULL,0xeada7dd6cde0eb1eULL,0xf57d4f7fee6ed178ULL,0x06f067aa72176fbaULL,0x0a637dc5a2c898a6ULL,0x113f9804bef90daeULL,0x1b710b35131c471bULL,
    0x28db77f523047d84ULL,0x32caab7b40c72493ULL,0x3c9ebe0a15c9bebcULL,0x431d67c49c100d4cULL,0x4cc5d4becb3e42b6ULL,0x597f299cfc657e2aULL,0x5fcb6fab3ad6faecULL,0x6c44198c4a475817ULL};
static uint64_t ror64(uint64_t x, int r) { return (x >> r) | (x << (64 - r)); }
/* trace: 80 schedule words then 80*8 state words = 720 u64 per block */
void orc_sha512_compress(uint64_t *h, const uint8_t *blk, uint64_t *trace) {
    uint64_t w[80], s[8];
    for (int i = 0; i < 16; i++) { w[i] = 0; for (int b = 0; b < 8; b++) w[i] = (w[i] << 8) | blk[8*i+b]; }
    for (int i = 16; i < 80; i++) {
        uint64_t s0 = ror64(w[i-15],1) ^ ror64(w[i-15],8) ^ (w[i-15] >> 7);
        uint64_t s1 = ror64(w[i-2],19) ^ ror64(w[i-2],61) ^ (w[i-2] >> 6);
        w[i] = w[i-16] + s0 + w[i-7] + s1;
    }
    memcpy(s, h, 64);
    for (int i = 0; i < 80; i++) {
        uint64_t S1 = ror64(s[4],14) ^ ror64(s[4],18) ^ ror64(s[4],41);
        uint64_t ch = (s[4] & s[5]) ^ (~s[4] & s[6]);
        uint64_t t1 = s[7] + S1 + ch + K512[i] + w[i];
        uint64_t S0 = ror64(s[0],28) ^ ror64(s[0],34) ^ ror64(s[0],39);
        uint64_t mj = (s[0] & s[1]) ^ (s[0] & s[2]) ^ (s[1] & s[2]);
        uint64_t t2 = S0 + mj;
        s[7] = s[6]; s[6] = s[5]; s[5] = s[4]; s[4] = s[3] + t1;
        s[3] = s[2]; s[2] = s[1]; s[1] = s[0]; s[0] = t1 + t2;
        if (trace) memcpy(trace + 80 + 8 * i, s, 64);
    }
    if (trace) memcpy(trace, w, 640);
    for (int i = 0; i < 8; i++) h[i] += s[i];
}
static const uint64_t H512[8] = {0x6a09e667f3bcc908ULL,0xbb67ae8584caa73bULL,0x3c6ef372fe94f82bULL,0xa54ff53a5f1d36f1ULL,0x510e527fade682d1ULL,0x9b05688c2b3e6c1fULL,0x1f83d9abfb41bd6bULL,0x5be0cd19137e2179ULL};
uint64_t orc_sha512_nblocks(uint64_t len) { return (len + 17 + 127) / 128; }
void orc_sha512(const uint8_t *msg, uint64_t len, uint8_t *out64, uint64_t *trace) {
    uint64_t h[8]; memcpy(h, H512, 64);
    uint64_t nb = orc_sha512_nblocks(len);
    uint8_t *buf = (uint8_t *)calloc(nb, 128);
    memcpy(buf, msg, len); buf[len] = 0x80;
    uint64_t bits = len * 8;
    for (int i = 0; i < 8; i++) buf[nb * 128 - 1 - i] = (uint8_t)(bits >> (8 * i));
    for (uint64_t b = 0; b < nb; b++) orc_sha512_compress(h, buf + 128 * b, trace ? trace + 720 * b : NULL);
    free(buf);
    for (int i = 0; i < 8; i++) for (int b = 0; b < 8; b++) out64[8*i+b] = (uint8_t)(h[i] >> (56 - 8*b));
}

/* Tendermint "simple" Merkle tree over byte-string leaves (RFC 6962 domain separation):
 * leafHash = SHA256(0x00 || leaf), innerHash = SHA256(0x01 || l || r), split at the
 * largest power of two strictly less than n; empty tree = SHA256("").
 * leaves are `n` items of fixed `leaf_len` bytes. */
static void tm_root_rec(const uint8_t *leaves, uint64_t leaf_len, uint64_t n, uint8_t *out) {
    if (n == 1) {
        uint8_t *b = (uint8_t *)malloc(leaf_len + 1);
        b[0] = 0; memcpy(b + 1, leaves, leaf_len);
        orc_sha256(b, leaf_len + 1, out, NULL);
        free(b); return;
    }
    uint64_t k = 1; while (k * 2 < n) k *= 2;
    uint8_t b[65]; b[0] = 1;
    tm_root_rec(leaves, leaf_len, k, b + 1);
    tm_root_rec(leaves + k * leaf_len, leaf_len, n - k, b + 33);
    orc_sha256(b, 65, out, NULL);
}
void orc_tm_merkle_root(const uint8_t *leaves, uint64_t leaf_len, uint64_t n, uint8_t *out32) {
    if (n == 0) { orc_sha256((const uint8_t *)"", 0, out32, NULL); return; }
    tm_root_rec(leaves, leaf_len, n, out32);
}
