/*
 * oracle/gl_fast.c — the CPU BASELINE leg of the oracle (TEST/BENCH INFRASTRUCTURE ONLY).
 *
 * Same transform as orc_ntt in gl_oracle.c (which is deliberately naive: unsigned __int128
 * and `%`), restated with the arithmetic a tuned CPU prover would use, so that
 * bench.py's "cpu_baseline" is a fair CPU number rather than a strawman:
 *   - Goldilocks product reduced by hand (2^64 = 2^32 - 1, 2^96 = -1), no division;
 *   - iterative radix-2 DIT with a precomputed twiddle table per stage laid out
 *     contiguously (unit-stride twiddle reads), bit-reversal by table;
 *   - branch-free field operations (round 3: the data-dependent corrections of add / sub / mul were branches — a coin flip per
 *     butterfly — and most of the run time: 3.2 -> 7.2-7.7 GB/s on the GPU box's 16-thread quota; a four-step, cache-blocked form was
 *     measured on the same box and is no faster: the loop is arithmetic-bound, not memory-bound);
 *   - OpenMP across the batch (one transform per thread at a time).
 * tests/test_oracle.py checks it bit-for-bit against orc_ntt and the golden vectors.
 * PARITY UNPINNED w.r.t. the reference (no reference source exists in the mount); this is
 * the "port" kind of cpu_baseline, not plonky2.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;
#define P 0xFFFFFFFF00000001ULL
#define EPS 0xFFFFFFFFULL

/* branch-free: the corrections below depend on the data (a coin flip per butterfly for random inputs); as branches they were most of the time */
static inline uint64_t f_add(uint64_t a, uint64_t b) {
    uint64_t s = a + b;
    return s + ((0 - (uint64_t)((s < a) | (s >= P))) & EPS);
}
static inline uint64_t f_sub(uint64_t a, uint64_t b) {
    return (a - b) - ((0 - (uint64_t)(a < b)) & EPS);
}
static inline uint64_t f_mul(uint64_t a, uint64_t b) {
    u128 x = (u128)a * b;
    uint64_t lo = (uint64_t)x, hi = (uint64_t)(x >> 64);
    uint64_t hh = hi >> 32, hl = hi & EPS;
    uint64_t t0 = (lo - hh) - ((0 - (uint64_t)(lo < hh)) & EPS);
    uint64_t t1 = hl * EPS;
    uint64_t r = t0 + t1;
    r += (0 - (uint64_t)(r < t1)) & EPS;
    return r - ((0 - (uint64_t)(r >= P)) & P);
}
static uint64_t f_pow(uint64_t a, uint64_t e) {
    uint64_t r = 1;
    while (e) { if (e & 1) r = f_mul(r, a); a = f_mul(a, a); e >>= 1; }
    return r;
}

/* in-place radix-2 DIT of one contiguous 2^log_m-point vector: tw = per-stage twiddles (stage s at tw + 2^(s-1) - 1), rev = bit-reversal table */
static void fft_small(uint64_t *a, unsigned log_m, const uint64_t *tw, const uint32_t *rev) {
    const uint64_t m_ = 1ULL << log_m;
    for (uint64_t i = 0; i < m_; i++) { uint32_t r = rev[i]; if (r > i) { uint64_t t = a[i]; a[i] = a[r]; a[r] = t; } }
    for (unsigned s = 1; s <= log_m; s++) {
        const uint64_t half = 1ULL << (s - 1), m = half << 1;
        const uint64_t *ts = tw + half - 1;
        for (uint64_t k = 0; k < m_; k += m)
            for (uint64_t j = 0; j < half; j++) {
                uint64_t u = a[k + j], v = f_mul(a[k + j + half], ts[j]);
                a[k + j] = f_add(u, v);
                a[k + j + half] = f_sub(u, v);
            }
    }
}
static void fill_tables(unsigned log_m, uint64_t w, uint64_t *tw, uint32_t *rev) {
    const uint64_t m = 1ULL << log_m;
    for (unsigned s = 1; s <= log_m; s++) {
        uint64_t half = 1ULL << (s - 1), ws = f_pow(w, m >> s), t = 1;
        for (uint64_t j = 0; j < half; j++) { tw[half - 1 + j] = t; t = f_mul(t, ws); }
    }
    rev[0] = 0;
    for (uint64_t i = 1; i < m; i++) rev[i] = (rev[i >> 1] >> 1) | ((uint32_t)(i & 1) << (log_m - 1));
}

void orc_ntt_fast(uint64_t *data, unsigned log_n, uint64_t batch, int inverse) {
    if (log_n == 0) return;
    const uint64_t n = 1ULL << log_n;
    extern uint64_t orc_two_adic_generator(void);        /* gl_oracle.c: the tests set it to the product build's generator */
    uint64_t w = f_pow(orc_two_adic_generator(), 1ULL << (32 - log_n));
    if (inverse) w = f_pow(w, P - 2);
    const uint64_t ninv = f_pow(n % P, P - 2);
    /* per-stage twiddles, stage s (half = 2^(s-1)) stored at tw + half - 1 */
    uint64_t *tw = (uint64_t *)malloc(sizeof(uint64_t) * n);
    uint32_t *rev = (uint32_t *)malloc(sizeof(uint32_t) * n);
    fill_tables(log_n, w, tw, rev);
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t b = 0; b < (int64_t)batch; b++) {
        uint64_t *a = data + (uint64_t)b * n;
        fft_small(a, log_n, tw, rev);
        if (inverse) for (uint64_t i = 0; i < n; i++) a[i] = f_mul(a[i], ninv);
    }
    free(tw);
    free(rev);
}

/* ------------------------------------------------------------------ the commitment stage, for the prove path's CPU baseline
 * PolynomialBatch.from_values restated with the fast arithmetic above: values [n_polys][n] -> coefficients (inverse transform) ->
 * coset LDE on shift * <w_N> (N = n << rate_bits, natural order, then bit-reversed index order as the prover commits) -> Merkle tree over the
 * COLUMNS (leaf i = the n_polys values at index i; overwrite-mode sponge, rate 8; two-to-one compression; cap of 2^cap_h nodes).
 * Same permutation as orc_poseidon_permute (4 full, 22 partial, 4 full rounds; add the round's 12 constants, x^7, circulant + diagonal MDS),
 * constants injected.  tests/test_oracle.py checks the cap against the naive oracle's composition (orc_ntt / orc_lde_coset / orc_merkle). */
static uint64_t FRC[360], FCIRC[12], FDIAG[12];
void orc_fast_set_poseidon(const uint64_t *rc, const uint64_t *circ, const uint64_t *diag) {
    for (int i = 0; i < 360; i++) FRC[i] = rc[i] % P;
    for (int i = 0; i < 12; i++) { FCIRC[i] = circ[i] % P; FDIAG[i] = diag[i] % P; }
}
static inline uint64_t f_sbox7(uint64_t x) {
    uint64_t x2 = f_mul(x, x), x3 = f_mul(x2, x), x4 = f_mul(x2, x2);
    return f_mul(x4, x3);
}
static inline void f_mds(uint64_t *s) {
    /* entries are small in every constant set this build uses, but nothing here assumes it: 128-bit accumulation, one reduction per row */
    uint64_t r[12];
    for (int row = 0; row < 12; row++) {
        u128 acc = (u128)s[row] * FDIAG[row];
        uint64_t carry = 0;                              /* overflows of the 128-bit accumulator (13 products of up to 128 bits each) */
        for (int i = 0; i < 12; i++) {
            u128 t = (u128)s[(i + row) % 12] * FCIRC[i];
            u128 n = acc + t;
            if (n < acc) carry++;
            acc = n;
        }
        /* acc + carry * 2^128;  2^128 mod p = (2^64)^2 = (2^32 - 1)^2 mod p */
        uint64_t lo = (uint64_t)acc, hi = (uint64_t)(acc >> 64);
        uint64_t v = f_add(lo % P, f_mul(hi % P, EPS));
        if (carry) v = f_add(v, f_mul(carry, f_mul(EPS, EPS)));
        r[row] = v;
    }
    memcpy(s, r, sizeof r);
}
void orc_poseidon_permute_fast(uint64_t *s) {
    int rnd = 0;
    for (int phase = 0; phase < 3; phase++) {
        int cnt = phase == 1 ? 22 : 4;
        for (int r = 0; r < cnt; r++, rnd++) {
            for (int i = 0; i < 12; i++) s[i] = f_add(s[i] % P, FRC[rnd * 12 + i]);
            if (phase == 1) s[0] = f_sbox7(s[0]);
            else for (int i = 0; i < 12; i++) s[i] = f_sbox7(s[i]);
            f_mds(s);
        }
    }
}
/* values: [n_polys][n] (destroyed: holds the coefficients afterwards); cap_out: 4 << cap_h words.  Returns 0, or -1 when out of memory. */
int orc_commit_fast(uint64_t *values, unsigned log_n, uint64_t n_polys, unsigned rate_bits, unsigned cap_h, uint64_t shift, uint64_t *cap_out) {
    const uint64_t n = 1ULL << log_n, N = n << rate_bits;
    const unsigned log_N = log_n + rate_bits;
    orc_ntt_fast(values, log_n, n_polys, 1);
    uint64_t *lde = (uint64_t *)malloc(sizeof(uint64_t) * N * n_polys);
    uint64_t *sp = (uint64_t *)malloc(sizeof(uint64_t) * n);
    uint64_t *dig = (uint64_t *)malloc(sizeof(uint64_t) * 4 * 2 * N);
    if (!lde || !sp || !dig) { free(lde); free(sp); free(dig); return -1; }
    sp[0] = 1;
    for (uint64_t j = 1; j < n; j++) sp[j] = f_mul(sp[j - 1], shift % P);
#pragma omp parallel for schedule(static)
    for (int64_t b = 0; b < (int64_t)n_polys; b++) {
        uint64_t *o = lde + (uint64_t)b * N;
        for (uint64_t j = 0; j < n; j++) o[j] = f_mul(values[(uint64_t)b * n + j], sp[j]);
        memset(o + n, 0, sizeof(uint64_t) * (N - n));
    }
    orc_ntt_fast(lde, log_N, n_polys, 0);
    /* leaf i (bit-reversed position) = column bitrev(i) of the natural-order LDE */
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < (int64_t)N; i++) {
        uint64_t src = 0;
        for (unsigned k = 0; k < log_N; k++) src |= (((uint64_t)i >> k) & 1) << (log_N - 1 - k);
        uint64_t s[12] = {0};
        if (n_polys <= 4) { for (uint64_t j = 0; j < n_polys; j++) s[j] = lde[j * N + src]; }
        else {
            for (uint64_t off = 0; off < n_polys; off += 8) {
                uint64_t c = n_polys - off < 8 ? n_polys - off : 8;
                for (uint64_t j = 0; j < c; j++) s[j] = lde[(off + j) * N + src];
                orc_poseidon_permute_fast(s);
            }
        }
        memcpy(dig + 4 * (uint64_t)i, s, 32);
    }
    uint64_t *prev = dig, cnt = N;
    for (unsigned lvl = log_N; lvl > cap_h; lvl--) {
        uint64_t *cur = prev + 4 * cnt;
        cnt >>= 1;
#pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < (int64_t)cnt; i++) {
            uint64_t s[12] = {0};
            memcpy(s, prev + 8 * (uint64_t)i, 64);
            orc_poseidon_permute_fast(s);
            memcpy(cur + 4 * (uint64_t)i, s, 32);
        }
        prev = cur;
    }
    memcpy(cap_out, prev, sizeof(uint64_t) * 4 * (1ULL << (cap_h < log_N ? cap_h : log_N)));
    free(lde); free(sp); free(dig);
    return 0;
}
