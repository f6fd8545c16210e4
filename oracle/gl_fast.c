/*
 * oracle/gl_fast.c — the CPU BASELINE leg of the oracle (TEST/BENCH INFRASTRUCTURE ONLY).
 *
 * Same transform as orc_ntt in gl_oracle.c (which is deliberately naive: unsigned __int128
 * and `%`), restated with the arithmetic a tuned CPU prover would use, so that
 * bench.py's "cpu_baseline" is a fair CPU number rather than a strawman:
 *   - Goldilocks product reduced by hand (2^64 = 2^32 - 1, 2^96 = -1), no division;
 *   - iterative radix-2 DIT with a precomputed twiddle table per stage laid out
 *     contiguously (unit-stride twiddle reads), bit-reversal by table;
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

static inline uint64_t f_add(uint64_t a, uint64_t b) {
    uint64_t s = a + b;
    return (s < a || s >= P) ? s + EPS : s;
}
static inline uint64_t f_sub(uint64_t a, uint64_t b) {
    uint64_t d = a - b;
    return a < b ? d - EPS : d;
}
static inline uint64_t f_mul(uint64_t a, uint64_t b) {
    u128 x = (u128)a * b;
    uint64_t lo = (uint64_t)x, hi = (uint64_t)(x >> 64);
    uint64_t hh = hi >> 32, hl = hi & EPS;
    uint64_t t0 = lo - hh;
    if (lo < hh) t0 -= EPS;
    uint64_t t1 = hl * EPS;
    uint64_t r = t0 + t1;
    if (r < t1) r += EPS;
    return r >= P ? r - P : r;
}
static uint64_t f_pow(uint64_t a, uint64_t e) {
    uint64_t r = 1;
    while (e) { if (e & 1) r = f_mul(r, a); a = f_mul(a, a); e >>= 1; }
    return r;
}

void orc_ntt_fast(uint64_t *data, unsigned log_n, uint64_t batch, int inverse) {
    if (log_n == 0) return;
    const uint64_t n = 1ULL << log_n;
    uint64_t w = f_pow(7, (P - 1) >> log_n);
    if (inverse) w = f_pow(w, P - 2);
    /* per-stage twiddles, stage s (half = 2^(s-1)) stored at tw + half - 1 */
    uint64_t *tw = (uint64_t *)malloc(sizeof(uint64_t) * n);
    for (unsigned s = 1; s <= log_n; s++) {
        uint64_t half = 1ULL << (s - 1), ws = f_pow(w, n >> s), t = 1;
        for (uint64_t j = 0; j < half; j++) { tw[half - 1 + j] = t; t = f_mul(t, ws); }
    }
    uint32_t *rev = (uint32_t *)malloc(sizeof(uint32_t) * n);
    rev[0] = 0;
    for (uint64_t i = 1; i < n; i++) rev[i] = (rev[i >> 1] >> 1) | ((uint32_t)(i & 1) << (log_n - 1));
    const uint64_t ninv = f_pow(n % P, P - 2);
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t b = 0; b < (int64_t)batch; b++) {
        uint64_t *a = data + (uint64_t)b * n;
        for (uint64_t i = 0; i < n; i++) { uint32_t r = rev[i]; if (r > i) { uint64_t t = a[i]; a[i] = a[r]; a[r] = t; } }
        for (unsigned s = 1; s <= log_n; s++) {
            const uint64_t half = 1ULL << (s - 1), m = half << 1;
            const uint64_t *ts = tw + half - 1;
            for (uint64_t k = 0; k < n; k += m)
                for (uint64_t j = 0; j < half; j++) {
                    uint64_t u = a[k + j], v = f_mul(a[k + j + half], ts[j]);
                    a[k + j] = f_add(u, v);
                    a[k + j + half] = f_sub(u, v);
                }
        }
        if (inverse) for (uint64_t i = 0; i < n; i++) a[i] = f_mul(a[i], ninv);
    }
    free(tw);
    free(rev);
}
