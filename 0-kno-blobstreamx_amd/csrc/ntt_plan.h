// ntt_plan.h — host-side planning for the Goldilocks NTT: how a size-2^log_n transform is
// cut into passes of glp_ntt_pass_kernel, which twiddle tables each pass needs, and the
// buffer each pass reads and writes.  Pure C++ (no HIP runtime) so that tests/emu can drive
// the same plans on the CPU.
#pragma once
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "gl_field.cuh"

#define GLP_MAX_PASSES 4
#define GLP_MIN_LOG_R 6     // smallest tile transform served by the pass kernel
#define GLP_MAX_LOG_R 12
#define GLP_FULL_TW_MIN_BATCH 8      // per-element inter-pass twiddle tables: batched transforms only
#define GLP_FULL_TW_MAX_LOG_N 22     // ... and tables of at most 32 MiB
#define GLP_TW_SPLIT 12     // two-level twiddle tables: w^e = lo[e & 4095] * hi[e >> 12]

enum { GLP_STRIP = 0, GLP_FINAL_T = 1, GLP_FINAL_ROWS = 2 };   // pass kernel modes (ntt_kernels.cuh)
enum { GLP_BUF_SRC = 0, GLP_BUF_DST = 1, GLP_BUF_SCRATCH = 2 };

struct GlpPass {
    int log_r;     // tile transform size
    int mode;      // GLP_STRIP / GLP_FINAL_T / GLP_FINAL_ROWS
    int log_c;     // columns per tile
    int log_e;     // elements per work-item (4, 5 or 6)
    int log_m;     // STRIP: axis stride (log2); FINAL: 0
    int in_buf;    // GLP_BUF_*
    int out_buf;
};

struct GlpPlan {
    int log_n;
    int rev;       // bit-reversed output
    int npass;
    GlpPass p[GLP_MAX_PASSES];
    int needs_scratch;
};

// default columns-per-tile (measured, profiles/r01_ntt_*_plan_sweep.jsonl): tiles of 2^12
// elements (32 KiB of LDS, 4-5 workgroups per CU) up to R = 2^8, 2^13 up to R = 2^10, then 2^14
static inline int glp_default_log_c(int log_r) {
    int tile_log = log_r <= 8 ? 12 : (log_r <= 10 ? 13 : 14);
    return tile_log - log_r;
}

// radix-32 register steps exist (are instantiated) only for the tile sizes they shorten
static inline int glp_has_e5(int log_r) { return log_r >= 9 && log_r <= 12; }
// radix-64 work-items (round 3): 2^11 = 32 * 64 and 2^12 = 64 * 64 tiles in TWO register steps and ONE split exchange, so that 2^22 and 2^24 run
// in two passes instead of three.  Plain natural-order transforms only (no coset / bit-reversed forms), tile widths 2^2 and 2^3 (compile-time).
static inline int glp_has_e6(int log_r) { return log_r == 11 || log_r == 12; }
static inline int glp_default_log_e(int log_r) { (void)log_r; return 4; }

// Parse "r:c[:e],r:c[:e],..." (log2 radix : log2 columns [: log2 elements per work-item]);
// returns number of passes or 0.
static inline int glp_parse_plan(const char* s, int* lr, int* lc, int* le = nullptr) {
    int n = 0;
    while (s && *s && n < GLP_MAX_PASSES) {
        int r = 0, c = -1, e = -1, used = 0;
        if (sscanf(s, "%d:%d:%d%n", &r, &c, &e, &used) < 3) {
            e = -1;
            if (sscanf(s, "%d:%d%n", &r, &c, &used) < 2) {
                if (sscanf(s, "%d%n", &r, &used) < 1) return 0;
                c = -1;
            }
        }
        lr[n] = r; lc[n] = c; if (le) le[n] = e; n++;
        s += used;
        if (*s == ',') s++;
    }
    return n;
}

// Build the plan.  `ovr` optionally forces the radices ("12:2,12:2").  Returns 0 on success.
// in_place: src == dst.
static inline int glp_make_plan(int log_n, int rev, int in_place, const char* ovr, GlpPlan* pl, unsigned long long batch = 1) {
    memset(pl, 0, sizeof(*pl));
    pl->log_n = log_n;
    pl->rev = rev;
    if (log_n < GLP_MIN_LOG_R || log_n > 32) return -1;
    int lr[GLP_MAX_PASSES], lc[GLP_MAX_PASSES], le[GLP_MAX_PASSES], np = 0;
    for (int i = 0; i < GLP_MAX_PASSES; i++) le[i] = -1;
    if (ovr && *ovr) {
        np = glp_parse_plan(ovr, lr, lc, le);
        int sum = 0;
        for (int i = 0; i < np; i++) sum += lr[i];
        if (np == 0 || sum != log_n) np = 0;   // ignore an override that does not fit this size
    }
    if (np == 0) {
        if (log_n <= GLP_MAX_LOG_R) { np = 1; lr[0] = log_n; }
        else {
            np = (log_n + GLP_MAX_LOG_R - 1) / GLP_MAX_LOG_R;
            if (log_n > 20 && np < 3) np = 3;  // v0 heuristic: wide strips (C=16) over 2 narrow passes
            int base = log_n / np, extra = log_n % np;
            for (int i = 0; i < np; i++) lr[i] = base + (i < extra ? 1 : 0);
        }
        for (int i = 0; i < np; i++) { lc[i] = -1; le[i] = -1; }
        // Round 3: large natural-order batches of 2^20 run both 2^10 tiles on radix-32 work-items (32 * 32: ONE tile-twiddle layer and one
        // exchange per tile instead of two), 16 columns wide, with the split LDS exchange (ntt_kernels.cuh GlpSplit): -11 % VALU instructions,
        // 1.255 vs 1.373 ms at 128 x 2^20 in one run (profiles/r03_ntt_e5_probe.jsonl).  Small batches stay on the radix-16 kernels
        // (2^20 x 1: 49 vs 28 us), and so do bit-reversed / coset transforms (their general kernel has no registers to spare at 32 elements).
        const bool big = !rev && (batch << log_n) >= (1ull << 25);
        if (log_n == 20 && big) { lc[0] = 4; lc[1] = 3; le[0] = le[1] = 5; }
        // 2^22 and 2^24 in large batches (same-box A/B, profiles/r03_ntt_mixed_plans_probe.jsonl): 2^22 = 2^10 * 2^12 in TWO passes — the 2^10 strip
        // kernel of the headline plan, then a FINAL_T pass of 2^12 = 64 * 64 on radix-64 work-items — 1.40 vs 1.48 ms at 32 x 2^22; 2^24 keeps three
        // passes but starts with the same 2^10 strip kernel (10 + 7 + 7): 1.55 vs 1.58 ms at 8 x 2^24.  The pure radix-64 two-pass plans tie or lose.
        // single (and paired) 2^20 transforms are latency-bound — 1024 waves of 16-element work-items are ONE wave per SIMD, each walking its
        // dependent chain with nothing to overlap its loads, exchanges and stores with: 4-element work-items (five register steps, 16 waves per CU)
        // run 2^20 x 1 in 21.6 instead of 26.5 us, 8-element work-items 2^20 x 2 in 31.8 instead of 35.0 us; from four polynomials on the
        // 16-element kernels win again (same-box A/B: profiles/r03_ntt_small_workitems_probe.jsonl)
        if (log_n == 20 && !rev && batch == 1) { lc[0] = lc[1] = 2; le[0] = le[1] = 2; }
        if (log_n == 20 && !rev && batch == 2) { lc[0] = lc[1] = 3; le[0] = le[1] = 3; }
        if (log_n == 22 && big) { np = 2; lr[0] = 10; lr[1] = 12; lc[0] = 4; lc[1] = 3; le[0] = 5; le[1] = 6; }
        if (log_n == 24 && big) { np = 3; lr[0] = 10; lr[1] = lr[2] = 7; lc[0] = 4; lc[1] = lc[2] = 5; le[0] = 5; le[1] = le[2] = 4; }
    }
    if (np > GLP_MAX_PASSES) return -1;
    pl->npass = np;
    int rem = log_n;  // log2 of the remaining sub-problem size
    for (int i = 0; i < np; i++) {
        GlpPass* ps = &pl->p[i];
        if (lr[i] < GLP_MIN_LOG_R || lr[i] > GLP_MAX_LOG_R) return -1;
        ps->log_r = lr[i];
        rem -= lr[i];
        int last = (i == np - 1);
        ps->mode = last ? ((rev || np == 1) ? GLP_FINAL_ROWS : GLP_FINAL_T) : GLP_STRIP;
        ps->log_m = last ? 0 : rem;
        int e = le[i] >= 2 ? le[i] : glp_default_log_e(lr[i]);
        if (e != 4 && !((e == 3 || e == 2) && lr[i] == 10) && !(e == 5 && glp_has_e5(lr[i])) && !(e == 6 && glp_has_e6(lr[i]) && !rev)) return -1;
        ps->log_e = e;
        const int tmin = 6 + e;                              // 64 .. 1024 threads; radix-32 FINAL_T tiles: <= 256 (GlpBounds, ntt_kernels.cuh)
        const int tmax = e == 6 ? 9 + e : ((e == 5 && ps->mode == GLP_FINAL_T) ? 8 + e : 10 + e);
        int c = lc[i] >= 0 ? lc[i] : glp_default_log_c(lr[i]);
        if (e == 6) c = c < 2 ? 2 : (c > 3 ? 3 : c);       // the instantiated widths
        if (lc[i] < 0) {
            // small batches: prefer more, narrower tiles until the launch fills the chip
            // (>= 2 workgroups per CU), but never narrower than 32-byte segments
            while (c > 2 && c + lr[i] > tmin && ((batch << log_n) >> (lr[i] + c)) < 512) c--;
        }
        if (c + lr[i] < tmin) c = tmin - lr[i];             // at least one wavefront of threads
        if (c + lr[i] > tmax) c = tmax - lr[i];             // at most 1024 threads
        if (!last && c > rem) c = rem;                      // strip no wider than the axis stride
        if (last && ps->mode == GLP_FINAL_T && c > log_n - lr[i]) c = log_n - lr[i];
        if (c + lr[i] < tmin || c < 0) return -1;
        // LDS footprint must fit 160 KiB: max(R*(C+1), C*(R+1)) * 8
        while (c > 0) {
            unsigned long long R = 1ull << lr[i], C = 1ull << c;
            unsigned long long el = R * C + (R > C ? R : C);
            const unsigned long long bytes_per = (e >= 5 && ps->mode != GLP_FINAL_ROWS) ? 4 : 8;   // split exchange: 32-bit halves (GlpSplit)
            if (el * bytes_per <= 160 * 1024) break;
            c--;
        }
        if (c + lr[i] < tmin) return -1;
        ps->log_c = c;
    }
    // buffer routing.  STRIP passes may run in place; FINAL_T may not.
    if (np == 1 || rev) {
        for (int i = 0; i < np; i++) {
            pl->p[i].in_buf = (i == 0) ? GLP_BUF_SRC : GLP_BUF_DST;
            pl->p[i].out_buf = GLP_BUF_DST;
        }
        pl->needs_scratch = 0;
    } else {
        pl->needs_scratch = 1;
        for (int i = 0; i < np; i++) {
            int last = (i == np - 1);
            if (last) { pl->p[i].in_buf = GLP_BUF_SCRATCH; pl->p[i].out_buf = GLP_BUF_DST; }
            else if (i == np - 2) { pl->p[i].in_buf = (i == 0) ? GLP_BUF_SRC : (in_place ? GLP_BUF_DST : GLP_BUF_SCRATCH); pl->p[i].out_buf = GLP_BUF_SCRATCH; }
            else {
                // earlier strips: in place on dst when the caller's buffer is in place,
                // otherwise move to scratch at the first pass and stay there
                if (in_place) { pl->p[i].in_buf = GLP_BUF_DST; pl->p[i].out_buf = GLP_BUF_DST; }
                else { pl->p[i].in_buf = (i == 0) ? GLP_BUF_SRC : GLP_BUF_SCRATCH; pl->p[i].out_buf = GLP_BUF_SCRATCH; }
            }
        }
    }
    return 0;
}

static inline size_t glp_pass_lds_bytes(const GlpPass* ps) {
    unsigned long long R = 1ull << ps->log_r, C = 1ull << ps->log_c;
    // radix-32 work-items exchange the tile as 32-bit halves (ntt_kernels.cuh, GlpSplit): half the footprint, except FINAL_ROWS (64-bit restaging)
    if (ps->log_e >= 5 && ps->mode != GLP_FINAL_ROWS) return (size_t)(R * (C + 1) * 4);
    unsigned long long a = R * (C + 1), b = (ps->mode == GLP_FINAL_ROWS) ? C * (R + 1) : 0;
    return (size_t)((a > b ? a : b) * 8);
}
static inline unsigned glp_pass_threads(const GlpPass* ps) { return 1u << (ps->log_r + ps->log_c - ps->log_e); }
static inline unsigned long long glp_pass_grid(const GlpPass* ps, int log_n, unsigned long long batch) {
    if (ps->mode == GLP_STRIP) return batch << (log_n - ps->log_r - ps->log_c);
    unsigned long long rows = batch << (log_n - ps->log_r);
    return (rows + (1ull << ps->log_c) - 1) >> ps->log_c;
}

// Twiddle table for w_N (N = 2^log_N): lo[e] = w^e for e < min(N, 4096); hi[e] = w^(4096 e)
// for e < N/4096 (when N > 4096).  `inv` uses w^-1.
static inline size_t glp_table_lo_len(int log_N) { return log_N <= GLP_TW_SPLIT ? (1ull << log_N) : (1ull << GLP_TW_SPLIT); }
static inline size_t glp_table_hi_len(int log_N) { return log_N <= GLP_TW_SPLIT ? 0 : (1ull << (log_N - GLP_TW_SPLIT)); }
static inline void glp_fill_table(int log_N, int inv, u64* lo, u64* hi) {
    u64 w = gl_root_of_unity((unsigned)log_N);
    if (inv) w = gl_inv(w);
    size_t nlo = glp_table_lo_len(log_N), nhi = glp_table_hi_len(log_N);
    u64 t = 1;
    for (size_t i = 0; i < nlo; i++) { lo[i] = t; t = gl_mul(t, w); }
    if (nhi) {
        u64 wh = gl_pow(w, 1ull << GLP_TW_SPLIT);
        t = 1;
        for (size_t i = 0; i < nhi; i++) { hi[i] = t; t = gl_mul(t, wh); }
    }
}

// Input-scale tables of a coset LDE (GlpNttPassArgs::in_row / in_col).  Coset k of the 2^rb evaluates on
// s_k * <w_n>, s_k = shift * w_N^k (N = n << rb).  First pass with tile size R = 2^log_r on the axis of
// stride m = 2^log_m (m = n / R; 1 for single-pass sizes):
//   row[(k << log_r) | r] = s_k^(r * m),   col[(k << log_m) | c] = s_k^c   (col may be null when m = 1)
static inline void glp_fill_coset_tables(int log_n, int rb, u64 shift, int log_r, int log_m, u64* row, u64* col) {
    const u64 wN = gl_root_of_unity((unsigned)(log_n + rb));
    u64 s = shift;
    for (u64 k = 0; k < (1ull << rb); k++, s = gl_mul(s, wN)) {
        const u64 sm = gl_pow(s, 1ull << log_m);
        u64 t = 1;
        for (u64 r = 0; r < (1ull << log_r); r++) { row[(k << log_r) | r] = t; t = gl_mul(t, sm); }
        if (col) {
            t = 1;
            for (u64 c = 0; c < (1ull << log_m); c++) { col[(k << log_m) | c] = t; t = gl_mul(t, s); }
        }
    }
}
