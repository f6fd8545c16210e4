// ntt_exec.h — turns a GlpPlan into kernel launches.  Templated on a Backend so that the
// HIP host (glprover.hip) and the CPU emulation used by the tests (tests/emu) share the
// exact pass sequencing, argument construction and twiddle-table selection.
//
// Backend concept:
//   const u64* table_lo(int log_N, int inv);   // device-visible tables (glp_fill_table)
//   const u64* table_hi(int log_N, int inv);   // nullptr when log_N <= 12
//   const u64* table_full(int log_N, int log_m, int inv);   // per-element inter-pass twiddles or nullptr
//   int launch_pass(const GlpPass&, int inv, unsigned long long grid, unsigned block, size_t lds, const GlpNttPassArgs&);
//   int coset_tables(int log_n, int rb, u64 shift, int log_r, int log_m, const u64** row, const u64** col);   // glp_fill_coset_tables
//   int launch_small(const u64* src, u64* dst, u64 ss, u64 ds, u32 log_n, u32 batch, const u64* tw, u64 scale, u32 rev);
#pragma once
#include "ntt_kernels.cuh"
#include "ntt_plan.h"

struct GlpNttCall {
    const u64* src;
    u64* dst;
    u64* scratch;          // batch << log_n elements when the plan needs it
    u64 src_poly_stride;   // elements
    u64 dst_poly_stride;
    u32 batch;
    int log_n;
    int inverse;
    int rev;               // bit-reversed output order
    // coset LDE (see GlpNttPassArgs::coset_log): batch counts virtual polynomials (real << coset_log),
    // src rows hold n coefficients, dst rows n << coset_log values; requires rev
    u32 coset_log = 0;
    u64 coset_shift = 0;
};

template <class Backend>
int glp_exec_ntt(Backend& be, const GlpPlan* pl, const GlpNttCall& c) {
    const u64 n = 1ull << c.log_n;
    const u64 scale = c.inverse ? gl_inv(n % GL_P) : 1ull;
    if (c.batch == 0) return 0;
    if (c.coset_log && (!c.rev || c.inverse || c.log_n < GLP_MIN_LOG_R || pl->needs_scratch)) return -5;
    if (c.log_n < GLP_MIN_LOG_R) {
        return be.launch_small(c.src, c.dst, c.src_poly_stride, c.dst_poly_stride, (u32)c.log_n, c.batch,
                               be.table_lo(c.log_n, c.inverse), scale, (u32)c.rev);
    }
    int rem = c.log_n;
    for (int i = 0; i < pl->npass; i++) {
        const GlpPass& ps = pl->p[i];
        GlpNttPassArgs a;
        memset(&a, 0, sizeof(a));
        const u64* in; u64 in_stride;
        u64* out; u64 out_stride;
        switch (ps.in_buf) {
            case GLP_BUF_SRC: in = c.src; in_stride = c.src_poly_stride; break;
            case GLP_BUF_DST: in = c.dst; in_stride = c.dst_poly_stride; break;
            default: in = c.scratch; in_stride = n; break;
        }
        switch (ps.out_buf) {
            case GLP_BUF_DST: out = c.dst; out_stride = c.dst_poly_stride; break;
            case GLP_BUF_SCRATCH: out = c.scratch; out_stride = n; break;
            default: return -2;
        }
        if ((ps.in_buf == GLP_BUF_SCRATCH || ps.out_buf == GLP_BUF_SCRATCH) && !c.scratch) return -3;
        a.src = in; a.dst = out;
        a.src_poly_stride = in_stride; a.dst_poly_stride = out_stride;
        a.tw_tile = be.table_lo(ps.log_r, c.inverse);
        const int log_N = rem;            // size of the sub-problem this pass splits
        rem -= ps.log_r;
        if (ps.mode == GLP_STRIP) {
            a.tw_lo = be.table_lo(log_N, c.inverse);
            a.tw_hi = be.table_hi(log_N, c.inverse);
            // batched transforms of moderate size: one table multiply per element instead of the
            // running product (the table tile is shared by all polynomials through L2)
            if (!c.coset_log && c.batch >= GLP_FULL_TW_MIN_BATCH && log_N <= GLP_FULL_TW_MAX_LOG_N) a.tw_full = be.table_full(log_N, ps.log_m, c.inverse);
        }
        const int last = (i == pl->npass - 1);
        a.scale = last ? scale : 1ull;
        a.log_n = (u32)c.log_n;
        a.log_m = (u32)ps.log_m;
        a.log_c = (u32)ps.log_c;
        a.batch = c.batch;
        a.rev = (u32)c.rev;
        a.nprev = 0;
        if (ps.mode == GLP_FINAL_T) {
            a.nprev = (u32)i;
            for (int j = 0; j < i && j < 3; j++) a.log_rprev[j] = (u32)pl->p[j].log_r;
        }
        if (c.coset_log) {
            a.coset_log = c.coset_log;
            a.src_coset = (ps.in_buf != GLP_BUF_SRC) ? 1u : 0u;
            if (i == 0) {
                const int lm = (ps.mode == GLP_STRIP) ? ps.log_m : 0;
                if (ps.log_r + lm != c.log_n) return -5;
                int rc = be.coset_tables(c.log_n, (int)c.coset_log, c.coset_shift, ps.log_r, lm, &a.in_row, lm ? &a.in_col : nullptr);
                if (rc) return rc;
            }
        }
        unsigned long long grid = glp_pass_grid(&ps, c.log_n, c.batch);
        if (grid == 0 || grid > 0x7fffffffull) return -4;
        a.xcd_group_log = 0;
        if (ps.mode == GLP_STRIP && ps.log_c < 4 && ps.log_m >= 4) {
            const u32 g = 4u - (u32)ps.log_c;
            if (grid % (8ull << g) == 0) a.xcd_group_log = g;
        }
        // FINAL_T tiles narrower than a line WRITE half (quarter ...) lines: the same pairing puts the tiles that complete each other's lines on one
        // XCD (same-box A/B at 128 x 2^20, three alternations: FINAL_T pass 0.494-0.509 -> 0.459-0.482 ms; GLP_FINALT_XCD=0 switches it off)
        if (ps.mode == GLP_FINAL_T && ps.log_c < 4 && !(getenv("GLP_FINALT_XCD") && !atoi(getenv("GLP_FINALT_XCD")))) {
            const u32 g = 4u - (u32)ps.log_c;
            if (grid % (8ull << g) == 0) a.xcd_group_log = g;
        }
        a.poly_minor = (ps.mode == GLP_STRIP && a.tw_full && c.batch > 1) ? 1u : 0u;
        if (a.poly_minor && a.xcd_group_log) {
            // the pairing of line-sharing strips must survive the (position, polynomial) order:
            // tiles per polynomial must be a multiple of the group
            const unsigned long long tpp = 1ull << (c.log_n - ps.log_r - ps.log_c);
            if (tpp % (1ull << a.xcd_group_log)) a.xcd_group_log = 0;
        }
        int rc = be.launch_pass(ps, c.inverse, grid, glp_pass_threads(&ps), glp_pass_lds_bytes(&ps), a);
        if (rc) return rc;
    }
    return 0;
}
