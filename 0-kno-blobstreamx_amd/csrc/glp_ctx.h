// glp_ctx.h — the opaque context behind include/glprover.h (internal to libglprover.so).
#pragma once
#include <hip/hip_runtime.h>
#include <chrono>
#include <map>
#include <string>
#include <tuple>
#include <utility>
#include <vector>
#include "../../include/glprover.h"
#include "gl_field.cuh"
#include "ntt_plan.h"

struct glp_table { u64* lo; u64* hi; };
struct glp_hash_state;
struct glp_comm_state;

struct glp_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;        // own_stream or an adopted one
    hipEvent_t t0 = nullptr, t1 = nullptr;
    hipStream_t aux_stream = nullptr;    // second stream of large NTT batches (glp_ntt_impl), forked from / joined into `stream` by the two events
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    hipEvent_t pass_ev[2 * GLP_MAX_PASSES] = {};
    int profiling = 0;
    int last_npass = 0;
    char err[512] = {0};
    std::map<int, glp_table> tables;     // key = log_N*2 + inv
    std::map<int, u64*> full_tables;     // key = (log_N*64 + log_m)*2 + inv: per-element inter-pass twiddles
    std::map<uint32_t, std::string> plan_override;
    // input-scale tables of the coset LDE, key = (shift, log_n, rate_bits, log_r, log_m) -> (row, col) on the device
    std::map<std::tuple<u64, int, int, int, int>, std::pair<u64*, u64*>> coset_tables;
    u64* scratch = nullptr;
    size_t scratch_bytes = 0;
    size_t scratch_cap = 0;              // upper bound for NTT scratch; larger batches are chunked
    u64* shift_lo = nullptr;             // cached shift^j tables of the last glp_lde_coset
    u64* shift_hi = nullptr;
    u64 shift_val = 0;
    int shift_log_n = -1;
    glp_hash_state* hash = nullptr;      // Poseidon constants etc. (hash.hip)
    glp_comm_state* comm = nullptr;      // RCCL communicator of the MapReduce exchange (comm.hip); null until glp_comm_init
    // device-memory pool for the prover drivers' temporaries: hipMalloc/hipFree cost milliseconds and
    // hipFree synchronises; all work of a ctx is ordered on one stream, so a block released by the host
    // can be handed to later work of the same stream without waiting (stream-ordered reuse; glp_set_stream
    // drains the stream it leaves).  The maps are touched only under the process-wide pool lock (glprover.hip).
    std::multimap<size_t, void*> pool_free;      // size -> block
    std::map<void*, size_t> pool_live;           // block -> size
    size_t pool_cached_bytes = 0;
    size_t pool_cap = 0;                          // resolved on first release: GLP_POOL_CAP_MB (per ctx), else 60 % of device memory
    int pool_cap_env = 0;                         // 0 unresolved, 1 from the environment (per ctx), 2 device share (divided by live ctxs)
    // prover stage timers (filled only while profiling is on: each mark synchronises the stream)
    std::vector<std::pair<std::string, float>> stages;
    std::chrono::steady_clock::time_point stage_t0;
    std::string stage_name;
};

// stage timing tree of the prover drivers (the TimingTree of the upstream prover, recalled): a
// mark closes the running stage and opens the next; no-ops unless glp_set_profiling(ctx, 1)
static inline void glp_stage_mark(glp_ctx* c, const char* next_name) {
    if (!c->profiling) return;
    hipStreamSynchronize(c->stream);
    const auto now = std::chrono::steady_clock::now();
    if (!c->stage_name.empty())
        c->stages.emplace_back(c->stage_name, std::chrono::duration<float, std::milli>(now - c->stage_t0).count());
    c->stage_name = next_name ? next_name : "";
    c->stage_t0 = now;
}

void glp_set_err(glp_ctx* c, const char* fmt, ...);
void* glp_pool_alloc(glp_ctx* c, size_t bytes);          // nullptr on failure (error text set)
void glp_pool_release(glp_ctx* c, void* p);              // back to the pool (no hipFree, no sync)
void glp_pool_trim(glp_ctx* c);                          // hipFree every cached block
void glp_pool_register(glp_ctx* c);                      // process-wide registry of live ctxs (shared cap, sibling trim on OOM)
void glp_pool_unregister(glp_ctx* c);

// RAII block from the ctx pool
struct GlpPoolBuf {
    glp_ctx* c = nullptr;
    void* p = nullptr;
    GlpPoolBuf() {}
    explicit GlpPoolBuf(glp_ctx* ctx) : c(ctx) {}
    GlpPoolBuf(const GlpPoolBuf&) = delete;
    GlpPoolBuf& operator=(const GlpPoolBuf&) = delete;
    ~GlpPoolBuf() { reset(); }
    void reset() { if (p && c) glp_pool_release(c, p); p = nullptr; }
    hipError_t alloc(size_t bytes) { reset(); p = glp_pool_alloc(c, bytes); return p ? hipSuccess : hipErrorOutOfMemory; }
    void adopt(void* q) { reset(); p = q; }              // q must come from glp_pool_alloc of the same ctx
    void* release() { void* q = p; p = nullptr; return q; }
    u64* u() const { return (u64*)p; }
};
void glp_hash_destroy(glp_ctx* c);

// HIP's current device is a per-host-thread setting and a ctx may be driven from a worker thread (MapReduce map
// step: several ctxs per GPU, one thread each): every entry point re-selects the ctx's device (a no-op when it
// already is current)
#define GLP_BIND(c)                                                        \
    do {                                                                   \
        if (hipSetDevice((c)->device) != hipSuccess) return GLP_E_HIP;     \
    } while (0)

#define GLP_HIPCHK(c, expr)                                                                    \
    do {                                                                                       \
        hipError_t e__ = (expr);                                                               \
        if (e__ != hipSuccess) {                                                               \
            glp_set_err((c), "%s:%d %s: %s", __FILE__, __LINE__, #expr, hipGetErrorString(e__)); \
            return GLP_E_HIP;                                                                  \
        }                                                                                      \
    } while (0)
