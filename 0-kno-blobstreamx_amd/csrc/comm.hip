// comm.hip — the exchange step of MapReduce behind the C ABI (SURVEY.md §8a row a11, §8(b) `glp_allgather_proofs`, §8e):
// one RCCL communicator per ctx (one process per GPU), the all-gather of the fixed-size padded leaf-proof blocks and
// the all-reduce(MIN) that combines the ranks' verdicts.  Upstream name (recalled, unverified; reference file:line
// NONE — the mount is empty): the plonky2x `mapreduce` generator ships leaf proofs between provers over HTTP; on one
// 8 x MI355X node the exchange is ONE collective over xGMI instead.
//
// Payloads are O(100 KiB) per leaf: the collective is latency-bound, so a rank packs all its leaves into one block and
// the exchange is one ncclAllGather of bytes on the ctx's stream (ordered with the prover's work on that stream).
// Host buffers in and out (the proofs are host objects: the transcript and the serialisation run on the host); the
// staging blocks come from the ctx pool.
//
// RCCL is bound at FIRST USE (dlopen of librccl.so.1), not at link time: a process that never calls glp_comm_* never loads it,
// and a host that already carries an RCCL (e.g. PyTorch's bundled one) shares that instance instead of getting a second
// runtime in the process.  (Linked directly, loading this library before PyTorch left PyTorch unable to see the GPU.)
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <string.h>
#include <mutex>
#include "glp_ctx.h"

static_assert(sizeof(ncclUniqueId) == GLP_COMM_ID_BYTES, "glprover.h's GLP_COMM_ID_BYTES must be sizeof(ncclUniqueId)");

namespace {
struct Rccl {
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
    char why[256] = {0};
};
Rccl g_rccl;
std::once_flag g_rccl_once;
const Rccl& rccl() {
    std::call_once(g_rccl_once, [] {
        void* h = nullptr;
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* nm : names) { h = dlopen(nm, RTLD_NOW | RTLD_LOCAL); if (h) break; }
        if (!h) { snprintf(g_rccl.why, sizeof(g_rccl.why), "cannot load librccl: %s", dlerror()); return; }
        bool all = true;
        auto sym = [&](const char* n) { void* p = dlsym(h, n); if (!p) { all = false; snprintf(g_rccl.why, sizeof(g_rccl.why), "librccl lacks %s", n); } return p; };
        g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))sym("ncclGetUniqueId");
        g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))sym("ncclCommInitRank");
        g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))sym("ncclCommDestroy");
        g_rccl.AllGather = (decltype(g_rccl.AllGather))sym("ncclAllGather");
        g_rccl.AllReduce = (decltype(g_rccl.AllReduce))sym("ncclAllReduce");
        g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))sym("ncclGetErrorString");
        g_rccl.ok = all;
    });
    return g_rccl;
}
}  // namespace
#define ncclGetErrorString rccl().GetErrorString

struct glp_comm_state {
    ncclComm_t comm = nullptr;
    int rank = 0, nranks = 0;
    // staging blocks of the exchange, resident for the communicator's life (hipMalloc, not the pool: they must not depend on what the prover
    // holds): a collective entry point that has its staging cannot fail locally BEFORE it enters RCCL and leave the peers waiting
    void* d_in = nullptr;
    void* d_out = nullptr;
    size_t cap = 0;                  // bytes per rank block that d_in / d_out (x nranks) hold
    void* d_red = nullptr;           // all-reduce operand (GLP_COMM_REDUCE_WORDS words)
};
#define GLP_COMM_REDUCE_WORDS 4096
#define GLP_COMM_DEFAULT_BLOCK (4u << 20)

static int comm_reserve(glp_ctx* c, size_t block) {
    glp_comm_state* st = c->comm;
    if (block <= st->cap) return GLP_OK;
    void *a = nullptr, *b = nullptr;
    if (hipMalloc(&a, block) != hipSuccess || hipMalloc(&b, block * (size_t)st->nranks) != hipSuccess) {
        if (a) hipFree(a);
        glp_set_err(c, "glp_comm_reserve: cannot allocate staging for %zu-byte blocks x %d ranks", block, st->nranks);
        return GLP_E_NOMEM;
    }
    if (st->d_in) hipFree(st->d_in);
    if (st->d_out) hipFree(st->d_out);
    st->d_in = a; st->d_out = b; st->cap = block;
    return GLP_OK;
}

#define GLP_NCCLCHK(c, expr)                                                                       \
    do {                                                                                           \
        ncclResult_t r__ = (expr);                                                                 \
        if (r__ != ncclSuccess) {                                                                  \
            glp_set_err((c), "%s:%d %s: %s", __FILE__, __LINE__, #expr, ncclGetErrorString(r__));  \
            return GLP_E_HIP;                                                                      \
        }                                                                                          \
    } while (0)

extern "C" int glp_comm_unique_id(uint8_t* id_out) {
    if (!id_out) return GLP_E_INVALID;
    if (!rccl().ok) return GLP_E_UNSUPPORTED;
    ncclUniqueId id;
    if (rccl().GetUniqueId(&id) != ncclSuccess) return GLP_E_HIP;
    memcpy(id_out, &id, sizeof(id));
    return GLP_OK;
}

extern "C" int glp_comm_init(glp_ctx* c, const uint8_t* id_in, int rank, int nranks) {
    if (!c) return GLP_E_INVALID;
    GLP_BIND(c);
    if (!id_in || nranks < 1 || rank < 0 || rank >= nranks) { glp_set_err(c, "glp_comm_init: bad argument"); return GLP_E_INVALID; }
    if (c->comm) { glp_set_err(c, "glp_comm_init: this ctx already has a communicator (glp_comm_destroy first)"); return GLP_E_STATE; }
    if (!rccl().ok) { glp_set_err(c, "glp_comm_init: %s", rccl().why); return GLP_E_UNSUPPORTED; }
    ncclUniqueId id;
    memcpy(&id, id_in, sizeof(id));
    glp_comm_state* st = new glp_comm_state();
    ncclResult_t r = rccl().CommInitRank(&st->comm, nranks, id, rank);     // collective: every rank calls it with the same id
    if (r != ncclSuccess) { delete st; glp_set_err(c, "ncclCommInitRank(rank %d of %d): %s", rank, nranks, ncclGetErrorString(r)); return GLP_E_HIP; }
    st->rank = rank; st->nranks = nranks;
    c->comm = st;
    // staging for the usual payloads now, where a failure is a failed init on this rank (no peer is inside a data collective yet)
    if (hipMalloc(&st->d_red, GLP_COMM_REDUCE_WORDS * 8) != hipSuccess) { glp_set_err(c, "glp_comm_init: staging allocation failed"); glp_comm_destroy(c); return GLP_E_NOMEM; }
    int rc = comm_reserve(c, GLP_COMM_DEFAULT_BLOCK);
    if (rc != GLP_OK) { glp_comm_destroy(c); return rc; }
    return GLP_OK;
}

// make room for exchanges of `padded_len`-byte blocks WITHOUT entering a collective: ranks that are about to exchange larger blocks than
// the default (4 MiB) call this first and agree on the result (e.g. through glp_allreduce_min_u64, whose staging always exists), so that
// no rank can fail locally inside glp_allgather_proofs while its peers wait in RCCL
extern "C" int glp_comm_reserve(glp_ctx* c, size_t padded_len) {
    if (!c) return GLP_E_INVALID;
    GLP_BIND(c);
    if (!c->comm) { glp_set_err(c, "glp_comm_reserve: no communicator (glp_comm_init)"); return GLP_E_STATE; }
    GLP_HIPCHK(c, hipStreamSynchronize(c->stream));
    return comm_reserve(c, padded_len);
}

extern "C" int glp_comm_rank(glp_ctx* c, int* rank, int* nranks) {
    if (!c || !c->comm) return GLP_E_STATE;
    if (rank) *rank = c->comm->rank;
    if (nranks) *nranks = c->comm->nranks;
    return GLP_OK;
}

extern "C" int glp_comm_destroy(glp_ctx* c) {
    if (!c) return GLP_E_INVALID;
    if (!c->comm) return GLP_OK;
    GLP_BIND(c);
    const hipError_t es = hipStreamSynchronize(c->stream);
    const ncclResult_t rd = rccl().CommDestroy(c->comm->comm);
    if (c->comm->d_in) hipFree(c->comm->d_in);
    if (c->comm->d_out) hipFree(c->comm->d_out);
    if (c->comm->d_red) hipFree(c->comm->d_red);
    delete c->comm;
    c->comm = nullptr;                                   // the communicator is gone either way; the status says whether it went cleanly
    if (es != hipSuccess) { glp_set_err(c, "glp_comm_destroy: stream synchronize: %s", hipGetErrorString(es)); return GLP_E_HIP; }
    if (rd != ncclSuccess) { glp_set_err(c, "glp_comm_destroy: ncclCommDestroy: %s", ncclGetErrorString(rd)); return GLP_E_HIP; }
    return GLP_OK;
}

// every rank contributes `padded_len` bytes (its leaf proofs packed and zero padded to the agreed size); h_all receives
// nranks * padded_len bytes, rank r's block at offset r * padded_len.  Synchronous.
extern "C" int glp_allgather_proofs(glp_ctx* c, const uint8_t* h_mine, size_t padded_len, uint8_t* h_all) {
    if (!c) return GLP_E_INVALID;
    GLP_BIND(c);
    if (!c->comm) { glp_set_err(c, "glp_allgather_proofs: no communicator (glp_comm_init)"); return GLP_E_STATE; }
    if (!h_mine || !h_all || padded_len == 0) { glp_set_err(c, "glp_allgather_proofs: bad argument"); return GLP_E_INVALID; }
    const size_t nr = (size_t)c->comm->nranks;
    // Staging is resident (glp_comm_init / glp_comm_reserve).  A block larger than what was reserved grows it here — the one local failure
    // left before the collective; callers that exchange more than the default reserve first and agree on the result.  Any non-OK return
    // from this function means the ranks may be out of step: the caller aborts the job on every rank (there is no recovery inside RCCL).
    if (padded_len > c->comm->cap) {
        GLP_HIPCHK(c, hipStreamSynchronize(c->stream));
        int rc = comm_reserve(c, padded_len);
        if (rc != GLP_OK) return rc;
    }
    void *d_in = c->comm->d_in, *d_out = c->comm->d_out;
    GLP_HIPCHK(c, hipMemcpyAsync(d_in, h_mine, padded_len, hipMemcpyHostToDevice, c->stream));
    GLP_NCCLCHK(c, rccl().AllGather(d_in, d_out, padded_len, ncclUint8, c->comm->comm, c->stream));
    GLP_HIPCHK(c, hipMemcpyAsync(h_all, d_out, padded_len * nr, hipMemcpyDeviceToHost, c->stream));
    GLP_HIPCHK(c, hipStreamSynchronize(c->stream));
    return GLP_OK;
}

// element-wise minimum over the ranks, in place (the Reduce step's verdicts: 1 = every leaf this rank checked verifies)
extern "C" int glp_allreduce_min_u64(glp_ctx* c, uint64_t* h_io, size_t n) {
    if (!c) return GLP_E_INVALID;
    GLP_BIND(c);
    if (!c->comm) { glp_set_err(c, "glp_allreduce_min_u64: no communicator (glp_comm_init)"); return GLP_E_STATE; }
    if (!h_io || n == 0 || n > GLP_COMM_REDUCE_WORDS) { glp_set_err(c, "glp_allreduce_min_u64: bad argument (1..%d words)", GLP_COMM_REDUCE_WORDS); return GLP_E_INVALID; }
    void* d = c->comm->d_red;                            // resident since glp_comm_init: nothing can fail locally before the collective
    GLP_HIPCHK(c, hipMemcpyAsync(d, h_io, n * 8, hipMemcpyHostToDevice, c->stream));
    GLP_NCCLCHK(c, rccl().AllReduce(d, d, n, ncclUint64, ncclMin, c->comm->comm, c->stream));
    GLP_HIPCHK(c, hipMemcpyAsync(h_io, d, n * 8, hipMemcpyDeviceToHost, c->stream));
    GLP_HIPCHK(c, hipStreamSynchronize(c->stream));
    return GLP_OK;
}
