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
};

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
    return GLP_OK;
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
    hipStreamSynchronize(c->stream);
    rccl().CommDestroy(c->comm->comm);
    delete c->comm;
    c->comm = nullptr;
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
    GlpPoolBuf d_in(c), d_out(c);
    GLP_HIPCHK(c, d_in.alloc(padded_len));
    GLP_HIPCHK(c, d_out.alloc(padded_len * nr));
    GLP_HIPCHK(c, hipMemcpyAsync(d_in.p, h_mine, padded_len, hipMemcpyHostToDevice, c->stream));
    GLP_NCCLCHK(c, rccl().AllGather(d_in.p, d_out.p, padded_len, ncclUint8, c->comm->comm, c->stream));
    GLP_HIPCHK(c, hipMemcpyAsync(h_all, d_out.p, padded_len * nr, hipMemcpyDeviceToHost, c->stream));
    GLP_HIPCHK(c, hipStreamSynchronize(c->stream));
    return GLP_OK;
}

// element-wise minimum over the ranks, in place (the Reduce step's verdicts: 1 = every leaf this rank checked verifies)
extern "C" int glp_allreduce_min_u64(glp_ctx* c, uint64_t* h_io, size_t n) {
    if (!c) return GLP_E_INVALID;
    GLP_BIND(c);
    if (!c->comm) { glp_set_err(c, "glp_allreduce_min_u64: no communicator (glp_comm_init)"); return GLP_E_STATE; }
    if (!h_io || n == 0) { glp_set_err(c, "glp_allreduce_min_u64: bad argument"); return GLP_E_INVALID; }
    GlpPoolBuf d(c);
    GLP_HIPCHK(c, d.alloc(n * 8));
    GLP_HIPCHK(c, hipMemcpyAsync(d.p, h_io, n * 8, hipMemcpyHostToDevice, c->stream));
    GLP_NCCLCHK(c, rccl().AllReduce(d.p, d.p, n, ncclUint64, ncclMin, c->comm->comm, c->stream));
    GLP_HIPCHK(c, hipMemcpyAsync(h_io, d.p, n * 8, hipMemcpyDeviceToHost, c->stream));
    GLP_HIPCHK(c, hipStreamSynchronize(c->stream));
    return GLP_OK;
}
