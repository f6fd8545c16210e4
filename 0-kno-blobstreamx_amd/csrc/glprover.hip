// glprover.hip — C-ABI implementation (include/glprover.h): context, device memory,
// twiddle-table cache, NTT/LDE/transpose entry points.  Host orchestration is C++17 because
// the reference's host language (Rust) has no toolchain in this image;
// a Rust host binds the same header.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <map>
#include <mutex>
#include <string>
#include <vector>
#include "../../include/glprover.h"
#include "glp_ctx.h"
#include "ntt_exec.h"
#include "ntt_launch.h"

void glp_set_err(glp_ctx* c, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(c->err, sizeof(c->err), fmt, ap);
    va_end(ap);
}

// ---------------------------------------------------------------------------------------
// device-memory pool (glp_ctx.h)
// ---------------------------------------------------------------------------------------
// Several ctxs may share one GPU (MapReduce map step: K provers per device, one host thread each).  Every pool operation
// runs under ONE process-wide lock and the live ctxs are registered, so that (a) the cache cap is a share of the device
// (60 % of its memory divided by the ctxs living on it) instead of 60 % per ctx, and (b) a ctx whose hipMalloc fails can
// give back its SIBLINGS' cached blocks too, not only its own, before reporting GLP_E_NOMEM.
static std::mutex g_pool_mu;
static std::vector<glp_ctx*> g_ctxs;
void glp_pool_register(glp_ctx* c) { std::lock_guard<std::mutex> lk(g_pool_mu); g_ctxs.push_back(c); }
void glp_pool_unregister(glp_ctx* c) {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    for (size_t i = 0; i < g_ctxs.size(); i++) if (g_ctxs[i] == c) { g_ctxs.erase(g_ctxs.begin() + i); break; }
}
static size_t pool_round(size_t bytes) {
    const size_t g = bytes < (1u << 20) ? 4096 : (2u << 20);     // 4 KiB / 2 MiB granules
    return ((bytes ? bytes : 1) + g - 1) / g * g;
}
// caller holds g_pool_mu.  The cached blocks of `c` may still be read by kernels enqueued on c's stream: drain it first.
static void pool_trim_locked(glp_ctx* c) {
    if (c->pool_free.empty()) return;
    hipStreamSynchronize(c->stream);
    for (auto& kv : c->pool_free) hipFree(kv.second);
    c->pool_free.clear();
    c->pool_cached_bytes = 0;
}
static size_t pool_cap_locked(glp_ctx* c) {
    // cached (released, reusable) blocks may hold up to GLP_POOL_CAP_MB per ctx, default 60 % of the device's memory shared
    // by the ctxs on that device: the MI355X has 288 GB and a 2^23-row x 80-wire proof recycles ~110 GB of temporaries;
    // with the earlier fixed 64 GiB cap that size fell into hipFree + hipMalloc of tens of GB per proof (2.9-4.2 s, not 0.7 s)
    if (c->pool_cap_env == 0) {
        const char* cap = getenv("GLP_POOL_CAP_MB");
        size_t free_b = 0, total_b = 0;
        if (cap && atoll(cap) > 0) { c->pool_cap_env = 1; c->pool_cap = (size_t)atoll(cap) << 20; }
        else { c->pool_cap_env = 2; c->pool_cap = (hipMemGetInfo(&free_b, &total_b) == hipSuccess && total_b) ? total_b / 10 * 6 : ((size_t)64 << 30); }
    }
    if (c->pool_cap_env == 1) return c->pool_cap;
    size_t same = 0;
    for (glp_ctx* o : g_ctxs) if (o->device == c->device) same++;
    return c->pool_cap / (same ? same : 1);
}
void* glp_pool_alloc(glp_ctx* c, size_t bytes) {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    const size_t sz = pool_round(bytes);
    auto it = c->pool_free.lower_bound(sz);
    if (it != c->pool_free.end() && it->first <= sz + sz / 8) {   // reuse a block at most 12.5 % larger
        void* p = it->second;
        c->pool_live[p] = it->first;
        c->pool_cached_bytes -= it->first;
        c->pool_free.erase(it);
        return p;
    }
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, sz);
    if (e != hipSuccess) {
        pool_trim_locked(c);                                       // give this ctx's cached blocks back and retry ...
        e = hipMalloc(&p, sz);
    }
    if (e != hipSuccess) {
        for (glp_ctx* o : g_ctxs) if (o != c && o->device == c->device) pool_trim_locked(o);   // ... then the siblings' caches
        e = hipMalloc(&p, sz);
    }
    if (e != hipSuccess) { glp_set_err(c, "device allocation of %zu bytes failed: %s", sz, hipGetErrorString(e)); return nullptr; }
    c->pool_live[p] = sz;
    return p;
}
void glp_pool_release(glp_ctx* c, void* p) {
    if (!p) return;
    std::lock_guard<std::mutex> lk(g_pool_mu);
    auto it = c->pool_live.find(p);
    if (it == c->pool_live.end()) { hipFree(p); return; }          // not ours: plain free
    c->pool_free.emplace(it->second, p);
    c->pool_cached_bytes += it->second;
    c->pool_live.erase(it);
    if (c->pool_cached_bytes > pool_cap_locked(c)) pool_trim_locked(c);
}
void glp_pool_trim(glp_ctx* c) {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    pool_trim_locked(c);
}
extern "C" int glp_trim_pool(glp_ctx* c) {
    if (!c) return GLP_E_INVALID;
    GLP_BIND(c);
    glp_pool_trim(c);
    return GLP_OK;
}

// ---------------------------------------------------------------------------------------
// small utility kernels
// ---------------------------------------------------------------------------------------
// out[b][j] = j < n ? coeffs[b][j] * shift^j : 0   (shift^j = s_lo[j & 4095] * s_hi[j >> 12])
__global__ void __launch_bounds__(256) glp_lde_prep_kernel(const u64* __restrict__ coeffs, u64* __restrict__ out,
                                                           u32 log_n, u32 log_N, u32 batch, const u64* __restrict__ s_lo,
                                                           const u64* __restrict__ s_hi) {
    const u64 N = 1ull << log_N, n = 1ull << log_n;
    const u64 total = (u64)batch << log_N;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (u64)gridDim.x * blockDim.x) {
        const u64 b = i >> log_N, j = i & (N - 1);
        u64 v = 0;
        if (j < n) {
            v = coeffs[b * n + j];
            u64 s = s_lo[j & 4095u];
            if (s_hi) s = gl_mul(s, s_hi[j >> 12]);
            v = gl_mul(v, s);
        }
        out[i] = v;
    }
}

// element-wise field operations (row a1): the same device functions the kernels inline
__global__ void __launch_bounds__(256) glp_field_op_kernel(int op, const u64* __restrict__ a, const u64* __restrict__ b,
                                                           u64* __restrict__ out, u64 n) {
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) {
        const u64 x = a[i], y = b ? b[i] : 0;
        u64 r = 0;
        switch (op) {
            case 0: r = gl_add(x, y); break;
            case 1: r = gl_sub(x, y); break;
            case 2: r = gl_mul(x, y); break;
            case 3: {
                const int sh = (int)(y % 192);
                r = x;
                glp_static_for<0, 192>([&](auto s_) {
                    constexpr int S = decltype(s_)::value;
                    if (sh == S) r = gl_mul_pow2<S>(x);
                });
                break;
            }
            case 4: r = x ? gl_inv(x) : 0; break;
            // the reduction primitives under the products, with ARBITRARY u64 operands (parity tests only)
            case 5: r = gl_reduce128(x, y); break;                       // (x * 2^64 + y) mod p
            case 6: r = gl_canon(gl_reduce128_t<false>(x, y)); break;    // same through the non-canonical form
            case 7: r = gl_canon(gl_mul_nc(x, y)); break;                // product of any two u64 representatives
            case 8: r = gl_canon(gl_fold_small(x >> 7, y >> 7)); break;  // (x>>7) + (y>>7) * 2^32 mod p, both < 2^57
            case 9: r = gl_canon(gl_mad_eps<false>((u32)y, x)); break;   // x + (y mod 2^32) * (2^32 - 1)
            case 10: r = gl_mad_eps<true>((u32)y, x); break;
        }
        out[i] = r;
    }
}

// [rows][cols] -> [cols][rows], 32x32 u64 tiles through LDS (row pad = 1 element)
__global__ void __launch_bounds__(256) glp_transpose_kernel(const u64* __restrict__ in, u64* __restrict__ out, u64 rows,
                                                            u64 cols, u64 tiles_c) {
    __shared__ u64 t[32][33];
    const u64 tile_r = blockIdx.x / tiles_c, tile_c = blockIdx.x % tiles_c;
    const u32 tx = threadIdx.x & 31u, ty = threadIdx.x >> 5;   // 32 x 8
    for (u32 k = 0; k < 32; k += 8) {
        const u64 r = tile_r * 32 + ty + k, c = tile_c * 32 + tx;
        if (r < rows && c < cols) t[ty + k][tx] = in[r * cols + c];
    }
    __syncthreads();
    for (u32 k = 0; k < 32; k += 8) {
        const u64 c = tile_c * 32 + ty + k, r = tile_r * 32 + tx;
        if (r < rows && c < cols) out[c * rows + r] = t[tx][ty + k];
    }
}

// ---------------------------------------------------------------------------------------
// twiddle tables
// ---------------------------------------------------------------------------------------
static int ensure_table(glp_ctx* c, int log_N, int inv) {
    const int key = log_N * 2 + (inv ? 1 : 0);
    if (c->tables.count(key)) return GLP_OK;
    const size_t nlo = glp_table_lo_len(log_N), nhi = glp_table_hi_len(log_N);
    std::vector<u64> lo(nlo), hi(nhi ? nhi : 1);
    glp_fill_table(log_N, inv, lo.data(), hi.data());
    glp_table t{nullptr, nullptr};
    GLP_HIPCHK(c, hipMalloc((void**)&t.lo, nlo * 8));
    GLP_HIPCHK(c, hipMemcpy(t.lo, lo.data(), nlo * 8, hipMemcpyHostToDevice));
    if (nhi) {
        GLP_HIPCHK(c, hipMalloc((void**)&t.hi, nhi * 8));
        GLP_HIPCHK(c, hipMemcpy(t.hi, hi.data(), nhi * 8, hipMemcpyHostToDevice));
    }
    c->tables[key] = t;
    return GLP_OK;
}

int glp_ntt_table(glp_ctx* c, int log_N, int inv, const u64** lo, const u64** hi) {
    int rc = ensure_table(c, log_N, inv);
    if (rc != GLP_OK) return rc;
    const glp_table& t = c->tables[log_N * 2 + (inv ? 1 : 0)];
    *lo = t.lo; *hi = t.hi;
    return GLP_OK;
}

namespace {
struct HipBackend {
    glp_ctx* c;
    int rc = GLP_OK;
    int npass = 0;
    hipStream_t st = nullptr;            // launch stream (the ctx's stream unless set)
    hipStream_t stream() const { return st ? st : c->stream; }
    const u64* table_lo(int log_N, int inv) {
        if (ensure_table(c, log_N, inv) != GLP_OK) { rc = GLP_E_HIP; return nullptr; }
        return c->tables[log_N * 2 + (inv ? 1 : 0)].lo;
    }
    const u64* table_hi(int log_N, int inv) {
        if (ensure_table(c, log_N, inv) != GLP_OK) { rc = GLP_E_HIP; return nullptr; }
        return c->tables[log_N * 2 + (inv ? 1 : 0)].hi;
    }
    const u64* table_full(int log_N, int log_m, int inv) {
        // Measured on MI355X (profiles/r01_ntt_full_table_ab.txt): the per-element table makes the
        // 128 x 2^20 strip pass SLOWER (0.967 vs 0.842 ms): the extra 8 B/element of L2 traffic costs
        // more than the 15 multiplies it removes.  Kept as an opt-in experiment only.
        if (!getenv("GLP_FULL_TW")) return nullptr;
        const int key = (log_N * 64 + log_m) * 2 + (inv ? 1 : 0);
        auto it = c->full_tables.find(key);
        if (it != c->full_tables.end()) return it->second;
        if (ensure_table(c, log_N, inv) != GLP_OK) { rc = GLP_E_HIP; return nullptr; }
        const glp_table& t = c->tables[log_N * 2 + (inv ? 1 : 0)];
        u64* d = nullptr;
        if (hipMalloc((void**)&d, (size_t)8 << log_N) != hipSuccess) return nullptr;   // fall back to the running product
        u64 blocks = ((1ull << log_N) + 255) / 256;
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(glp_build_full_tw_kernel<0>, dim3((unsigned)blocks), dim3(256), 0, c->stream, d, (u32)log_N, (u32)log_m, t.lo, t.hi);
        if (hipGetLastError() != hipSuccess) { hipFree(d); return nullptr; }
        c->full_tables[key] = d;
        return d;
    }
    int coset_tables(int log_n, int rb, u64 shift, int log_r, int log_m, const u64** row, const u64** col) {
        const auto key = std::make_tuple(shift, log_n, rb, log_r, log_m);
        auto it = c->coset_tables.find(key);
        if (it == c->coset_tables.end()) {
            const size_t nr = (size_t)1 << (rb + log_r), nc = col ? ((size_t)1 << (rb + log_m)) : 0;
            std::vector<u64> hr(nr), hc(nc ? nc : 1);
            glp_fill_coset_tables(log_n, rb, shift, log_r, log_m, hr.data(), nc ? hc.data() : nullptr);
            u64 *dr = nullptr, *dc = nullptr;
            if (hipMalloc((void**)&dr, nr * 8) != hipSuccess) { glp_set_err(c, "coset table alloc"); return rc = GLP_E_NOMEM; }
            if (nc && hipMalloc((void**)&dc, nc * 8) != hipSuccess) { hipFree(dr); glp_set_err(c, "coset table alloc"); return rc = GLP_E_NOMEM; }
            // synchronous copies: the host vectors die at the end of this scope
            if (hipMemcpy(dr, hr.data(), nr * 8, hipMemcpyHostToDevice) != hipSuccess ||
                (nc && hipMemcpy(dc, hc.data(), nc * 8, hipMemcpyHostToDevice) != hipSuccess)) {
                hipFree(dr); if (dc) hipFree(dc);
                glp_set_err(c, "coset table upload");
                return rc = GLP_E_HIP;
            }
            it = c->coset_tables.emplace(key, std::make_pair(dr, dc)).first;
        }
        *row = it->second.first;
        if (col) *col = it->second.second;
        return GLP_OK;
    }
    void mark(int idx) {
        if (c->profiling && idx < 2 * GLP_MAX_PASSES) hipEventRecord(c->pass_ev[idx], c->stream);
    }
    int launch_pass(const GlpPass& ps, int inv, unsigned long long grid, unsigned block, size_t lds, const GlpNttPassArgs& a) {
        if (rc != GLP_OK) return rc;
        if (lds > 160 * 1024 || block > 1024 || block < 64) { glp_set_err(c, "bad launch geometry"); return GLP_E_INVALID; }
        mark(2 * npass);
        hipError_t e = glp_launch_ntt_pass(ps.log_r, ps.mode, inv, ps.log_e, (unsigned)grid, block, lds, stream(), &a);
        mark(2 * npass + 1);
        npass++;
        if (e != hipSuccess) { glp_set_err(c, "ntt pass launch: %s", hipGetErrorString(e)); return GLP_E_HIP; }
        return GLP_OK;
    }
    int launch_small(const u64* src, u64* dst, u64 ss, u64 ds, u32 log_n, u32 batch, const u64* tw, u64 scale, u32 rev) {
        if (rc != GLP_OK) return rc;
        mark(0);
        hipLaunchKernelGGL(glp_ntt_small_kernel<0>, dim3((batch + 255) / 256), dim3(256), 0, c->stream, src, dst, ss, ds, log_n,
                           batch, tw, scale, rev);
        mark(1);
        npass = 1;
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) { glp_set_err(c, "small ntt launch: %s", hipGetErrorString(e)); return GLP_E_HIP; }
        return GLP_OK;
    }
};
}  // namespace

static int ensure_scratch(glp_ctx* c, size_t bytes) {
    if (c->scratch_bytes >= bytes) return GLP_OK;
    if (c->scratch) { hipStreamSynchronize(c->stream); hipFree(c->scratch); c->scratch = nullptr; c->scratch_bytes = 0; }
    hipError_t e = hipMalloc((void**)&c->scratch, bytes);
    if (e != hipSuccess) { glp_set_err(c, "scratch alloc of %zu bytes: %s", bytes, hipGetErrorString(e)); return GLP_E_NOMEM; }
    c->scratch_bytes = bytes;
    return GLP_OK;
}

// ---------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------
extern "C" const char* glp_version(void) { return "glprover 0.1 (gfx950)"; }

// the build's two-adic subgroup (gl_field.cuh): generator of order 2^32 and the exponent of w_64 = 2^e
extern "C" int glp_field_params(uint64_t* two_adic_generator, uint32_t* w64_log2) {
    if (two_adic_generator) *two_adic_generator = GLP_TWO_ADIC_GENERATOR;
    if (w64_log2) *w64_log2 = GLP_W64_LOG2;
    // consistent: order exactly 2^32, and its 2^26-th power is the power of two the shift twiddles assume
    const u64 g = GLP_TWO_ADIC_GENERATOR;
    if (g >= GL_P || gl_pow(g, 1ull << 31) != GL_P - 1 || gl_pow(g, 1ull << 26) != gl_pow(2, GLP_W64_LOG2)) return GLP_E_STATE;
    return GLP_OK;
}

extern "C" int glp_create(glp_ctx** out, int device_id) {
    if (!out) return GLP_E_INVALID;
    *out = nullptr;
    if (glp_field_params(nullptr, nullptr) != GLP_OK) return GLP_E_STATE;       // a build with an inconsistent generator pair must not run
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return GLP_E_NODEVICE;
    if (device_id < 0 || device_id >= ndev) return GLP_E_INVALID;
    if (hipSetDevice(device_id) != hipSuccess) return GLP_E_NODEVICE;
    glp_ctx* c = new glp_ctx();
    c->device = device_id;
    c->err[0] = 0;
    if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess) { delete c; return GLP_E_HIP; }
    c->stream = c->own_stream;
    hipEventCreate(&c->t0);
    hipEventCreate(&c->t1);
    for (int i = 0; i < 2 * GLP_MAX_PASSES; i++) hipEventCreate(&c->pass_ev[i]);
    const char* cap = getenv("GLP_SCRATCH_CAP_MB");
    c->scratch_cap = (cap && atoll(cap) > 0) ? (size_t)atoll(cap) << 20 : (size_t)4 << 30;
    glp_pool_register(c);
    *out = c;
    return GLP_OK;
}

// HIP's current device is a per-thread setting: a host thread other than the one that created the ctx
// must bind before its first call (glp_plonk_prove / glp_fri_prove / glp_plonk_setup bind by themselves)
extern "C" int glp_bind_thread(glp_ctx* c) {
    if (!c) return GLP_E_INVALID;
    GLP_BIND(c);
    return GLP_OK;
}

extern "C" int glp_comm_destroy(glp_ctx* c);
extern "C" void glp_destroy(glp_ctx* c) {
    if (!c) return;
    hipSetDevice(c->device);
    glp_comm_destroy(c);
    hipStreamSynchronize(c->stream);
    for (auto& kv : c->tables) { if (kv.second.lo) hipFree(kv.second.lo); if (kv.second.hi) hipFree(kv.second.hi); }
    for (auto& kv : c->full_tables) if (kv.second) hipFree(kv.second);
    for (auto& kv : c->coset_tables) { if (kv.second.first) hipFree(kv.second.first); if (kv.second.second) hipFree(kv.second.second); }
    if (c->shift_lo) hipFree(c->shift_lo);
    if (c->shift_hi) hipFree(c->shift_hi);
    if (c->scratch) hipFree(c->scratch);
    glp_pool_unregister(c);
    glp_pool_trim(c);
    for (auto& kv : c->pool_live) hipFree(kv.first);             // blocks a driver still held (error paths)
    c->pool_live.clear();
    glp_hash_destroy(c);
    hipEventDestroy(c->t0);
    hipEventDestroy(c->t1);
    for (int i = 0; i < 2 * GLP_MAX_PASSES; i++) hipEventDestroy(c->pass_ev[i]);
    if (c->aux_stream) { hipStreamSynchronize(c->aux_stream); hipStreamDestroy(c->aux_stream); hipEventDestroy(c->ev_fork); hipEventDestroy(c->ev_join); }
    hipStreamDestroy(c->own_stream);
    delete c;
}

extern "C" const char* glp_last_error(const glp_ctx* c) { return c ? c->err : "null ctx"; }

extern "C" int glp_alloc(glp_ctx* c, void** d_ptr, size_t bytes) {
    if (!c || !d_ptr) return GLP_E_INVALID;
    GLP_BIND(c);
    hipError_t e = hipMalloc(d_ptr, bytes ? bytes : 1);
    if (e != hipSuccess) { glp_set_err(c, "hipMalloc(%zu): %s", bytes, hipGetErrorString(e)); return GLP_E_NOMEM; }
    return GLP_OK;
}
extern "C" int glp_free(glp_ctx* c, void* d_ptr) {
    if (!c) return GLP_E_INVALID;
    GLP_BIND(c);
    if (!d_ptr) return GLP_OK;
    GLP_HIPCHK(c, hipStreamSynchronize(c->stream));
    GLP_HIPCHK(c, hipFree(d_ptr));
    return GLP_OK;
}
extern "C" int glp_h2d(glp_ctx* c, void* d, const void* h, size_t bytes) {
    if (!c || (!d && bytes) || (!h && bytes)) return GLP_E_INVALID;
    GLP_BIND(c);
    GLP_HIPCHK(c, hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, c->stream));
    GLP_HIPCHK(c, hipStreamSynchronize(c->stream));
    return GLP_OK;
}
extern "C" int glp_d2h(glp_ctx* c, void* h, const void* d, size_t bytes) {
    if (!c || (!d && bytes) || (!h && bytes)) return GLP_E_INVALID;
    GLP_BIND(c);
    GLP_HIPCHK(c, hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, c->stream));
    GLP_HIPCHK(c, hipStreamSynchronize(c->stream));
    return GLP_OK;
}
extern "C" int glp_sync(glp_ctx* c) {
    if (!c) return GLP_E_INVALID;
    GLP_BIND(c);
    GLP_HIPCHK(c, hipStreamSynchronize(c->stream));
    return GLP_OK;
}
extern "C" int glp_set_stream(glp_ctx* c, void* s) {
    if (!c) return GLP_E_INVALID;
    GLP_BIND(c);
    hipStream_t next = s ? (hipStream_t)s : c->own_stream;
    if (next == c->stream) return GLP_OK;
    // pool blocks and the NTT scratch are handed out again without waiting because all work of a ctx is ordered on ONE
    // stream; when that stream changes, drain the old one first or released blocks could be reused under kernels still running
    GLP_HIPCHK(c, hipStreamSynchronize(c->stream));
    c->stream = next;
    return GLP_OK;
}
extern "C" int glp_timer_start(glp_ctx* c) {
    if (!c) return GLP_E_INVALID;
    GLP_BIND(c);
    GLP_HIPCHK(c, hipEventRecord(c->t0, c->stream));
    return GLP_OK;
}
extern "C" int glp_timer_stop(glp_ctx* c, float* ms) {
    if (!c || !ms) return GLP_E_INVALID;
    GLP_BIND(c);
    GLP_HIPCHK(c, hipEventRecord(c->t1, c->stream));
    GLP_HIPCHK(c, hipEventSynchronize(c->t1));
    GLP_HIPCHK(c, hipEventElapsedTime(ms, c->t0, c->t1));
    return GLP_OK;
}

extern "C" int glp_ntt_set_plan(glp_ctx* c, uint32_t log_n, const char* plan) {
    if (!c || log_n > 32) return GLP_E_INVALID;
    if (plan && *plan) {
        GlpPlan pl;
        if (log_n < GLP_MIN_LOG_R) return GLP_E_INVALID;
        int lr[GLP_MAX_PASSES], lc[GLP_MAX_PASSES];
        int np = glp_parse_plan(plan, lr, lc, nullptr), sum = 0;
        for (int i = 0; i < np; i++) sum += lr[i];
        if (np == 0 || sum != (int)log_n || glp_make_plan((int)log_n, 0, 1, plan, &pl) != 0) {
            glp_set_err(c, "plan '%s' does not fit log_n=%u", plan, log_n);
            return GLP_E_INVALID;
        }
        c->plan_override[log_n] = plan;
    } else {
        c->plan_override.erase(log_n);
    }
    return GLP_OK;
}

static const char* plan_override_for(glp_ctx* c, uint32_t log_n) {
    auto it = c->plan_override.find(log_n);
    if (it != c->plan_override.end()) return it->second.c_str();
    const char* env = getenv("GLP_NTT_PLAN");   // applies to every size it sums to
    return env;
}

extern "C" int glp_ntt_describe_plan(glp_ctx* c, uint32_t log_n, uint32_t batch, uint32_t flags, char* buf, size_t len) {
    if (!c || !buf || len == 0) return GLP_E_INVALID;
    if (log_n < GLP_MIN_LOG_R) { snprintf(buf, len, "small(n=%u)", 1u << log_n); return GLP_OK; }
    GlpPlan pl;
    if (glp_make_plan((int)log_n, (flags & GLP_NTT_BITREV) ? 1 : 0, 1, plan_override_for(c, log_n), &pl, batch ? batch : 1) != 0) return GLP_E_UNSUPPORTED;
    size_t off = 0;
    static const char* mn[] = {"strip", "finalT", "finalRows"};
    for (int i = 0; i < pl.npass && off < len; i++)
        off += (size_t)snprintf(buf + off, len - off, "%s%s(R=2^%d,C=2^%d%s)", i ? "+" : "", mn[pl.p[i].mode], pl.p[i].log_r, pl.p[i].log_c,
                                pl.p[i].log_e == 5 ? ",E=32" : (pl.p[i].log_e == 6 ? ",E=64" : (pl.p[i].log_e == 3 ? ",E=8" : (pl.p[i].log_e == 2 ? ",E=4" : ""))));
    return GLP_OK;
}

extern "C" int glp_set_profiling(glp_ctx* c, int on) {
    if (!c) return GLP_E_INVALID;
    GLP_BIND(c);
    c->profiling = on ? 1 : 0;
    c->last_npass = 0;
    return GLP_OK;
}
extern "C" int glp_last_stage_ms(glp_ctx* c, char* names, size_t names_len, float* ms, int* n_inout) {
    if (!c || !names || !ms || !n_inout || names_len == 0) return GLP_E_INVALID;
    const int cap = *n_inout;
    size_t off = 0;
    int k = 0;
    names[0] = 0;
    for (const auto& st : c->stages) {
        if (k >= cap) break;
        const int w = snprintf(names + off, names_len - off, "%s%s", k ? ";" : "", st.first.c_str());
        if (w < 0 || (size_t)w >= names_len - off) break;
        off += (size_t)w;
        ms[k++] = st.second;
    }
    *n_inout = k;
    return GLP_OK;
}
extern "C" int glp_last_pass_ms(glp_ctx* c, float* ms, int* n_out) {
    if (!c || !ms || !n_out) return GLP_E_INVALID;
    *n_out = 0;
    if (!c->profiling) return GLP_E_STATE;
    GLP_HIPCHK(c, hipStreamSynchronize(c->stream));
    for (int i = 0; i < c->last_npass; i++) GLP_HIPCHK(c, hipEventElapsedTime(&ms[i], c->pass_ev[2 * i], c->pass_ev[2 * i + 1]));
    *n_out = c->last_npass;
    return GLP_OK;
}

int glp_ntt_impl(glp_ctx* c, const uint64_t* src, uint64_t* dst, uint32_t log_n, uint32_t batch, uint64_t ss, uint64_t ds,
                 uint32_t flags) {
    if (!c) return GLP_E_INVALID;
    GLP_BIND(c);
    if (!src || !dst || log_n > 32 || (flags & ~(GLP_NTT_INVERSE | GLP_NTT_BITREV))) { glp_set_err(c, "glp_ntt: bad argument"); return GLP_E_INVALID; }
    const u64 n = 1ull << log_n;
    if (batch > 1 && (ss < n || ds < n)) { glp_set_err(c, "glp_ntt: poly stride < n"); return GLP_E_INVALID; }
    if (src != dst) {
        // partial overlap is not supported
        const u64 span_s = (u64)(batch ? batch - 1 : 0) * ss + n, span_d = (u64)(batch ? batch - 1 : 0) * ds + n;
        if (src < dst + span_d && dst < src + span_s) { glp_set_err(c, "glp_ntt: src/dst overlap"); return GLP_E_INVALID; }
    } else if (ss != ds) { glp_set_err(c, "glp_ntt: in place needs equal strides"); return GLP_E_INVALID; }
    if (batch == 0) return GLP_OK;
    if (log_n == 0) {
        if (src != dst) GLP_HIPCHK(c, hipMemcpy2DAsync(dst, ds * 8, src, ss * 8, 8, batch, hipMemcpyDeviceToDevice, c->stream));
        return GLP_OK;
    }
    const int inv = (flags & GLP_NTT_INVERSE) ? 1 : 0, rev = (flags & GLP_NTT_BITREV) ? 1 : 0;
    GlpPlan pl;
    memset(&pl, 0, sizeof(pl));
    if (log_n >= GLP_MIN_LOG_R) {
        if (glp_make_plan((int)log_n, rev, src == dst, plan_override_for(c, log_n), &pl, batch) != 0) {
            glp_set_err(c, "glp_ntt: no plan for log_n=%u", log_n);
            return GLP_E_UNSUPPORTED;
        }
    }
    // bounded scratch: process the batch in chunks of `chunk` polynomials
    u32 chunk = batch;
    if (pl.needs_scratch) {
        u64 per_poly = n * 8;
        u64 maxp = c->scratch_cap / per_poly;
        if (maxp == 0) maxp = 1;
        if (chunk > maxp) chunk = (u32)maxp;
        int rc = ensure_scratch(c, (size_t)chunk * per_poly);
        if (rc != GLP_OK) return rc;
    }
    HipBackend be{c};
    // Two halves on two streams (round 3): a pass kernel's tail — the last wave of workgroups draining while the dependent next pass cannot
    // start — and the launch gap cost ~7 % of a two-pass transform of a large batch (20-step mean 1.155 ms vs 1.067 ms of kernel time).  The
    // polynomials are independent, so the batch is cut in two: the second half runs the same passes on an auxiliary stream (forked from and
    // joined back into the ctx's stream by events: the call stays stream-ordered for the caller) and each half's tails and gaps are covered by
    // the other's kernels.  MEASURED (same-box A/B, three alternations at 128 x 2^20: profiles/r03_ab_ntt_two_streams.txt): 1.108 ms without,
    // 1.119 ms with — no gain: the difference between a transform's wall time and its kernels' isolated times is not idle tail (the chip runs
    // isolated kernels at a higher clock than a sustained stream of them).  Kept as an OPT-IN experiment (GLP_NTT_SPLIT=1), off by default.
    if (pl.npass >= 2 && pl.needs_scratch && chunk == batch && batch >= 2 && !c->profiling && ((u64)batch << log_n) >= (1ull << 26) &&
        getenv("GLP_NTT_SPLIT") && atoi(getenv("GLP_NTT_SPLIT"))) {
        if (!c->aux_stream) {
            if (hipStreamCreateWithFlags(&c->aux_stream, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) != hipSuccess) { glp_set_err(c, "glp_ntt: auxiliary stream"); return GLP_E_HIP; }
        }
        const u32 h = batch / 2;
        GLP_HIPCHK(c, hipEventRecord(c->ev_fork, c->stream));
        GLP_HIPCHK(c, hipStreamWaitEvent(c->aux_stream, c->ev_fork, 0));
        GlpNttCall a{src, dst, c->scratch, ss, ds, h, (int)log_n, inv, rev};
        GlpNttCall b{src + (u64)h * ss, dst + (u64)h * ds, c->scratch + (u64)h * n, ss, ds, batch - h, (int)log_n, inv, rev};
        int rc = glp_exec_ntt(be, &pl, a);                    // (also makes the tables resident, on the ctx's stream, before the fork is consumed)
        HipBackend be2{c};
        be2.st = c->aux_stream;
        if (rc == GLP_OK) rc = glp_exec_ntt(be2, &pl, b);
        GLP_HIPCHK(c, hipEventRecord(c->ev_join, c->aux_stream));
        GLP_HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_join, 0));
        if (rc != GLP_OK) { if (rc > -10 && c->err[0] == 0) glp_set_err(c, "glp_ntt: exec rc=%d", rc); return rc < -6 ? GLP_E_INVALID : rc; }
        c->last_npass = be.npass;
        return GLP_OK;
    }
    for (u32 b0 = 0; b0 < batch; b0 += chunk) {
        const u32 nb = (batch - b0 < chunk) ? batch - b0 : chunk;
        GlpNttCall call{src + (u64)b0 * ss, dst + (u64)b0 * ds, c->scratch, ss, ds, nb, (int)log_n, inv, rev};
        be.npass = 0;
        int rc = glp_exec_ntt(be, &pl, call);
        if (rc != GLP_OK) { if (rc > -10 && c->err[0] == 0) glp_set_err(c, "glp_ntt: exec rc=%d", rc); return rc < -6 ? GLP_E_INVALID : rc; }
    }
    c->last_npass = be.npass;
    return GLP_OK;
}

extern "C" int glp_ntt_ex(glp_ctx* c, const uint64_t* src, uint64_t* dst, uint32_t log_n, uint32_t batch, uint64_t ss,
                          uint64_t ds, uint32_t flags) {
    return glp_ntt_impl(c, src, dst, log_n, batch, ss, ds, flags);
}
extern "C" int glp_ntt(glp_ctx* c, uint64_t* d_io, uint32_t log_n, uint32_t batch, int inverse) {
    const u64 n = log_n <= 32 ? (1ull << log_n) : 0;
    return glp_ntt_impl(c, d_io, d_io, log_n, batch, n, n, inverse ? GLP_NTT_INVERSE : 0u);
}

extern "C" int glp_lde_coset(glp_ctx* c, const uint64_t* coeffs, uint64_t* out, uint32_t log_n, uint32_t rate_bits,
                             uint32_t batch, uint64_t shift, uint32_t flags) {
    if (!c) return GLP_E_INVALID;
    GLP_BIND(c);
    if (!coeffs || !out || log_n + rate_bits > 32 || shift == 0 || shift >= GL_P || (flags & ~GLP_NTT_BITREV)) {
        glp_set_err(c, "glp_lde_coset: bad argument");
        return GLP_E_INVALID;
    }
    if (batch == 0) return GLP_OK;
    const u32 log_N = log_n + rate_bits;
    const u64 n = 1ull << log_n;
    if ((flags & GLP_NTT_BITREV) && rate_bits > 0 && log_n >= GLP_MIN_LOG_R && ((u64)batch << rate_bits) <= 0x7fffffffull &&
        !getenv("GLP_LDE_PADDED")) {
        // Bit-reversed output (what the prover commits to): 2^rate_bits size-n transforms per polynomial, one
        // per coset, each landing in its contiguous block of the output row — no zero padding, no size-N
        // passes: 2 passes over 8n points instead of pad + 3 passes (n = 2^20, rate 8).
        GlpPlan pl;
        const u32 vbatch = batch << rate_bits;
        if (glp_make_plan((int)log_n, 1, 0, plan_override_for(c, log_n), &pl, vbatch) != 0 || pl.needs_scratch) {
            glp_set_err(c, "glp_lde_coset: no plan for log_n=%u", log_n);
            return GLP_E_UNSUPPORTED;
        }
        HipBackend be{c};
        GlpNttCall call{coeffs, out, nullptr, n, n << rate_bits, vbatch, (int)log_n, 0, 1};
        call.coset_log = rate_bits;
        call.coset_shift = shift;
        int rc = glp_exec_ntt(be, &pl, call);
        if (rc != GLP_OK) { if (rc > -10 && c->err[0] == 0) glp_set_err(c, "glp_lde_coset: exec rc=%d", rc); return rc < -6 ? GLP_E_INVALID : rc; }
        c->last_npass = be.npass;
        return GLP_OK;
    }
    // natural-order output (and tiny sizes): zero-pad + scale, then one size-N transform
    // shift^j tables (two-level), cached for the last (shift, log_n)
    if (c->shift_val != shift || c->shift_log_n != (int)log_n) {
        const size_t nlo = n < 4096 ? n : 4096, nhi = n > 4096 ? (n >> 12) : 0;
        std::vector<u64> lo(nlo), hi(nhi ? nhi : 1);
        u64 t = 1;
        for (size_t i = 0; i < nlo; i++) { lo[i] = t; t = gl_mul(t, shift); }
        if (nhi) { u64 sh = gl_pow(shift, 4096); t = 1; for (size_t i = 0; i < nhi; i++) { hi[i] = t; t = gl_mul(t, sh); } }
        GLP_HIPCHK(c, hipStreamSynchronize(c->stream));
        if (c->shift_lo) { hipFree(c->shift_lo); c->shift_lo = nullptr; }
        if (c->shift_hi) { hipFree(c->shift_hi); c->shift_hi = nullptr; }
        GLP_HIPCHK(c, hipMalloc((void**)&c->shift_lo, nlo * 8));
        GLP_HIPCHK(c, hipMemcpy(c->shift_lo, lo.data(), nlo * 8, hipMemcpyHostToDevice));
        if (nhi) {
            GLP_HIPCHK(c, hipMalloc((void**)&c->shift_hi, nhi * 8));
            GLP_HIPCHK(c, hipMemcpy(c->shift_hi, hi.data(), nhi * 8, hipMemcpyHostToDevice));
        }
        c->shift_val = shift;
        c->shift_log_n = (int)log_n;
    }
    const u64 total = (u64)batch << log_N;
    u64 blocks = (total + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(glp_lde_prep_kernel, dim3((unsigned)blocks), dim3(256), 0, c->stream, coeffs, out, log_n, log_N, batch,
                       c->shift_lo, c->shift_hi);
    GLP_HIPCHK(c, hipGetLastError());
    const u64 N = 1ull << log_N;
    return glp_ntt_impl(c, out, out, log_N, batch, N, N, flags & GLP_NTT_BITREV);
}

extern "C" int glp_field_op(glp_ctx* c, int op, const uint64_t* a, const uint64_t* b, uint64_t* out, uint64_t n) {
    if (!c) return GLP_E_INVALID;
    GLP_BIND(c);
    if (op < 0 || op > 10 || ((!a || !out || (!b && op != 4)) && n)) { glp_set_err(c, "glp_field_op: bad argument"); return GLP_E_INVALID; }
    if (n == 0) return GLP_OK;
    u64 blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(glp_field_op_kernel, dim3((unsigned)blocks), dim3(256), 0, c->stream, op, a, b, out, n);
    GLP_HIPCHK(c, hipGetLastError());
    return GLP_OK;
}

// witness placement: wire cell i takes the variable its index entry names (0xFFFFFFFF = an unused cell, zero)
__global__ void __launch_bounds__(256) glp_witness_place_kernel(u64* __restrict__ dst, const u64* __restrict__ src, u64 n_src, const u32* __restrict__ index,
                                                         u64 n, u32* __restrict__ bad) {
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) {
        const u32 k = index[i];
        u64 v = 0;
        if (k != 0xFFFFFFFFu) {
            if (k < n_src) v = src[k];
            else atomicOr(bad, 1u);
        }
        dst[i] = v;
    }
}

extern "C" int glp_gather_u64(glp_ctx* c, uint64_t* d_dst, const uint64_t* d_src, size_t n_src, const uint32_t* d_index, size_t n) {
    if (!c) return GLP_E_INVALID;
    GLP_BIND(c);
    if ((!d_dst || !d_index || (!d_src && n_src)) && n) { glp_set_err(c, "glp_gather_u64: bad argument"); return GLP_E_INVALID; }
    if (n == 0) return GLP_OK;
    GlpPoolBuf flag(c);
    if (flag.alloc(256) != hipSuccess) return GLP_E_HIP;
    GLP_HIPCHK(c, hipMemsetAsync(flag.p, 0, 4, c->stream));
    u64 blocks = (n + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(glp_witness_place_kernel, dim3((unsigned)blocks), dim3(256), 0, c->stream, d_dst, d_src, (u64)n_src, d_index, (u64)n, (u32*)flag.p);
    GLP_HIPCHK(c, hipGetLastError());
    u32 bad = 0;
    GLP_HIPCHK(c, hipMemcpyAsync(&bad, flag.p, 4, hipMemcpyDeviceToHost, c->stream));
    GLP_HIPCHK(c, hipStreamSynchronize(c->stream));
    if (bad) { glp_set_err(c, "glp_gather_u64: an index entry is out of range"); return GLP_E_INVALID; }
    return GLP_OK;
}

extern "C" int glp_transpose(glp_ctx* c, const uint64_t* in, uint64_t* out, uint64_t rows, uint64_t cols) {
    if (!c) return GLP_E_INVALID;
    GLP_BIND(c);
    if (!in || !out || in == out) { glp_set_err(c, "glp_transpose: bad argument"); return GLP_E_INVALID; }
    if (rows == 0 || cols == 0) return GLP_OK;
    const u64 tr = (rows + 31) / 32, tc = (cols + 31) / 32;
    if (tr * tc > 0x7fffffffull) { glp_set_err(c, "glp_transpose: too large"); return GLP_E_UNSUPPORTED; }
    hipLaunchKernelGGL(glp_transpose_kernel, dim3((unsigned)(tr * tc)), dim3(256), 0, c->stream, in, out, rows, cols, tc);
    GLP_HIPCHK(c, hipGetLastError());
    return GLP_OK;
}
