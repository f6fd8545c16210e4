// tests/emu/hip_emu.h — TEST INFRASTRUCTURE: a minimal CPU emulation of the HIP execution
// model, just enough to run the product's kernel bodies (0-kno-blobstreamx_amd/csrc/*.cuh)
// unchanged on the host, under AddressSanitizer, before they ever touch a GPU.
// One workgroup at a time; every work-item is a real std::thread; __syncthreads() is a
// std::barrier; dynamic LDS is a heap block shared by the workgroup's threads (so ASan
// sees out-of-range LDS indices); __shfl* go through a per-wave scratch + wave barrier.
// It is never compiled into the product library.
#pragma once
#define GLP_EMU 1
#include <barrier>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <thread>
#include <vector>

struct glp_emu_dim3 { unsigned x = 1, y = 1, z = 1; };
inline thread_local glp_emu_dim3 threadIdx, blockIdx;
inline glp_emu_dim3 blockDim, gridDim;

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __restrict__
#define __shared__ static   // one workgroup at a time: a function-local static is shared by its threads

namespace glp_emu {
inline std::barrier<>* block_barrier = nullptr;
inline void* dyn_lds = nullptr;
inline std::vector<std::unique_ptr<std::barrier<>>> wave_barriers;
inline std::vector<uint64_t> wave_scratch;  // 64 slots per wave
}  // namespace glp_emu

inline void __syncthreads() { glp_emu::block_barrier->arrive_and_wait(); }
inline void* glp_emu_dyn_lds() { return glp_emu::dyn_lds; }
inline unsigned long long __umul64hi(unsigned long long a, unsigned long long b) {
    return (unsigned long long)(((unsigned __int128)a * b) >> 64);
}
inline unsigned __brev(unsigned v) {
    unsigned r = 0;
    for (int i = 0; i < 32; i++) r |= ((v >> i) & 1u) << (31 - i);
    return r;
}
#include <mutex>
inline void glp_emu_atomic_min(unsigned long long* p, unsigned long long v) {
    static std::mutex m;
    std::lock_guard<std::mutex> g(m);
    if (v < *p) *p = v;
}
// 64-wide wavefront shuffles (all lanes of the wave must call)
inline unsigned long long glp_emu_shfl(unsigned long long v, unsigned src_lane) {
    unsigned wave = threadIdx.x / 64, lane = threadIdx.x % 64;
    auto& bar = *glp_emu::wave_barriers[wave];
    glp_emu::wave_scratch[wave * 64 + lane] = v;
    bar.arrive_and_wait();
    unsigned long long r = glp_emu::wave_scratch[wave * 64 + (src_lane & 63)];
    bar.arrive_and_wait();
    return r;
}
inline unsigned long long __shfl(unsigned long long v, int src) { return glp_emu_shfl(v, (unsigned)src); }
inline unsigned long long __shfl_xor(unsigned long long v, int mask) { return glp_emu_shfl(v, (threadIdx.x % 64) ^ (unsigned)mask); }

// Run `body` once per work-item of a 1-D grid of 1-D blocks.
inline void glp_emu_launch(unsigned grid, unsigned block, size_t lds_bytes, const std::function<void()>& body) {
    gridDim.x = grid; blockDim.x = block;
    for (unsigned b = 0; b < grid; b++) {
        std::barrier<> bar(block);
        glp_emu::block_barrier = &bar;
        // exact-size heap block: ASan flags any LDS access past the requested bytes
        std::unique_ptr<unsigned char[]> lds(new unsigned char[lds_bytes ? lds_bytes : 1]);
        std::memset(lds.get(), 0xA5, lds_bytes);
        glp_emu::dyn_lds = lds.get();
        unsigned nw = (block + 63) / 64;
        glp_emu::wave_barriers.clear();
        for (unsigned w = 0; w < nw; w++) {
            unsigned lanes = (w + 1) * 64 <= block ? 64 : block - w * 64;
            glp_emu::wave_barriers.emplace_back(new std::barrier<>(lanes));
        }
        glp_emu::wave_scratch.assign(nw * 64, 0);
        std::vector<std::thread> ths;
        ths.reserve(block);
        for (unsigned t = 0; t < block; t++)
            ths.emplace_back([&, t, b] {
                threadIdx.x = t; blockIdx.x = b;
                body();
            });
        for (auto& th : ths) th.join();
    }
    glp_emu::block_barrier = nullptr;
    glp_emu::dyn_lds = nullptr;
}
