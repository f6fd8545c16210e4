// fri_kernels.cuh — kernels of the FRI opening proof (upstream names
// recalled, unverified: plonky2::fri::prover::{fri_proof, fri_committed_trees,
// fri_proof_of_work, fri_prover_query_rounds}, PolynomialBatch::prove_openings — reference
// file:line NONE, the mount is empty).  The protocol these serve is BUILD-DEFINED and written
// down in DESIGN.md §3.5; it is self-verifying (tests/fri_verifier.py), not plonky2's format.
//
//   glp_ext_powers_kernel   z^j for j < n (extension element), two-level
//   glp_eval_ext_kernel     f_p(z) = sum_j c_{p,j} z^j for every polynomial of a batch
//   glp_fri_combine_kernel  G(x_i) = (sum_k alpha^k f_k(x_i) - Y) / (x_i - z) on the LDE domain
//   glp_pow_kernel          proof-of-work grinding (smallest nonce in a window)
//   glp_gather_kernel       out[k] = src[offset[k]]   (query openings, Merkle paths)
// Plain HIP C++ without AMD builtins (tests/emu runs these bodies on the CPU).
#pragma once
#include "gl_field.cuh"
#include "hash_kernels.cuh"

GL_HD gl_ext2 gl_ext_inv(gl_ext2 x) {
    // (a + bX)^-1 = (a - bX) / (a^2 - 7 b^2)
    const u64 bb = gl_mul(x.b, x.b);
    u64 b7 = gl_add(gl_add(gl_add(bb, bb), gl_add(bb, bb)), gl_add(gl_add(bb, bb), bb));
    const u64 nrm = gl_sub(gl_mul(x.a, x.a), b7);
    const u64 ni = gl_inv(nrm);
    return {gl_mul(x.a, ni), gl_mul(gl_neg(x.b), ni)};
}
GL_HD gl_ext2 gl_ext_pow(gl_ext2 x, u64 e) {
    gl_ext2 r{1, 0};
    while (e) {
        if (e & 1) r = gl_ext_mul(r, x);
        x = gl_ext_mul(x, x);
        e >>= 1;
    }
    return r;
}

// zp[j] = z^j (2 u64 each), j < n.  lo[j] = z^j (j < 256), hi[j] = z^(256 j): host-built tables.
template <int UNUSED = 0>
__global__ void __launch_bounds__(256) glp_ext_powers_kernel(u64* __restrict__ zp, u64 n, const u64* __restrict__ lo,
                                                             const u64* __restrict__ hi) {
    for (u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (u64)gridDim.x * blockDim.x) {
        const u64 l = j & 255u, h = j >> 8;
        const gl_ext2 v = gl_ext_mul(gl_ext2{lo[2 * l], lo[2 * l + 1]}, gl_ext2{hi[2 * h], hi[2 * h + 1]});
        zp[2 * j] = v.a;
        zp[2 * j + 1] = v.b;
    }
}

// partial[p][c] = sum over chunk c of coeffs[p][j] * z^j   (chunk = 4096 coefficients)
// grid = n_polys * n_chunks, block = 256; LDS tree reduction.
#define GLP_EVAL_CHUNK 4096u
template <int UNUSED = 0>
__global__ void __launch_bounds__(256) glp_eval_ext_kernel(const u64* __restrict__ coeffs, u64 poly_stride, u64 n,
                                                           u32 n_chunks, const u64* __restrict__ zp, u64* __restrict__ partial) {
    __shared__ u64 ra[256], rb[256];
    const u32 p = blockIdx.x / n_chunks, c = blockIdx.x % n_chunks;
    const u64 j0 = (u64)c * GLP_EVAL_CHUNK;
    u64 sa = 0, sb = 0;
    for (u32 t = threadIdx.x; t < GLP_EVAL_CHUNK; t += 256) {
        const u64 j = j0 + t;
        if (j < n) {
            const u64 cf = coeffs[(u64)p * poly_stride + j];
            sa = gl_add(sa, gl_mul(cf, zp[2 * j]));
            sb = gl_add(sb, gl_mul(cf, zp[2 * j + 1]));
        }
    }
    ra[threadIdx.x] = sa;
    rb[threadIdx.x] = sb;
    __syncthreads();
    for (u32 s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) {
            ra[threadIdx.x] = gl_add(ra[threadIdx.x], ra[threadIdx.x + s]);
            rb[threadIdx.x] = gl_add(rb[threadIdx.x], rb[threadIdx.x + s]);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        partial[2 * (u64)blockIdx.x] = ra[0];
        partial[2 * (u64)blockIdx.x + 1] = rb[0];
    }
}

// One polynomial batch's contribution to the combined codeword, accumulated into acc[i]
// (extension, [N][2]):  acc[i] += sum_p alpha_pow[p] * lde[p][i].   lde is polynomial-major,
// bit-reversed index order; alpha_pow: [n_polys][2] (already offset by the batch's position).
// When `finish` is set:  acc[i] = (acc[i] - Y) / (x_i - z),  x_i = shift * w_N^{rev(i)}.
// Each work-item owns 4 consecutive points so the 4 extension inversions share one field
// inversion (Montgomery's trick).
struct GlpCombineArgs {
    const u64* lde; u64 poly_stride; u32 n_polys;
    const u64* alpha_pow;
    u64* acc; u32 log_N; u32 first; u32 finish;
    gl_ext2 Y, z; u64 shift;
    const u64* w_lo; const u64* w_hi;       // forward two-level table of w_N
};
template <int UNUSED = 0>
__global__ void __launch_bounds__(256) glp_fri_combine_kernel(GlpCombineArgs a) {
    const u64 N = 1ull << a.log_N;
    const u64 i0 = ((u64)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i0 >= N) return;
    gl_ext2 s[4];
    glp_static_for<0, 4>([&](auto k_) {
        constexpr int k = decltype(k_)::value;
        s[k] = a.first ? gl_ext2{0, 0} : gl_ext2{a.acc[2 * (i0 + k)], a.acc[2 * (i0 + k) + 1]};
    });
    for (u32 p = 0; p < a.n_polys; p++) {
        const u64 aa = a.alpha_pow[2 * p], ab = a.alpha_pow[2 * p + 1];
        const u64* row = a.lde + (u64)p * a.poly_stride + i0;
        glp_static_for<0, 4>([&](auto k_) {
            constexpr int k = decltype(k_)::value;
            const u64 v = row[k];
            s[k].a = gl_add(s[k].a, gl_mul(v, aa));
            s[k].b = gl_add(s[k].b, gl_mul(v, ab));
        });
    }
    if (a.finish) {
        gl_ext2 d[4];
        u64 nrm[4];
        glp_static_for<0, 4>([&](auto k_) {
            constexpr int k = decltype(k_)::value;
            const u64 i = i0 + k;
            u64 e = 0;                         // bit reversal over log_N bits (portable form)
            for (u32 b = 0; b < a.log_N; b++) e |= ((i >> b) & 1ull) << (a.log_N - 1 - b);
            u64 x = a.w_lo[e & 4095u];
            if (a.w_hi) x = gl_mul(x, a.w_hi[e >> 12]);
            x = gl_mul(x, a.shift);
            d[k] = gl_ext2{gl_sub(x, a.z.a), gl_neg(a.z.b)};          // x - z
            const u64 bb = gl_mul(d[k].b, d[k].b);
            const u64 b7 = gl_add(gl_add(gl_add(bb, bb), gl_add(bb, bb)), gl_add(gl_add(bb, bb), bb));
            nrm[k] = gl_sub(gl_mul(d[k].a, d[k].a), b7);              // norm, nonzero unless x == z
        });
        // batch inversion of the 4 norms
        const u64 p01 = gl_mul(nrm[0], nrm[1]), p012 = gl_mul(p01, nrm[2]), p0123 = gl_mul(p012, nrm[3]);
        u64 inv = gl_inv(p0123);
        u64 ni[4];
        ni[3] = gl_mul(inv, p012); inv = gl_mul(inv, nrm[3]);
        ni[2] = gl_mul(inv, p01); inv = gl_mul(inv, nrm[2]);
        ni[1] = gl_mul(inv, nrm[0]);
        ni[0] = gl_mul(inv, nrm[1]);
        glp_static_for<0, 4>([&](auto k_) {
            constexpr int k = decltype(k_)::value;
            const gl_ext2 dinv{gl_mul(d[k].a, ni[k]), gl_mul(gl_neg(d[k].b), ni[k])};
            s[k] = gl_ext_mul(gl_ext_sub(s[k], a.Y), dinv);
        });
    }
    glp_static_for<0, 4>([&](auto k_) {
        constexpr int k = decltype(k_)::value;
        a.acc[2 * (i0 + k)] = s[k].a;
        a.acc[2 * (i0 + k) + 1] = s[k].b;
    });
}

// Proof of work: nonce is accepted when the first output word of
// Poseidon(seed[0..4], nonce, 0, ..., 0) has its top pow_bits bits clear.  Every work-item
// tries `per_thread` consecutive nonces of the window [base, base + count); the smallest
// accepted nonce of the window ends up in *found (initialised to ~0 by the host).
template <bool SMALL>
__global__ void __launch_bounds__(256) glp_pow_kernel(const u64* __restrict__ seed, u64 base, u64 count, u32 pow_bits,
                                                      unsigned long long* found, GlpPoseidonConsts k) {
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count) return;
    const u64 nonce = base + t;
    u64 s[12];
    glp_static_for<0, 4>([&](auto j_) { constexpr int j = decltype(j_)::value; s[j] = seed[j]; });
    s[4] = nonce % GL_P;
    glp_static_for<5, 12>([&](auto j_) { constexpr int j = decltype(j_)::value; s[j] = 0; });
    glp_poseidon_permute<SMALL>(s, k);
    if ((s[0] >> (64 - pow_bits)) == 0) {
#if defined(GLP_EMU)
        glp_emu_atomic_min(found, (unsigned long long)nonce);
#else
        atomicMin(found, (unsigned long long)nonce);
#endif
    }
}

template <int UNUSED = 0>
__global__ void __launch_bounds__(256) glp_gather_kernel(const u64* __restrict__ src, const u64* __restrict__ offs, u64 n,
                                                         u64* __restrict__ out) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = src[offs[i]];
}
