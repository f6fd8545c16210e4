// tests/isa/field_probe.hip — TEST INFRASTRUCTURE: instantiates every inline-asm primitive of gl_field.cuh in one gfx950 kernel so that
// tests/test_isa_hazards.py can inspect the ISA the toolchain emits around them (compiled device-only to assembly, never run).
#include <hip/hip_runtime.h>
#include "gl_field.cuh"

__global__ void glp_isa_probe(const u64* a, const u64* b, u64* out, u64 n) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u64 x = a[i], y = b[i];
    u64 r = gl_sub(x, y);
    r = gl_add(r, gl_mul(x, y));
    r = gl_add(r, gl_canon(gl_mul_nc(r, y)));
    r = gl_sub(r, gl_canon(gl_fold_small(x >> 7, y >> 7)));
    r = gl_add(r, gl_mad_eps<true>((u32)y, x));
    r = gl_add(r, gl_canon(gl_mad_eps<false>((u32)x, y)));
    r = gl_add(r, gl_mul_pow2<12>(x));
    r = gl_add(r, gl_mul_pow2<48>(y));
    r = gl_add(r, gl_mul_pow2<84>(r));
    r = gl_add(r, gl_mul_pow2<156>(x));
    r = gl_add(r, gl_reduce128(x, y));
    r = gl_add(r, gl_add_u32(x, (u32)y) % GL_P);
    // address arithmetic after the blocks (s_add_u32 / s_addc_u32 live across them is what the SCC clobber protects)
    out[i + (r & 3)] = r;
}
