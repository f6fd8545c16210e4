"""Host-side binding of libglprover.so (include/glprover.h) for the prover hot path.

The reference's host language is Rust and its prover is plonky2 (neither is in
/root/reference, which holds only `.gitignore:1` and `changelog.md:1-2`, nor in this image),
so this module mirrors the *names* of the upstream operator surface for the path — recalled,
unverified: ``fft`` / ``ifft`` / ``coset_fft`` / ``lde`` of
``plonky2_field::fft`` and ``PolynomialBatch::from_values`` / ``from_coeffs`` of
``plonky2::fri::oracle`` — on top of the C ABI.  It is test/bench plumbing: ctypes for the
calls, numpy (or torch tensors via ``data_ptr()``) for buffers.

There is no CPU fallback.  Importing works anywhere (so the ABI can be inspected without a
GPU), but ``Prover()`` raises ``GlpError`` unless a gfx950 device is usable and the library
was built (``python -c "import __graft_entry__ as g; g.build()"``).
"""
import ctypes
import os

import numpy as np

P = 2**64 - 2**32 + 1
COSET_SHIFT = 7  # multiplicative generator, the coset shift used for LDEs
NTT_INVERSE = 1
NTT_BITREV = 2

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libglprover.so")

_u64p = ctypes.POINTER(ctypes.c_uint64)
_vp = ctypes.c_void_p


class GlpError(RuntimeError):
    pass


_ERR = {-1: "GLP_E_INVALID", -2: "GLP_E_NODEVICE", -3: "GLP_E_HIP", -4: "GLP_E_NOMEM",
        -5: "GLP_E_UNSUPPORTED", -6: "GLP_E_STATE"}

_lib = None


def load_library():
    """dlopen libglprover.so; raises GlpError (never falls back) when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise GlpError(f"{LIB_PATH} not built: run __graft_entry__.build() (hipcc --offload-arch=gfx950)")
    lib = ctypes.CDLL(LIB_PATH)
    sig = {
        "glp_create": (ctypes.c_int, [ctypes.POINTER(_vp), ctypes.c_int]),
        "glp_destroy": (None, [_vp]),
        "glp_last_error": (ctypes.c_char_p, [_vp]),
        "glp_version": (ctypes.c_char_p, []),
        "glp_alloc": (ctypes.c_int, [_vp, ctypes.POINTER(_vp), ctypes.c_size_t]),
        "glp_free": (ctypes.c_int, [_vp, _vp]),
        "glp_h2d": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_size_t]),
        "glp_d2h": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_size_t]),
        "glp_sync": (ctypes.c_int, [_vp]),
        "glp_set_stream": (ctypes.c_int, [_vp, _vp]),
        "glp_timer_start": (ctypes.c_int, [_vp]),
        "glp_timer_stop": (ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_float)]),
        "glp_ntt": (ctypes.c_int, [_vp, _vp, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_int]),
        "glp_ntt_ex": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint64,
                                      ctypes.c_uint64, ctypes.c_uint32]),
        "glp_lde_coset": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32,
                                         ctypes.c_uint64, ctypes.c_uint32]),
        "glp_transpose": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_uint64, ctypes.c_uint64]),
        "glp_ntt_set_plan": (ctypes.c_int, [_vp, ctypes.c_uint32, ctypes.c_char_p]),
        "glp_ntt_describe_plan": (ctypes.c_int, [_vp, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_char_p,
                                                 ctypes.c_size_t]),
        "glp_set_profiling": (ctypes.c_int, [_vp, ctypes.c_int]),
        "glp_last_pass_ms": (ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int)]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    # optional entry points (bound when the library exports them)
    opt = {
        "glp_set_poseidon_constants": (ctypes.c_int, [_vp, _vp, ctypes.c_size_t, _vp, _vp]),
        "glp_poseidon_permute": (ctypes.c_int, [_vp, _vp, ctypes.c_uint64]),
        "glp_merkle": (ctypes.c_int, [_vp, _vp, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, _vp, _vp]),
        "glp_merkle_from_polys": (ctypes.c_int, [_vp, _vp, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32,
                                                 ctypes.c_uint32, _vp, _vp]),
        "glp_fri_fold2": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_uint32, ctypes.c_uint64, _vp]),
        "glp_sha256_trace": (ctypes.c_int, [_vp, _vp, ctypes.c_uint64, ctypes.c_uint32, _vp, _vp]),
        "glp_sha512_trace": (ctypes.c_int, [_vp, _vp, ctypes.c_uint64, ctypes.c_uint32, _vp, _vp]),
    }
    for name, (res, args) in opt.items():
        if hasattr(lib, name):
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
    _lib = lib
    return lib


class DeviceBuffer:
    """A hipMalloc'd block owned by a Prover (freed with it or by .free())."""

    def __init__(self, prover, nbytes):
        self.prover = prover
        self.nbytes = int(nbytes)
        p = _vp()
        prover._chk(prover.lib.glp_alloc(prover.ctx, ctypes.byref(p), self.nbytes), "glp_alloc")
        self.ptr = p.value
        prover._bufs.add(self)

    def free(self):
        if self.ptr:
            self.prover._chk(self.prover.lib.glp_free(self.prover.ctx, self.ptr), "glp_free")
            self.ptr = None
            self.prover._bufs.discard(self)

    def upload(self, arr):
        a = np.ascontiguousarray(arr)
        assert a.nbytes <= self.nbytes
        self.prover._chk(self.prover.lib.glp_h2d(self.prover.ctx, self.ptr, a.ctypes.data, a.nbytes), "glp_h2d")
        return self

    def download(self, shape, dtype=np.uint64, offset_bytes=0):
        out = np.empty(shape, dtype=dtype)
        assert offset_bytes + out.nbytes <= self.nbytes
        self.prover._chk(self.prover.lib.glp_d2h(self.prover.ctx, out.ctypes.data, self.ptr + offset_bytes, out.nbytes),
                         "glp_d2h")
        return out


def _ptr(x):
    """device pointer of a DeviceBuffer / torch tensor / int"""
    if isinstance(x, DeviceBuffer):
        return x.ptr
    if hasattr(x, "data_ptr"):
        return x.data_ptr()
    return int(x)


class Prover:
    """One glp_ctx: one GPU, one stream.  Device-level calls take device pointers
    (DeviceBuffer, torch tensor or int); the numpy helpers below them copy in and out."""

    def __init__(self, device=0):
        self.lib = load_library()
        self._bufs = set()
        ctx = _vp()
        rc = self.lib.glp_create(ctypes.byref(ctx), int(device))
        if rc != 0:
            raise GlpError(f"glp_create(device={device}) failed: {_ERR.get(rc, rc)} — no CPU fallback exists")
        self.ctx = ctx

    def close(self):
        if getattr(self, "ctx", None):
            for b in list(self._bufs):
                b.free()
            self.lib.glp_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc, what):
        if rc != 0:
            msg = self.lib.glp_last_error(self.ctx)
            raise GlpError(f"{what}: {_ERR.get(rc, rc)}: {msg.decode() if msg else ''}")

    # ---- memory / stream -------------------------------------------------------------
    def alloc(self, nbytes):
        return DeviceBuffer(self, nbytes)

    def to_device(self, arr):
        a = np.ascontiguousarray(arr)
        return DeviceBuffer(self, max(a.nbytes, 8)).upload(a)

    def sync(self):
        self._chk(self.lib.glp_sync(self.ctx), "glp_sync")

    def set_stream(self, hip_stream):
        self._chk(self.lib.glp_set_stream(self.ctx, hip_stream), "glp_set_stream")

    def timer_start(self):
        self._chk(self.lib.glp_timer_start(self.ctx), "glp_timer_start")

    def timer_stop(self):
        ms = ctypes.c_float()
        self._chk(self.lib.glp_timer_stop(self.ctx, ctypes.byref(ms)), "glp_timer_stop")
        return ms.value

    # ---- device-level transforms -------------------------------------------------------
    def ntt_(self, d_io, log_n, batch=1, inverse=False):
        self._chk(self.lib.glp_ntt(self.ctx, _ptr(d_io), log_n, batch, 1 if inverse else 0), "glp_ntt")

    def ntt_ex(self, d_src, d_dst, log_n, batch=1, src_stride=None, dst_stride=None, flags=0):
        n = 1 << log_n
        self._chk(self.lib.glp_ntt_ex(self.ctx, _ptr(d_src), _ptr(d_dst), log_n, batch, src_stride or n,
                                      dst_stride or n, flags), "glp_ntt_ex")

    def lde_coset_(self, d_coeffs, d_out, log_n, rate_bits, batch=1, shift=COSET_SHIFT, flags=0):
        self._chk(self.lib.glp_lde_coset(self.ctx, _ptr(d_coeffs), _ptr(d_out), log_n, rate_bits, batch, shift, flags),
                  "glp_lde_coset")

    def transpose_(self, d_in, d_out, rows, cols):
        self._chk(self.lib.glp_transpose(self.ctx, _ptr(d_in), _ptr(d_out), rows, cols), "glp_transpose")

    def set_plan(self, log_n, plan):
        self._chk(self.lib.glp_ntt_set_plan(self.ctx, log_n, plan.encode() if plan else None), "glp_ntt_set_plan")

    def describe_plan(self, log_n, flags=0):
        buf = ctypes.create_string_buffer(256)
        self._chk(self.lib.glp_ntt_describe_plan(self.ctx, log_n, flags, buf, 256), "glp_ntt_describe_plan")
        return buf.value.decode()

    def set_profiling(self, on):
        self._chk(self.lib.glp_set_profiling(self.ctx, 1 if on else 0), "glp_set_profiling")

    def last_pass_ms(self):
        ms = (ctypes.c_float * 8)()
        n = ctypes.c_int()
        self._chk(self.lib.glp_last_pass_ms(self.ctx, ms, ctypes.byref(n)), "glp_last_pass_ms")
        return [ms[i] for i in range(n.value)]

    # ---- numpy in / numpy out (upstream names, recalled: plonky2_field::fft) -----------
    def _xform(self, x, flags):
        x = np.ascontiguousarray(x, dtype=np.uint64)
        one = x.ndim == 1
        a = x.reshape(1, -1) if one else x
        batch, n = a.shape
        log_n = n.bit_length() - 1
        assert 1 << log_n == n, "length must be a power of two"
        d = self.to_device(a)
        self.ntt_ex(d, d, log_n, batch, flags=flags)
        out = d.download(a.shape)
        d.free()
        return out[0] if one else out

    def fft(self, coeffs):
        """coefficients -> evaluations on <w_n>, natural order"""
        return self._xform(coeffs, 0)

    def ifft(self, values):
        """evaluations -> coefficients (inverse transform, scaled by 1/n)"""
        return self._xform(values, NTT_INVERSE)

    def fft_bitrev(self, coeffs):
        return self._xform(coeffs, NTT_BITREV)

    def lde(self, coeffs, rate_bits, shift=COSET_SHIFT, bitrev=False):
        """coset low-degree extension: [batch][n] coefficients -> [batch][n << rate_bits] values"""
        x = np.ascontiguousarray(coeffs, dtype=np.uint64)
        one = x.ndim == 1
        a = x.reshape(1, -1) if one else x
        batch, n = a.shape
        log_n = n.bit_length() - 1
        assert 1 << log_n == n
        d_in = self.to_device(a)
        d_out = self.alloc(a.nbytes << rate_bits)
        self.lde_coset_(d_in, d_out, log_n, rate_bits, batch, shift, NTT_BITREV if bitrev else 0)
        out = d_out.download((batch, n << rate_bits))
        d_in.free()
        d_out.free()
        return out[0] if one else out

    def coset_fft(self, coeffs, shift=COSET_SHIFT):
        return self.lde(coeffs, 0, shift)

    def transpose(self, mat):
        m = np.ascontiguousarray(mat, dtype=np.uint64)
        rows, cols = m.shape
        d_in = self.to_device(m)
        d_out = self.alloc(m.nbytes)
        self.transpose_(d_in, d_out, rows, cols)
        out = d_out.download((cols, rows))
        d_in.free()
        d_out.free()
        return out
