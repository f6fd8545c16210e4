import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

P = 2**64 - 2**32 + 1
u64p = ctypes.POINTER(ctypes.c_uint64)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def ptr(a):
    return a.ctypes.data_as(u64p)


def rand_field(rng, shape):
    """uniform canonical field elements, full 64-bit range"""
    hi = rng.integers(0, 1 << 32, size=shape, dtype=np.uint64)
    lo = rng.integers(0, 1 << 32, size=shape, dtype=np.uint64)
    return ((hi << np.uint64(32)) | lo) % np.uint64(P)


@pytest.fixture(scope="session")
def oracle():
    lib = graft.load_oracle()
    lib.orc_add.restype = lib.orc_sub.restype = lib.orc_mul.restype = ctypes.c_uint64
    lib.orc_pow.restype = lib.orc_inv.restype = lib.orc_root.restype = ctypes.c_uint64
    for f in (lib.orc_add, lib.orc_sub, lib.orc_mul, lib.orc_pow):
        f.argtypes = [ctypes.c_uint64, ctypes.c_uint64]
    lib.orc_inv.argtypes = [ctypes.c_uint64]
    lib.orc_root.argtypes = [ctypes.c_uint]
    lib.orc_dft_naive.argtypes = [u64p, u64p, ctypes.c_uint, ctypes.c_int]
    lib.orc_ntt.argtypes = [u64p, ctypes.c_uint, ctypes.c_uint64, ctypes.c_int]
    lib.orc_ntt_par.argtypes = [u64p, ctypes.c_uint, ctypes.c_int]
    lib.orc_bitrev_rows.argtypes = [u64p, ctypes.c_uint, ctypes.c_uint64]
    lib.orc_lde_coset.argtypes = [u64p, u64p, ctypes.c_uint, ctypes.c_uint, ctypes.c_uint64, ctypes.c_uint64]
    lib.orc_transpose.argtypes = [u64p, u64p, ctypes.c_uint64, ctypes.c_uint64]
    return lib


@pytest.fixture(scope="session")
def pkg():
    return graft.load_package()


@pytest.fixture(scope="session")
def emu():
    d = os.path.join(ROOT, "tests", "emu")
    subprocess.run(["make", "-s"], cwd=d, check=True)
    lib = ctypes.CDLL(os.path.join(d, "libglp_emu.so"))
    lib.emu_ntt.argtypes = [u64p, u64p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_int, ctypes.c_uint, ctypes.c_int,
                            ctypes.c_int, ctypes.c_char_p]
    for name in ("emu_gl_add", "emu_gl_sub", "emu_gl_mul", "emu_gl_reduce128"):
        f = getattr(lib, name)
        f.restype = ctypes.c_uint64
        f.argtypes = [ctypes.c_uint64, ctypes.c_uint64]
    lib.emu_gl_mul_pow2.restype = ctypes.c_uint64
    lib.emu_gl_mul_pow2.argtypes = [ctypes.c_uint64, ctypes.c_int]
    return lib


@pytest.fixture(scope="session")
def prover(pkg):
    """the HIP product path; fails loudly (no fallback) when the GPU or the library is missing"""
    pr = pkg.Prover(0)
    yield pr
    pr.close()
