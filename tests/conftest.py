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


def field_params():
    """(two-adic generator, log2 w_64) of the PRODUCT build under test (GLP_LIB may select the alternative-generator library: csrc `make altgen`)"""
    return graft.load_package().field_params()


def default_generator():
    return field_params()[0] == pow(7, (P - 1) >> 32, P)


def root_of_unity(k):
    """the primitive 2^k-th root of unity of the build under test"""
    return pow(field_params()[0], 1 << (32 - k), P)


@pytest.fixture(scope="session")
def oracle():
    lib = graft.load_oracle()
    lib.orc_set_two_adic_generator.argtypes = [ctypes.c_uint64]
    lib.orc_two_adic_generator.restype = ctypes.c_uint64
    try:
        lib.orc_set_two_adic_generator(field_params()[0])      # the oracle computes in the subgroup generator the product build was compiled with
    except Exception:  # noqa: BLE001 — product library not built: the oracle keeps its default (generator 7)
        pass
    lib.orc_add.restype = lib.orc_sub.restype = lib.orc_mul.restype = ctypes.c_uint64
    lib.orc_pow.restype = lib.orc_inv.restype = lib.orc_root.restype = ctypes.c_uint64
    for f in (lib.orc_add, lib.orc_sub, lib.orc_mul, lib.orc_pow):
        f.argtypes = [ctypes.c_uint64, ctypes.c_uint64]
    lib.orc_inv.argtypes = [ctypes.c_uint64]
    lib.orc_root.argtypes = [ctypes.c_uint]
    lib.orc_dft_naive.argtypes = [u64p, u64p, ctypes.c_uint, ctypes.c_int]
    lib.orc_ntt.argtypes = [u64p, ctypes.c_uint, ctypes.c_uint64, ctypes.c_int]
    lib.orc_ntt_par.argtypes = [u64p, ctypes.c_uint, ctypes.c_int]
    lib.orc_ntt_fast.argtypes = [u64p, ctypes.c_uint, ctypes.c_uint64, ctypes.c_int]
    lib.orc_bitrev_rows.argtypes = [u64p, ctypes.c_uint, ctypes.c_uint64]
    lib.orc_lde_coset.argtypes = [u64p, u64p, ctypes.c_uint, ctypes.c_uint, ctypes.c_uint64, ctypes.c_uint64]
    lib.orc_transpose.argtypes = [u64p, u64p, ctypes.c_uint64, ctypes.c_uint64]
    lib.orc_poseidon_set_constants.argtypes = [u64p, u64p, u64p]
    lib.orc_poseidon_permute.argtypes = [u64p]
    lib.orc_hash_no_pad.argtypes = [u64p, ctypes.c_uint64, u64p]
    lib.orc_two_to_one.argtypes = [u64p, u64p, u64p]
    lib.orc_merkle.argtypes = [u64p, ctypes.c_uint64, ctypes.c_uint, ctypes.c_uint, u64p, u64p]
    lib.orc_fri_fold2.argtypes = [u64p, u64p, ctypes.c_uint, ctypes.c_uint64, u64p]
    lib.orc_sha256.argtypes = [ctypes.c_char_p, ctypes.c_uint64, ctypes.c_char_p, ctypes.c_void_p]
    lib.orc_sha512.argtypes = [ctypes.c_char_p, ctypes.c_uint64, ctypes.c_char_p, ctypes.c_void_p]
    lib.orc_hash_or_noop.argtypes = [u64p, ctypes.c_uint64, u64p]
    return lib


def poseidon_consts(kind):
    """(rc, circ, diag) as uint64 arrays: 'small' = the package default (small-integer MDS,
    fast path), 'big' = random 64-bit MDS entries (generic path)."""
    import importlib
    pc = importlib.import_module(graft.PKG_NAME + ".poseidon_constants")
    rc, circ, diag = pc.default_constants()
    rc = np.array(rc, dtype=np.uint64)
    if kind == "small":
        return rc, np.array(circ, dtype=np.uint64), np.array(diag, dtype=np.uint64)
    rng = np.random.default_rng(4242)
    if kind == "medium":   # largest entries the fast MDS path admits: all < 2^24 and sum < 2^24
        circ = rng.integers(1 << 19, (1 << 20) - 1, 12).astype(np.uint64)
        diag = rng.integers(1 << 19, (1 << 20) - 1, 12).astype(np.uint64)
        assert int(circ.sum()) + int(diag.max()) < (1 << 24)
        return rc, circ, diag
    return rc, rand_field(rng, 12), rand_field(rng, 12)


def oracle_merkle(oracle, leaves, cap_h):
    """leaves [n][leaf_len] -> (digests [*,4], cap [2^cap_h,4]) by the CPU oracle"""
    a = np.ascontiguousarray(leaves, dtype=np.uint64)
    n, leaf_len = a.shape
    log_leaves = n.bit_length() - 1
    dig = np.zeros(((2 << log_leaves) - (1 << cap_h), 4), dtype=np.uint64)
    cap = np.zeros((1 << cap_h, 4), dtype=np.uint64)
    oracle.orc_merkle(ptr(a), leaf_len, log_leaves, cap_h, ptr(dig), ptr(cap))
    return dig, cap


@pytest.fixture(scope="session")
def pkg():
    return graft.load_package()


@pytest.fixture(scope="session")
def emu():
    d = os.path.join(ROOT, "tests", "emu")
    # GLP_EMU_ASAN=1 (with LD_PRELOAD=libasan.so, ASAN_OPTIONS=detect_leaks=0): run the kernel bodies
    # under AddressSanitizer + UBSan — the sanitizer leg of the CPU build (GPU ASan is not available)
    asan = os.environ.get("GLP_EMU_ASAN") == "1"
    alt = os.environ.get("GLP_EMU_ALTGEN") == "1"          # kernel bodies compiled on the alternative two-adic generator (set together with GLP_LIB)
    subprocess.run(["make", "-s"] + (["asan"] if asan else ["altgen"] if alt else []), cwd=d, check=True)
    lib = ctypes.CDLL(os.path.join(d, "libglp_emu_asan.so" if asan else "libglp_emu_altgen.so" if alt else "libglp_emu.so"))
    lib.emu_ntt.argtypes = [u64p, u64p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_int, ctypes.c_uint, ctypes.c_int,
                            ctypes.c_int, ctypes.c_char_p]
    lib.emu_lde_coset_bitrev.argtypes = [u64p, u64p, ctypes.c_int, ctypes.c_int, ctypes.c_uint, ctypes.c_uint64, ctypes.c_char_p]
    for name in ("emu_gl_add", "emu_gl_sub", "emu_gl_mul", "emu_gl_reduce128"):
        f = getattr(lib, name)
        f.restype = ctypes.c_uint64
        f.argtypes = [ctypes.c_uint64, ctypes.c_uint64]
    lib.emu_gl_mul_pow2.restype = ctypes.c_uint64
    lib.emu_gl_mul_pow2.argtypes = [ctypes.c_uint64, ctypes.c_int]
    lib.emu_poseidon_permute.argtypes = [u64p, ctypes.c_uint64, u64p, ctypes.c_int]
    lib.emu_poseidon_grouped_available.argtypes = [u64p]
    lib.emu_merkle.argtypes = [u64p, ctypes.c_uint64, ctypes.c_int, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, u64p, u64p,
                               ctypes.c_int, ctypes.c_int]
    lib.emu_fri_fold2.argtypes = [u64p, u64p, ctypes.c_uint32, ctypes.c_uint64, u64p]
    lib.emu_sha256_trace.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p,
                                     ctypes.c_void_p]
    lib.emu_sha512_trace.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p,
                                     ctypes.c_void_p]
    lib.emu_challenger.argtypes = [u64p, ctypes.c_int, u64p, ctypes.c_uint64, u64p]
    lib.emu_eval_at_ext.argtypes = [u64p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32, u64p, u64p]
    lib.emu_fri_combine.argtypes = [u64p, ctypes.c_uint32, ctypes.c_uint32, u64p, u64p, u64p, ctypes.c_uint64, u64p, ctypes.c_int,
                                    ctypes.c_int]
    lib.emu_pow.argtypes = [u64p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint32, u64p, ctypes.c_int,
                            ctypes.POINTER(ctypes.c_ulonglong)]
    lib.emu_plonk_zs.argtypes = [u64p, u64p, ctypes.c_uint32, ctypes.c_uint32, u64p, u64p, u64p]
    lib.emu_plonk_quotient.argtypes = [u64p, u64p, u64p, u64p, u64p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32,
                                       ctypes.c_uint32, u64p, u64p, u64p, u64p, u64p]
    lib.emu_poseidon_gate_fill_rows.argtypes = [u64p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_uint32, u64p]
    lib.emu_sha_gate_fill_rows.argtypes = [u64p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32]
    lib.emu_bitrev_scale.argtypes = [u64p, u64p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint64]
    lib.emu_ed25519_witness.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_uint64,
                                        ctypes.c_void_p, ctypes.c_void_p]
    lib.emu_tm_merkle_root_var.argtypes = [ctypes.c_char_p, u64p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_char_p]
    lib.emu_tm_merkle_root.argtypes = [ctypes.c_char_p, ctypes.c_uint32, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_char_p]
    return lib


@pytest.fixture(scope="session")
def prover(pkg):
    """the HIP product path; fails loudly (no fallback) when the GPU or the library is missing"""
    pr = pkg.Prover(0)
    yield pr
    pr.close()
