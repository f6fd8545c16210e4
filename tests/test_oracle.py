"""CPU oracle vs the committed golden vectors (tests/golden, generated from Python big-int
and hashlib by gen_golden.py).  These pin the oracle to the published definitions; the
reference mount holds nothing to pin it to (parity with plonky2 unpinned)."""
import ctypes
import hashlib
import json
import os

import numpy as np

from conftest import P, ptr

G = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    with open(os.path.join(G, name)) as f:
        return json.load(f)


def test_field_ops(oracle):
    f = load("field.json")
    assert int(f["p"]) == P
    for a, b, s, d, m in f["binary"]:
        a, b = int(a), int(b)
        assert oracle.orc_add(a, b) == int(s)
        assert oracle.orc_sub(a, b) == int(d)
        assert oracle.orc_mul(a, b) == int(m)
    for a, ai in f["inv"]:
        assert oracle.orc_inv(int(a)) == int(ai)
    for k, r in f["roots"]:
        assert oracle.orc_root(k) == int(r)
    # the structure the kernels rely on: 2 has order 192 and w_64 = 2^39
    assert oracle.orc_pow(2, 96) == P - 1
    assert oracle.orc_root(6) == 2**39 % P


def test_ntt_matches_golden_and_naive(oracle):
    for c in load("ntt.json")["cases"]:
        log_n = c["log_n"]
        x = np.array([int(v) for v in c["x"]], dtype=np.uint64)
        fwd = np.array([int(v) for v in c["fwd"]], dtype=np.uint64)
        inv = np.array([int(v) for v in c["inv"]], dtype=np.uint64)
        want_inv = inv
        a = x.copy()
        oracle.orc_ntt(ptr(a), log_n, 1, 0)
        assert np.array_equal(a, fwd)
        a = x.copy()
        oracle.orc_ntt(ptr(a), log_n, 1, 1)
        assert np.array_equal(a, inv)
        a = x.copy()
        oracle.orc_ntt_par(ptr(a), log_n, 0)
        assert np.array_equal(a, fwd)
        for inv, want in ((0, fwd), (1, want_inv)):          # the cpu_baseline leg (gl_fast.c)
            a = x.copy()
            oracle.orc_ntt_fast(ptr(a), log_n, 1, inv)
            assert np.array_equal(a, want)
        if log_n <= 8:
            out = np.zeros_like(x)
            oracle.orc_dft_naive(ptr(x.copy()), ptr(out), log_n, 0)
            assert np.array_equal(out, fwd)


def test_lde_matches_golden(oracle):
    for c in load("lde.json")["cases"]:
        coeffs = np.array([int(v) for v in c["coeffs"]], dtype=np.uint64)
        want = np.array([int(v) for v in c["values"]], dtype=np.uint64)
        out = np.zeros(len(want), dtype=np.uint64)
        oracle.orc_lde_coset(ptr(coeffs), ptr(out), c["log_n"], c["rate_bits"], 1, int(c["shift"]))
        assert np.array_equal(out, want)


def test_lde_bitrev_matches_golden(oracle):
    """the oracle's LDE + bit reversal against the big-int vectors of the bit-reversed layout"""
    for c in load("lde_bitrev.json")["cases"]:
        coeffs = np.array([int(v) for v in c["coeffs"]], dtype=np.uint64)
        want = np.array([int(v) for v in c["values_bitrev"]], dtype=np.uint64)
        out = np.zeros(len(want), dtype=np.uint64)
        oracle.orc_lde_coset(ptr(coeffs), ptr(out), c["log_n"], c["rate_bits"], 1, int(c["shift"]))
        oracle.orc_bitrev_rows(ptr(out), c["log_n"] + c["rate_bits"], 1)
        assert np.array_equal(out, want)


def test_sha2_known_answers(oracle):
    oracle.orc_sha256.argtypes = [ctypes.c_char_p, ctypes.c_uint64, ctypes.c_char_p, ctypes.c_void_p]
    oracle.orc_sha512.argtypes = [ctypes.c_char_p, ctypes.c_uint64, ctypes.c_char_p, ctypes.c_void_p]
    for c in load("sha2.json")["cases"]:
        m = bytes.fromhex(c["msg"])
        o = ctypes.create_string_buffer(32)
        oracle.orc_sha256(m, len(m), o, None)
        assert o.raw.hex() == c["sha256"] == hashlib.sha256(m).hexdigest()
        o = ctypes.create_string_buffer(64)
        oracle.orc_sha512(m, len(m), o, None)
        assert o.raw.hex() == c["sha512"]
    # FIPS 180-4 "abc"
    assert load("sha2.json")["cases"][1]["sha256"] == "ba7816bf8f01cfea414140de5dae2223b00361a396177a9cb410ff61f20015ad"


def test_sha256_trace_is_consistent(oracle):
    """the per-round trace reproduces the digest: final state of last block + chaining"""
    oracle.orc_sha256.argtypes = [ctypes.c_char_p, ctypes.c_uint64, ctypes.c_char_p, ctypes.c_void_p]
    m = b"a" * 100
    nb = (len(m) + 9 + 63) // 64
    tr = np.zeros(576 * nb, dtype=np.uint32)
    o = ctypes.create_string_buffer(32)
    oracle.orc_sha256(m, len(m), o, tr.ctypes.data)
    h = np.array([0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19], dtype=np.uint32)
    for b in range(nb):
        last = tr[576 * b + 64 + 8 * 63: 576 * b + 64 + 8 * 64]
        h = (h + last).astype(np.uint32)
    assert b"".join(int(v).to_bytes(4, "big") for v in h) == o.raw


def test_tendermint_merkle(oracle):
    oracle.orc_tm_merkle_root.argtypes = [ctypes.c_char_p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_char_p]
    t = load("tendermint_merkle.json")
    for c in t["cases"]:
        leaves = bytes.fromhex(c["leaves"])
        o = ctypes.create_string_buffer(32)
        oracle.orc_tm_merkle_root(leaves, t["leaf_len"], c["n"], o)
        assert o.raw.hex() == c["root"]


def test_fast_baseline_matches_naive_oracle_at_2_16(oracle):
    from conftest import rand_field
    rng = np.random.default_rng(16)
    x = rand_field(rng, (3, 1 << 16))
    a, b = x.copy(), x.copy()
    oracle.orc_ntt(ptr(a), 16, 3, 0)
    oracle.orc_ntt_fast(ptr(b), 16, 3, 0)
    assert np.array_equal(a, b)


def test_fast_commitment_matches_the_naive_composition(oracle):
    """oracle/gl_fast.c::orc_commit_fast (the prove path's CPU baseline: inverse transform, coset LDE, bit-reversed columns as leaves, Poseidon
    Merkle tree) against the same stage composed from the NAIVE oracle functions, for a wide batch (sponge), a narrow one (<= 4: the padded leaf
    itself) and both test constant sets"""
    from conftest import oracle_merkle, poseidon_consts, rand_field
    u64p = ctypes.POINTER(ctypes.c_uint64)
    oracle.orc_commit_fast.argtypes = [u64p, ctypes.c_uint, ctypes.c_uint64, ctypes.c_uint, ctypes.c_uint, ctypes.c_uint64, u64p]
    oracle.orc_commit_fast.restype = ctypes.c_int
    oracle.orc_lde_coset.argtypes = [u64p, u64p, ctypes.c_uint, ctypes.c_uint, ctypes.c_uint64, ctypes.c_uint64]
    rng = np.random.default_rng(77)
    for kind, log_n, n_polys, rb, cap_h in (("small", 6, 11, 3, 2), ("big", 5, 3, 2, 1), ("small", 4, 20, 3, 4)):
        consts = poseidon_consts(kind)
        oracle.orc_poseidon_set_constants(*(ptr(a) for a in consts))
        oracle.orc_fast_set_poseidon(*(ptr(a) for a in consts))
        n, N = 1 << log_n, 1 << (log_n + rb)
        vals = rand_field(rng, (n_polys, n))
        coeffs = vals.copy()
        oracle.orc_ntt(ptr(coeffs), log_n, n_polys, 1)
        lde = np.zeros((n_polys, N), dtype=np.uint64)
        oracle.orc_lde_coset(ptr(coeffs), ptr(lde), log_n, rb, n_polys, 7)
        oracle.orc_bitrev_rows(ptr(lde), log_n + rb, n_polys)
        leaves = np.ascontiguousarray(lde.T)
        _, want = oracle_merkle(oracle, leaves, cap_h)
        got = np.zeros(4 << cap_h, dtype=np.uint64)
        work = vals.copy()
        assert oracle.orc_commit_fast(ptr(work), log_n, n_polys, rb, cap_h, 7, ptr(got)) == 0
        assert np.array_equal(got, np.asarray(want).reshape(-1)), (kind, log_n, n_polys)
        assert np.array_equal(work, coeffs)                       # the values are left as coefficients
