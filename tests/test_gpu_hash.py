"""GPU parity tests for Poseidon / Merkle / FRI fold / SHA-2 traces / PolynomialBatch, through
the C ABI, against the CPU oracle.  Poseidon constants are injected; parity is with the
oracle's restatement of the permutation structure, not with plonky2 (unpinned)."""
import ctypes
import hashlib
import json
import os

import numpy as np
import pytest

from conftest import P, oracle_merkle, poseidon_consts, ptr, rand_field

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def use_consts(prover, oracle, kind):
    rc, circ, diag = poseidon_consts(kind)
    prover.set_poseidon_constants(rc, circ, diag)
    oracle.orc_poseidon_set_constants(ptr(rc), ptr(circ), ptr(diag))


def test_hashing_needs_constants(pkg):
    pr = pkg.Prover(0)
    d = pr.alloc(96)
    with pytest.raises(pkg.GlpError, match="GLP_E_STATE"):
        pr.poseidon_permute_(d, 1)
    pr.close()


@pytest.mark.parametrize("kind", ["small", "medium", "big"])
def test_poseidon_permutation(prover, oracle, kind):
    use_consts(prover, oracle, kind)
    rng = np.random.default_rng(31)
    st = rand_field(rng, (1000, 12))
    st[0, :] = 0
    st[1, :] = P - 1
    ref = st.copy()
    for i in range(ref.shape[0]):
        row = ref[i].copy()
        oracle.orc_poseidon_permute(ptr(row))
        ref[i] = row
    assert np.array_equal(prover.poseidon_permute(st), ref)


@pytest.mark.parametrize("kind,leaf_len,log_leaves,cap_h", [("small", 135, 12, 4), ("small", 3, 8, 0), ("small", 4, 5, 5),
                                                            ("small", 8, 10, 1), ("small", 9, 6, 6), ("small", 1, 0, 0),
                                                            ("big", 20, 9, 3), ("small", 16, 14, 4)])
def test_merkle_tree(prover, oracle, kind, leaf_len, log_leaves, cap_h):
    use_consts(prover, oracle, kind)
    rng = np.random.default_rng(leaf_len * 100 + log_leaves)
    leaves = rand_field(rng, (1 << log_leaves, leaf_len))
    dig_ref, cap_ref = oracle_merkle(oracle, leaves, cap_h)
    dig, cap = prover.merkle_tree(leaves, cap_h)
    assert np.array_equal(dig, dig_ref) and np.array_equal(cap, cap_ref)
    dig2, cap2 = prover.merkle_tree(np.ascontiguousarray(leaves.T), cap_h, poly_major=True)
    assert np.array_equal(dig2, dig_ref) and np.array_equal(cap2, cap_ref)


def test_polynomial_batch_from_values(prover, oracle, pkg):
    """values -> ifft -> coset LDE x8 (bit-reversed) -> Merkle cap, vs the same chain on the oracle"""
    use_consts(prover, oracle, "small")
    rng = np.random.default_rng(41)
    n_polys, log_n, rate_bits, cap_h = 19, 10, 3, 4
    vals = rand_field(rng, (n_polys, 1 << log_n))
    pb = pkg.PolynomialBatch.from_values(prover, vals, rate_bits, cap_h)
    coeffs = vals.copy()
    oracle.orc_ntt(ptr(coeffs), log_n, n_polys, 1)
    assert np.array_equal(pb.coeffs.download(vals.shape), coeffs)
    N = 1 << (log_n + rate_bits)
    lde = np.zeros((n_polys, N), dtype=np.uint64)
    oracle.orc_lde_coset(ptr(coeffs), ptr(lde), log_n, rate_bits, n_polys, 7)
    oracle.orc_bitrev_rows(ptr(lde), log_n + rate_bits, n_polys)
    assert np.array_equal(pb.lde.download(lde.shape), lde)
    dig_ref, cap_ref = oracle_merkle(oracle, np.ascontiguousarray(lde.T), cap_h)
    assert np.array_equal(pb.cap, cap_ref)
    assert np.array_equal(pb.digests.download(dig_ref.shape), dig_ref)
    pb.free()


@pytest.mark.parametrize("log_n", [1, 2, 5, 13, 16])
def test_fri_fold(prover, oracle, log_n):
    rng = np.random.default_rng(log_n)
    ev = rand_field(rng, (1 << log_n, 2))
    beta = rand_field(rng, 2)
    ref = np.zeros((1 << (log_n - 1), 2), dtype=np.uint64)
    oracle.orc_fri_fold2(ptr(ev), ptr(ref), log_n, 7, ptr(beta))
    assert np.array_equal(prover.fri_fold2(ev, 7, beta), ref)


def test_fri_fold_chain_to_constant(prover):
    """folding the LDE of a degree < 2^k polynomial k times leaves a constant codeword —
    the property FRI rests on, checked end to end on the GPU path at a larger size"""
    rng = np.random.default_rng(5)
    log_d, rate_bits = 10, 3
    coeffs = rand_field(rng, 1 << log_d)
    ev0 = prover.lde(coeffs, rate_bits, bitrev=True)
    ev = np.stack([ev0, np.zeros_like(ev0)], axis=1)
    shift = 7
    for _ in range(log_d):
        beta = rand_field(rng, 2)
        ev = prover.fri_fold2(ev, shift, beta)
        shift = shift * shift % P
    assert ev.shape[0] == 1 << rate_bits
    assert (ev == ev[0]).all()


def test_sha2_traces(prover, oracle, pkg):
    with open(os.path.join(G, "sha2.json")) as f:
        cases = json.load(f)["cases"]
    for block, name, fn, orc_fn, wdt, tw in ((64, "sha256", prover.sha256_trace, oracle.orc_sha256, np.uint32, 576),
                                              (128, "sha512", prover.sha512_trace, oracle.orc_sha512, np.uint64, 720)):
        groups = {}
        for c in cases:
            m = bytes.fromhex(c["msg"])
            p = pkg.sha_pad(m, block)
            groups.setdefault(len(p) // block, []).append((m, p, c[name]))
        for nb, items in groups.items():
            padded = np.frombuffer(b"".join(p for _, p, _ in items), dtype=np.uint8).reshape(len(items), nb * block)
            dig, tr = fn(padded, nb)
            width = 4 if block == 64 else 8
            for i, (m, _, want) in enumerate(items):
                assert b"".join(int(v).to_bytes(width, "big") for v in dig[i]).hex() == want
                ref_tr = np.zeros((nb, tw), dtype=wdt)
                o = ctypes.create_string_buffer(32 if block == 64 else 64)
                orc_fn(m, len(m), o, ref_tr.ctypes.data)
                assert np.array_equal(tr[i], ref_tr)


def test_sha256_many_messages_vs_hashlib(prover, pkg):
    """a validator-set sized batch: 2000 leaves of 0x00 || 40 bytes (Tendermint leaf hashing)"""
    rng = np.random.default_rng(8)
    msgs = [b"\x00" + rng.integers(0, 256, 40, dtype=np.uint8).tobytes() for _ in range(2000)]
    padded = np.frombuffer(b"".join(pkg.sha_pad(m, 64, 1) for m in msgs), dtype=np.uint8).reshape(len(msgs), 64)
    dig, _ = prover.sha256_trace(padded, 1, want_trace=False)
    for i in (0, 1, 999, 1999):
        assert b"".join(int(v).to_bytes(4, "big") for v in dig[i]) == hashlib.sha256(msgs[i]).digest()


def test_tendermint_merkle_root(prover):
    """BASELINE configs[0] shape: validator-set hashing, 100 validators of 40-byte leaves, and the
    other committed cases, vs hashlib-generated golden roots"""
    with open(os.path.join(G, "tendermint_merkle.json")) as f:
        t = json.load(f)
    for c in t["cases"]:
        assert prover.tm_merkle_root(bytes.fromhex(c["leaves"]), t["leaf_len"]).hex() == c["root"], c["n"]
