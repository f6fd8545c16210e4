"""Poseidon / Merkle / FRI fold / SHA-2 kernel bodies run on the CPU (tests/emu) and compared
with the oracle.  Poseidon parity is against the oracle's NAIVE restatement of the permutation
structure with injected constants — not against plonky2 (constants unavailable; unpinned)."""
import ctypes
import hashlib
import json
import os

import numpy as np
import pytest

from conftest import P, oracle_merkle, poseidon_consts, ptr, rand_field

G = os.path.join(os.path.dirname(__file__), "golden")


def consts384(kind):
    rc, circ, diag = poseidon_consts(kind)
    return np.concatenate([rc, circ, diag]).astype(np.uint64), (rc, circ, diag)


@pytest.mark.parametrize("kind", ["small", "medium", "big"])
def test_emulated_poseidon_permutation(emu, oracle, kind):
    c384, (rc, circ, diag) = consts384(kind)
    oracle.orc_poseidon_set_constants(ptr(rc), ptr(circ), ptr(diag))
    rng = np.random.default_rng(21)
    st = rand_field(rng, (70, 12))
    st[0, :] = 0
    st[1, :] = P - 1
    st[2, :] = (1 << 32) - 1
    st[3, :] = P - (1 << 32)
    ref = st.copy()
    for i in range(ref.shape[0]):
        row = ref[i].copy()
        oracle.orc_poseidon_permute(ptr(row))
        ref[i] = row
    got = st.copy()
    assert emu.emu_poseidon_permute(ptr(got), got.shape[0], ptr(c384), 0 if kind == "big" else 1) == 0
    assert np.array_equal(got, ref)
    if kind == "small":
        # grouped partial rounds (three rounds per dot-product group) give the same permutation
        assert emu.emu_poseidon_grouped_available(ptr(c384)) == 1
        got = st.copy()
        assert emu.emu_poseidon_permute(ptr(got), got.shape[0], ptr(c384), 2) == 0
        assert np.array_equal(got, ref)
    if kind == "medium":
        assert emu.emu_poseidon_grouped_available(ptr(c384)) == 0      # cubes of 2^20 entries are not small


@pytest.mark.parametrize("kind,leaf_len,log_leaves,cap_h", [("small", 135, 5, 2), ("small", 3, 4, 0), ("small", 4, 5, 5),
                                                            ("small", 8, 3, 1), ("small", 9, 6, 6), ("big", 20, 5, 3)])
def test_emulated_merkle(emu, oracle, kind, leaf_len, log_leaves, cap_h):
    c384, (rc, circ, diag) = consts384(kind)
    oracle.orc_poseidon_set_constants(ptr(rc), ptr(circ), ptr(diag))
    rng = np.random.default_rng(leaf_len * 100 + log_leaves)
    leaves = rand_field(rng, (1 << log_leaves, leaf_len))
    dig_ref, cap_ref = oracle_merkle(oracle, leaves, cap_h)
    small = 1 if kind == "small" else 0
    dig = np.zeros_like(dig_ref)
    assert emu.emu_merkle(ptr(leaves), leaf_len, 0, leaf_len, log_leaves, cap_h, ptr(dig), ptr(c384), small, 0) == 0
    assert np.array_equal(dig, dig_ref)
    assert np.array_equal(dig[-(1 << cap_h):], cap_ref)
    # the lane-cooperative kernels (small levels: 16 lanes per node, the last levels fused in one launch) build the same tree
    for coop_max in ((4096, 16) if leaf_len in (135, 20) else (16,)):      # (1024 emulated threads per fused launch: the slow part of this file)
        dig3 = np.zeros_like(dig_ref)
        assert emu.emu_merkle(ptr(leaves), leaf_len, 0, leaf_len, log_leaves, cap_h, ptr(dig3), ptr(c384), small, coop_max) == 0
        assert np.array_equal(dig3, dig_ref), coop_max
    # polynomial-major source gives the same tree
    polys = np.ascontiguousarray(leaves.T)
    dig2 = np.zeros_like(dig_ref)
    assert emu.emu_merkle(ptr(polys), 1 << log_leaves, 1, leaf_len, log_leaves, cap_h, ptr(dig2), ptr(c384), 2 if small else 0, 32 if leaf_len == 135 else 0) == 0
    assert np.array_equal(dig2, dig_ref)


@pytest.mark.parametrize("log_n", [1, 2, 5, 13])
def test_emulated_fri_fold(emu, oracle, log_n):
    rng = np.random.default_rng(log_n)
    ev = rand_field(rng, (1 << log_n, 2))
    beta = rand_field(rng, 2)
    ref = np.zeros((1 << (log_n - 1), 2), dtype=np.uint64)
    oracle.orc_fri_fold2(ptr(ev), ptr(ref), log_n, 7, ptr(beta))
    out = np.zeros_like(ref)
    assert emu.emu_fri_fold2(ptr(ev), ptr(out), log_n, 7, ptr(beta)) == 0
    assert np.array_equal(out, ref)


def test_fri_fold_is_polynomial_folding(oracle):
    """the oracle's fold equals even/odd coefficient folding: fold(f)(y) = fe(y) + beta*fo(y)"""
    rng = np.random.default_rng(2)
    log_n = 4
    n = 1 << log_n
    coeffs = [int(v) for v in rand_field(rng, n)]
    beta = [int(rand_field(rng, 1)[0]), 0]
    w = pow(7, (P - 1) >> log_n, P)
    rev = lambda i, b: int(format(i, f"0{b}b")[::-1], 2) if b else 0
    xs = [7 * pow(w, rev(i, log_n), P) % P for i in range(n)]
    ev = np.array([[sum(c * pow(x, j, P) for j, c in enumerate(coeffs)) % P, 0] for x in xs], dtype=np.uint64)
    out = np.zeros((n // 2, 2), dtype=np.uint64)
    oracle.orc_fri_fold2(ptr(ev), ptr(out), log_n, 7, ptr(np.array(beta, dtype=np.uint64)))
    folded = [(coeffs[2 * j] + beta[0] * coeffs[2 * j + 1]) % P for j in range(n // 2)]
    w2 = w * w % P
    ys = [49 * pow(w2, rev(i, log_n - 1), P) % P for i in range(n // 2)]
    want = [sum(c * pow(y, j, P) for j, c in enumerate(folded)) % P for y in ys]
    assert [int(v) for v in out[:, 0]] == want and not out[:, 1].any()


def k_tables():
    import re
    src = open(os.path.join(os.path.dirname(__file__), "..", "oracle", "gl_oracle.c")).read()
    k256 = [int(x, 16) for x in re.findall(r"0x[0-9a-f]{8}\b", src.split("K256[64]")[1].split("};")[0])]
    k512 = [int(x, 16) for x in re.findall(r"0x[0-9a-f]{16}", src.split("K512[80]")[1].split("};")[0])]
    assert len(k256) == 64 and len(k512) == 80
    return np.array(k256, dtype=np.uint32), np.array(k512, dtype=np.uint64)


def test_emulated_sha2_traces(emu, oracle, pkg):
    k256, k512 = k_tables()
    with open(os.path.join(G, "sha2.json")) as f:
        cases = json.load(f)["cases"]
    for block, name, emu_fn, orc_fn, wdt, tw in ((64, "sha256", emu.emu_sha256_trace, oracle.orc_sha256, np.uint32, 576),
                                                   (128, "sha512", emu.emu_sha512_trace, oracle.orc_sha512, np.uint64, 720)):
        by_blocks = {}
        for c in cases:
            m = bytes.fromhex(c["msg"])
            p = pkg.sha_pad(m, block)
            by_blocks.setdefault(len(p) // block, []).append((m, p, c[name]))
        for nb, items in by_blocks.items():
            padded = np.frombuffer(b"".join(p for _, p, _ in items), dtype=np.uint8).reshape(len(items), nb * block).copy()
            dig = np.zeros((len(items), 8), dtype=wdt)
            tr = np.zeros((len(items), nb, tw), dtype=wdt)
            kt = k256 if block == 64 else k512
            assert emu_fn(padded.ctypes.data, len(items), nb, dig.ctypes.data, tr.ctypes.data, kt.ctypes.data) == 0
            for i, (m, _, want) in enumerate(items):
                width = 4 if block == 64 else 8
                assert b"".join(int(v).to_bytes(width, "big") for v in dig[i]).hex() == want
                ref_tr = np.zeros((nb, tw), dtype=wdt)
                o = ctypes.create_string_buffer(32 if block == 64 else 64)
                orc_fn(m, len(m), o, ref_tr.ctypes.data)
                assert np.array_equal(tr[i], ref_tr)
            if block == 64:
                assert hashlib.sha256(items[0][0]).hexdigest() == items[0][2]


def test_emulated_tendermint_merkle_root(emu):
    """validator-set style hashing (BASELINE configs[0]) vs the committed hashlib-generated vectors"""
    k256, _ = k_tables()
    with open(os.path.join(G, "tendermint_merkle.json")) as f:
        t = json.load(f)
    for c in t["cases"]:
        if c["n"] == 0:
            continue
        leaves = bytes.fromhex(c["leaves"])
        out = ctypes.create_string_buffer(32)
        assert emu.emu_tm_merkle_root(leaves, t["leaf_len"], c["n"], k256.ctypes.data, out) == 0
        assert out.raw.hex() == c["root"], c["n"]
    # other leaf lengths, including the two-block boundary (55/56 bytes with the prefix)
    import hashlib

    def tm_root(items):
        if len(items) == 1:
            return hashlib.sha256(b"\x00" + items[0]).digest()
        k = 1
        while k * 2 < len(items):
            k *= 2
        return hashlib.sha256(b"\x01" + tm_root(items[:k]) + tm_root(items[k:])).digest()

    rng = np.random.default_rng(3)
    for leaf_len in (1, 32, 54, 55, 56, 63, 64, 118):
        n = int(rng.integers(1, 20))
        items = [rng.integers(0, 256, leaf_len, dtype=np.uint8).tobytes() for _ in range(n)]
        out = ctypes.create_string_buffer(32)
        assert emu.emu_tm_merkle_root(b"".join(items), leaf_len, n, k256.ctypes.data, out) == 0
        assert out.raw == tm_root(items), (leaf_len, n)
