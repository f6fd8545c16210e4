"""Rows a6 / a7 on the CPU emulation: permutation products (Z, partial products) and the
quotient on the LDE domain, against the Python big-int restatements in tests/plonk_ref.py."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import fri_verifier as fv  # noqa: E402
import plonk_ref as pr  # noqa: E402
from conftest import P, poseidon_consts, ptr, rand_field  # noqa: E402


def lde_bitrev(oracle, vals, log_n, rb):
    """values on the trace domain -> coset LDE values in bit-reversed order (via the oracle)"""
    co = np.ascontiguousarray(vals).copy()
    k = co.shape[0]
    oracle.orc_ntt(ptr(co), log_n, k, 1)
    out = np.zeros((k, 1 << (log_n + rb)), dtype=np.uint64)
    oracle.orc_lde_coset(ptr(co), ptr(out), log_n, rb, k, 7)
    oracle.orc_bitrev_rows(ptr(out), log_n + rb, k)
    return out, co


@pytest.mark.parametrize("log_n,W", [(4, 8), (5, 16), (11, 8)])
def test_partial_products_and_z(emu, log_n, W):
    rng = np.random.default_rng(log_n * 10 + W)
    circ = pr.build_circuit(rng, log_n, W)
    beta = [int(v) for v in rand_field(rng, 2)]
    gamma = [int(v) for v in rand_field(rng, 2)]
    want = pr.ref_zs(circ, beta, gamma)
    got = np.zeros_like(want)
    b, g = np.array(beta, dtype=np.uint64), np.array(gamma, dtype=np.uint64)
    assert emu.emu_plonk_zs(ptr(circ["wires"]), ptr(circ["sigmas"]), log_n, W, ptr(b), ptr(g), ptr(got)) == 0
    assert np.array_equal(got, want)


def test_broken_copy_constraint_is_detected_by_reference():
    rng = np.random.default_rng(1)
    circ = pr.build_circuit(rng, 4, 8, copy_prob=0.9)
    circ["wires"][0, 3] ^= np.uint64(1)
    with pytest.raises(AssertionError):
        pr.ref_zs(circ, [3, 5], [7, 11])


def _emu_quotient(emu, oracle, circ, beta, gamma, alpha, rb=3):
    """(got, want): the product's K7 body under emulation vs the Python restatement, on oracle-made LDEs"""
    log_n, W, R = circ["log_n"], circ["W"], circ["R"]
    zs = pr.ref_zs(circ, beta, gamma)
    L = {}
    for name, vals in (("consts", circ["consts"]), ("sigmas", circ["sigmas"]), ("wires", circ["wires"]), ("zs", zs)):
        L[name], _ = lde_bitrev(oracle, vals, log_n, rb)
    want = pr.ref_quotient(circ, {k: [[int(x) for x in r] for r in v] for k, v in L.items()}, beta, gamma, alpha, rb)
    n, N = 1 << log_n, 1 << (log_n + rb)
    pi = None
    if circ["n_public"]:
        pv = np.zeros((1, n), dtype=np.uint64)
        pv[0, :circ["n_public"]] = circ["public"]
        pi, _ = lde_bitrev(oracle, pv, log_n, rb)
    pos = np.concatenate([np.array(a, dtype=np.uint64) for a in circ["pos_consts"]]) if circ["flags"] & pr.FLAG_POSEIDON else None
    got = np.zeros((2, N), dtype=np.uint64)
    arr = lambda v: np.array(v, dtype=np.uint64)
    assert emu.emu_plonk_quotient(ptr(L["consts"]), ptr(L["sigmas"]), ptr(L["wires"]), ptr(L["zs"]), ptr(pi) if pi is not None else None, log_n, rb,
                                  W, R, circ["flags"], ptr(pos) if pos is not None else None, ptr(arr(beta)), ptr(arr(gamma)), ptr(arr(alpha)),
                                  ptr(got)) == 0
    return got, want


def _assert_polynomial_quotient(emu, oracle, got, log_n, rb=3):
    """the quotient is a genuine polynomial: un-bit-reverse, inverse NTT, unshift -> degree < 8n, and (because the
    constraints hold on the trace domain) its top coefficients vanish too"""
    nat = np.zeros_like(got)
    sinv = pow(7, P - 2, P)
    assert emu.emu_bitrev_scale(ptr(got), ptr(nat), log_n + rb, 2, 1) == 0
    oracle.orc_ntt(ptr(nat), log_n + rb, 2, 1)
    co = [[int(v) * pow(sinv, j, P) % P for j, v in enumerate(r)] for r in nat]
    n = 1 << log_n
    for t in range(2):
        assert any(co[t]), "quotient should not be identically zero"
        assert not any(co[t][8 * n - 8:]), "degree bound 8n - 9 violated"
    return sinv


@pytest.mark.parametrize("log_n,W", [(3, 8), (4, 16)])
def test_quotient_on_lde_domain(emu, oracle, log_n, W):
    rb = 3
    rng = np.random.default_rng(log_n + W)
    circ = pr.build_circuit(rng, log_n, W)
    beta = [int(v) for v in rand_field(rng, 2)]
    gamma = [int(v) for v in rand_field(rng, 2)]
    alpha = [int(v) for v in rand_field(rng, 2)]
    got, want = _emu_quotient(emu, oracle, circ, beta, gamma, alpha, rb)
    assert [[int(v) for v in r] for r in got] == want
    sinv = _assert_polynomial_quotient(emu, oracle, got, log_n, rb)
    N = 1 << (log_n + rb)
    # and the scaled variant of the same kernel
    nat2 = np.zeros_like(got)
    assert emu.emu_bitrev_scale(ptr(got), ptr(nat2), log_n + rb, 2, sinv) == 0
    for t in range(2):
        for i in (0, 1, 5, N - 1):
            assert int(nat2[t][i]) == int(got[t][fv.rev(i, log_n + rb)]) * pow(sinv, i, P) % P


def test_poseidon_row_is_the_permutation(emu, oracle):
    """the Poseidon row's witness: the Python restatement, the product's fill kernel (emulated) and the oracle's permutation agree,
    and a filled row satisfies all 123 constraints (swap bit 0 and 1) while a corrupted one does not"""
    consts = poseidon_consts("small")
    oracle.orc_poseidon_set_constants(*(ptr(a) for a in consts))
    ic = pr.int_consts(consts)
    rng = np.random.default_rng(12)
    log_n, W = 3, 136
    n = 1 << log_n
    wires = rand_field(rng, (W, n))
    wires[:12, 0] = 0
    wires[:12, 1] = P - 1
    rows = np.array([0, 1, 3, 6], dtype=np.uint32)
    wires[pr.POS_SWAP, rows] = [0, 1, 1, 0]                      # the swap bit is an INPUT of the row (a routed cell): rows 1 and 3 swap
    before = wires.copy()
    flat = np.concatenate([np.array(a, dtype=np.uint64) for a in consts])
    assert emu.emu_poseidon_gate_fill_rows(ptr(wires), log_n, rows.ctypes.data, len(rows), ptr(flat)) == 0
    for r in range(n):
        if r not in rows:
            assert np.array_equal(wires[:, r], before[:, r])
            continue
        swap = int(before[pr.POS_SWAP, r])
        row = pr.poseidon_row([int(v) for v in before[:12, r]], ic, swap=swap)
        assert [int(v) for v in wires[:pr.POS_WIRES, r]] == row
        st = before[:12, r].copy()
        if swap:
            st[:4], st[4:8] = before[4:8, r], before[:4, r]
        oracle.orc_poseidon_permute(ptr(st))
        assert [int(v) for v in st] == row[12:24]
        assert not any(pr.poseidon_constraints(pr.Base, row, ic))
        assert np.array_equal(wires[pr.POS_WIRES:, r], before[pr.POS_WIRES:, r])
        for j in (0, 5, 12, 24, 25, 60, 61, 82, 130, 131, 134):
            if j == 24 and r < 2:
                continue                                         # rows 0 and 1 have equal input halves: either swap bit satisfies them
            bad = list(row)
            bad[j] = (bad[j] + 1) % P
            assert any(pr.poseidon_constraints(pr.Base, bad, ic)), f"wire {j} is not constrained"
    # a swap "bit" that is not a bit cannot satisfy the row
    row2 = pr.poseidon_row([int(v) for v in before[:12, 3]], ic, swap=2)
    assert any(pr.poseidon_constraints(pr.Base, row2, ic))


@pytest.mark.parametrize("log_n,W,R,n_public,pos_rows", [(3, 16, 8, 2, ()), (4, 24, 16, 5, ()), (3, 136, 80, 3, (1, 4, 5)), (4, 136, 24, 0, (0, 15))])
def test_quotient_with_public_inputs_advice_wires_and_poseidon_rows(emu, oracle, log_n, W, R, n_public, pos_rows):
    """the extended gate set under emulation: constant term c2, advice (unrouted) wires, public-input rows and Poseidon rows —
    K6 over the routed wires and K7 against the Python restatements, and the quotient is a polynomial of degree < 8n"""
    consts = poseidon_consts("small")
    rng = np.random.default_rng(1000 * log_n + W + n_public)
    circ = pr.build_circuit(rng, log_n, W, n_routed=R, n_public=n_public, poseidon_rows=pos_rows, consts=consts)
    beta = [int(v) for v in rand_field(rng, 2)]
    gamma = [int(v) for v in rand_field(rng, 2)]
    alpha = [int(v) for v in rand_field(rng, 2)]
    want_zs = pr.ref_zs(circ, beta, gamma)
    got_zs = np.zeros_like(want_zs)
    b, g = np.array(beta, dtype=np.uint64), np.array(gamma, dtype=np.uint64)
    routed = np.ascontiguousarray(circ["wires"][:R])
    assert emu.emu_plonk_zs(ptr(routed), ptr(circ["sigmas"]), log_n, R, ptr(b), ptr(g), ptr(got_zs)) == 0
    assert np.array_equal(got_zs, want_zs)
    got, want = _emu_quotient(emu, oracle, circ, beta, gamma, alpha)
    assert [[int(v) for v in r] for r in got] == want
    _assert_polynomial_quotient(emu, oracle, got, log_n)
    # a wrong public input, a broken gate output and a corrupted Poseidon wire each make the quotient a non-polynomial
    def broken(mut):
        c2 = dict(circ)
        c2["wires"] = circ["wires"].copy()
        c2["public"] = list(circ["public"])
        mut(c2)
        g2, w2 = _emu_quotient(emu, oracle, c2, beta, gamma, alpha)
        assert [[int(v) for v in r] for r in g2] == w2
        with pytest.raises(AssertionError):
            _assert_polynomial_quotient(emu, oracle, g2, log_n)
    if n_public:
        broken(lambda c: c["public"].__setitem__(n_public - 1, (c["public"][n_public - 1] + 1) % P))
    if pos_rows:
        def corrupt(c):
            c["wires"][100, pos_rows[0]] ^= np.uint64(1)            # an advice wire of a Poseidon row (no copy constraint)
        broken(corrupt)


@pytest.mark.parametrize("log_n,W,R,n_public,pos_rows,sha_rows", [(3, 144, 16, 1, (), (2, 3, 4, 5, 6, 7)), (4, 144, 24, 0, (1,), (0, 2, 3, 5, 8, 9, 13, 14, 15))])
def test_quotient_with_sha_rows(emu, oracle, log_n, W, R, n_public, pos_rows, sha_rows):
    """SHA-256 rows under emulation (ASan build included): the filler kernel rebuilds the bit wires, K7 + K7s equal the Python restatement
    point for point, the quotient is a polynomial of degree < 8n, and a flipped bit wire or output word makes it a non-polynomial"""
    consts = poseidon_consts("small")
    rng = np.random.default_rng(31 * log_n + len(sha_rows))
    circ = pr.build_circuit(rng, log_n, W, n_routed=R, n_public=n_public, poseidon_rows=pos_rows, consts=consts, sha_rows=sha_rows)
    rows = np.array(sorted(sha_rows), dtype=np.uint32)
    kinds = np.array([int(np.argmax(circ["consts"][6:10, r])) for r in rows], dtype=np.uint32)
    blank = circ["wires"].copy()
    blank[12:pr.SHA_WIRES, rows] = 0
    assert emu.emu_sha_gate_fill_rows(ptr(blank), log_n, rows.ctypes.data, kinds.ctypes.data, len(rows)) == 0
    assert np.array_equal(blank, circ["wires"])
    beta = [int(v) for v in rand_field(rng, 2)]
    gamma = [int(v) for v in rand_field(rng, 2)]
    alpha = [int(v) for v in rand_field(rng, 2)]
    got, want = _emu_quotient(emu, oracle, circ, beta, gamma, alpha)
    assert [[int(v) for v in r] for r in got] == want
    _assert_polynomial_quotient(emu, oracle, got, log_n)
    for wire, k in ((12 + 3, 0), (108 + 30, 1), (140, 2)):
        c2 = dict(circ)
        c2["wires"] = circ["wires"].copy()
        c2["wires"][wire, rows[k]] ^= np.uint64(1)
        g2, w2 = _emu_quotient(emu, oracle, c2, beta, gamma, alpha)
        assert [[int(v) for v in r] for r in g2] == w2
        with pytest.raises(AssertionError):
            _assert_polynomial_quotient(emu, oracle, g2, log_n)


@pytest.mark.parametrize("log_n,W,R,pos_rows,sha_rows,ext_rows", [(3, 16, 16, (), (), (1, 2, 5)), (4, 144, 24, (1,), (0, 2, 3), (4, 5, 6, 12))])
def test_quotient_with_ext_rows(emu, oracle, log_n, W, R, pos_rows, sha_rows, ext_rows):
    """extension-arithmetic rows under emulation: K7 equals the restatement, the quotient is a polynomial, a wrong product component is not"""
    consts = poseidon_consts("small")
    rng = np.random.default_rng(17 * log_n + W)
    beta = [int(v) for v in rand_field(rng, 2)]
    gamma = [int(v) for v in rand_field(rng, 2)]
    alpha = [int(v) for v in rand_field(rng, 2)]
    for copy_prob in (0.5, 0.0):
        circ = pr.build_circuit(rng, log_n, W, copy_prob=copy_prob, n_routed=R, n_public=1, poseidon_rows=pos_rows, consts=consts, sha_rows=sha_rows,
                                ext_rows=ext_rows)
        got, want = _emu_quotient(emu, oracle, circ, beta, gamma, alpha)
        assert [[int(v) for v in r] for r in got] == want
        _assert_polynomial_quotient(emu, oracle, got, log_n)
    for wire in (6, 7):                         # on the copy-free circuit: only the row's own equations notice
        c2 = dict(circ)
        c2["wires"] = circ["wires"].copy()
        c2["wires"][wire, ext_rows[0]] ^= np.uint64(1)
        g2, w2 = _emu_quotient(emu, oracle, c2, beta, gamma, alpha)
        assert [[int(v) for v in r] for r in g2] == w2
        with pytest.raises(AssertionError):
            _assert_polynomial_quotient(emu, oracle, g2, log_n)
