"""Rows a6 / a7 on the CPU emulation: permutation products (Z, partial products) and the
quotient on the LDE domain, against the Python big-int restatements in tests/plonk_ref.py."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import fri_verifier as fv  # noqa: E402
import plonk_ref as pr  # noqa: E402
from conftest import P, ptr, rand_field  # noqa: E402


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


@pytest.mark.parametrize("log_n,W", [(3, 8), (4, 16)])
def test_quotient_on_lde_domain(emu, oracle, log_n, W):
    rb = 3
    rng = np.random.default_rng(log_n + W)
    circ = pr.build_circuit(rng, log_n, W)
    beta = [int(v) for v in rand_field(rng, 2)]
    gamma = [int(v) for v in rand_field(rng, 2)]
    alpha = [int(v) for v in rand_field(rng, 2)]
    zs = pr.ref_zs(circ, beta, gamma)
    L = {}
    for name, vals in (("consts", circ["consts"]), ("sigmas", circ["sigmas"]), ("wires", circ["wires"]), ("zs", zs)):
        L[name], _ = lde_bitrev(oracle, vals, log_n, rb)
    want = pr.ref_quotient(circ, {k: [[int(x) for x in r] for r in v] for k, v in L.items()}, beta, gamma, alpha, rb)
    N = 1 << (log_n + rb)
    got = np.zeros((2, N), dtype=np.uint64)
    arr = lambda v: np.array(v, dtype=np.uint64)
    assert emu.emu_plonk_quotient(ptr(L["consts"]), ptr(L["sigmas"]), ptr(L["wires"]), ptr(L["zs"]), log_n, rb, W, ptr(arr(beta)),
                                  ptr(arr(gamma)), ptr(arr(alpha)), ptr(got)) == 0
    assert [[int(v) for v in r] for r in got] == want
    # the quotient is a genuine polynomial: un-bit-reverse, inverse NTT, unshift -> degree < 8n,
    # and (because the constraints hold on the trace domain) its top coefficients vanish too
    nat = np.zeros_like(got)
    sinv = pow(7, P - 2, P)
    assert emu.emu_bitrev_scale(ptr(got), ptr(nat), log_n + rb, 2, 1) == 0
    oracle.orc_ntt(ptr(nat), log_n + rb, 2, 1)
    co = [[int(v) * pow(sinv, j, P) % P for j, v in enumerate(r)] for r in nat]
    n = 1 << log_n
    for t in range(2):
        assert any(co[t]), "quotient should not be identically zero"
        assert not any(co[t][8 * n - 8:]), "degree bound 8n - 9 violated"
    # and the scaled variant of the same kernel
    nat2 = np.zeros_like(got)
    assert emu.emu_bitrev_scale(ptr(got), ptr(nat2), log_n + rb, 2, sinv) == 0
    for t in range(2):
        for i in (0, 1, 5, N - 1):
            assert int(nat2[t][i]) == int(got[t][fv.rev(i, log_n + rb)]) * pow(sinv, i, P) % P
