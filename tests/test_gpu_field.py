"""Row a1 on the GPU: the device field arithmetic the kernels inline, through glp_field_op,
against the Python big-int golden vectors and the oracle."""
import json
import os

import numpy as np
import pytest

from conftest import P, rand_field

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def test_golden_field_vectors(prover):
    with open(os.path.join(G, "field.json")) as f:
        fld = json.load(f)
    a = np.array([int(r[0]) for r in fld["binary"]], dtype=np.uint64)
    b = np.array([int(r[1]) for r in fld["binary"]], dtype=np.uint64)
    for col, op in ((2, "add"), (3, "sub"), (4, "mul")):
        want = np.array([int(r[col]) for r in fld["binary"]], dtype=np.uint64)
        assert np.array_equal(prover.field_op(op, a, b), want), op
    x = np.array([int(r[0]) for r in fld["pow2"]], dtype=np.uint64)
    s = np.array([r[1] for r in fld["pow2"]], dtype=np.uint64)
    want = np.array([int(r[2]) for r in fld["pow2"]], dtype=np.uint64)
    assert np.array_equal(prover.field_op("mul_pow2", x, s), want)
    x = np.array([int(r[0]) for r in fld["inv"]], dtype=np.uint64)
    want = np.array([int(r[1]) for r in fld["inv"]], dtype=np.uint64)
    assert np.array_equal(prover.field_op("inv", x), want)


def test_random_million_vs_bigint(prover):
    rng = np.random.default_rng(99)
    n = 1 << 20
    a, b = rand_field(rng, n), rand_field(rng, n)
    # bias a slice towards the reduction edge cases: values within 2^33 of 0, 2^32 multiples, p
    edge = np.array([0, 1, P - 1, P - 2, 2**32 - 1, 2**32, 2**32 + 1, 2**64 - 2**33, P - 2**32, 2**63], dtype=np.uint64)
    a[:1000] = rng.choice(edge, 1000)
    b[:1000] = rng.choice(edge, 1000)
    ao, bo = a.astype(object), b.astype(object)
    assert np.array_equal(prover.field_op("add", a, b), ((ao + bo) % P).astype(np.uint64))
    assert np.array_equal(prover.field_op("sub", a, b), ((ao - bo) % P).astype(np.uint64))
    assert np.array_equal(prover.field_op("mul", a, b), ((ao * bo) % P).astype(np.uint64))
    sh = rng.integers(0, 192, n).astype(np.uint64)
    pw = np.array([pow(2, int(s), P) for s in range(192)], dtype=object)
    assert np.array_equal(prover.field_op("mul_pow2", a, sh), ((ao * pw[sh.astype(np.int64)]) % P).astype(np.uint64))


def _edge_words(rng, n):
    """64-bit words biased to the places where a carry, a borrow or the final >= p test flips"""
    specials = np.array([0, 1, 2, 2**32 - 2, 2**32 - 1, 2**32, 2**32 + 1, 2**33 - 1, 2**33, 2**63, 2**64 - 2**33, 2**64 - 2**33 + 1,
                         P - 2**32, P - 2, P - 1, P, P + 1, 2**64 - 2**32, 2**64 - 2**32 + 2, 2**64 - 2, 2**64 - 1], dtype=np.uint64)
    x = rng.integers(0, 2**64, n, dtype=np.uint64, endpoint=False)
    k = n // 2
    x[:k] = rng.choice(specials, k)
    # near-special: a special word plus or minus a small amount, and words with one half all-ones / all-zeros
    j = k // 2
    with np.errstate(over="ignore"):
        x[:j] = x[:j] + rng.integers(0, 4, j, dtype=np.uint64) - np.uint64(2)
    x[k:k + n // 8] &= np.uint64(0xFFFFFFFF00000000)
    x[k + n // 8:k + n // 4] |= np.uint64(0x00000000FFFFFFFF)
    rng.shuffle(x)
    return x


def test_reduction_primitives_on_arbitrary_words(prover):
    """the carry-chain (inline asm) forms of the 128-bit reduction, the lazy product, the accumulator fold
    and w*(2^32-1)+t, on ARBITRARY 64-bit operands incl. every carry/borrow boundary, vs Python big-ints"""
    rng = np.random.default_rng(2026)
    n = 1 << 18
    a, b = _edge_words(rng, n), _edge_words(rng, n)
    ao, bo = a.astype(object), b.astype(object)
    want = (((ao << 64) | bo) % P).astype(np.uint64)
    assert np.array_equal(prover.field_op("reduce128", a, b), want)
    assert np.array_equal(prover.field_op("reduce128_lazy", a, b), want)
    assert np.array_equal(prover.field_op("mul_any", a, b), ((ao * bo) % P).astype(np.uint64))
    assert np.array_equal(prover.field_op("fold_small", a, b), (((ao >> 7) + ((bo >> 7) << 32)) % P).astype(np.uint64))
    mad = ((ao + (bo & 0xFFFFFFFF) * (2**32 - 1)) % P).astype(np.uint64)
    assert np.array_equal(prover.field_op("mad_eps_lazy", a, b), mad)
    assert np.array_equal(prover.field_op("mad_eps", a, b), mad)
    # sub accepts any minuend and a subtrahend <= p; add needs canonical operands
    bc = (bo % P).astype(np.uint64)
    assert np.array_equal(prover.field_op("sub", (ao % P).astype(np.uint64), bc), ((ao - bo) % P).astype(np.uint64))
    got = prover.field_op("sub", a, bc).astype(object)
    assert np.all((got - (ao - bo)) % P == 0)
