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
