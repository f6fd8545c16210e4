"""The FRI opening proof on the GPU (rows a5, a8, a12): kernels vs Python big-int restatements,
the full prover accepted by the independent verifier (tests/fri_verifier.py), determinism, and
rejection of every kind of tampering.  The protocol is build-defined."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import fri_verifier as fv  # noqa: E402
from conftest import P, poseidon_consts, ptr, rand_field  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture()
def setup(prover, oracle):
    rc, circ, diag = poseidon_consts("small")
    prover.set_poseidon_constants(rc, circ, diag)
    oracle.orc_poseidon_set_constants(ptr(rc), ptr(circ), ptr(diag))
    return prover, oracle


def test_challenger_vs_python(setup):
    prover, oracle = setup
    rng = np.random.default_rng(1)
    ch, py = prover.challenger(), fv.Challenger(fv.Hasher(oracle))
    for step in range(40):
        vals = rand_field(rng, int(rng.integers(0, 19)))
        ch.observe(vals)
        for v in vals:
            py.observe(int(v))
        k = int(rng.integers(0, 11))
        assert [int(v) for v in ch.challenges(k)] == [py.challenge() for _ in range(k)]


def test_eval_at_ext_and_pow(setup):
    prover, oracle = setup
    rng = np.random.default_rng(2)
    co = rand_field(rng, (5, 1 << 13))
    z = rand_field(rng, 2)
    d = prover.to_device(co)
    got = prover.eval_at_ext(d, 13, 5, z)
    zz = (int(z[0]), int(z[1]))
    for p in range(5):
        acc = (0, 0)
        for cf in reversed([int(v) for v in co[p]]):
            acc = fv.eadd(fv.emul(acc, zz), (cf, 0))
        assert (int(got[p, 0]), int(got[p, 1])) == acc
    d.free()
    h = fv.Hasher(oracle)
    seed = [int(v) for v in rand_field(rng, 4)]
    nonce = prover.pow_grind(seed, 12)
    assert (h.permute(seed + [nonce] + [0] * 7)[0] >> 52) == 0
    assert all((h.permute(seed + [m] + [0] * 7)[0] >> 52) != 0 for m in range(0, min(nonce, 300)))


CONFIGS = [
    # log_n, n_polys per batch, rate_bits, cap_h, arity_bits, final_bits, queries, pow_bits
    (10, [5], 3, 4, 4, 5, 10, 8),
    (12, [7, 2], 3, 4, 4, 5, 12, 10),
    (9, [1], 1, 2, 2, 3, 8, 4),
    (6, [3], 2, 0, 1, 2, 6, 0),
    (5, [2], 3, 8, 4, 5, 5, 6),          # no fold layer: final polynomial is the whole thing
    (14, [20, 4, 8], 3, 4, 4, 5, 28, 16),
]


def commit(pkg, prover, rng, log_n, n_polys, rb, cap_h):
    vals = rand_field(rng, (n_polys, 1 << log_n))
    return pkg.PolynomialBatch.from_values(prover, vals, rb, cap_h), vals


@pytest.mark.parametrize("log_n,polys,rb,cap_h,a,fb,nq,pw", CONFIGS)
def test_prove_then_verify(setup, pkg, oracle, log_n, polys, rb, cap_h, a, fb, nq, pw):
    prover, _ = setup
    rng = np.random.default_rng(log_n * 13 + len(polys))
    batches, values = zip(*[commit(pkg, prover, rng, log_n, k, rb, cap_h) for k in polys])
    proof = prover.fri_prove(list(batches), rb, cap_h, arity_bits=a, final_poly_bits=fb, num_queries=nq, pow_bits=pw)
    info = fv.parse_and_verify(proof, oracle)
    assert info["n_polys"] == polys and len(info["queries"]) == nq
    assert prover.fri_verify(proof, min_queries=nq, min_pow_bits=pw, min_rate_bits=rb), prover.last_reject     # native verifier
    # ... and says WHAT it accepted: the caller binds the proof to its statement by comparing this with what it expects
    st = prover.last_statement
    assert (st["log_n"], st["rate_bits"], st["n_polys"], st["num_queries"], st["pow_bits"]) == (log_n, rb, polys, nq, pw)
    assert st["caps"] == [[int(v) for v in b.cap.reshape(-1)] for b in batches]
    assert st["zeta"] == info["zeta"] and st["openings"] == info["openings"] and st["point_mult"] == [1]
    # the defaults demand the prover's standard parameters (28 queries, 16 PoW bits, rate 1/8): weaker proofs are refused
    assert prover.fri_verify(proof) == (nq >= 28 and pw >= 16 and rb >= 3)
    assert not prover.fri_verify(proof, nq, pw, min_rate_bits=rb + 1) and "rate" in prover.last_reject
    # the claimed openings are the true evaluations: f(zeta) from the coefficients (oracle ifft)
    zeta = info["zeta"]
    k = 0
    for vals in values:
        co = vals.copy()
        oracle.orc_ntt(ptr(co), log_n, co.shape[0], 1)
        for row in co[:2]:
            acc = (0, 0)
            for cf in reversed([int(v) for v in row]):
                acc = fv.eadd(fv.emul(acc, zeta), (cf, 0))
            assert info["openings"][k] == acc
            k += 1
        k += co.shape[0] - min(2, co.shape[0])
    # determinism: same inputs, same bytes
    assert prover.fri_prove(list(batches), rb, cap_h, arity_bits=a, final_poly_bits=fb, num_queries=nq, pow_bits=pw) == proof
    for b in batches:
        b.free()


def test_two_opening_points(setup, pkg, oracle):
    """zeta and w_n*zeta (the "next row" point), second point only for the second batch"""
    prover, _ = setup
    rng = np.random.default_rng(21)
    log_n, rb, cap_h = 11, 3, 4
    b0, v0 = commit(pkg, prover, rng, log_n, 6, rb, cap_h)
    b1, v1 = commit(pkg, prover, rng, log_n, 3, rb, cap_h)
    g = pow(7, (P - 1) >> log_n, P)
    proof = prover.fri_prove([b0, b1], rb, cap_h, num_queries=10, pow_bits=6, point_mults=(1, g), open_masks=[1, 3])
    info = fv.parse_and_verify(proof, oracle)
    assert prover.fri_verify(proof, 10, 6), prover.last_reject
    assert prover.last_statement["point_mult"] == [1, g] and prover.last_statement["open_mask"] == [1, 3]
    assert sorted(info["openings_at"].keys()) == [(0, 0), (0, 1), (1, 1)]
    co = v1.copy()
    oracle.orc_ntt(ptr(co), log_n, 3, 1)
    z1 = info["points"][1]
    assert z1 == fv.escale(info["zeta"], g)
    acc = (0, 0)
    for cf in reversed([int(v) for v in co[2]]):
        acc = fv.eadd(fv.emul(acc, z1), (cf, 0))
    assert info["openings_at"][(1, 1)][2] == acc
    b0.free()
    b1.free()


def test_tampered_proofs_are_rejected(setup, pkg, oracle):
    prover, _ = setup
    rng = np.random.default_rng(77)
    log_n, rb, cap_h = 10, 3, 4
    pb, _ = commit(pkg, prover, rng, log_n, 4, rb, cap_h)
    proof = prover.fri_prove([pb], rb, cap_h, num_queries=8, pow_bits=8)
    fv.parse_and_verify(proof, oracle)
    words = np.frombuffer(proof, dtype="<u8").copy()
    n = len(words)
    rejected = 0
    # flip one bit in a spread of words covering header, caps, openings, layer caps, final
    # polynomial, nonce, query leaves and paths
    targets = sorted(set([1, 9, 11, 12, 20, 80, 81, 90, 100, 150, 200, 230, 231, 240, n // 2, n // 2 + 1, n - 300, n - 40, n - 2,
                          n - 1] + list(range(240, n, max(1, n // 60)))))
    for t in targets:
        bad = words.copy()
        bad[t] ^= np.uint64(1)
        try:
            fv.parse_and_verify(bad.tobytes(), oracle)
        except fv.VerifyError:
            rejected += 1
        except Exception:
            rejected += 1      # malformed sizes after a header flip count as rejection too
        assert not prover.fri_verify(bad.tobytes(), 1, 0, 1), f"native verifier accepted a proof with word {t} flipped"
    assert rejected == len(targets)
    assert prover.fri_verify(proof, 8, 8) and not prover.fri_verify(proof[:-8], 8, 8) and not prover.fri_verify(proof, 9, 8)
    with pytest.raises(fv.VerifyError):
        fv.parse_and_verify(proof[:-8], oracle)
    pb.free()


def test_inconsistent_batch_is_refused_or_unverifiable(setup, pkg, oracle):
    """a batch whose LDE is not the LDE of its coefficients (degree bound violated) cannot
    produce an accepted proof"""
    prover, _ = setup
    rng = np.random.default_rng(5)
    log_n, rb, cap_h = 8, 3, 2
    pb, _ = commit(pkg, prover, rng, log_n, 2, rb, cap_h)
    junk = rand_field(rng, (2, 1 << (log_n + rb)))
    pb.lde.upload(junk)     # digests no longer match either; prover must fail or the verifier reject
    try:
        proof = prover.fri_prove([pb], rb, cap_h, num_queries=8, pow_bits=4)
    except pkg.GlpError:
        pb.free()
        return
    with pytest.raises(fv.VerifyError):
        fv.parse_and_verify(proof, oracle)
    pb.free()
