"""Rows a6 / a7 and the prover driver on the GPU: permutation products and quotient inside a
complete proof of the build-defined circuit, accepted by the independent verifier
(tests/plonk_ref.py on top of tests/fri_verifier.py); invalid witnesses and tampered proofs
are rejected."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import fri_verifier as fv  # noqa: E402
import plonk_ref as pref  # noqa: E402
from conftest import P, poseidon_consts, ptr, rand_field  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture()
def setup(prover, oracle):
    rc, circ, diag = poseidon_consts("small")
    prover.set_poseidon_constants(rc, circ, diag)
    oracle.orc_poseidon_set_constants(ptr(rc), ptr(circ), ptr(diag))
    return prover, oracle


@pytest.mark.parametrize("log_n,W,nq,pw", [(6, 8, 8, 4), (8, 16, 10, 6), (10, 32, 12, 8), (12, 80, 28, 12)])
def test_prove_and_verify(setup, pkg, log_n, W, nq, pw):
    prover, oracle = setup
    rng = np.random.default_rng(log_n * 100 + W)
    circ = pref.build_circuit(rng, log_n, W)
    ck = pkg.PlonkCircuit(prover, circ["consts"], circ["sigmas"])
    proof = ck.prove(circ["wires"], num_queries=nq, pow_bits=pw)
    info = pref.verify_plonk(proof, oracle)
    assert info["log_n"] == log_n and info["W"] == W
    assert ck.prove(circ["wires"], num_queries=nq, pow_bits=pw) == proof      # deterministic
    # the product's own (native) verifier agrees, bound to this circuit and to the security parameters
    assert ck.verify(proof, min_queries=nq, min_pow_bits=pw), prover.last_reject
    assert prover.plonk_verify(proof, pkg.UNBOUND, nq, pw)                    # explicitly unbound
    assert prover.plonk_verify(proof, ck.cap()) == (nq >= 28 and pw >= 16)    # defaults demand 28 queries / 16 PoW bits
    with pytest.raises(pkg.GlpError):
        prover.plonk_verify(proof, None)
    assert not ck.verify(proof, min_queries=nq + 1, min_pow_bits=pw) and "fewer queries" in prover.last_reject
    other = circ["consts"].copy()
    other[1, 0] = (int(other[1, 0]) + 1) % P
    ck2 = pkg.PlonkCircuit(prover, other, circ["sigmas"])                     # a different circuit: different key
    assert not prover.plonk_verify(proof, ck2.cap(), nq, pw) and "preprocessed commitment" in prover.last_reject
    ck2.free()
    ck.free()


def _lde_bitrev(oracle, vals, log_n, rb):
    """values on the trace domain -> coset LDE values, bit-reversed order, by the ORACLE (inverse NTT, pad/scale, NTT)"""
    co = np.ascontiguousarray(vals).copy()
    k = co.shape[0]
    oracle.orc_ntt(ptr(co), log_n, k, 1)
    out = np.zeros((k, 1 << (log_n + rb)), dtype=np.uint64)
    oracle.orc_lde_coset(ptr(co), ptr(out), log_n, rb, k, 7)
    oracle.orc_bitrev_rows(ptr(out), log_n + rb, k)
    return out


@pytest.mark.parametrize("log_n,W", [(6, 8), (8, 16), (10, 24), (12, 8)])
def test_k6_partial_products_direct_parity(setup, pkg, log_n, W):
    """row a6 on the GPU, directly: Z and the partial products the HIP kernels compute for caller-chosen (beta, gamma)
    (glp_plonk_debug_stage) against the by-the-definition Python restatement, every value"""
    prover, _ = setup
    rng = np.random.default_rng(1000 * log_n + W)
    circ = pref.build_circuit(rng, log_n, W)
    ck = pkg.PlonkCircuit(prover, circ["consts"], circ["sigmas"])
    beta = [int(v) for v in rand_field(rng, 2)]
    gamma = [int(v) for v in rand_field(rng, 2)]
    got = ck.debug_stage(circ["wires"], "zs", beta + gamma)
    assert np.array_equal(got, pref.ref_zs(circ, beta, gamma))
    ck.free()


@pytest.mark.parametrize("log_n,W", [(6, 8), (8, 16), (10, 8)])
def test_k7_quotient_direct_parity(setup, pkg, log_n, W):
    """row a7 on the GPU, directly: the quotient evaluations on the 8n-point coset for caller-chosen challenges against
    plonk_ref.ref_quotient evaluated on ORACLE-made LDEs of the constants, sigmas, wires and the reference Z's"""
    prover, oracle = setup
    rb = 3
    rng = np.random.default_rng(77 * log_n + W)
    circ = pref.build_circuit(rng, log_n, W)
    ck = pkg.PlonkCircuit(prover, circ["consts"], circ["sigmas"])
    beta = [int(v) for v in rand_field(rng, 2)]
    gamma = [int(v) for v in rand_field(rng, 2)]
    alpha = [int(v) for v in rand_field(rng, 2)]
    got = ck.debug_stage(circ["wires"], "quotient", beta + gamma + alpha)
    zs = pref.ref_zs(circ, beta, gamma)
    L = {name: [[int(x) for x in r] for r in _lde_bitrev(oracle, vals, log_n, rb)]
         for name, vals in (("consts", circ["consts"]), ("sigmas", circ["sigmas"]), ("wires", circ["wires"]), ("zs", zs))}
    want = pref.ref_quotient(circ, L, beta, gamma, alpha, rb)
    assert [[int(v) for v in r] for r in got] == want
    ck.free()


def test_invalid_witness_cannot_be_proved(setup, pkg):
    """a gate violation or a broken copy constraint makes the quotient a non-polynomial: the
    prover refuses (degree check of the final FRI polynomial) or the verifier rejects"""
    prover, oracle = setup
    rng = np.random.default_rng(9)
    circ = pref.build_circuit(rng, 8, 16, copy_prob=0.8)
    ck = pkg.PlonkCircuit(prover, circ["consts"], circ["sigmas"])
    pref.verify_plonk(ck.prove(circ["wires"], 8, 4), oracle)
    for kind in ("gate", "copy"):
        bad = circ["wires"].copy()
        if kind == "gate":
            rows = np.nonzero(circ["consts"][0])[0]
            bad[3, rows[5]] = (int(bad[3, rows[5]]) + 1) % P          # output of an active gate
        else:
            bad[0, 200] = (int(bad[0, 200]) + 1) % P                  # an input cell (likely in a copy class)
            q, c0, c1 = (int(circ["consts"][k, 200]) for k in range(3))
            bad[3, 200] = (c0 * int(bad[0, 200]) * int(bad[1, 200]) + c1 * int(bad[2, 200])) % P if q else bad[3, 200]
        try:
            proof = ck.prove(bad, 8, 4)
        except pkg.GlpError:
            continue
        with pytest.raises(fv.VerifyError):
            pref.verify_plonk(proof, oracle)
        assert not ck.verify(proof, 8, 4)
    ck.free()


def test_tampered_plonk_proof_rejected(setup, pkg):
    prover, oracle = setup
    rng = np.random.default_rng(10)
    circ = pref.build_circuit(rng, 7, 8)
    ck = pkg.PlonkCircuit(prover, circ["consts"], circ["sigmas"])
    proof = ck.prove(circ["wires"], 6, 4)
    pref.verify_plonk(proof, oracle)
    words = np.frombuffer(proof, dtype="<u8").copy()
    n = len(words)
    for t in sorted(set([1, 2, 4, 6, 70, 140, 200, 270, 300, 330, 400, n // 2, n - 3] + list(range(280, n, max(1, n // 40))))):
        bad = words.copy()
        bad[t] ^= np.uint64(1)
        with pytest.raises(Exception):
            pref.verify_plonk(bad.tobytes(), oracle)
        assert not ck.verify(bad.tobytes(), 6, 4), f"native verifier accepted a proof with word {t} flipped"
    assert ck.verify(proof, 6, 4)
    assert not ck.verify(proof[:-8], 6, 4) and not ck.verify(proof + bytes(8), 6, 4)
    ck.free()


def test_stage_timing_tree(setup, pkg):
    prover, oracle = setup
    rng = np.random.default_rng(3)
    circ = pref.build_circuit(rng, 9, 16)
    ck = pkg.PlonkCircuit(prover, circ["consts"], circ["sigmas"])
    prover.set_profiling(True)
    proof = ck.prove(circ["wires"], 8, 4)
    stages = prover.last_stage_ms()
    prover.set_profiling(False)
    pref.verify_plonk(proof, oracle)
    names = [n for n, _ in stages]
    assert names[:5] == ["commit_wires(ifft+lde+merkle)", "perm_products(K6)", "commit_zs", "quotient(K7)+to_coeffs",
                         "commit_quotient(lde+merkle)"]
    assert [n for n in names if n.startswith("fri:")] == ["fri:evaluate_openings", "fri:combine", "fri:fold_layers+merkle",
                                                          "fri:proof_of_work", "fri:queries"]
    assert all(ms >= 0 for _, ms in stages)
    assert prover.last_stage_ms() == stages          # stable until the next prove
    ck.free()


def test_provers_driven_from_worker_threads(pkg):
    """the MapReduce map step drives several ctxs on one GPU from one host thread each: proofs made concurrently
    on worker threads are byte-identical to the one made on the main thread, and every one verifies"""
    from concurrent.futures import ThreadPoolExecutor
    rng = np.random.default_rng(42)
    circ = pref.build_circuit(rng, 9, 16)
    rc, cc, dg = poseidon_consts("small")
    provers, cks = [], []
    for _ in range(3):
        pr = pkg.Prover(0)
        pr.set_poseidon_constants(rc, cc, dg)
        provers.append(pr)
        cks.append(pkg.PlonkCircuit(pr, circ["consts"], circ["sigmas"]))
    want = cks[0].prove(circ["wires"], 8, 4)

    def job(k):
        provers[k].bind_thread()
        return [cks[k].prove(circ["wires"], 8, 4) for _ in range(4)]

    with ThreadPoolExecutor(3) as ex:
        got = [p for f in [ex.submit(job, k) for k in range(3)] for p in f.result()]
    assert all(p == want for p in got)
    assert all(cks[i % 3].verify(p, 8, 4) for i, p in enumerate(got))
    for ck, pr in zip(cks, provers):
        ck.free()
        pr.close()


def test_stream_switch_between_proofs(setup, pkg):
    """glp_set_stream drains the stream it leaves (the pool and the NTT scratch are reused in stream order): proofs made on the
    ctx's own stream, on an adopted torch stream and back again are byte-identical"""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")               # the HIP runtime the library is already using (same soname)
    side = ctypes.c_void_p()
    assert hip.hipStreamCreate(ctypes.byref(side)) == 0
    prover, oracle = setup
    rng = np.random.default_rng(55)
    circ = pref.build_circuit(rng, 10, 16)
    ck = pkg.PlonkCircuit(prover, circ["consts"], circ["sigmas"])
    want = ck.prove(circ["wires"], 8, 4)
    for _ in range(3):
        prover.set_stream(side)
        assert ck.prove(circ["wires"], 8, 4) == want
        prover.set_stream(None)
        assert ck.prove(circ["wires"], 8, 4) == want
    x = rand_field(rng, (4, 1 << 16))
    prover.set_stream(side)
    a = prover.fft(x)
    prover.set_stream(None)
    assert np.array_equal(prover.fft(x), a)
    pref.verify_plonk(want, oracle)
    ck.free()
    prover.sync()
    assert hip.hipStreamDestroy(side) == 0
