"""The extended gate set on the GPU (VERDICT r1 "next" item 3): public inputs bound into the transcript and the verifier's
statement, the constant term of the arithmetic gate, advice (unrouted) wires and Poseidon rows — proofs accepted by the native
verifier and by the independent Python verifier (tests/plonk_ref.py), statements that differ in one public word rejected, and
Poseidon rows reproducing glp_poseidon_permute / the oracle's permutation as wire values."""
import importlib
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import fri_verifier as fv  # noqa: E402
import plonk_ref as pref  # noqa: E402
from conftest import P, poseidon_consts, ptr, rand_field  # noqa: E402
import __graft_entry__ as graft  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture()
def setup(prover, oracle):
    rc, circ, diag = poseidon_consts("small")
    prover.set_poseidon_constants(rc, circ, diag)
    oracle.orc_poseidon_set_constants(ptr(rc), ptr(circ), ptr(diag))
    return prover, oracle


def test_public_inputs_are_a_skip_statement(setup, pkg):
    """a 2^12-row circuit whose public inputs are the packed public values of a light-client skip (trusted block, trusted header
    hash, target block | target header hash, data commitment): proves, verifies, and is REJECTED for any other statement"""
    prover, oracle = setup
    graft.load_package()
    bs = importlib.import_module(graft.PKG_NAME + ".blobstream")
    rng = np.random.default_rng(2026)
    h32 = lambda: rng.integers(0, 256, 32, dtype=np.uint8).tobytes()
    packed = bs.pack_skip_inputs(1_000_000, h32(), 1_001_024) + bs.pack_outputs(h32(), h32())
    words = bs.public_words(packed)
    assert len(words) == 28 and all(w < 2**32 for w in words)
    circ = pref.build_circuit(rng, 12, 16, n_public=len(words), public_values=words)
    assert circ["public"] == words
    ck = pkg.PlonkCircuit(prover, circ["consts"], circ["sigmas"], n_public=len(words))
    proof = ck.prove(circ["wires"], 28, 16, public=words)
    assert pkg.proof_public_inputs(proof) == words
    assert ck.verify(proof, public=words), prover.last_reject
    info = pref.verify_plonk(proof, oracle, public=words)
    assert info["public"] == words
    # the verifier's statement differs in one word
    for k in (0, 9, 27):
        other = list(words)
        other[k] ^= 1
        assert not ck.verify(proof, public=other) and "public inputs" in prover.last_reject
        with pytest.raises(fv.VerifyError):
            pref.verify_plonk(proof, oracle, public=other)
    assert not ck.verify(proof, public=words[:-1]) and not ck.verify(proof, public=words + [0])
    assert not prover.plonk_verify(proof, ck.cap(), 28, 16, public=None)          # "no public inputs" is a different statement
    assert prover.plonk_verify(proof, ck.cap(), 28, 16, public=pkg.UNBOUND)       # explicit opt-out
    with pytest.raises(pkg.GlpError):
        ck.verify(proof)                                                         # forgetting the statement is an error
    # the proof's own copy of a public word flipped: the transcript diverges
    w = np.frombuffer(proof, dtype="<u8").copy()
    w[8 + 5] ^= np.uint64(1)
    assert not ck.verify(w.tobytes(), public=pkg.proof_public_inputs(w.tobytes()))
    assert not ck.verify(w.tobytes(), public=words)
    # a witness that does not match the claimed statement cannot be proved (or does not verify)
    lie = list(words)
    lie[3] = (lie[3] + 1) % P
    try:
        bad = ck.prove(circ["wires"], 28, 16, public=lie)
    except pkg.GlpError:
        bad = None
    if bad is not None:
        assert not ck.verify(bad, public=lie)
    with pytest.raises(pkg.GlpError):
        ck.prove(circ["wires"], 28, 16, public=words[:-1])
    ck.free()


@pytest.mark.parametrize("log_n,W,R,n_public,n_pos", [(8, 136, 80, 4, 40), (10, 136, 24, 0, 200), (6, 160, 136, 2, 9)])
def test_poseidon_gate_circuit(setup, pkg, log_n, W, R, n_public, n_pos):
    """Poseidon rows: the wire values of every row reproduce the permutation (glp_poseidon_permute and the oracle), the product's
    GPU witness filler rebuilds them from the 12 inputs, and the circuit proves and verifies with both verifiers"""
    prover, oracle = setup
    consts = poseidon_consts("small")
    rng = np.random.default_rng(log_n * 1000 + n_pos)
    n = 1 << log_n
    rows = sorted(int(v) for v in rng.choice(np.arange(n_public, n), size=n_pos, replace=False))
    circ = pref.build_circuit(rng, log_n, W, n_routed=R, n_public=n_public, poseidon_rows=rows, consts=consts)
    wires = circ["wires"]
    # (1) wire values = the permutation, three ways
    states = np.ascontiguousarray(wires[:12, rows].T)
    swapped = wires[pref.POS_SWAP, rows] == 1                   # rows whose swap bit is set permute (in[4..8), in[0..4), in[8..12))
    states[swapped] = states[swapped][:, [4, 5, 6, 7, 0, 1, 2, 3, 8, 9, 10, 11]]
    assert swapped.any() and not swapped.all()
    got = prover.poseidon_permute(states)
    assert np.array_equal(got, wires[12:24, rows].T)
    for k in (0, len(rows) // 2, len(rows) - 1):
        st = states[k].copy()
        oracle.orc_poseidon_permute(ptr(st))
        assert np.array_equal(st, wires[12:24, rows[k]])
    # (2) the GPU witness filler: blank wires 12..129 of the Poseidon rows, refill on the device
    blank = wires.copy()
    blank[12:pref.POS_SWAP, rows] = 0                           # everything the filler derives: not the inputs, not the swap bit (wire 24)
    blank[pref.POS_SWAP + 1:pref.POS_WIRES, rows] = 0
    dw = prover.to_device(blank)
    prover.poseidon_gate_fill_rows(dw, log_n, W, rows)
    assert np.array_equal(dw.download(wires.shape), wires)
    # (3) prove on the refilled device wires, verify natively and independently
    ck = pkg.PlonkCircuit(prover, circ["consts"], circ["sigmas"], n_wires=W, n_public=n_public, poseidon=True)
    proof = ck.prove_(dw, 10, 6, public=circ["public"])
    dw.free()
    assert ck.verify(proof, 10, 6, public=circ["public"] if n_public else None), prover.last_reject
    info = pref.verify_plonk(proof, oracle, pos_consts=consts, public=circ["public"])
    assert info["flags"] == pref.FLAG_POSEIDON and info["R"] == R
    # (4) one corrupted wire of a Poseidon row (an S-box input, not reachable by a copy constraint when it is advice): no proof
    bad = wires.copy()
    bad[W - 8 if W - 8 < pref.POS_WIRES else 100, rows[1]] ^= np.uint64(1)
    try:
        p2 = ck.prove(bad, 10, 6, public=circ["public"])
    except pkg.GlpError:
        p2 = None
    if p2 is not None:
        assert not ck.verify(p2, 10, 6, public=circ["public"] if n_public else None)
        with pytest.raises(fv.VerifyError):
            pref.verify_plonk(p2, oracle, pos_consts=consts)
    # (5) direct K6 / K7 parity on this gate set too
    beta = [int(v) for v in rand_field(rng, 2)]
    gamma = [int(v) for v in rand_field(rng, 2)]
    assert np.array_equal(ck.debug_stage(wires, "zs", beta + gamma, public=circ["public"]), pref.ref_zs(circ, beta, gamma))
    ck.free()


def test_k7_direct_parity_extended_gates(setup, pkg):
    """row a7 directly on the GPU for a circuit with every gate kind (small enough for the big-int restatement)"""
    prover, oracle = setup
    consts = poseidon_consts("small")
    rng = np.random.default_rng(91)
    log_n, W, R, rb = 5, 136, 32, 3
    circ = pref.build_circuit(rng, log_n, W, n_routed=R, n_public=3, poseidon_rows=(4, 5, 20), consts=consts)
    ck = pkg.PlonkCircuit(prover, circ["consts"], circ["sigmas"], n_wires=W, n_public=3, poseidon=True)
    beta = [int(v) for v in rand_field(rng, 2)]
    gamma = [int(v) for v in rand_field(rng, 2)]
    alpha = [int(v) for v in rand_field(rng, 2)]
    got = ck.debug_stage(circ["wires"], "quotient", beta + gamma + alpha, public=circ["public"])
    zs = pref.ref_zs(circ, beta, gamma)

    def lde(vals):
        co = np.ascontiguousarray(vals).copy()
        k = co.shape[0]
        oracle.orc_ntt(ptr(co), log_n, k, 1)
        out = np.zeros((k, 1 << (log_n + rb)), dtype=np.uint64)
        oracle.orc_lde_coset(ptr(co), ptr(out), log_n, rb, k, 7)
        oracle.orc_bitrev_rows(ptr(out), log_n + rb, k)
        return [[int(x) for x in r] for r in out]

    L = {name: lde(vals) for name, vals in (("consts", circ["consts"]), ("sigmas", circ["sigmas"]), ("wires", circ["wires"]), ("zs", zs))}
    assert [[int(v) for v in r] for r in got] == pref.ref_quotient(circ, L, beta, gamma, alpha, rb)
    ck.free()


@pytest.mark.parametrize("log_n,W,R,n_public,n_sha,n_pos", [(8, 144, 80, 2, 60, 10), (6, 160, 16, 0, 20, 0), (9, 144, 24, 0, 300, 0)])
def test_sha_gate_circuit(setup, pkg, log_n, W, R, n_public, n_sha, n_pos):
    """SHA-256 rows (E / A / W / ADD kinds at random): the GPU witness filler rebuilds every bit wire from the routed words, the circuit
    proves and verifies with the native and the independent verifier (alone and mixed with Poseidon rows), a flipped bit wire or a
    wrong output word yields no accepted proof, and K6 matches the restatement"""
    prover, oracle = setup
    consts = poseidon_consts("small")
    rng = np.random.default_rng(log_n * 77 + n_sha)
    n = 1 << log_n
    special = [int(v) for v in rng.choice(np.arange(n_public, n), size=n_sha + n_pos, replace=False)]
    sha_rows, pos_rows = sorted(special[:n_sha]), sorted(special[n_sha:])
    circ = pref.build_circuit(rng, log_n, W, n_routed=R, n_public=n_public, poseidon_rows=pos_rows, consts=consts, sha_rows=sha_rows)
    wires = circ["wires"]
    assert circ["consts"].shape[0] == pref.NCONST_SHA and circ["flags"] & pref.FLAG_SHA
    kinds = [int(np.argmax(circ["consts"][6:10, r])) for r in sha_rows]
    assert sorted(set(kinds)) == [0, 1, 2, 3]
    blank = wires.copy()
    blank[12:pref.SHA_WIRES, sha_rows] = 0
    dw = prover.to_device(blank)
    prover.sha_gate_fill_rows(dw, log_n, W, sha_rows, kinds)
    assert np.array_equal(dw.download(wires.shape), wires)
    ck = pkg.PlonkCircuit(prover, circ["consts"], circ["sigmas"], n_wires=W, n_public=n_public, poseidon=bool(pos_rows), sha=True)
    proof = ck.prove_(dw, 10, 6, public=circ["public"])
    dw.free()
    assert ck.verify(proof, 10, 6, public=circ["public"] if n_public else None), prover.last_reject
    info = pref.verify_plonk(proof, oracle, pos_consts=consts if pos_rows else None, public=circ["public"])
    assert info["flags"] == circ["flags"]
    for wire, row in ((12 + 5, sha_rows[0]), (140, sha_rows[1]), (108 + 31, sha_rows[2]), (4, sha_rows[3]), (76, sha_rows[4])):
        bad = wires.copy()
        bad[wire, row] ^= np.uint64(1)
        try:
            p2 = ck.prove(bad, 10, 6, public=circ["public"])
        except pkg.GlpError:
            p2 = None
        if p2 is not None:
            assert not ck.verify(p2, 10, 6, public=circ["public"] if n_public else None), (wire, row)
    beta = [int(v) for v in rand_field(rng, 2)]
    gamma = [int(v) for v in rand_field(rng, 2)]
    assert np.array_equal(ck.debug_stage(wires, "zs", beta + gamma, public=circ["public"]), pref.ref_zs(circ, beta, gamma))
    ck.free()


def test_k7_direct_parity_sha_rows(setup, pkg):
    """row a7 directly on the GPU for a circuit with SHA rows of every kind next to Poseidon rows, arithmetic gates and public inputs:
    the quotient values of K7 + K7s equal the big-int restatement point for point"""
    prover, oracle = setup
    consts = poseidon_consts("small")
    rng = np.random.default_rng(92)
    log_n, W, R, rb = 5, 144, 32, 3
    for attempt in range(20):
        circ = pref.build_circuit(rng, log_n, W, n_routed=R, n_public=3, poseidon_rows=(4, 20), consts=consts, sha_rows=(5, 6, 7, 8, 9, 21, 22, 23, 30))
        if len({int(np.argmax(circ["consts"][6:10, r])) for r in (5, 6, 7, 8, 9, 21, 22, 23, 30)}) == 4:
            break
    ck = pkg.PlonkCircuit(prover, circ["consts"], circ["sigmas"], n_wires=W, n_public=3, poseidon=True, sha=True)
    beta = [int(v) for v in rand_field(rng, 2)]
    gamma = [int(v) for v in rand_field(rng, 2)]
    alpha = [int(v) for v in rand_field(rng, 2)]
    got = ck.debug_stage(circ["wires"], "quotient", beta + gamma + alpha, public=circ["public"])
    zs = pref.ref_zs(circ, beta, gamma)

    def lde(vals):
        co = np.ascontiguousarray(vals).copy()
        k = co.shape[0]
        oracle.orc_ntt(ptr(co), log_n, k, 1)
        out = np.zeros((k, 1 << (log_n + rb)), dtype=np.uint64)
        oracle.orc_lde_coset(ptr(co), ptr(out), log_n, rb, k, 7)
        oracle.orc_bitrev_rows(ptr(out), log_n + rb, k)
        return [[int(x) for x in r] for r in out]

    L = {name: lde(vals) for name, vals in (("consts", circ["consts"]), ("sigmas", circ["sigmas"]), ("wires", circ["wires"]), ("zs", zs))}
    assert [[int(v) for v in r] for r in got] == pref.ref_quotient(circ, L, beta, gamma, alpha, rb)
    ck.free()


@pytest.mark.parametrize("log_n,W,R,n_pos,n_sha", [(7, 24, 16, 0, 0), (8, 144, 80, 12, 20)])
def test_ext_gate_circuit(setup, pkg, log_n, W, R, n_pos, n_sha):
    """extension-arithmetic rows (every 8-wire chunk a multiply-add in F_p[X]/(X^2 - 7), sharing the chunk's two gate slots), alone and next to
    Poseidon and SHA rows: proves, verifies with both verifiers, a wrong product component yields no accepted proof, K6 and K7 match the restatements"""
    prover, oracle = setup
    consts = poseidon_consts("small")
    rng = np.random.default_rng(log_n * 31 + W)
    n = 1 << log_n
    special = [int(v) for v in rng.choice(np.arange(2, n), size=30 + n_pos + n_sha, replace=False)]
    ext_rows, pos_rows, sha_rows = sorted(special[:30]), sorted(special[30:30 + n_pos]), sorted(special[30 + n_pos:])
    circ = pref.build_circuit(rng, log_n, W, n_routed=R, n_public=2, poseidon_rows=pos_rows, consts=consts, sha_rows=sha_rows, ext_rows=ext_rows)
    assert circ["flags"] & pref.FLAG_EXT and circ["consts"].shape[0] == pref.n_const(circ["flags"])
    ck = pkg.PlonkCircuit(prover, circ["consts"], circ["sigmas"], n_wires=W, n_public=2, poseidon=bool(pos_rows), sha=bool(sha_rows), ext=True)
    proof = ck.prove(circ["wires"], 10, 6, public=circ["public"])
    assert ck.verify(proof, 10, 6, public=circ["public"]), prover.last_reject
    info = pref.verify_plonk(proof, oracle, pos_consts=consts if pos_rows else None, public=circ["public"])
    assert info["flags"] == circ["flags"]
    for wire in (6, 7, R - 1):
        bad = circ["wires"].copy()
        bad[wire, ext_rows[3]] ^= np.uint64(1)
        try:
            p2 = ck.prove(bad, 10, 6, public=circ["public"])
        except pkg.GlpError:
            p2 = None
        assert p2 is None or not ck.verify(p2, 10, 6, public=circ["public"]), wire
    beta = [int(v) for v in rand_field(rng, 2)]
    gamma = [int(v) for v in rand_field(rng, 2)]
    assert np.array_equal(ck.debug_stage(circ["wires"], "zs", beta + gamma, public=circ["public"]), pref.ref_zs(circ, beta, gamma))
    ck.free()


def test_k7_direct_parity_ext_rows(setup, pkg):
    """row a7 directly on the GPU for a circuit with extension rows next to every other kind: quotient values equal the big-int restatement"""
    prover, oracle = setup
    consts = poseidon_consts("small")
    rng = np.random.default_rng(93)
    log_n, W, R, rb = 5, 144, 32, 3
    circ = pref.build_circuit(rng, log_n, W, n_routed=R, n_public=3, poseidon_rows=(4, 20), consts=consts, sha_rows=(5, 6, 7, 8), ext_rows=(9, 10, 11, 25, 26))
    ck = pkg.PlonkCircuit(prover, circ["consts"], circ["sigmas"], n_wires=W, n_public=3, poseidon=True, sha=True, ext=True)
    beta = [int(v) for v in rand_field(rng, 2)]
    gamma = [int(v) for v in rand_field(rng, 2)]
    alpha = [int(v) for v in rand_field(rng, 2)]
    got = ck.debug_stage(circ["wires"], "quotient", beta + gamma + alpha, public=circ["public"])
    zs = pref.ref_zs(circ, beta, gamma)

    def lde(vals):
        co = np.ascontiguousarray(vals).copy()
        k = co.shape[0]
        oracle.orc_ntt(ptr(co), log_n, k, 1)
        out = np.zeros((k, 1 << (log_n + rb)), dtype=np.uint64)
        oracle.orc_lde_coset(ptr(co), ptr(out), log_n, rb, k, 7)
        oracle.orc_bitrev_rows(ptr(out), log_n + rb, k)
        return [[int(x) for x in r] for r in out]

    L = {name: lde(vals) for name, vals in (("consts", circ["consts"]), ("sigmas", circ["sigmas"]), ("wires", circ["wires"]), ("zs", zs))}
    assert [[int(v) for v in r] for r in got] == pref.ref_quotient(circ, L, beta, gamma, alpha, rb)
    ck.free()
