"""The committed golden proofs (tests/golden/proofs.json, made on an MI355X by tests/golden/gen_proofs.py).
CPU: the native HOST verifier (no GPU, no ctx) and the independent Python verifiers accept them and reject every
flipped word.  GPU: the prover regenerates them byte for byte from the seeded inputs — any change to the
transcript order, the arithmetic or the layout shows up as a diff against a committed fixture."""
import json
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import fri_verifier as fv  # noqa: E402
import plonk_ref as pref  # noqa: E402
from conftest import poseidon_consts, ptr  # noqa: E402

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def golden():
    with open(os.path.join(G, "proofs.json")) as f:
        return json.load(f)


def test_host_verifier_accepts_golden_and_rejects_tampering(pkg, golden):
    consts = poseidon_consts("small")
    pl, fr = golden["plonk"], golden["fri"]
    proof, cap = bytes.fromhex(pl["proof"]), np.array(pl["circuit_cap"], dtype=np.uint64)
    assert pkg.plonk_verify_host(consts, proof, cap, pl["queries"], pl["pow_bits"]) == (True, None)
    assert pkg.plonk_verify_host(consts, proof, pkg.UNBOUND, pl["queries"], pl["pow_bits"]) == (True, None)
    # the wrappers' defaults demand the standard security parameters and a circuit binding: the small golden proof
    # (8 queries, 4 PoW bits) is refused by default, and omitting the circuit cap is an error, not "accept anything"
    ok, why = pkg.plonk_verify_host(consts, proof, cap)
    assert not ok and "fewer queries" in why
    with pytest.raises(pkg.GlpError):
        pkg.plonk_verify_host(consts, proof, None)
    ok, why = pkg.plonk_verify_host(consts, proof, cap, pl["queries"] + 1, pl["pow_bits"])
    assert not ok and "fewer queries" in why
    other = cap.copy()
    other[0] ^= np.uint64(1)
    ok, why = pkg.plonk_verify_host(consts, proof, other, pl["queries"], pl["pow_bits"])
    assert not ok and "preprocessed commitment" in why
    words = np.frombuffer(proof, dtype="<u8")
    for t in range(0, len(words), 7):
        bad = words.copy()
        bad[t] ^= np.uint64(1 << (t % 63))
        assert not pkg.plonk_verify_host(consts, bad.tobytes(), cap, 1, 0)[0], f"word {t}"
    assert not pkg.plonk_verify_host(consts, proof[:-8], cap, 1, 0)[0] and not pkg.plonk_verify_host(consts, proof + bytes(8), cap, 1, 0)[0]
    fproof = bytes.fromhex(fr["proof"])
    assert pkg.fri_verify_host(consts, fproof, fr["queries"], fr["pow_bits"]) == (True, None)
    assert not pkg.fri_verify_host(consts, fproof)[0]                       # defaults: 28 queries / 16 bits required
    ok, why, st = pkg.fri_verify_host(consts, fproof, fr["queries"], fr["pow_bits"], fr["rate_bits"], want_statement=True)
    assert ok and (st["log_n"], st["rate_bits"], st["n_polys"], st["num_queries"]) == (fr["log_n"], fr["rate_bits"], fr["polys"], fr["queries"])
    assert len(st["caps"]) == len(fr["polys"]) and len(st["openings"]) == sum(fr["polys"])
    ok, why = pkg.fri_verify_host(consts, fproof, fr["queries"], fr["pow_bits"], fr["rate_bits"] + 1)
    assert not ok and "rate" in why
    fwords = np.frombuffer(fproof, dtype="<u8")
    for t in range(0, len(fwords), 5):
        bad = fwords.copy()
        bad[t] ^= np.uint64(1 << (t % 63))
        assert not pkg.fri_verify_host(consts, bad.tobytes(), 1, 0, 1)[0], f"word {t}"
    # different constants: the transcript diverges
    big = poseidon_consts("big")
    assert not pkg.fri_verify_host(big, fproof, 1, 0, 1)[0]
    with pytest.raises(pkg.GlpError):
        pkg.fri_verify_host((consts[0][:100], consts[1], consts[2]), fproof)


def forge_rate1_proof(oracle, y=(123456789, 987654321), nq=28):
    """a 'proof' at rate_bits = 0 for a FALSE opening.  f = one polynomial of 4 coefficients committed on the
    4-point coset 7<w_4>; the claimed value y != f(zeta).  At rate 1 every word of length 4 is a codeword of degree < 4, so the
    quotient (f(x) - y)/(x - zeta) on the 4 points always interpolates to a 'final polynomial' and every query passes."""
    P = fv.P
    h = fv.Hasher(oracle)
    ch = fv.Challenger(h)
    log_n, rb, cap0, a, fb, pw, shift = 2, 0, 0, 1, 2, 0, 7
    N = 4
    vals = [11, 22, 33, 44]                                    # f on the domain, bit-reversed order: any 4 values
    words = [fv.TAG, log_n, rb, cap0, a, fb, nq, pw, shift, 1, 1, 1, 1, 1]        # header, mults [1], (n_polys, mask) = (1, 1)
    for w in words:
        ch.observe(w)
    leaves = [h.hash_or_noop([v]) for v in vals]
    lvl1 = [h.two_to_one(leaves[0], leaves[1]), h.two_to_one(leaves[2], leaves[3])]
    root = h.two_to_one(lvl1[0], lvl1[1])
    words += root
    for v in root:
        ch.observe(v)
    zeta = ch.ext_challenge()
    words += list(y)
    for v in y:
        ch.observe(v)
    ch.ext_challenge()                                         # alpha: one polynomial, alpha^0 = 1
    w4 = fv.root(2)
    xs = [shift * pow(w4, fv.rev(i, 2), P) % P for i in range(N)]
    G = [fv.emul(fv.esub((vals[i], 0), y), fv.einv(fv.esub((xs[i], 0), zeta))) for i in range(N)]
    # interpolate G (4 ext values on 4 points): Lagrange, big-int
    coeffs = [(0, 0)] * N
    for i in range(N):
        num = [1]                                              # prod_{j != i} (X - x_j), base-field coefficients
        den = 1
        for j in range(N):
            if j != i:
                num = [(-xs[j] * (num[k] if k < len(num) else 0) + (num[k - 1] if k else 0)) % P for k in range(len(num) + 1)]
                den = den * (xs[i] - xs[j]) % P
        sc = pow(den, P - 2, P)
        for k in range(N):
            coeffs[k] = fv.eadd(coeffs[k], fv.escale(G[i], num[k] * sc % P))
    for cf in coeffs:
        words += list(cf)
        for v in cf:
            ch.observe(v)
    for _ in range(4):
        ch.challenge()                                         # PoW seed; pow_bits = 0 accepts any nonce
    words.append(0)
    ch.observe(0)
    idxs = [ch.challenge() & (N - 1) for _ in range(nq)]
    for idx in idxs:
        words += [idx, vals[idx]] + leaves[idx ^ 1] + lvl1[(idx >> 1) ^ 1]
    return np.array(words, dtype=np.uint64).tobytes()


def test_forged_rate1_proof_is_rejected(pkg, oracle):
    """the forged rate-1 proof is well formed in every other respect (the Python verifier accepts it once its rate guard is
    switched off) and is REJECTED by the native host verifier and by the Python verifier, whatever the caller's minimums"""
    rc, circ, diag = poseidon_consts("small")
    oracle.orc_poseidon_set_constants(ptr(rc), ptr(circ), ptr(diag))
    forged = forge_rate1_proof(oracle)
    info = fv.parse_and_verify(forged, oracle, min_rate_bits=0)           # guard off: "accepted", which is the attack
    assert info["rate_bits"] == 0 and info["openings"] == [(123456789, 987654321)]
    with pytest.raises(fv.VerifyError, match="rate"):
        fv.parse_and_verify(forged, oracle)
    for mq, mp, mr in ((28, 0, 0), (1, 0, 0), (28, 0, 1)):
        ok, why = pkg.fri_verify_host((rc, circ, diag), forged, mq, mp, mr)
        assert not ok and "rate" in why, (mq, mp, mr, why)


def test_host_verifier_gates_proof_statement_binding(pkg, oracle, golden):
    """the golden proof of a circuit with public inputs, advice wires and Poseidon rows: accepted only for ITS statement"""
    consts = poseidon_consts("small")
    g = golden["gates"]
    proof, cap, pub = bytes.fromhex(g["proof"]), np.array(g["circuit_cap"], dtype=np.uint64), g["public"]
    q, pw = g["queries"], g["pow_bits"]
    assert pkg.proof_public_inputs(proof) == pub
    assert pkg.plonk_verify_host(consts, proof, cap, q, pw, public=pub) == (True, None)
    assert pkg.plonk_verify_host(consts, proof, cap, q, pw, public=pkg.UNBOUND) == (True, None)
    ok, why = pkg.plonk_verify_host(consts, proof, cap, q, pw, public=[pub[0], (pub[1] + 1) % (2**64 - 2**32 + 1)])
    assert not ok and "public inputs" in why
    ok, why = pkg.plonk_verify_host(consts, proof, cap, q, pw)                   # "no public inputs" is another statement
    assert not ok and "public inputs" in why
    words = np.frombuffer(proof, dtype="<u8")
    for t in range(0, len(words), 11):
        bad = words.copy()
        bad[t] ^= np.uint64(1 << (t % 63))
        assert not pkg.plonk_verify_host(consts, bad.tobytes(), cap, 1, 0, public=pkg.UNBOUND)[0], f"word {t}"
    # the Poseidon rows are about THESE constants: with others the identity at zeta fails (and the transcript diverges)
    assert not pkg.plonk_verify_host(poseidon_consts("medium"), proof, cap, q, pw, public=pub)[0]
    # independent Python verifier
    rc, circ, diag = consts
    oracle.orc_poseidon_set_constants(ptr(rc), ptr(circ), ptr(diag))
    info = pref.verify_plonk(proof, oracle, pos_consts=consts, public=pub)
    assert (info["W"], info["R"], info["flags"]) == (g["W"], g["R"], pref.FLAG_POSEIDON)
    with pytest.raises(fv.VerifyError):
        pref.verify_plonk(proof, oracle, pos_consts=consts, public=[pub[1], pub[0]])


def test_host_verifier_sha_rows_proof(pkg, oracle, golden):
    """the golden proof of a circuit with SHA-256 rows of every kind (+ a Poseidon row, a public input): the native host verifier and the
    independent Python verifier accept it for its statement; every flipped word is refused"""
    consts = poseidon_consts("small")
    g = golden["sha"]
    proof, cap, pub = bytes.fromhex(g["proof"]), np.array(g["circuit_cap"], dtype=np.uint64), g["public"]
    q, pw = g["queries"], g["pow_bits"]
    assert pkg.plonk_verify_host(consts, proof, cap, q, pw, public=pub) == (True, None)
    ok, why = pkg.plonk_verify_host(consts, proof, cap, q, pw, public=[(pub[0] + 1) % (2**64 - 2**32 + 1)])
    assert not ok and "public inputs" in why
    words = np.frombuffer(proof, dtype="<u8")
    assert int(words[7]) == pref.FLAG_POSEIDON | pref.FLAG_SHA
    for t in range(0, len(words), 7):
        bad = words.copy()
        bad[t] ^= np.uint64(1 << (t % 63))
        assert not pkg.plonk_verify_host(consts, bad.tobytes(), cap, 1, 0, public=pkg.UNBOUND)[0], f"word {t}"
    rc, circ, diag = consts
    oracle.orc_poseidon_set_constants(ptr(rc), ptr(circ), ptr(diag))
    info = pref.verify_plonk(proof, oracle, pos_consts=consts, public=pub)
    assert (info["W"], info["R"], info["flags"]) == (g["W"], g["R"], 3)
    # an opened wire value changed consistently is impossible; an opened SELECTOR changed (claiming another row kind) breaks the identity:
    # covered by the flipped words above — the openings are part of the proof


def test_python_verifiers_accept_golden(oracle, golden):
    rc, circ, diag = poseidon_consts("small")
    oracle.orc_poseidon_set_constants(ptr(rc), ptr(circ), ptr(diag))
    info = pref.verify_plonk(bytes.fromhex(golden["plonk"]["proof"]), oracle)
    assert info["log_n"] == golden["plonk"]["log_n"] and info["W"] == golden["plonk"]["W"]
    cap_pre = [v for d in info["fri"]["caps"][0] for v in d]
    assert cap_pre == golden["plonk"]["circuit_cap"]
    finfo = fv.parse_and_verify(bytes.fromhex(golden["fri"]["proof"]), oracle)
    assert finfo["n_polys"] == golden["fri"]["polys"] and len(finfo["queries"]) == golden["fri"]["queries"]


@pytest.mark.gpu
def test_gpu_prover_reproduces_golden_bytes(pkg, prover, golden):
    sys.path.insert(0, G)
    import gen_proofs
    out = gen_proofs.make(pkg, prover)
    assert out["plonk"]["circuit_cap"] == golden["plonk"]["circuit_cap"]
    assert out["plonk"]["proof"] == golden["plonk"]["proof"]
    assert out["gates"]["circuit_cap"] == golden["gates"]["circuit_cap"] and out["gates"]["public"] == golden["gates"]["public"]
    assert out["gates"]["proof"] == golden["gates"]["proof"]
    assert out["sha"]["circuit_cap"] == golden["sha"]["circuit_cap"] and out["sha"]["proof"] == golden["sha"]["proof"]
    assert out["fri"]["proof"] == golden["fri"]["proof"]
    # and the device-ctx verifier agrees with the host one
    rc, circ, diag = poseidon_consts("small")
    prover.set_poseidon_constants(rc, circ, diag)
    assert prover.plonk_verify(bytes.fromhex(golden["plonk"]["proof"]), np.array(golden["plonk"]["circuit_cap"], dtype=np.uint64), 8, 4)
    assert prover.fri_verify(bytes.fromhex(golden["fri"]["proof"]), 6, 5)


def test_host_verifiers_survive_truncated_and_garbage_proofs(pkg, golden):
    """robustness of the parsers (they read lengths and counts from untrusted bytes): every golden proof cut at many lengths, padded with
    junk, filled with random words or with extreme header fields is REFUSED — no crash, no acceptance, no huge allocation"""
    consts = poseidon_consts("small")
    rng = np.random.default_rng(99)
    for name in ("plonk", "gates", "sha"):
        g = golden[name]
        proof, cap = bytes.fromhex(g["proof"]), np.array(g["circuit_cap"], dtype=np.uint64)
        verify = lambda b: pkg.plonk_verify_host(consts, b, cap, 1, 0, public=pkg.UNBOUND)[0]
        assert verify(proof)
        for cut in list(range(0, 200, 8)) + list(range(200, len(proof), 8 * 53)) + [len(proof) - 8]:
            assert not verify(proof[:cut]), f"{name}: accepted a proof cut to {cut} bytes"
        assert not verify(proof + bytes(8)) and not verify(proof + proof[:64])
        words = np.frombuffer(proof, dtype="<u8")
        for pos, val in ((1, 2**63), (1, 25), (2, 2**32), (3, 2**40), (5, 2**20), (6, 2**62), (7, 255)):       # log_n, W, R, cap_h, n_public, flags
            bad = words.copy()
            bad[pos] = np.uint64(val)
            assert not verify(bad.tobytes()), f"{name}: header word {pos} = {val}"
        for _ in range(20):
            junk = rng.integers(0, 2**63, len(words), dtype=np.uint64)
            junk[0] = words[0]
            junk[1:8] = words[1:8]                                                  # a plausible header in front of random words
            assert not verify(junk.tobytes())
    fri = bytes.fromhex(golden["fri"]["proof"])
    fverify = lambda b: pkg.fri_verify_host(consts, b, 1, 0, 1)[0]
    assert fverify(fri)
    for cut in list(range(0, 160, 8)) + list(range(160, len(fri), 8 * 41)):
        assert not fverify(fri[:cut])
    fw = np.frombuffer(fri, dtype="<u8")
    for pos in range(1, 24):
        for val in (2**63, 2**32 + 1, 0):
            bad = fw.copy()
            if int(bad[pos]) == val:
                continue
            bad[pos] = np.uint64(val)
            assert not fverify(bad.tobytes()), f"fri header word {pos} = {val}"
    # digests / public-input readers on short input
    for cut in (0, 8, 56, 64, 72):
        with pytest.raises(Exception):
            pkg.proof_digest_host(consts, proof[:cut])
