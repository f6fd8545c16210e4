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
    assert pkg.plonk_verify_host(consts, proof) == (True, None)
    ok, why = pkg.plonk_verify_host(consts, proof, cap, pl["queries"] + 1, pl["pow_bits"])
    assert not ok and "fewer queries" in why
    other = cap.copy()
    other[0] ^= np.uint64(1)
    ok, why = pkg.plonk_verify_host(consts, proof, other)
    assert not ok and "preprocessed commitment" in why
    words = np.frombuffer(proof, dtype="<u8")
    for t in range(0, len(words), 7):
        bad = words.copy()
        bad[t] ^= np.uint64(1 << (t % 63))
        assert not pkg.plonk_verify_host(consts, bad.tobytes(), cap)[0], f"word {t}"
    assert not pkg.plonk_verify_host(consts, proof[:-8], cap)[0] and not pkg.plonk_verify_host(consts, proof + bytes(8), cap)[0]
    fproof = bytes.fromhex(fr["proof"])
    assert pkg.fri_verify_host(consts, fproof, fr["queries"], fr["pow_bits"]) == (True, None)
    fwords = np.frombuffer(fproof, dtype="<u8")
    for t in range(0, len(fwords), 5):
        bad = fwords.copy()
        bad[t] ^= np.uint64(1 << (t % 63))
        assert not pkg.fri_verify_host(consts, bad.tobytes())[0], f"word {t}"
    # different constants: the transcript diverges
    big = poseidon_consts("big")
    assert not pkg.fri_verify_host(big, fproof)[0]
    with pytest.raises(pkg.GlpError):
        pkg.fri_verify_host((consts[0][:100], consts[1], consts[2]), fproof)


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
    assert out["fri"]["proof"] == golden["fri"]["proof"]
    # and the device-ctx verifier agrees with the host one
    rc, circ, diag = poseidon_consts("small")
    prover.set_poseidon_constants(rc, circ, diag)
    assert prover.plonk_verify(bytes.fromhex(golden["plonk"]["proof"]), np.array(golden["plonk"]["circuit_cap"], dtype=np.uint64), 8, 4)
    assert prover.fri_verify(bytes.fromhex(golden["fri"]["proof"]), 6, 5)
