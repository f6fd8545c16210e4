#!/usr/bin/env python3
"""tests/golden/gen_proofs.py — writes tests/golden/proofs.json ON AN MI355X: one proof of the build-defined
circuit and one stand-alone FRI opening proof, made by the GPU prover from seeded inputs, with the package's
default Poseidon constants.  The fixtures pin the proof bytes (transcript order, arithmetic, layout): the GPU
tests regenerate them and demand identical bytes; the CPU tests verify them with the native host verifier and the
independent Python verifiers.  Usage (GPU box, repo root):  python tests/golden/gen_proofs.py"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
import __graft_entry__ as graft  # noqa: E402
import plonk_ref as pref  # noqa: E402
from conftest import poseidon_consts, rand_field  # noqa: E402

PLONK = {"seed": 20261004, "log_n": 6, "W": 8, "queries": 8, "pow_bits": 4}
GATES = {"seed": 424242, "log_n": 5, "W": 136, "R": 24, "n_public": 2, "pos_rows": [3, 17, 18], "queries": 4, "pow_bits": 3}
SHA = {"seed": 515151, "log_n": 5, "W": 144, "R": 24, "n_public": 1, "pos_rows": [2], "sha_rows": [3, 4, 5, 6, 9, 10, 11, 12, 20, 21, 22, 30],
       "queries": 4, "pow_bits": 3}
FRI = {"seed": 77001, "log_n": 8, "polys": [3, 2], "rate_bits": 3, "cap_height": 2, "arity_bits": 2, "final_poly_bits": 3, "queries": 6,
       "pow_bits": 5}


def make(pkg, prover):
    rc, circ, diag = poseidon_consts("small")
    prover.set_poseidon_constants(rc, circ, diag)
    c = pref.build_circuit(np.random.default_rng(PLONK["seed"]), PLONK["log_n"], PLONK["W"])
    ck = pkg.PlonkCircuit(prover, c["consts"], c["sigmas"])
    plonk_proof = ck.prove(c["wires"], PLONK["queries"], PLONK["pow_bits"])
    cap = ck.cap()
    ck.free()
    g = pref.build_circuit(np.random.default_rng(GATES["seed"]), GATES["log_n"], GATES["W"], n_routed=GATES["R"], n_public=GATES["n_public"],
                           poseidon_rows=GATES["pos_rows"], consts=(rc, circ, diag))
    gk = pkg.PlonkCircuit(prover, g["consts"], g["sigmas"], n_wires=GATES["W"], n_public=GATES["n_public"], poseidon=True)
    gates_proof = gk.prove(g["wires"], GATES["queries"], GATES["pow_bits"], public=g["public"])
    gates_cap = gk.cap()
    gk.free()
    h = pref.build_circuit(np.random.default_rng(SHA["seed"]), SHA["log_n"], SHA["W"], n_routed=SHA["R"], n_public=SHA["n_public"],
                           poseidon_rows=SHA["pos_rows"], consts=(rc, circ, diag), sha_rows=SHA["sha_rows"])
    assert len({int(np.argmax(h["consts"][6:10, r])) for r in SHA["sha_rows"]}) == 4, "pick a seed that draws every row kind"
    hk = pkg.PlonkCircuit(prover, h["consts"], h["sigmas"], n_wires=SHA["W"], n_public=SHA["n_public"], poseidon=True, sha=True)
    sha_proof = hk.prove(h["wires"], SHA["queries"], SHA["pow_bits"], public=h["public"])
    sha_cap = hk.cap()
    hk.free()
    rng = np.random.default_rng(FRI["seed"])
    batches = [pkg.PolynomialBatch.from_values(prover, rand_field(rng, (k, 1 << FRI["log_n"])), FRI["rate_bits"], FRI["cap_height"])
               for k in FRI["polys"]]
    fri_proof = prover.fri_prove(batches, FRI["rate_bits"], FRI["cap_height"], arity_bits=FRI["arity_bits"],
                                 final_poly_bits=FRI["final_poly_bits"], num_queries=FRI["queries"], pow_bits=FRI["pow_bits"])
    for b in batches:
        b.free()
    return {"note": "made by tests/golden/gen_proofs.py on an MI355X; Poseidon constants = poseidon_constants.default_constants()",
            "plonk": dict(PLONK, circuit_cap=[int(v) for v in cap], proof=plonk_proof.hex()),
            "gates": dict(GATES, public=[int(v) for v in g["public"]], circuit_cap=[int(v) for v in gates_cap], proof=gates_proof.hex()),
            "sha": dict(SHA, public=[int(v) for v in h["public"]], circuit_cap=[int(v) for v in sha_cap], proof=sha_proof.hex()),
            "fri": dict(FRI, proof=fri_proof.hex())}


if __name__ == "__main__":
    pkg = graft.load_package()
    pr = pkg.Prover(0)
    out = make(pkg, pr)
    pr.close()
    with open(sys.argv[1] if len(sys.argv) > 1 else os.path.join(HERE, "proofs.json"), "w") as f:
        json.dump(out, f)
    print("wrote proofs.json:", len(out["plonk"]["proof"]) // 2, "+", len(out["gates"]["proof"]) // 2, "+", len(out["fri"]["proof"]) // 2, "bytes")
