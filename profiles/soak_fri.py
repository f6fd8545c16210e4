#!/usr/bin/env python3
"""Soak of the FRI opening proof: random configurations (size, batches, blow-up 1..3 bits, cap height, arity 2..16, final-polynomial bound, queries,
PoW bits, 1..3 opening points with random per-batch masks) proved on the GPU and checked by the native verifier and tests/fri_verifier.py; one
flipped word per proof must be refused by both.  Unsupported combinations (GLP_E_UNSUPPORTED / INVALID) are counted, not failures.
python3 profiles/soak_fri.py [seconds=300]"""
import ctypes
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as graft  # noqa: E402
import fri_verifier as fv  # noqa: E402
from conftest import P, poseidon_consts, rand_field  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
pkg = graft.load_package()
orc = graft.load_oracle()
u64p = ctypes.POINTER(ctypes.c_uint64)
orc.orc_poseidon_permute.argtypes = [u64p]
orc.orc_poseidon_set_constants.argtypes = [u64p, u64p, u64p]
consts = poseidon_consts("small")
orc.orc_poseidon_set_constants(*(a.ctypes.data_as(u64p) for a in consts))
pr = pkg.Prover(0)
pr.set_poseidon_constants(*consts)
t0, n_ok, n_unsup, seed, shapes = time.time(), 0, 0, 0, set()
while time.time() - t0 < budget:
    seed += 1
    rng = np.random.default_rng(31000 + seed)
    log_n = int(rng.integers(3, 13))
    rb = int(rng.integers(1, 4))
    cap_h = int(rng.integers(0, min(6, log_n + rb + 1)))
    a = int(rng.integers(1, 5))
    fb = int(rng.integers(0, min(6, log_n) + 1))
    nq, pw = int(rng.integers(1, 12)), int(rng.integers(0, 9))
    polys = [int(rng.integers(1, 9)) for _ in range(int(rng.integers(1, 4)))]
    n_pts = int(rng.integers(1, 4))
    g = pow(7, (P - 1) >> log_n, P)
    mults = [1, g, pow(g, 2, P)][:n_pts]
    masks = [int(rng.integers(1, 1 << n_pts)) for _ in polys]
    masks[0] |= 1
    batches = [pkg.PolynomialBatch.from_values(pr, rand_field(rng, (k, 1 << log_n)), rb, cap_h) for k in polys]
    try:
        proof = pr.fri_prove(batches, rb, cap_h, arity_bits=a, final_poly_bits=fb, num_queries=nq, pow_bits=pw, point_mults=tuple(mults), open_masks=masks)
    except pkg.GlpError:
        n_unsup += 1
        for b in batches:
            b.free()
        continue
    finally:
        pass
    for b in batches:
        b.free()
    assert pr.fri_verify(proof, nq, pw, min_rate_bits=rb), (seed, pr.last_reject)
    info = fv.parse_and_verify(proof, orc)
    assert info["n_polys"] == polys and info["rate_bits"] == rb
    w = np.frombuffer(proof, dtype="<u8").copy()
    w[int(rng.integers(0, len(w)))] ^= np.uint64(1 << int(rng.integers(0, 62)))
    assert not pr.fri_verify(w.tobytes(), 1, 0, 1), ("flipped word accepted natively", seed)
    try:
        fv.parse_and_verify(w.tobytes(), orc)
        raise AssertionError(("flipped word accepted by the Python verifier", seed))
    except fv.VerifyError:
        pass
    shapes.add((log_n, rb, cap_h, a, fb, n_pts, len(polys)))
    n_ok += 1
print({"fri_proofs_verified_twice": n_ok, "unsupported_configurations": n_unsup, "distinct_configurations": len(shapes), "seconds": round(time.time() - t0, 1)})
