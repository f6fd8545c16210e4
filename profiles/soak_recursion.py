#!/usr/bin/env python3
"""Soak of the in-circuit verifier: random leaf circuits (every row-kind mix, random shapes, cap heights, query counts, public inputs) proved on the
GPU, then verified IN-CIRCUIT by a recursion node over two such proofs (recorded, witnessed by the C evaluator, proved, verified natively and by
the Python verifier); a tampered leaf proof must be refused.  python3 profiles/soak_recursion.py [seconds=300]"""
import ctypes
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as graft  # noqa: E402
import plonk_ref as pref  # noqa: E402
from conftest import poseidon_consts  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
pkg = graft.load_package()
vc = importlib.import_module(graft.PKG_NAME + ".verifier_circuit")
orc = graft.load_oracle()
u64p = ctypes.POINTER(ctypes.c_uint64)
orc.orc_poseidon_permute.argtypes = [u64p]
orc.orc_poseidon_set_constants.argtypes = [u64p, u64p, u64p]
consts = poseidon_consts("small")
orc.orc_poseidon_set_constants(*(a.ctypes.data_as(u64p) for a in consts))
pr = pkg.Prover(0)
pr.set_poseidon_constants(*consts)
t0, n_ok, shapes, seed = time.time(), 0, set(), 0
while time.time() - t0 < budget:
    seed += 1
    rng = np.random.default_rng(5000 + seed)
    log_n = int(rng.integers(5, 11))
    n = 1 << log_n
    use_pos, use_sha, use_ext = bool(rng.integers(0, 2)), bool(rng.integers(0, 3) == 0), bool(rng.integers(0, 2))
    W = 144 if use_sha else (136 if use_pos else int(rng.choice([8, 16, 40])))
    R = int(rng.choice([r for r in (8, 16, 24, 32) if r <= W and (not use_pos or r >= 24) and (not use_sha or r >= 16)]))
    n_public = int(rng.integers(0, 4))
    rows = [int(v) for v in rng.permutation(np.arange(n_public, n))]
    k_pos, k_sha, k_ext = (int(rng.integers(1, 6)) if u else 0 for u in (use_pos, use_sha, use_ext))
    pos_rows, sha_rows, ext_rows = rows[:k_pos], rows[k_pos:k_pos + k_sha], rows[k_pos + k_sha:k_pos + k_sha + k_ext]
    cap_h = int(rng.integers(0, 5))
    nq, pw = int(rng.integers(2, 6)), int(rng.integers(0, 5))
    leaves, key = [], None
    for k in range(2):
        circ = pref.build_circuit(np.random.default_rng(7000 + seed), log_n, W, copy_prob=0.3, n_routed=R, n_public=n_public, poseidon_rows=pos_rows,
                                  consts=consts, sha_rows=sha_rows, ext_rows=ext_rows,
                                  public_values=[int(v) for v in np.random.default_rng(9000 + seed * 2 + k).integers(0, 1 << 60, n_public)] if n_public else None)
        if k == 0:
            ck = pkg.PlonkCircuit(pr, circ["consts"], circ["sigmas"], n_wires=W, n_public=n_public, poseidon=bool(pos_rows), sha=bool(sha_rows),
                                  ext=bool(ext_rows), cap_height=cap_h)
            key = ck.cap()
        leaves.append(ck.prove(circ["wires"], nq, pw, public=circ["public"]))
    rp = vc.RecursionProgram(pr, leaves, key, nq, pw, W, consts, n_routed=R, n_public=n_public, cap_height=cap_h, child_is_recursion=bool(pos_rows),
                             child_sha=bool(sha_rows), child_ext=bool(ext_rows), ext_gate=bool(seed % 2))
    node, public = rp.prove(leaves, 4, 2)
    assert pr.plonk_verify(node, rp.key(), 4, 2, public=public), (seed, pr.last_reject)
    pref.verify_plonk(node, orc, pos_consts=consts, public=public)
    bad = np.frombuffer(leaves[1], dtype="<u8").copy()
    bad[int(rng.integers(8, len(bad)))] ^= np.uint64(1 << int(rng.integers(0, 60)))
    try:
        rp.prove([leaves[0], bad.tobytes()], 4, 2)
        raise AssertionError(("tampered leaf proof folded", seed))
    except ValueError:
        pass
    rp.free()
    ck.free()
    shapes.add((log_n, W, R, n_public, cap_h, bool(pos_rows), bool(sha_rows), bool(ext_rows)))
    n_ok += 1
print({"recursion_nodes_built_proved_verified": n_ok, "distinct_leaf_shapes": len(shapes), "seconds": round(time.time() - t0, 1)})
