#!/usr/bin/env python3
"""Where the arithmetic gates of the in-circuit verifier go: one real 2^16 x 80 leaf proof laid down by verify_in_circuit with every gate attributed
to the innermost verifier_circuit / recursion function on the call stack.  python3 profiles/recursion_gate_census.py -> JSON"""
import collections
import importlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402
import bench  # noqa: E402

pkg = graft.load_package()
vcm = importlib.import_module(graft.PKG_NAME + ".verifier_circuit")
rec = importlib.import_module(graft.PKG_NAME + ".recursion")
pc = importlib.import_module(graft.PKG_NAME + ".poseidon_constants")
rc, circ, diag = (np.array(a, dtype=np.uint64) for a in pc.default_constants())
pr = pkg.Prover(0)
pr.set_poseidon_constants(rc, circ, diag)
consts, sigmas, wires = bench.synthetic_circuit(pr, 16, 80)
ck = pkg.PlonkCircuit(pr, consts, sigmas)
proof = ck.prove_(pr.to_device(wires), 28, 16)
b = rec.CircuitBuilder(pr)
census, pos_census = collections.Counter(), collections.Counter()
orig_arith, orig_pos = b.arith, b.poseidon
SKIP = {"arith", "mul", "add", "sub", "lin", "k", "constant", "select", "e_mul", "e_add", "e_sub", "e_scale", "e_scale_const", "e_muladd_base", "e_inv", "inv",
        "e_select", "e_eq", "assert_bool", "counted", "counted_pos", "<lambda>", "<listcomp>", "e_from_base", "bit_select_const", "poseidon", "_duplex"}


def site():
    f = sys._getframe(2)
    names = []
    while f is not None and len(names) < 12:
        fn = f.f_code.co_name
        if fn == "verify_in_circuit":
            names.append(f"verify_in_circuit:{f.f_lineno // 20 * 20}")
            break
        if fn not in SKIP:
            names.append(fn)
        f = f.f_back
    return " < ".join(names[:2]) if names else "?"


def counted(*a, **k):
    census[site()] += 1
    return orig_arith(*a, **k)


def counted_pos(*a, **k):
    pos_census[site()] += 1
    return orig_pos(*a, **k)


b.arith, b.poseidon = counted, counted_pos
vcm.verify_in_circuit(b, proof, ck.cap(), 28, 16, 80)
tot = sum(census.values())
print(json.dumps({"arith_gates_per_leaf_proof": tot, "poseidon_rows_per_leaf_proof": sum(pos_census.values()),
                  "arith_by_site": {k: v for k, v in census.most_common(25)}, "poseidon_by_site": dict(pos_census.most_common(10))}, indent=1))
