#!/usr/bin/env python3
"""Prove the data commitment of a block range as a MapReduce of proofs and check it like a consumer would (needs an MI355X).
    python examples/data_commitment_range.py [blocks=1024]
Map: leaves of 64 blocks on the SHA row gates; Reduce: nodes that verify their children in-circuit (0-kno-blobstreamx_amd/data_commitment_mr.py)."""
import hashlib
import importlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft  # noqa: E402

blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
pkg = graft.load_package()
dm = importlib.import_module(graft.PKG_NAME + ".data_commitment_mr")
pc = importlib.import_module(graft.PKG_NAME + ".poseidon_constants")
consts = tuple(np.array(a, dtype=np.uint64) for a in pc.default_constants())     # stand-in constants: the library ships none
provers = [pkg.Prover(0) for _ in range(3)]                                       # one main + two more for the Map step
for p in provers:
    p.set_poseidon_constants(*consts)

rng = np.random.default_rng(1)
heights = [1_000_000 + i for i in range(blocks)]
data_roots = [rng.integers(0, 256, 32, dtype=np.uint8).tobytes() for _ in heights]

mr = dm.DataCommitmentMapReduce(provers[0], consts, leaf_blocks=64, fan_in=8, map_provers=provers[1:])
for label in ("first run (records the circuits)", "second run"):
    t0 = time.perf_counter()
    out = mr.prove_range(heights, data_roots)
    print(f"{label}: {time.perf_counter() - t0:.2f} s  (map {out['map_seconds']} s, reduce {out['reduce_seconds']} s, proof {len(out['root_proof'])} bytes)")

# the consumer: knows the tuples, holds the root proof and the root circuit's key — nothing else
level = [hashlib.sha256(b"\x00" + int(h).to_bytes(32, "big") + r).digest() for h, r in zip(heights, data_roots)]
while len(level) > 1:
    level = [hashlib.sha256(b"\x01" + level[i] + level[i + 1]).digest() for i in range(0, len(level), 2)]
assert out["commitment"] == level[0], "the proof's commitment is the RFC 6962 root of the tuples"
assert mr.verify(out["root_proof"], out["key"], heights, data_roots, out["commitment"])
print("commitment", out["commitment"].hex(), "verified")
mr.free()
for p in provers:
    p.close()
