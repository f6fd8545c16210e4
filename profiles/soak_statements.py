#!/usr/bin/env python3
"""Soak of the SHA-row statements: random validator sets (1..10 validators, powers drawn around the varint boundaries 2^7k and at random), random
signer subsets above 2/3, random header field lengths — validator_set_circuit / commit_check_circuit / step_circuit laid down, proved on the GPU,
verified natively; hashes compared with hashlib and with the GPU witness kernel.  python3 profiles/soak_statements.py [seconds=240]"""
import hashlib
import importlib
import os
import struct
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
pkg = graft.load_package()
gd = importlib.import_module(graft.PKG_NAME + ".gadgets")
bs = importlib.import_module(graft.PKG_NAME + ".blobstream")
pc = importlib.import_module(graft.PKG_NAME + ".poseidon_constants")
consts = tuple(np.array(a, dtype=np.uint64) for a in pc.default_constants())
pr = pkg.Prover(0)
pr.set_poseidon_constants(*consts)


def tree(leaves):
    if len(leaves) == 1:
        return hashlib.sha256(b"\x00" + leaves[0]).digest()
    k = 1 << ((len(leaves) - 1).bit_length() - 1)
    return hashlib.sha256(b"\x01" + tree(leaves[:k]) + tree(leaves[k:])).digest()


t0, n_ok, seed, kinds = time.time(), 0, 0, {"validator_set": 0, "commit_check": 0, "step": 0}
while time.time() - t0 < budget:
    seed += 1
    rng = np.random.default_rng(77000 + seed)
    n = int(rng.integers(1, 11))
    keys = [rng.integers(0, 256, 32, dtype=np.uint8).tobytes() for _ in range(n)]
    powers = []
    for _ in range(n):
        k = int(rng.integers(1, 7))
        edge = (1 << (7 * k)) + int(rng.integers(-2, 3))
        powers.append(max(1, edge) if rng.random() < 0.5 else int(rng.integers(1, 1 << int(rng.integers(1, 49)))))
    order = np.argsort(powers)[::-1]
    signed, acc = [False] * n, 0
    for i in order:
        signed[i] = True
        acc += powers[i]
        if 3 * acc > 2 * sum(powers):
            break
    vh = tree([bs.encode_validator(k, p) for k, p in zip(keys, powers)])
    assert vh == bs.validator_set_hash(pr, keys, powers)
    fields = lambda: [rng.integers(0, 256, int(rng.integers(1, 90)), dtype=np.uint8).tobytes() for _ in range(14)]
    which = seed % 3
    if which == 0:
        ck, dw, public, digest = gd.validator_set_circuit(pr, keys, powers, signed)
        assert digest == vh and public[8] == acc and public[9] == sum(powers)
        kinds["validator_set"] += 1
    elif which == 1:
        hf = fields()
        ck, dw, public, hh, vh2 = gd.commit_check_circuit(pr, hf, 7, keys, powers, signed)
        hf[7] = b"\x0a\x20" + vh
        assert vh2 == vh and hh == tree(hf)
        kinds["commit_check"] += 1
    else:
        hf_t, hf_v = fields(), fields()
        hf_v[4] = rng.integers(0, 256, 72, dtype=np.uint8).tobytes()
        h0 = int(rng.integers(1, 1 << 40))
        ck, dw, public, hb_t, hb_v = gd.step_circuit(pr, hf_t, hf_v, (keys, powers), signed, trusted_height=h0)
        hf_t[2], hf_t[8] = b"\x08" + bs.encode_varint(h0), b"\x0a\x20" + vh
        hf_v[2], hf_v[7] = b"\x08" + bs.encode_varint(h0 + 1), b"\x0a\x20" + vh
        hf_v[4] = b"\x0a\x20" + tree(hf_t) + hf_v[4][34:]
        assert hb_t == tree(hf_t) and hb_v == tree(hf_v) and public[-2:] == [h0, h0 + 1]
        assert public[16:20] == gd.signer_digest_host(consts, keys, signed)
        kinds["step"] += 1
    proof = ck.prove_(dw, 6, 4, public=public)
    assert ck.verify(proof, 6, 4, public=public), (seed, pr.last_reject)
    other = list(public)
    other[int(rng.integers(0, len(other)))] ^= 1
    assert not ck.verify(proof, 6, 4, public=other)
    dw.free()
    ck.free()
    n_ok += 1
print({"statement_circuits_built_proved_verified": n_ok, "by_kind": kinds, "seconds": round(time.time() - t0, 1)})
