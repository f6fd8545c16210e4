#!/usr/bin/env python3
"""profiles/soak_signatures.py <seconds> — soak of the signature-set MapReduce (signature_mr.py) on one GPU: random validator sets (2..14 validators:
the digest pads to a power of two, the tree proves only the groups that hold a validator and enters the all-padding groups as constants, random flags, random keys and votes), each proved to a root that must verify for the host-computed signer digest and
block hash; every second set is then corrupted in one random way — a flipped signature bit of a flagged slot, a flagged slot signed by another key,
a vote naming another block, a flag without a signature — and must be REFUSED (ValueError before or while folding).  6 queries / 4 PoW bits."""
import hashlib
import importlib
import json
import os
import random
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
pkg = graft.load_package()
pc = importlib.import_module(graft.PKG_NAME + ".poseidon_constants")
sm = importlib.import_module(graft.PKG_NAME + ".signature_mr")
gd = importlib.import_module(graft.PKG_NAME + ".gadgets")
ec = importlib.import_module(graft.PKG_NAME + ".ed25519_circuit")
consts = tuple(np.array(a, dtype=np.uint64) for a in pc.default_constants())
provers = [pkg.Prover(0) for _ in range(3)]
for p in provers:
    p.set_poseidon_constants(*consts)
mr = sm.SignatureSetMapReduce(provers[0], consts, msg_len=64, hash_offset=12, fan_in=2, num_queries=6, pow_bits=4, map_provers=provers[1:])
rng = random.Random(2024)
t_end = time.time() + budget
stats = {"sets": 0, "slots": 0, "flagged": 0, "accepted": 0, "corrupted": 0, "refused": 0, "by_kind": {}}
t_mark = time.time()
while time.time() < t_end:
    if time.time() - t_mark > 60:                            # a progress line a minute (a silent run looks hung to the GPU runner)
        t_mark = time.time()
        print(json.dumps({"progress": {k: stats[k] for k in ("sets", "accepted", "corrupted", "refused")}}), flush=True)
    n = rng.randint(2, 14)
    block = hashlib.sha256(str(rng.random()).encode()).digest()
    seeds = [bytes(rng.randrange(256) for _ in range(32)) for _ in range(n)]
    msgs = [mr.vote_bytes(block, rng.randrange(1000)) for _ in range(n)]
    flags = [rng.random() < 0.7 for _ in range(n)]
    keys, sigs = [], []
    for sd, m, f in zip(seeds, msgs, flags):
        pub, sig = ec.keypair_and_sign(sd, m)
        keys.append(pub)
        sigs.append(sig if f else None)
    out = mr.prove_set(keys, sigs, msgs, flags)
    total = 1 << (n - 1).bit_length()
    want = gd.signer_digest_host(consts, keys, flags, pad_to=total)
    ok = out["block_hash"] == block and out["signer_digest"] == want and mr.verify_set(out["root_proof"], out["key"], block, want)
    other = list(want)
    other[0] ^= 1
    ok = ok and not mr.verify_set(out["root_proof"], out["key"], block, other)
    stats["sets"] += 1
    stats["slots"] += total
    stats["leaves_proved"] = stats.get("leaves_proved", 0) + out["slots"]
    stats["trimmed_sets"] = stats.get("trimmed_sets", 0) + (out["slots"] < total)
    stats["flagged"] += sum(flags)
    stats["accepted"] += bool(ok)
    if not ok:
        print(json.dumps({"FAILED": "accept", "n": n, "flags": flags}), flush=True)
        break
    if stats["sets"] % 2 == 0 and any(flags):
        i = rng.choice([j for j, f in enumerate(flags) if f])
        kind = rng.choice(["sig_bit", "other_key", "other_block", "flag_without_sig"])
        k2, s2, m2, f2 = list(keys), list(sigs), list(msgs), list(flags)
        if kind == "sig_bit":
            b = bytearray(s2[i])
            b[rng.randrange(64)] ^= 1 << rng.randrange(8)
            s2[i] = bytes(b)
        elif kind == "other_key":
            k2[i] = ec.keypair_and_sign(bytes(32), b"")[0]
        elif kind == "other_block":
            m2[i] = mr.vote_bytes(hashlib.sha256(b"x" + block).digest(), 5)
            s2[i] = ec.keypair_and_sign(seeds[i], m2[i])[1]
        else:
            j = rng.randrange(n)
            if f2[j]:
                s2[j] = bytes(64)
            else:
                f2[j], s2[j] = True, bytes(64)
        stats["corrupted"] += 1
        stats["by_kind"][kind] = stats["by_kind"].get(kind, 0) + 1
        try:
            mr.prove_set(k2, s2, m2, f2)
            print(json.dumps({"FAILED": "a corrupted set was proved", "kind": kind}), flush=True)
            break
        except ValueError:
            stats["refused"] += 1
stats["seconds"] = round(budget, 1)
stats["all_accepted"] = stats["accepted"] == stats["sets"]
stats["all_corrupted_refused"] = stats["refused"] == stats["corrupted"]
print(json.dumps(stats), flush=True)
mr.free()
for p in provers:
    p.close()
