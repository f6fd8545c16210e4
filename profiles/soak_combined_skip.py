#!/usr/bin/env python3
"""profiles/soak_combined_skip.py <seconds> — soak of the COMPLETE statement (combined_skip_mr.py + signature_mr.py) on one GPU, one fixed shape
(skip = 8, 2-header leaves, 4 trusted / 5 target validators of which 3 are shared; 6 queries / 4 PoW bits): per iteration a fresh random case —
keys, powers, headers, which validators signed — proved through the header-chain MapReduce, the signature-set MapReduce and the outer circuit.
When the signers hold > 2/3 of the target power and the shared signers > 1/3 of the trusted power the root must verify for exactly the statement
recomputed on the host (hashlib, host signer digest) under the key a SECOND object on other ctxs derived for itself; otherwise the case must be REFUSED.  Every second provable case is then corrupted in one random
way — a broken chain link, a header at another height, a target header naming another validator set, a forged signature, a vote for another block —
and must be refused."""
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
pc, cs, sm, dm, gd, bs, ec = (importlib.import_module(graft.PKG_NAME + m) for m in
                              (".poseidon_constants", ".combined_skip_mr", ".signature_mr", ".data_commitment_mr", ".gadgets", ".blobstream", ".ed25519_circuit"))
consts = tuple(np.array(a, dtype=np.uint64) for a in pc.default_constants())
provers = [pkg.Prover(0) for _ in range(2)]
for p in provers:
    p.set_poseidon_constants(*consts)
SKIP, IDX = 8, [0, 1, 2, None, None]
sigs = sm.SignatureSetMapReduce(provers[1], consts, msg_len=48, hash_offset=8, fan_in=2, num_queries=6, pow_bits=4)
mr = cs.CombinedSkipMapReduce(provers[0], consts, skip=SKIP, batch=2, fan_in=2, num_queries=6, pow_bits=4, max_skip=100, signatures=sigs)


def tm_root(heights, roots):
    lvl = [hashlib.sha256(b"\x00" + int(h).to_bytes(32, "big") + r).digest() for h, r in zip(heights, roots)]
    while len(lvl) > 1:
        lvl = [hashlib.sha256(b"\x01" + lvl[i] + lvl[i + 1]).digest() for i in range(0, len(lvl), 2)]
    return lvl[0]


def statement(case):
    tf, (tk, tp), chain, (vk, vp), signed, idx, h0 = case
    return dict(trusted_hash=dm.HeaderChainMapReduce.header_hash(tf), target_hash=dm.HeaderChainMapReduce.header_hash(chain[-1]),
                signer_digest=gd.signer_digest_host(consts, vk, signed, pad_to=8), trusted_block=h0, target_block=h0 + SKIP,
                commitment=tm_root([h0 + 1 + k for k in range(SKIP)], [f[6][2:] for f in chain]))


# the VERIFIER's key: from a setup of its own on another ctx (never the key a prover hands over)
vp_ = [pkg.Prover(0) for _ in range(2)]
for p in vp_:
    p.set_poseidon_constants(*consts)
vsigs = sm.SignatureSetMapReduce(vp_[1], consts, msg_len=48, hash_offset=8, fan_in=2, num_queries=6, pow_bits=4)
vmr = cs.CombinedSkipMapReduce(vp_[0], consts, skip=SKIP, batch=2, fan_in=2, num_queries=6, pow_bits=4, max_skip=100, signatures=vsigs)
VKEY = vmr.expected_key(4, 5, IDX, power_groups=3)
rng = random.Random(777)
stats = {"cases": 0, "provable": 0, "accepted": 0, "rule_fails": 0, "rule_fails_refused": 0, "corrupted": 0, "refused": 0, "by_kind": {}}
t_end, t_mark = time.time() + budget, time.time()
failed = None
while time.time() < t_end and failed is None:
    if time.time() - t_mark > 60:
        t_mark = time.time()
        print(json.dumps({"progress": {k: stats[k] for k in ("cases", "accepted", "rule_fails_refused", "corrupted", "refused")}}), flush=True)
    *case, seeds = mr.synthetic_case(4, 5, IDX, trusted_height=2_500_000 + rng.randrange(1 << 20), power_groups=3, seed=rng.randrange(1 << 30), real_keys=True)
    tf, (tk, tp), chain, (vk, vp), _, idx, h0 = case
    signed = [rng.random() < 0.8 for _ in vk]
    case[4] = signed
    rule = 3 * sum(p for p, s in zip(vp, signed) if s) > 2 * sum(vp) and 3 * sum(tp[t] for t, s in zip(idx, signed) if s and t is not None) > sum(tp)
    votes = mr.synthetic_votes(case, seeds)
    stats["cases"] += 1
    if not rule:
        stats["rule_fails"] += 1
        try:
            mr.prove_skip(*case, votes=votes)
            failed = {"FAILED": "a case below the power thresholds was proved", "signed": signed}
        except ValueError:
            stats["rule_fails_refused"] += 1
        continue
    stats["provable"] += 1
    out = mr.prove_skip(*case, votes=votes)
    want = statement(tuple(case))
    ok = all(out[k] == want[k] for k in want) and np.array_equal(out["key"], VKEY) and vmr.verify(out["root_proof"], VKEY, **want)
    ok = ok and not vmr.verify(out["root_proof"], VKEY, **dict(want, target_block=want["target_block"] + 1))
    stats["accepted"] += bool(ok)
    if not ok:
        failed = {"FAILED": "accept", "signed": signed}
        break
    if stats["provable"] % 2 == 0:
        kind = rng.choice(["link", "height", "other_set", "forged_signature", "other_block"])
        c2, v2 = [list(x) if isinstance(x, list) else x for x in case], (list(votes[0]), list(votes[1]))
        c2[2] = [list(f) for f in chain]
        k = rng.randrange(1, SKIP)
        if kind == "link":
            f4 = bytearray(c2[2][k][4])
            f4[2 + rng.randrange(32)] ^= 1 << rng.randrange(8)
            c2[2][k][4] = bytes(f4)
        elif kind == "height":
            c2[2][k][2] = b"\x08" + bs.encode_varint(h0 + 1 + k + 1)
        elif kind == "other_set":
            c2[2][-1][7] = b"\x0a\x20" + hashlib.sha256(b"another set").digest()
        else:
            i = rng.choice([j for j, s in enumerate(signed) if s])
            if kind == "forged_signature":
                b = bytearray(v2[0][i])
                b[rng.randrange(64)] ^= 1 << rng.randrange(8)
                v2[0][i] = bytes(b)
            else:
                v2[1][i] = sigs.vote_bytes(hashlib.sha256(b"x" + want["target_hash"]).digest(), i)
                v2[0][i] = ec.keypair_and_sign(seeds[i], v2[1][i])[1]
        stats["corrupted"] += 1
        stats["by_kind"][kind] = stats["by_kind"].get(kind, 0) + 1
        try:
            mr.prove_skip(*c2, votes=v2)
            failed = {"FAILED": "a corrupted case was proved", "kind": kind}
        except ValueError:
            stats["refused"] += 1
if failed:
    print(json.dumps(failed), flush=True)
stats["seconds"] = round(budget, 1)
stats["all_ok"] = failed is None and stats["accepted"] == stats["provable"] and stats["refused"] == stats["corrupted"] and stats["rule_fails_refused"] == stats["rule_fails"]
print(json.dumps(stats), flush=True)
for o in (mr, sigs, vmr, vsigs):
    o.free()
for p in provers + vp_:
    p.close()
