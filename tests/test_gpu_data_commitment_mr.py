"""DataCommitment over a block range as a MapReduce of proofs (data_commitment_mr.py): leaves on the SHA row gates, nodes that verify their
children in-circuit and combine the statements (SHA-256 inner nodes over the subtree roots, Poseidon tree over the tuple digests).  The root
proof's commitment equals hashlib's RFC 6962 root over the whole range, and the proof verifies only for the tuples it was made from."""
import hashlib
import importlib
import os
import struct
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import plonk_ref as pref  # noqa: E402
from conftest import poseidon_consts, ptr  # noqa: E402
import __graft_entry__ as graft  # noqa: E402


def _root(heights, roots):
    lvl = [hashlib.sha256(b"\x00" + int(h).to_bytes(32, "big") + r).digest() for h, r in zip(heights, roots)]
    while len(lvl) > 1:
        lvl = [hashlib.sha256(b"\x01" + lvl[i] + lvl[i + 1]).digest() for i in range(0, len(lvl), 2)]
    return lvl[0]


def test_tuples_digest_is_a_poseidon_tree(oracle):
    """host restatement of D: hash_no_pad per leaf (oracle permutation), binary two_to_one tree above"""
    graft.load_package()
    dm = importlib.import_module(graft.PKG_NAME + ".data_commitment_mr")
    consts = poseidon_consts("small")
    oracle.orc_poseidon_set_constants(*(ptr(a) for a in consts))
    rng = np.random.default_rng(8)
    hs = list(range(50, 58))
    rs = [rng.integers(0, 256, 32, dtype=np.uint8).tobytes() for _ in hs]

    def perm(state):
        s = np.array(state, dtype=np.uint64)
        oracle.orc_poseidon_permute(ptr(s))
        return [int(v) for v in s]
    leaves = []
    for k in range(0, 8, 2):
        words = [w for h, r in zip(hs[k:k + 2], rs[k:k + 2]) for w in dm.tuple_words(h, r)]
        st = [0] * 12
        for off in range(0, len(words), 8):
            st = perm(words[off:off + 8] + st[8:])
        leaves.append(st[:4])
    while len(leaves) > 1:
        leaves = [perm(leaves[k] + leaves[k + 1] + [0] * 4)[:4] for k in range(0, len(leaves), 2)]
    assert dm.tuples_digest(consts, hs, rs, 2) == leaves[0]


@pytest.mark.gpu
def test_data_commitment_of_a_range_by_mapreduce(prover, oracle, pkg):
    dm = importlib.import_module(graft.PKG_NAME + ".data_commitment_mr")
    consts = poseidon_consts("small")
    prover.set_poseidon_constants(*consts)
    oracle.orc_poseidon_set_constants(*(ptr(a) for a in consts))
    rng = np.random.default_rng(4100)
    heights = [7_000_000 + k for k in range(8)]
    roots = [rng.integers(0, 256, 32, dtype=np.uint8).tobytes() for _ in heights]
    mr = dm.DataCommitmentMapReduce(prover, consts, leaf_blocks=2, fan_in=2, num_queries=6, pow_bits=4)
    out = mr.prove_range(heights, roots)
    assert out["leaves"] == 4 and [lv["nodes"] for lv in out["levels"]] == [2, 1]
    assert out["commitment"] == _root(heights, roots)
    assert len(out["public"]) == 12 and out["public"][8:] == dm.tuples_digest(consts, heights, roots, 2)
    assert mr.verify(out["root_proof"], out["key"], heights, roots, out["commitment"]), prover.last_reject
    pref.verify_plonk(out["root_proof"], oracle, pos_consts=consts, public=out["public"])
    # another commitment, another tuple, another order: different statements
    wrong = bytearray(out["commitment"])
    wrong[5] ^= 1
    assert not mr.verify(out["root_proof"], out["key"], heights, roots, bytes(wrong))
    other = list(roots)
    other[3] = bytes(32)
    assert not mr.verify(out["root_proof"], out["key"], heights, other, out["commitment"])
    assert not mr.verify(out["root_proof"], out["key"], heights[::-1], roots[::-1], out["commitment"])
    # the verifier derives the key itself (expected_key on its own object and ctx), it does not take it from the prover
    p2 = pkg.Prover(0)
    p2.set_poseidon_constants(*consts)
    vr = dm.DataCommitmentMapReduce(p2, consts, leaf_blocks=2, fan_in=2, num_queries=6, pow_bits=4)
    vkey = vr.expected_key(8)
    assert np.array_equal(vkey, out["key"]) and vr.verify(out["root_proof"], vkey, heights, roots, out["commitment"])
    vr.free()
    p2.close()
    # a second range through the recorded programs (no builder run): leaf + both node levels are replays
    rec_before = dict(mr.record_seconds)
    h2 = [9_000 + 3 * k for k in range(8)]
    r2 = [rng.integers(0, 256, 32, dtype=np.uint8).tobytes() for _ in h2]
    out2 = mr.prove_range(h2, r2)
    assert mr.record_seconds == rec_before and out2["commitment"] == _root(h2, r2)
    assert mr.verify(out2["root_proof"], out2["key"], h2, r2, out2["commitment"]) and np.array_equal(out2["key"], out["key"])
    assert not mr.verify(out2["root_proof"], out2["key"], heights, roots, out["commitment"])
    # a leaf proof with a flipped word cannot be folded
    leaf, _ = mr.prove_leaf(heights[:2], roots[:2])
    leaf_b, _ = mr.prove_leaf(heights[2:4], roots[2:4])
    bad = np.frombuffer(leaf_b, dtype="<u8").copy()
    bad[len(bad) // 3] ^= np.uint64(1)
    with pytest.raises(ValueError):
        mr.reduce([leaf, bad.tobytes()])
    # the distributed form on one rank (the exchange is a no-op): same proof shape, same commitment
    out3 = mr.prove_range_distributed(heights, roots)
    assert out3["commitment"] == out["commitment"] and out3["ranks"] == 1 and np.array_equal(out3["key"], out["key"])
    assert mr.verify(out3["root_proof"], out3["key"], heights, roots, out3["commitment"])
    # ... and what two ranks would do, in turn: each folds its contiguous half, then the root over the two node proofs
    halves = []
    for r in range(2):
        lv = [mr.prove_leaf(heights[k:k + 2], roots[k:k + 2])[0] for k in range(4 * r, 4 * r + 4, 2)]
        node, _, key1, lvl = mr.reduce(lv)
        halves.append(node)
    root4, pub4, key4, _ = mr.reduce(halves, child_key=key1, level=lvl)
    assert np.array_equal(key4, out["key"]) and pub4 == out["public"]
    assert mr.verify(root4, key4, heights, roots, out["commitment"])
    mr.free()


def _header_chain(rng, bs, dm, start_hash, first, n):
    lens = (4, 12, 5, 13, 72, 34, 34, 34, 34, 34, 34, 34, 34, 22)
    prev, out = start_hash, []
    for k in range(n):
        f = [rng.integers(0, 256, L, dtype=np.uint8).tobytes() for L in lens]
        f[2] = b"\x08" + bs.encode_varint(first + k)
        f[4] = b"\x0a\x20" + prev + f[4][34:]
        f[6] = b"\x0a\x20" + f[6][2:]
        out.append(f)
        prev = dm.HeaderChainMapReduce.header_hash(f)
    return out, prev


@pytest.mark.gpu
def test_header_chain_data_commitment_by_mapreduce(prover, oracle, pkg):
    """the header-chain form: 8 headers walked from a start hash in 4 leaves of 2, nodes that verify their children in-circuit and check that they
    are adjacent (hash and height), root = (start hash, end hash, commitment, first height).  The end hash and the commitment equal the hashlib
    restatement; the proof verifies only for its statement; a second chain replays the recorded programs."""
    dm = importlib.import_module(graft.PKG_NAME + ".data_commitment_mr")
    bs = importlib.import_module(graft.PKG_NAME + ".blobstream")
    consts = poseidon_consts("small")
    prover.set_poseidon_constants(*consts)
    oracle.orc_poseidon_set_constants(*(ptr(a) for a in consts))
    rng = np.random.default_rng(4200)
    start, first = hashlib.sha256(b"trusted header").digest(), 2_500_000
    headers, end = _header_chain(rng, bs, dm, start, first, 8)
    mr = dm.HeaderChainMapReduce(prover, consts, leaf_headers=2, fan_in=2, num_queries=6, pow_bits=4)
    out = mr.prove_chain(start, first, headers)
    assert out["leaves"] == 4 and [lv["nodes"] for lv in out["levels"]] == [2, 1]
    want = _root([first + k for k in range(8)], [h[6][2:] for h in headers])
    assert out["end_hash"] == end and out["commitment"] == want == bs.data_commitment(prover, [first + k for k in range(8)], [h[6][2:] for h in headers])
    assert out["public"][24] == first and len(out["public"]) == 25
    assert mr.verify_chain(out["root_proof"], out["key"], start, end, want, first), prover.last_reject
    pref.verify_plonk(out["root_proof"], oracle, pos_consts=consts, public=out["public"])
    assert not mr.verify_chain(out["root_proof"], out["key"], start, end, want, first + 1)
    assert not mr.verify_chain(out["root_proof"], out["key"], end, end, want, first)
    assert not mr.verify_chain(out["root_proof"], out["key"], start, end, bytes(32), first)
    # leaves that are not adjacent cannot be folded: swap two leaf proofs
    l0, _ = mr.prove_leaf(start, first, headers[:2])
    mid = dm.HeaderChainMapReduce.header_hash(headers[1])
    l1, _ = mr.prove_leaf(mid, first + 2, headers[2:4])
    node, public, _, _ = mr.reduce([l0, l1])
    assert public[:8] == list(struct.unpack(">8I", start)) and public[24] == first
    with pytest.raises(ValueError):
        mr.reduce([l1, l0])
    # a leaf proved with the wrong first height for its headers: its header hashes differ from the real chain's, so it does not connect
    l1_wrong, _ = mr.prove_leaf(mid, first + 3, headers[2:4])
    with pytest.raises(ValueError):
        mr.reduce([l0, l1_wrong])
    # the distributed form on one rank, and what two ranks would do in turn (each its contiguous half, then the root over the two node proofs)
    out3 = mr.prove_chain_distributed(start, first, headers)
    assert out3["end_hash"] == end and out3["commitment"] == want and out3["ranks"] == 1 and np.array_equal(out3["key"], out["key"])
    hashes = [start] + [dm.HeaderChainMapReduce.header_hash(h) for h in headers]
    halves = []
    for r in range(2):
        lv = mr._map_chain(hashes, first, headers, 4 * r, 4 * r + 4)
        nd, _, key1, lvl = mr.reduce(lv)
        halves.append(nd)
    root4, pub4, key4, _ = mr.reduce(halves, child_key=key1, level=lvl, span=mr.last_span)
    assert pub4 == out["public"] and np.array_equal(key4, out["key"])
    # the VERIFIER's key comes from its own setup — another object on another ctx, a synthetic chain of the same shape — never from the prover
    # (a key handed over with the proof names whatever circuit the prover chose to run)
    p2 = pkg.Prover(0)
    p2.set_poseidon_constants(*consts)
    vr = dm.HeaderChainMapReduce(p2, consts, leaf_headers=2, fan_in=2, num_queries=6, pow_bits=4)
    vkey = vr.expected_key(8)
    assert np.array_equal(vkey, out["key"])
    assert vr.verify_chain(out["root_proof"], vkey, start, end, want, first), p2.last_reject
    assert not np.array_equal(vr.expected_key(4), vkey)            # a chain of another length is another circuit tower
    vr.free()
    p2.close()
    # round 3: the DEFERRED form (what 8-header leaves use): the leaves expose their headers' data hashes, the level-1 nodes hash the tuples — the
    # root states exactly the same thing (same public inputs), through other circuits (another key)
    md = dm.HeaderChainMapReduce(prover, consts, leaf_headers=2, fan_in=2, num_queries=6, pow_bits=4, defer_commitment=True)
    outd = md.prove_chain(start, first, headers)
    assert outd["public"] == out["public"] and outd["end_hash"] == end and outd["commitment"] == want and not np.array_equal(outd["key"], out["key"])
    assert md.verify_chain(outd["root_proof"], outd["key"], start, end, want, first), prover.last_reject
    assert len(pkg.proof_public_inputs(md.prove_leaf(start, first, headers[:2])[0])) == 17 + 16
    d0, _ = md.prove_leaf(start, first, headers[:2])
    d1, _ = md.prove_leaf(mid, first + 2, headers[2:4])
    with pytest.raises(ValueError):
        md.reduce([d1, d0])                                                     # not adjacent
    d1_wrong, _ = md.prove_leaf(mid, first + 3, headers[2:4])
    with pytest.raises(ValueError):
        md.reduce([d0, d1_wrong])                                               # a leaf at another height does not connect
    with pytest.raises(ValueError):
        md.prove_chain(start, first, headers[:2])                               # one leaf: no node to hash its tuples
    md.free()
    # the bytes of a header that the circuit BUILDS (the hash inside last_block_id, the height field) are not inputs: headers that disagree with
    # them are refused on the host, at a leaf's first header (k = 2) as inside a leaf (k = 3) — found by profiles/soak_combined_skip.py: the
    # former used to be ignored and the implied chain proved instead
    for k in (2, 3, 0):
        bad = [list(h) for h in headers]
        f4 = bytearray(bad[k][4])
        f4[9] ^= 4
        bad[k][4] = bytes(f4)
        with pytest.raises(ValueError):
            mr.prove_chain(start, first, bad)
    bad = [list(h) for h in headers]
    bad[4][2] = b"\x08" + bs.encode_varint(first + 5)
    with pytest.raises(ValueError):
        mr.prove_chain(start, first, bad)
    # another chain through the recorded programs
    rec_before = dict(mr.record_seconds)
    s2 = hashlib.sha256(b"another").digest()
    h2, e2 = _header_chain(rng, bs, dm, s2, 3_000_000, 8)
    out2 = mr.prove_chain(s2, 3_000_000, h2)
    assert mr.record_seconds == rec_before and out2["end_hash"] == e2 and np.array_equal(out2["key"], out["key"])
    assert mr.verify_chain(out2["root_proof"], out2["key"], s2, e2, out2["commitment"], 3_000_000)
    mr.free()


@pytest.mark.gpu
def test_data_commitment_4096_blocks_at_baseline_size(prover, oracle, pkg):
    """BASELINE configs[4] at its stated size on one GPU: 4096 blocks = 64 leaves of 64 blocks -> 8 nodes -> root (16 382 constrained SHA-256
    compressions), full parameters; the commitment equals hashlib's and the proof verifies for exactly these tuples."""
    dm = importlib.import_module(graft.PKG_NAME + ".data_commitment_mr")
    consts = poseidon_consts("small")
    prover.set_poseidon_constants(*consts)
    extra = [pkg.Prover(0) for _ in range(2)]
    for p in extra:
        p.set_poseidon_constants(*consts)
    rng = np.random.default_rng(4096)
    heights = [5_000_000 + k for k in range(4096)]
    roots = [rng.integers(0, 256, 32, dtype=np.uint8).tobytes() for _ in heights]
    mr = dm.DataCommitmentMapReduce(prover, consts, leaf_blocks=64, fan_in=8, map_provers=extra)
    out = mr.prove_range(heights, roots)
    assert out["leaves"] == 64 and [lv["nodes"] for lv in out["levels"]] == [8, 1]
    assert out["commitment"] == _root(heights, roots)
    assert mr.verify(out["root_proof"], out["key"], heights, roots, out["commitment"]), prover.last_reject
    other = list(roots)
    other[4095] = bytes(32)
    assert not mr.verify(out["root_proof"], out["key"], heights, other, out["commitment"])
    mr.free()
    for p in extra:
        p.close()


@pytest.mark.gpu
def test_cloned_node_circuits_are_the_directly_built_ones(prover, oracle, pkg, monkeypatch):
    """A node circuit verifies its 2nd..Nth child through CLONES of the first child's recorded sub-circuit (CircuitBuilder.clone_segment).  The same
    range proved with cloning off (every child laid down through the gadget code): identical verifying keys at every level, identical root
    statement; each root is accepted under the other run's key; a bad leaf among the cloned children still stops the fold."""
    dm = importlib.import_module(graft.PKG_NAME + ".data_commitment_mr")
    consts = poseidon_consts("small")
    prover.set_poseidon_constants(*consts)
    oracle.orc_poseidon_set_constants(*(ptr(a) for a in consts))
    rng = np.random.default_rng(77)
    heights = [9_000_000 + k for k in range(16)]
    roots = [rng.integers(0, 256, 32, dtype=np.uint8).tobytes() for _ in heights]
    runs = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("GLP_RECORD_CLONE", mode)
        mr = dm.DataCommitmentMapReduce(prover, consts, leaf_blocks=2, fan_in=4, num_queries=6, pow_bits=4)
        out = mr.prove_range(heights, roots)                 # 8 distinct leaf proofs -> 2 nodes of 4 -> root of 2
        keys = sorted((k[0], k[1], bytes(np.ascontiguousarray(p.key(), dtype=np.uint64))) for k, p in mr.nodes.items())
        runs[mode] = (mr, out, keys)
    (mr0, out0, keys0), (mr1, out1, keys1) = runs["0"], runs["1"]
    assert [k[:2] for k in keys0] == [(1, 4), (2, 2)] and keys0 == keys1, "a cloned node circuit differs from the directly built one"
    assert np.array_equal(out0["key"], out1["key"]) and out0["public"] == out1["public"] and out0["commitment"] == _root(heights, roots)
    assert mr0.verify(out1["root_proof"], out0["key"], heights, roots, out1["commitment"]), prover.last_reject
    assert mr1.verify(out0["root_proof"], out1["key"], heights, roots, out0["commitment"]), prover.last_reject
    pref.verify_plonk(out1["root_proof"], oracle, pos_consts=consts, public=out1["public"])
    # a FRESH object recording its node from a batch whose third child is bad: the clone's witness is refused
    mr2 = dm.DataCommitmentMapReduce(prover, consts, leaf_blocks=2, fan_in=4, num_queries=6, pow_bits=4)
    leaves = mr2.prove_leaves(heights[:8], roots[:8])
    w = np.frombuffer(leaves[2], dtype="<u8").copy()
    w[len(w) // 2] ^= np.uint64(1)
    with pytest.raises(ValueError):
        mr2.reduce(leaves[:2] + [w.tobytes()] + leaves[3:])
    for m in (mr0, mr1, mr2):
        m.free()
