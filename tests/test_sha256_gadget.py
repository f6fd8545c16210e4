"""In-circuit SHA-256 (0-kno-blobstreamx_amd/gadgets.py).  CPU: the gadget's witness values reproduce hashlib for one- and two-block
messages and every gate it lays down is satisfied (the constraint system is what the GPU then proves).  GPU: the data-commitment circuit —
its public inputs are the (height, dataRoot) tuples and the commitment root, its constraints every SHA-256 compression of the RFC 6962
tree — proves, verifies with both verifiers, and exposes the root the GPU witness kernel and hashlib compute."""
import hashlib
import importlib
import os
import struct
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import fri_verifier as fv  # noqa: E402
import plonk_ref as pref  # noqa: E402
from conftest import P, poseidon_consts, ptr  # noqa: E402
import __graft_entry__ as graft  # noqa: E402


def _mods():
    graft.load_package()
    return (importlib.import_module(graft.PKG_NAME + ".gadgets"), importlib.import_module(graft.PKG_NAME + ".recursion"),
            importlib.import_module(graft.PKG_NAME + ".blobstream"))


class _NoGpu:
    """the gadget only needs a prover to BUILD (sigma on the GPU); laying down gates and witness values needs none"""


@pytest.mark.parametrize("msg", [b"", b"abc", bytes(range(55)), bytes(range(56)), b"\x00" + bytes(range(64))])
def test_gadget_values_match_hashlib_and_gates_hold(msg):
    gd, rec, _ = _mods()
    b = rec.CircuitBuilder(_NoGpu())
    g = gd.Sha256Gadget(b)
    bits = [g.one if (byte >> (7 - i)) & 1 else g.zero for byte in msg for i in range(8)]
    state = g.hash_bits(bits)
    assert b"".join(struct.pack(">I", b.value(w[1])) for w in state) == hashlib.sha256(msg).digest()
    # every gate holds on the witness, every bit is boolean where it was asserted, and the gate count is as documented
    n_gates = 0
    for (c0, c1, c2), rows in b.arith_rows.items():
        for row in rows:
            for x, y, z, w in row:
                assert (c0 * b.value(x) * b.value(y) + c1 * b.value(z) + c2 - b.value(w)) % P == 0
                n_gates += 1
    blocks = (len(msg) + 9 + 63) // 64
    assert 60_000 * blocks < n_gates < 70_000 * blocks + 200


@pytest.mark.gpu
def test_data_commitment_circuit_constrains_the_sha256_tree(prover, oracle, pkg):
    gd, rec, bs = _mods()
    rc, circ, diag = poseidon_consts("small")
    prover.set_poseidon_constants(rc, circ, diag)
    oracle.orc_poseidon_set_constants(ptr(rc), ptr(circ), ptr(diag))
    rng = np.random.default_rng(4096)
    heights = [1_000_000, 1_000_001]
    roots = [rng.integers(0, 256, 32, dtype=np.uint8).tobytes() for _ in heights]
    ck, dw, public, root = gd.data_commitment_circuit(prover, heights, roots)
    # the root the circuit exposes = the GPU witness kernel's = hashlib's (RFC 6962 tree over abi.encode(height, dataRoot))
    leaves = [bs.encode_data_root_tuple(h, r) for h, r in zip(heights, roots)]
    lh = [hashlib.sha256(b"\x00" + x).digest() for x in leaves]
    want = hashlib.sha256(b"\x01" + lh[0] + lh[1]).digest()
    assert root == want == bs.data_commitment(prover, heights, roots)
    assert public == bs.public_words(b"".join(leaves) + want) and len(public) == 40
    assert ck.log_n == 15 and ck.n_wires == 136
    proof = ck.prove_(dw, 28, 16, public=public)
    assert ck.verify(proof, 28, 16, public=public), prover.last_reject
    info = pref.verify_plonk(proof, oracle, pos_consts=(rc, circ, diag), public=public)
    assert info["log_n"] == 15
    # another root, or another tuple, is another statement
    for k in (0, 17, 39):
        other = list(public)
        other[k] ^= 1
        assert not ck.verify(proof, 28, 16, public=other)
    # and cannot be proved from this witness either: the claimed root must be the SHA-256 tree's
    lie = list(public)
    lie[-1] ^= 1
    try:
        bad = ck.prove_(dw, 28, 16, public=lie)
    except pkg.GlpError:
        bad = None
    assert bad is None or not ck.verify(bad, 28, 16, public=lie)
    dw.free()
    ck.free()


def _tm_root(heights, roots):
    lvl = [hashlib.sha256(b"\x00" + int(h).to_bytes(32, "big") + r).digest() for h, r in zip(heights, roots)]
    while len(lvl) > 1:
        lvl = [hashlib.sha256(b"\x01" + lvl[i] + lvl[i + 1]).digest() for i in range(0, len(lvl), 2)]
    return lvl[0]


def test_sha_rows_gadget_matches_hashlib_and_replays():
    """the SHA row gadget (48 W + 64 E + 64 A + 2 ADD rows per compression): the RFC 6962 tree over two tuples laid down on the builder gives
    hashlib's root; the recorded program recomputes it for other tuples (glp_witness_eval: ops SHA_E / SHA_A / SHA_W / ADD32 / BITS); an
    input that is not a 32-bit word is refused"""
    gd, rec, _ = _mods()
    consts = poseidon_consts("small")
    rng = np.random.default_rng(3)
    b = rec.CircuitBuilder(object(), n_wires=144)
    g = gd.Sha256Rows(b)
    heights, roots = [100, 101], [rng.integers(0, 256, 32, dtype=np.uint8).tobytes() for _ in range(2)]
    level = []
    for hgt, root in zip(heights, roots):
        tup = int(hgt).to_bytes(32, "big") + root
        level.append(g.hash_prefixed_64(0, [g.public_word(v) for v in struct.unpack(">16I", tup)]))
    top = g.hash_prefixed_64(1, level[0] + level[1])
    assert b"".join(struct.pack(">I", b.value(w)) for w in top) == _tm_root(heights, roots)
    prog = b.program()
    assert prog.stats["sha_rows"] == 6 * (48 + 64 + 64 + 2) + 2 * 4 + 3 * 16 and prog.stats["rows"] == 2048
    kinds = np.bincount(prog.sha_kinds, minlength=4)
    assert list(kinds[:3]) == [6 * 64, 6 * 64, 6 * 48] and prog.consts.shape[0] == 10
    assert np.array_equal(prog.consts[6:10, prog.sha_row_ids].argmax(axis=0), prog.sha_kinds)
    for _ in range(2):
        h2 = [int(rng.integers(0, 1 << 62)), int(rng.integers(0, 1 << 30))]
        r2 = [rng.integers(0, 256, 32, dtype=np.uint8).tobytes() for _ in range(2)]
        inp = [v for hgt, root in zip(h2, r2) for v in struct.unpack(">16I", int(hgt).to_bytes(32, "big") + root)]
        vals = prog.evaluate(consts, inp)
        assert b"".join(struct.pack(">I", int(vals[w])) for w in top) == _tm_root(h2, r2)
    with pytest.raises(ValueError):
        prog.evaluate(consts, [1 << 32] + inp[1:])


@pytest.mark.parametrize("length", [0, 1, 3, 55, 56, 64, 119, 130])
def test_sha_rows_hash_bytes_matches_hashlib(length):
    """hash_bytes: a message of range-checked byte variables of any length (padding as constants); the recorded program hashes other
    messages of that length, and a "byte" of 256 is refused (its range check cannot hold)"""
    gd, rec, _ = _mods()
    consts = poseidon_consts("small")
    rng = np.random.default_rng(length)
    msg = rng.integers(0, 256, length, dtype=np.uint8).tobytes()
    b = rec.CircuitBuilder(object(), n_wires=144)
    g = gd.Sha256Rows(b)
    digest = g.hash_bytes([g.byte(b.var(v)) for v in msg])
    assert b"".join(struct.pack(">I", b.value(w)) for w in digest) == hashlib.sha256(msg).digest()
    prog = b.program()
    assert prog.stats["sha_rows"] >= 178 * ((length + 8) // 64 + 1)
    if length:
        m2 = rng.integers(0, 256, length, dtype=np.uint8).tobytes()
        vals = prog.evaluate(consts, list(m2))
        assert b"".join(struct.pack(">I", int(vals[w])) for w in digest) == hashlib.sha256(m2).digest()
        with pytest.raises(ValueError):
            prog.evaluate(consts, [256] + list(m2[1:]))
        # bytes_of_word round trip on the digest words
        bs = [x for w in digest for x in g.bytes_of_word(w)]
        assert bytes(b.value(x) for x in bs) == hashlib.sha256(msg).digest()


class _HostPoseidon:
    """stands in for a Prover on the builder's side: the library's HOST permutation with the small test constants (no GPU)"""

    def poseidon_permute_host(self, states):
        graft.load_package()
        return importlib.import_module(graft.PKG_NAME).poseidon_permute_host(poseidon_consts("small"), states)


def _validators(rng, n):
    keys = [rng.integers(0, 256, 32, dtype=np.uint8).tobytes() for _ in range(n)]
    powers = [int(rng.integers(1, 1 << int(rng.integers(1, 49)))) for _ in range(n)]
    return keys, powers


def _validators_hash(bs, keys, powers):
    def tree(nodes):
        if len(nodes) == 1:
            return nodes[0]
        k = 1 << ((len(nodes) - 1).bit_length() - 1)
        return hashlib.sha256(b"\x01" + tree(nodes[:k]) + tree(nodes[k:])).digest()
    return tree([hashlib.sha256(b"\x00" + bs.encode_validator(k, p)).digest() for k, p in zip(keys, powers)])


@pytest.mark.parametrize("n", [1, 2, 3, 5])
def test_validator_set_statement_on_the_builder(n):
    """the validator-set hash (variable-length protobuf leaves, RFC 6962 tree with a non-power-of-two split) and the > 2/3 voting-power rule laid
    down on the builder: the hash equals the hashlib restatement, the recorded program replays for another set of the same shape, and a
    signer set below the threshold cannot be laid down"""
    gd, rec, bs = _mods()
    consts = poseidon_consts("small")
    rng = np.random.default_rng(100 + n)
    keys, powers = _validators(rng, n)
    order = np.argsort(powers)[::-1]
    signed = [False] * n
    acc = 0
    for i in order:                                                     # the largest holders sign until > 2/3 is reached
        signed[i] = True
        acc += powers[i]
        if 3 * acc > 2 * sum(powers):
            break
    b = rec.CircuitBuilder(object(), n_wires=144)
    g = gd.Sha256Rows(b)
    root, got, total = gd.validator_set_statement(b, g, keys, powers, signed)
    assert b"".join(struct.pack(">I", b.value(w)) for w in root) == _validators_hash(bs, keys, powers)
    assert b.value(got) == acc and b.value(total) == sum(powers)
    prog = b.program()
    # replay: same shape (same varint lengths), other keys, flags unchanged
    keys2 = [rng.integers(0, 256, 32, dtype=np.uint8).tobytes() for _ in range(n)]
    def program_inputs(ks, flags):
        """per validator its 32 key bytes then its varint groups; then the flags (the order the statement records them in)"""
        out = []
        for k2, p in zip(ks, powers):
            groups = []
            while True:
                groups.append(p & 0x7F)
                p >>= 7
                if not p:
                    break
            out += list(k2) + groups
        return out + [1 if sg else 0 for sg in flags]
    inputs = program_inputs(keys2, signed)
    vals = prog.evaluate(consts, inputs)
    assert b"".join(struct.pack(">I", int(vals[w])) for w in root) == _validators_hash(bs, keys2, powers)
    if n > 1:
        weak = [False] * n
        weak[int(order[-1])] = True                                     # only the smallest holder signs
        if 3 * powers[int(order[-1])] <= 2 * sum(powers):
            b3 = rec.CircuitBuilder(object(), n_wires=144)
            with pytest.raises(ValueError, match="assert_equal"):
                gd.validator_set_statement(b3, gd.Sha256Rows(b3), keys, powers, weak)
            # ... and the recorded program refuses those flags too (the range check of the difference fails)
            with pytest.raises(ValueError):
                prog.evaluate(consts, program_inputs(keys2, weak))


def _tm_tree(leaves):
    if len(leaves) == 1:
        return hashlib.sha256(b"\x00" + leaves[0]).digest()
    k = 1 << ((len(leaves) - 1).bit_length() - 1)
    return hashlib.sha256(b"\x01" + _tm_tree(leaves[:k]) + _tm_tree(leaves[k:])).digest()


def _header_fields(rng):
    """14 opaque field encodings of realistic lengths (version, chain id, height, time, last block id, then nine 34-byte BytesValue hashes and a
    22-byte address); index 7 is the validators_hash slot"""
    lens = [4, 12, 5, 13, 72, 34, 34, 34, 34, 34, 34, 34, 34, 22]
    return [rng.integers(0, 256, n, dtype=np.uint8).tobytes() for n in lens]


def test_header_hash_binds_the_validator_set():
    """a header's RFC 6962 root over 14 encoded fields laid down in-circuit, with field 7 bound to BytesValue(validators_hash) of a validator set
    hashed in the same circuit: equals the hashlib restatement; a header carrying another validators_hash gives another public hash"""
    gd, rec, bs = _mods()
    rng = np.random.default_rng(77)
    keys, powers = _validators(rng, 4)
    fields = _header_fields(rng)
    b = rec.CircuitBuilder(object(), n_wires=144)
    g = gd.Sha256Rows(b)
    vroot, got, total = gd.validator_set_statement(b, g, keys, powers, [True] * 4)
    field7 = [b.constant(0x0a), b.constant(0x20)] + [x for w in vroot for x in g.bytes_of_word(w)]
    hroot = gd.header_hash_statement(b, g, fields, bound={7: field7})
    vh = _validators_hash(bs, keys, powers)
    want = list(fields)
    want[7] = b"\x0a\x20" + vh
    assert b"".join(struct.pack(">I", b.value(w)) for w in hroot) == _tm_tree(want)
    assert _tm_tree(want) != _tm_tree(fields)


def _skip_case(rng, n_trusted=4, n_target=5):
    """a trusted set, a target set sharing its first three members (other powers), headers, flags: all target validators but the last sign"""
    tk, tp = _validators(rng, n_trusted)
    vk, vp = _validators(rng, n_target)
    idx = [None] * n_target
    for i in range(3):
        vk[i] = tk[i]
        idx[i] = i
    tp = [1000, 900, 800, 50][:n_trusted]
    signed = [True] * (n_target - 1) + [False]
    vp = [700, 600, 500, 400, 100][:n_target]
    return tk, tp, vk, vp, idx, signed, _header_fields(rng), _header_fields(rng)


def test_skip_statement_on_the_builder():
    """the whole non-cryptographic skip statement: both headers bound to their validator sets, > 2/3 of the target power flagged, > 1/3 of the
    TRUSTED power held by flagged validators present in both sets (same key bytes by copy constraints); each broken premise cannot be laid down"""
    gd, rec, bs = _mods()
    rng = np.random.default_rng(88)
    tk, tp, vk, vp, idx, signed, hf_t, hf_v = _skip_case(rng)
    b = rec.CircuitBuilder(_HostPoseidon(), n_wires=144)
    g = gd.Sha256Rows(b)
    ht, hv, sd = gd.skip_statement(b, g, hf_t, (tk, tp), hf_v, (vk, vp), signed, idx)
    assert [b.value(v) for v in sd] == gd.signer_digest_host(poseidon_consts("small"), vk, signed)
    want_t, want_v = list(hf_t), list(hf_v)
    want_t[8] = b"\x0a\x20" + _validators_hash(bs, tk, tp)
    want_v[7] = b"\x0a\x20" + _validators_hash(bs, vk, vp)
    to_bytes = lambda ws: b"".join(struct.pack(">I", b.value(w)) for w in ws)
    assert to_bytes(ht) == _tm_tree(want_t) and to_bytes(hv) == _tm_tree(want_v)

    def fails(**kw):
        args = dict(signed=signed, idx=idx, vk=vk)
        args.update(kw)
        bb = rec.CircuitBuilder(_HostPoseidon(), n_wires=144)
        with pytest.raises(ValueError):
            gd.skip_statement(bb, gd.Sha256Rows(bb), hf_t, (tk, tp), hf_v, (args["vk"], vp), args["signed"], args["idx"])
    fails(signed=[True, False, False, False, False])                          # 700 of 2300: not > 2/3 of the target set
    fails(signed=[False, False, True, True, True])                            # 1000 of 2300 target power... and only 800 of 2750 trusted: not > 1/3
    other = list(vk)
    other[1] = bytes(32)
    fails(vk=other)                                                           # "validator 1 is trusted validator 1" with another key
    # enough target power but too little TRUSTED power: validators 3 and 4 are not in the trusted set
    fails(signed=[False, True, True, True, True], idx=[None, 1, None, None, None])


def test_step_statement_links_the_headers():
    """the step statement: one validator set behind the trusted header's next_validators_hash and the target header's validators_hash, and the
    target's last_block_id carrying the trusted header's hash AS COMPUTED IN THE CIRCUIT; equals the hashlib restatement"""
    gd, rec, bs = _mods()
    rng = np.random.default_rng(90)
    keys, powers = _validators(rng, 4)
    hf_t, hf_v = _header_fields(rng), _header_fields(rng)
    b = rec.CircuitBuilder(_HostPoseidon(), n_wires=144)
    g = gd.Sha256Rows(b)
    ht, hv, sd = gd.step_statement(b, g, hf_t, hf_v, (keys, powers), [True] * 4)
    assert [b.value(v) for v in sd] == gd.signer_digest_host(poseidon_consts("small"), keys, [True] * 4)
    vh = _validators_hash(bs, keys, powers)
    want_t = list(hf_t)
    want_t[8] = b"\x0a\x20" + vh
    trusted_hash = _tm_tree(want_t)
    want_v = list(hf_v)
    want_v[7] = b"\x0a\x20" + vh
    want_v[4] = b"\x0a\x20" + trusted_hash + hf_v[4][34:]
    to_bytes = lambda ws: b"".join(struct.pack(">I", b.value(w)) for w in ws)
    assert to_bytes(ht) == trusted_hash and to_bytes(hv) == _tm_tree(want_v)
    # the recorded program follows another trusted header through the link: change one opaque byte of the trusted header.
    # Input order: per validator 32 key bytes + varint groups, the flags, the trusted header's witness bytes field by field, the tail of the
    # target's last_block_id, then the target's other fields
    prog = b.program()
    consts = poseidon_consts("small")
    vec = []
    for kk, p in zip(keys, powers):
        groups = []
        while True:
            groups.append(p & 0x7F)
            p >>= 7
            if not p:
                break
        vec += list(kk) + groups
    first_header_byte = len(vec) + 4
    vec += [1] * 4
    vec += [v for k, fb in enumerate(hf_t) if k != 8 for v in fb]
    vec += list(hf_v[4][34:])                                                # the opaque tail of the target's last_block_id is recorded first
    vec += [v for k, fb in enumerate(hf_v) if k not in (4, 7) for v in fb]
    assert len(vec) == prog.n_inputs
    vals = prog.evaluate(consts, vec)
    assert b"".join(struct.pack(">I", int(vals[w])) for w in hv) == _tm_tree(want_v)
    vec2 = list(vec)
    vec2[first_header_byte] ^= 1                                            # first byte of the trusted header's field 0
    vals2 = prog.evaluate(consts, vec2)
    new_t = b"".join(struct.pack(">I", int(vals2[w])) for w in ht)
    new_v = b"".join(struct.pack(">I", int(vals2[w])) for w in hv)
    assert new_t != trusted_hash and new_v != _tm_tree(want_v)                         # the target's hash moves with the trusted header's
    want_v2 = list(want_v)
    want_v2[4] = b"\x0a\x20" + new_t + hf_v[4][34:]
    assert new_v == _tm_tree(want_v2)


def _chain_reference(start, headers, first_height):
    """hashlib restatement: link every header to its predecessor's hash, then the data commitment over (height, data_hash)"""
    _, _, bs = _mods()
    hf = lambda h: b"\x08" + bs.encode_varint(h)
    start = list(start)
    start[2] = hf(first_height - 1)
    prev = _tm_tree(start)
    h_start = prev
    leaves = []
    for k, f in enumerate(headers):
        f = list(f)
        f[2] = hf(first_height + k)
        f[4] = b"\x0a\x20" + prev + f[4][34:]
        prev = _tm_tree(f)
        leaves.append(hashlib.sha256(b"\x00" + int(first_height + k).to_bytes(32, "big") + f[6][2:]).digest())
    while len(leaves) > 1:
        leaves = [hashlib.sha256(b"\x01" + leaves[j] + leaves[j + 1]).digest() for j in range(0, len(leaves), 2)]
    return h_start, prev, leaves[0]


def _chain_case(rng, n):
    def hdr():
        f = _header_fields(rng)
        f[6] = b"\x0a\x20" + f[6][2:]                                # data_hash: a BytesValue
        return f
    return hdr(), [hdr() for _ in range(n)]


def test_data_commitment_chain_statement():
    """the header-chain form: two headers after a start header, each linked through last_block_id to the in-circuit hash of its predecessor, their
    data_hash fields feeding the commitment; equals the hashlib restatement, and the commitment is the one blobstream.data_commitment's formula gives"""
    gd, rec, bs = _mods()
    rng = np.random.default_rng(95)
    start, headers = _chain_case(rng, 2)
    b = rec.CircuitBuilder(object(), n_wires=144)
    g = gd.Sha256Rows(b)
    hs, he, root = gd.data_commitment_chain_statement(b, g, start, headers, 1000)
    to_bytes = lambda ws: b"".join(struct.pack(">I", b.value(w)) for w in ws)
    want = _chain_reference(start, headers, 1000)
    assert (to_bytes(hs), to_bytes(he), to_bytes(root)) == want
    assert want[2] == _tm_root([1000, 1001], [h[6][2:] for h in headers])
    with pytest.raises(ValueError):
        bad = [list(h) for h in headers]
        bad[1][6] = bad[1][6][:-1]
        bb = rec.CircuitBuilder(object(), n_wires=144)
        gd.data_commitment_chain_statement(bb, gd.Sha256Rows(bb), start, bad, 1000)


@pytest.mark.gpu
def test_data_commitment_chain_circuit_proves(prover, oracle, pkg):
    gd, rec, bs = _mods()
    rc, circ, diag = poseidon_consts("small")
    prover.set_poseidon_constants(rc, circ, diag)
    oracle.orc_poseidon_set_constants(ptr(rc), ptr(circ), ptr(diag))
    rng = np.random.default_rng(96)
    start, headers = _chain_case(rng, 4)
    ck, dw, public, hb_s, hb_e, root = gd.data_commitment_chain_circuit(prover, start, headers, 2_000_000)
    want = _chain_reference(start, headers, 2_000_000)
    assert (hb_s, hb_e, root) == want
    assert root == bs.data_commitment(prover, [2_000_000 + k for k in range(4)], [h[6][2:] for h in headers])      # = the GPU witness kernel's
    assert public == [w for part in want for w in struct.unpack(">8I", part)]
    proof = ck.prove_(dw, 10, 6, public=public)
    assert ck.verify(proof, 10, 6, public=public), prover.last_reject
    pref.verify_plonk(proof, oracle, pos_consts=(rc, circ, diag), public=public)
    other = list(public)
    other[20] ^= 1                                                      # another commitment
    assert not ck.verify(proof, 10, 6, public=other)
    dw.free()
    ck.free()


def test_step_and_skip_bind_the_block_numbers():
    """with heights given, the headers' height fields are tied to block-number variables: a step needs target = trusted + 1, a skip
    trusted < target <= trusted + max_skip; the header hashes equal the hashlib restatement with those height encodings"""
    gd, rec, bs = _mods()
    rng = np.random.default_rng(97)
    keys, powers = _validators(rng, 3)
    hf_t, hf_v = _header_fields(rng), _header_fields(rng)
    hfield = lambda h: b"\x08" + bs.encode_varint(h)
    b = rec.CircuitBuilder(_HostPoseidon(), n_wires=144)
    g = gd.Sha256Rows(b)
    ht, hv, sd, blocks = gd.step_statement(b, g, hf_t, hf_v, (keys, powers), [True] * 3, trusted_height=12345)
    assert [b.value(v) for v in blocks] == [12345, 12346]
    vh = _validators_hash(bs, keys, powers)
    want_t = list(hf_t)
    want_t[2], want_t[8] = hfield(12345), b"\x0a\x20" + vh
    want_v = list(hf_v)
    want_v[2], want_v[7] = hfield(12346), b"\x0a\x20" + vh
    want_v[4] = b"\x0a\x20" + _tm_tree(want_t) + hf_v[4][34:]
    to_bytes = lambda ws: b"".join(struct.pack(">I", b.value(w)) for w in ws)
    assert to_bytes(ht) == _tm_tree(want_t) and to_bytes(hv) == _tm_tree(want_v)
    # skip: the gap rule
    tk, tp, vk, vp, idx, signed, hs_t, hs_v = _skip_case(rng)
    for heights, ok in (((1000, 1001), True), ((1000, 1000 + 500), True), ((1000, 1000), False), ((1000, 999), False), ((1000, 1000 + 501), False)):
        bb = rec.CircuitBuilder(_HostPoseidon(), n_wires=144)
        gg = gd.Sha256Rows(bb)
        if ok:
            out = gd.skip_statement(bb, gg, hs_t, (tk, tp), hs_v, (vk, vp), signed, idx, heights=heights, max_skip=500)
            assert [bb.value(v) for v in out[3]] == list(heights)
        else:
            with pytest.raises(ValueError):
                gd.skip_statement(bb, gg, hs_t, (tk, tp), hs_v, (vk, vp), signed, idx, heights=heights, max_skip=500)


@pytest.mark.gpu
def test_step_circuit_proves(prover, oracle, pkg):
    gd, rec, bs = _mods()
    rc, circ, diag = poseidon_consts("small")
    prover.set_poseidon_constants(rc, circ, diag)
    oracle.orc_poseidon_set_constants(ptr(rc), ptr(circ), ptr(diag))
    rng = np.random.default_rng(91)
    keys, powers = _validators(rng, 5)
    hf_t, hf_v = _header_fields(rng), _header_fields(rng)
    ck, dw, public, hb_t, hb_v = gd.step_circuit(prover, hf_t, hf_v, (keys, powers), [True] * 5)
    vh = bs.validator_set_hash(prover, keys, powers)
    want_t = list(hf_t)
    want_t[8] = b"\x0a\x20" + vh
    want_v = list(hf_v)
    want_v[7] = b"\x0a\x20" + vh
    want_v[4] = b"\x0a\x20" + _tm_tree(want_t) + hf_v[4][34:]
    assert hb_t == _tm_tree(want_t) and hb_v == _tm_tree(want_v)
    proof = ck.prove_(dw, 10, 6, public=public)
    assert ck.verify(proof, 10, 6, public=public), prover.last_reject
    pref.verify_plonk(proof, oracle, pos_consts=(rc, circ, diag), public=public)
    dw.free()
    ck.free()


@pytest.mark.gpu
def test_skip_circuit_proves(prover, oracle, pkg):
    gd, rec, bs = _mods()
    rc, circ, diag = poseidon_consts("small")
    prover.set_poseidon_constants(rc, circ, diag)
    oracle.orc_poseidon_set_constants(ptr(rc), ptr(circ), ptr(diag))
    rng = np.random.default_rng(89)
    tk, tp, vk, vp, idx, signed, hf_t, hf_v = _skip_case(rng)
    ck, dw, public, hb_t, hb_v = gd.skip_circuit(prover, hf_t, (tk, tp), hf_v, (vk, vp), signed, idx, heights=(4_000_000, 4_000_700), max_skip=1000)
    want_t, want_v = list(hf_t), list(hf_v)
    want_t[2], want_v[2] = b"\x08" + bs.encode_varint(4_000_000), b"\x08" + bs.encode_varint(4_000_700)
    want_t[8] = b"\x0a\x20" + bs.validator_set_hash(prover, tk, tp)           # the GPU witness kernel's hashes
    want_v[7] = b"\x0a\x20" + bs.validator_set_hash(prover, vk, vp)
    assert hb_t == _tm_tree(want_t) and hb_v == _tm_tree(want_v)
    assert public == list(struct.unpack(">8I", hb_t)) + list(struct.unpack(">8I", hb_v)) + gd.signer_digest_host((rc, circ, diag), vk, signed) \
        + [4_000_000, 4_000_700]                                        # ... and the two block numbers, tied to the headers' height fields
    proof = ck.prove_(dw, 10, 6, public=public)
    assert ck.verify(proof, 10, 6, public=public), prover.last_reject
    pref.verify_plonk(proof, oracle, pos_consts=(rc, circ, diag), public=public)
    other = list(public)
    other[15] ^= 1
    assert not ck.verify(proof, 10, 6, public=other)
    dw.free()
    ck.free()


@pytest.mark.gpu
def test_commit_check_circuit_proves(prover, oracle, pkg):
    gd, rec, bs = _mods()
    rc, circ, diag = poseidon_consts("small")
    prover.set_poseidon_constants(rc, circ, diag)
    oracle.orc_poseidon_set_constants(ptr(rc), ptr(circ), ptr(diag))
    rng = np.random.default_rng(654)
    keys, powers = _validators(rng, 5)
    fields = _header_fields(rng)
    ck, dw, public, hh, vh = gd.commit_check_circuit(prover, fields, 7, keys, powers, [True] * 5)
    want = list(fields)
    want[7] = b"\x0a\x20" + _validators_hash(bs, keys, powers)
    assert vh == _validators_hash(bs, keys, powers) and hh == _tm_tree(want)
    assert public[:8] == list(struct.unpack(">8I", hh)) and public[8] == public[9] == sum(powers)
    proof = ck.prove_(dw, 10, 6, public=public)
    assert ck.verify(proof, 10, 6, public=public), prover.last_reject
    pref.verify_plonk(proof, oracle, pos_consts=(rc, circ, diag), public=public)
    other = list(public)
    other[0] ^= 1                                                       # another header
    assert not ck.verify(proof, 10, 6, public=other)
    dw.free()
    ck.free()


@pytest.mark.gpu
def test_validator_set_circuit_proves(prover, oracle, pkg):
    gd, rec, bs = _mods()
    rc, circ, diag = poseidon_consts("small")
    prover.set_poseidon_constants(rc, circ, diag)
    oracle.orc_poseidon_set_constants(ptr(rc), ptr(circ), ptr(diag))
    rng = np.random.default_rng(321)
    keys, powers = _validators(rng, 6)
    signed = [True, True, True, True, True, False]
    if 3 * sum(p for p, s_ in zip(powers, signed) if s_) <= 2 * sum(powers):
        signed = [True] * 6
    ck, dw, public, digest = gd.validator_set_circuit(prover, keys, powers, signed)
    assert digest == _validators_hash(bs, keys, powers) == bs.validator_set_hash(prover, keys, powers)        # = the GPU witness kernel's
    assert public[:8] == list(struct.unpack(">8I", digest)) and public[9] == sum(powers)
    proof = ck.prove_(dw, 10, 6, public=public)
    assert ck.verify(proof, 10, 6, public=public), prover.last_reject
    pref.verify_plonk(proof, oracle, pos_consts=(rc, circ, diag), public=public)
    other = list(public)
    other[8] += 1                                                       # another signed power: another statement
    assert not ck.verify(proof, 10, 6, public=other)
    dw.free()
    ck.free()


@pytest.mark.gpu
def test_data_commitment_on_sha_rows(prover, oracle, pkg):
    """the DataCommitment statement on the SHA row gates: same public inputs and root as the bit-decomposition circuit, 2^12 rows instead
    of 2^16 for four blocks; proved, accepted by both verifiers, other statements refused"""
    gd, rec, bs = _mods()
    rc, circ, diag = poseidon_consts("small")
    prover.set_poseidon_constants(rc, circ, diag)
    oracle.orc_poseidon_set_constants(ptr(rc), ptr(circ), ptr(diag))
    rng = np.random.default_rng(4097)
    heights = [1_000_000 + k for k in range(4)]
    roots = [rng.integers(0, 256, 32, dtype=np.uint8).tobytes() for _ in heights]
    ck, dw, public, root = gd.data_commitment_rows_circuit(prover, heights, roots)
    assert root == _tm_root(heights, roots) == bs.data_commitment(prover, heights, roots)
    leaves = [bs.encode_data_root_tuple(h, r) for h, r in zip(heights, roots)]
    assert public == bs.public_words(b"".join(leaves) + root) and len(public) == 72
    assert ck.log_n == 12 and ck.n_wires == 144
    proof = ck.prove_(dw, 28, 16, public=public)
    assert ck.verify(proof, 28, 16, public=public), prover.last_reject
    info = pref.verify_plonk(proof, oracle, pos_consts=(rc, circ, diag), public=public)
    assert info["log_n"] == 12 and info["flags"] & pref.FLAG_SHA
    for k in (0, 17, 71):
        other = list(public)
        other[k] ^= 1
        assert not ck.verify(proof, 28, 16, public=other)
    lie = list(public)
    lie[-1] ^= 1
    try:
        bad = ck.prove_(dw, 28, 16, public=lie)
    except pkg.GlpError:
        bad = None
    assert bad is None or not ck.verify(bad, 28, 16, public=lie)
    # a corrupted intermediate word (a round's e_new on some E row): the copy constraints / row equations no longer hold
    w = dw.download((ck.n_wires, 1 << ck.log_n))
    w[7, 72 + 300] ^= np.uint64(1)
    try:
        bad = ck.prove(w, 28, 16, public=public)
    except pkg.GlpError:
        bad = None
    assert bad is None or not ck.verify(bad, 28, 16, public=public)
    dw.free()
    ck.free()


@pytest.mark.gpu
def test_recursion_over_data_commitment_proofs(prover, oracle, pkg):
    """two DataCommitment proofs (SHA rows: header flags 2, ten constant columns) verified completely IN-CIRCUIT — the 140
    SHA-row constraints at zeta included — by one recursion proof whose public inputs are the leaves' statements, digests and the root"""
    gd, rec, bs = _mods()
    vc = importlib.import_module(graft.PKG_NAME + ".verifier_circuit")
    consts = poseidon_consts("small")
    prover.set_poseidon_constants(*consts)
    oracle.orc_poseidon_set_constants(*(ptr(a) for a in consts))
    rng = np.random.default_rng(4098)
    nq, pw = 6, 4
    leaves, key, shape = [], None, None
    for k in range(2):
        heights = [5000 + 2 * k, 5001 + 2 * k]
        roots = [rng.integers(0, 256, 32, dtype=np.uint8).tobytes() for _ in heights]
        ck, dw, public, root = gd.data_commitment_rows_circuit(prover, heights, roots)
        proof = ck.prove_(dw, nq, pw, public=public)
        assert ck.verify(proof, nq, pw, public=public)
        if key is None:
            key, shape = ck.cap(), (ck.n_wires, ck.n_routed, len(public))
        assert np.array_equal(key, ck.cap()), "every range of the same size shares one circuit"
        leaves.append((proof, public, root))
        dw.free()
        ck.free()
    W, R, n_pub = shape
    rp = vc.RecursionProgram(prover, [p for p, _, _ in leaves], key, nq, pw, W, consts, n_routed=R, n_public=n_pub, cap_height=1, child_sha=True)
    proof, public = rp.prove([p for p, _, _ in leaves], 8, 4)
    digests = [prover.proof_digest(p) for p, _, _ in leaves]
    want = [v for (_, pub, _), d in zip(leaves, digests) for v in pub + d] + rec.merkle_root_host(prover, digests)
    assert public == want
    assert prover.plonk_verify(proof, rp.key(), 8, 4, public=public), prover.last_reject
    pref.verify_plonk(proof, oracle, pos_consts=consts, public=public)
    # a DataCommitment proof with one flipped word is refused by the recorded program
    bad = np.frombuffer(leaves[1][0], dtype="<u8").copy()
    bad[len(bad) // 2] ^= np.uint64(1)
    with pytest.raises(ValueError):
        rp.prove([leaves[0][0], bad.tobytes()], 8, 4)
    rp.free()


@pytest.mark.gpu
def test_hybrid_commit_check_with_real_signatures(prover, oracle, pkg):
    """the signature half, checked natively against what the proof exposes: a step circuit over validators whose keys are the RFC 8032 / OpenSSL
    fixtures' keys; the proof's signer digest binds (keys, flags), and blobstream.verify_signers accepts exactly when every flagged validator's
    real Ed25519 signature verifies on the GPU kernel — a forged signature, a flipped flag or another key set is refused"""
    import json
    gd, rec, bs = _mods()
    consts = poseidon_consts("small")
    prover.set_poseidon_constants(*consts)
    oracle.orc_poseidon_set_constants(*(ptr(a) for a in consts))
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ed25519.json")) as f:
        cases = [c for c in json.load(f)["cases"] if c["valid"]][:5]
    keys = [bytes.fromhex(c["pub"]) for c in cases]
    sigs = [bytes.fromhex(c["sig"]) for c in cases]
    msgs = [bytes.fromhex(c["msg"]) for c in cases]
    powers = [500, 400, 300, 200, 100]
    signed = [True, True, True, True, False]
    rng = np.random.default_rng(92)
    ck, dw, public, hb_t, hb_v = gd.step_circuit(prover, _header_fields(rng), _header_fields(rng), (keys, powers), signed)
    proof = ck.prove_(dw, 10, 6, public=public)
    assert ck.verify(proof, 10, 6, public=public), prover.last_reject
    digest = public[16:20]
    assert digest == gd.signer_digest_host(consts, keys, signed)
    given = [s_ if f else None for s_, f in zip(sigs, signed)]
    assert bs.verify_signers(prover, consts, digest, keys, signed, given, msgs)
    forged = list(given)
    forged[1] = forged[1][:10] + bytes([forged[1][10] ^ 1]) + forged[1][11:]
    assert not bs.verify_signers(prover, consts, digest, keys, signed, forged, msgs)
    assert not bs.verify_signers(prover, consts, digest, keys, [True] * 5, sigs, msgs)          # other flags than the proof was about
    assert not bs.verify_signers(prover, consts, digest, keys[::-1], signed[::-1], given[::-1], msgs[::-1])
    missing = list(given)
    missing[0] = None
    assert not bs.verify_signers(prover, consts, digest, keys, signed, missing, msgs)
    dw.free()
    ck.free()


@pytest.mark.gpu
def test_combined_skip_circuit(prover, oracle, pkg):
    """CombinedSkip's shape in one circuit: a trusted header, four chained headers ending in the target, two validator sets; the skip rules and
    the chain + data commitment meet in the two header hashes.  The headers given must be CONSISTENT (real field bytes where the other half binds
    them): a chain whose target header carries another validators_hash cannot be laid down."""
    gd, rec, bs = _mods()
    rc, circ, diag = poseidon_consts("small")
    prover.set_poseidon_constants(rc, circ, diag)
    oracle.orc_poseidon_set_constants(ptr(rc), ptr(circ), ptr(diag))
    rng = np.random.default_rng(98)
    tk, tp, vk, vp, idx, signed, _, _ = _skip_case(rng)
    h0 = 5_000_000
    hfield = lambda h: b"\x08" + bs.encode_varint(h)
    trusted, chain = _chain_case(rng, 4)
    trusted[2], trusted[8] = hfield(h0), b"\x0a\x20" + _validators_hash(bs, tk, tp)
    prev = _tm_tree(trusted)
    for k, f in enumerate(chain):                                       # make the headers a real chain (what a node would hand the prover)
        f[2] = hfield(h0 + 1 + k)
        f[4] = b"\x0a\x20" + prev + f[4][34:]
        if k == 3:
            f[7] = b"\x0a\x20" + _validators_hash(bs, vk, vp)
        prev = _tm_tree(f)
    ck, dw, public, hb_t, hb_v, root = gd.combined_skip_circuit(prover, trusted, (tk, tp), chain, (vk, vp), signed, idx, h0, max_skip=1000)
    assert hb_t == _tm_tree(trusted) and hb_v == prev
    assert root == bs.data_commitment(prover, [h0 + 1 + k for k in range(4)], [f[6][2:] for f in chain])
    assert public == list(struct.unpack(">8I", hb_t)) + list(struct.unpack(">8I", hb_v)) + gd.signer_digest_host((rc, circ, diag), vk, signed) \
        + [h0, h0 + 4] + list(struct.unpack(">8I", root))
    proof = ck.prove_(dw, 10, 6, public=public)
    assert ck.verify(proof, 10, 6, public=public), prover.last_reject
    pref.verify_plonk(proof, oracle, pos_consts=(rc, circ, diag), public=public)
    dw.free()
    ck.free()
    broken = [list(f) for f in chain]
    broken[3][7] = b"\x0a\x20" + bytes(32)                             # the chain's target header names another validator set
    with pytest.raises(ValueError):
        gd.combined_skip_circuit(prover, trusted, (tk, tp), broken, (vk, vp), signed, idx, h0, max_skip=1000)


def test_chain_leaf_height_words_cannot_alias():
    """the leaf of the header-chain MapReduce hashes abi.encode(height, data_hash) with the height as two 32-bit words (hi, lo)
    tied to the height variable by hi * 2^32 + lo == height IN THE FIELD.  Without a bound on hi the pair (2^32 - 1, height + 1) satisfies that
    equation too (it is height + p) and a prover could commit to another tuple.  hi is now shown to be below 2^17, so the sum cannot wrap: the
    aliased pair cannot be laid down, the honest one can, and the leaf's tuple hash is hashlib's."""
    graft.load_package()
    dm = importlib.import_module(graft.PKG_NAME + ".data_commitment_mr")
    gd, rec, bs = _mods()
    rng = np.random.default_rng(312)
    first = 2_500_000
    lens = (4, 12, 5, 13, 72, 34, 34, 34, 34, 34, 34, 34, 34, 22)
    start = hashlib.sha256(b"trusted").digest()

    def header(k, prev):
        f = [rng.integers(0, 256, L, dtype=np.uint8).tobytes() for L in lens]
        f[2] = b"\x08" + bs.encode_varint(first + k)
        f[4] = b"\x0a\x20" + prev + f[4][34:]
        f[6] = b"\x0a\x20" + f[6][2:]
        return f
    h0 = header(0, start)
    headers = [h0, header(1, dm.HeaderChainMapReduce.header_hash(h0))]

    def lay_down(alias):
        b = rec.CircuitBuilder(object(), n_wires=144)
        if alias:
            honest = b.bit_field

            def forged(x, shift, bits):
                if (shift, bits) == (32, 17):
                    return b.var(0xFFFFFFFF)                              # a malicious prover's free choice instead of the computed field
                if (shift, bits) == (0, 32) and b.value(x) >= first:
                    return b.var(b.value(x) + 1)
                return honest(x, shift, bits)
            b.bit_field = forged
        pub = dm._chain_leaf_statement(b, gd.Sha256Rows(b), list(struct.unpack(">8I", start)), first, headers, 4)
        return b, pub
    # the forged pair does satisfy the field equation the circuit used to rely on alone ...
    assert ((0xFFFFFFFF << 32) + first + 1) % P == first
    # ... and is refused now (the range check of hi * 2^15 sees a value above 32 bits)
    with pytest.raises(ValueError):
        lay_down(alias=True)
    b, pub = lay_down(alias=False)
    to_bytes = lambda ws: b"".join(struct.pack(">I", b.value(w)) for w in ws)
    assert to_bytes(pub[16:24]) == _tm_root([first, first + 1], [h[6][2:] for h in headers]) and b.value(pub[24]) == first
    assert to_bytes(pub[8:16]) == dm.HeaderChainMapReduce.header_hash(headers[1])
    with pytest.raises(ValueError):
        dm.HeaderChainMapReduce(object(), poseidon_consts("small"), height_varint_bytes=8)


def test_skip_statement_inputs_mirror_the_builder():
    """the recorded outer circuit of the CombinedSkip MapReduce replays the skip statement from a flat input vector
    (gadgets.skip_statement_inputs): it must list the free variables in the order skip_statement creates them.  Checked by evaluating the recorded
    program on that vector (C evaluator) against the builder's own values, for the recording case and for ANOTHER case of the same shape; a case
    that breaks the 2/3 rule is refused by the evaluator."""
    gd, rec, bs = _mods()
    rng = np.random.default_rng(188)
    tk, tp, vk, vp, idx, signed, hf_t, hf_v = _skip_case(rng)
    b = rec.CircuitBuilder(_HostPoseidon(), n_wires=144)
    b.auto_tag_list = 1
    gd.skip_statement(b, gd.Sha256Rows(b), hf_t, (tk, tp), hf_v, (vk, vp), signed, idx, heights=(5000, 5100), max_skip=1000)
    b.auto_tag_list = None
    assert all(t == (1, k) for k, t in enumerate(b.input_tags))
    prog = b.program()
    consts = poseidon_consts("small")
    inp = gd.skip_statement_inputs(hf_t, (tk, tp), hf_v, (vk, vp), signed, heights=(5000, 5100))
    assert len(inp) == prog.n_inputs
    assert np.array_equal(prog.evaluate(consts, inp), np.array(b.values, dtype=np.uint64))
    # through the tagged path, the way RecursionProgram.witness feeds it: word list 0 is a proof (unused here), list 1 the statement's inputs
    via_tags, _ = prog.inputs_from_words([np.zeros(1, dtype=np.uint64), np.array(inp, dtype=np.uint64)])
    assert np.array_equal(via_tags, np.array(inp, dtype=np.uint64))
    # another case of the same shape (other keys, fields, heights with as many varint groups; powers with as many groups)
    tk2, _, vk2, _, _, _, hf_t2, hf_v2 = _skip_case(rng)
    for i in range(3):
        vk2[i] = tk2[i]
    tp2, vp2 = [999, 901, 799, 51], [701, 599, 501, 399, 101]
    b2 = rec.CircuitBuilder(_HostPoseidon(), n_wires=144)
    gd.skip_statement(b2, gd.Sha256Rows(b2), hf_t2, (tk2, tp2), hf_v2, (vk2, vp2), signed, idx, heights=(6000, 6900), max_skip=1000)
    vals2 = prog.evaluate(consts, gd.skip_statement_inputs(hf_t2, (tk2, tp2), hf_v2, (vk2, vp2), signed, heights=(6000, 6900)))
    assert np.array_equal(vals2, np.array(b2.values, dtype=np.uint64))
    # too little signed power: the evaluator refuses (a range-checked difference is not a 32-bit word / a copy constraint fails)
    with pytest.raises(ValueError):
        prog.evaluate(consts, gd.skip_statement_inputs(hf_t2, (tk2, tp2), hf_v2, (vk2, vp2), [True, False, False, False, False], heights=(6000, 6900)))
    # a gap above max_skip is refused too
    with pytest.raises(ValueError):
        prog.evaluate(consts, gd.skip_statement_inputs(hf_t2, (tk2, tp2), hf_v2, (vk2, vp2), signed, heights=(6000, 7001)))
