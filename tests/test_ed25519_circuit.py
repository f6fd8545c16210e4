"""Ed25519 verification in-circuit (0-kno-blobstreamx_amd/ed25519_circuit.py): the non-native field arithmetic against Python integers, the
witness evaluator's product op against the builder, SHA-512 by bit decomposition against hashlib, the curve formulas against a textbook affine
implementation, and the whole statement on the RFC 8032 §7.1 vectors and the OpenSSL-made fixtures (tests/golden/ed25519.json): valid signatures can
be laid down, invalid ones cannot; a recorded circuit replays other signatures through the C evaluator and refuses forged ones.  GPU: one signature
proved and verified by both verifiers."""
import hashlib
import importlib
import json
import os
import random
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import plonk_ref as pref  # noqa: E402
from conftest import P, poseidon_consts, ptr  # noqa: E402
import __graft_entry__ as graft  # noqa: E402

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _mods():
    graft.load_package()
    return importlib.import_module(graft.PKG_NAME + ".ed25519_circuit"), importlib.import_module(graft.PKG_NAME + ".recursion")


def _gates_hold(b):
    for (c0, c1, c2), rows in b.arith_rows.items():
        for row in rows:
            for x, y, z, w in row:
                assert (c0 * b.value(x) * b.value(y) + c1 * b.value(z) + c2 - b.value(w)) % P == 0


def test_non_native_field_arithmetic():
    ec, rec = _mods()
    b = rec.CircuitBuilder(object(), n_wires=144)
    f = ec.NNF(b)
    rng = random.Random(25519)
    vals = [0, 1, ec.Q - 1, ec.Q - 19, (1 << 255) - 20, (1 << 254) + 12345] + [rng.randrange(ec.Q) for _ in range(6)]
    elems = [f.witness(v) for v in vals]
    for i in range(0, len(vals), 2):
        x, y, X, Y = vals[i], vals[i + 1], elems[i], elems[i + 1]
        Z = f.mul(X, Y)
        assert f.value(Z) == x * y % ec.Q and Z.bound == 1 << 24
        S, A = f.sub(X, Y), f.add(X, Y)
        assert f.value(S) % ec.Q == (x - y) % ec.Q and f.value(A) % ec.Q == (x + y) % ec.Q
        assert all(b.value(v) < S.bound for v in S.limbs) and all(b.value(v) < A.bound for v in A.limbs)
        W = f.mul(S, A)                                                   # loose operands
        assert f.value(W) == (x - y) * (x + y) % ec.Q
        f.assert_equal(W, f.sub(f.sqr(X), f.sqr(Y)))
        H = f.lincomb([], [X, Y])
        assert f.value(H) % ec.Q == (-x - y) % ec.Q
    with pytest.raises(ValueError):
        f.assert_equal(elems[0], elems[1])                                 # 0 != 1
    with pytest.raises(AssertionError):
        big = f.scale(f.scale(elems[2], 16), 16)                            # limbs up to 2^32: a product could leave the carry range
        f.mul(big, big)
    # canonical form and parity
    f.assert_le_const(elems[2].limbs, ec.Q - 1)
    with pytest.raises(ValueError):
        f.assert_le_const(f.witness(ec.Q).limbs, ec.Q - 1)
    assert b.value(f.parity(elems[3].limbs[0])) == (ec.Q - 19) & 1
    _gates_hold(b)
    # the C evaluator's product op (csrc/nnf25519.h) reproduces every hint the builder computed
    prog = b.program()
    inputs = [w for v in vals + [ec.Q] for w in ec.limbs_of(v)]
    assert len(inputs) == prog.n_inputs


def test_product_hints_by_the_c_evaluator():
    """witness op 14 (glp_witness_eval): remainder, quotient and carries of random, extreme and LOOSE products equal the builder's"""
    ec, rec = _mods()
    b = rec.CircuitBuilder(object(), n_wires=144)
    f = ec.NNF(b)
    rng = random.Random(7)
    vals = [0, 1, ec.Q - 1, (1 << 264) - 1, rng.randrange(1 << 264)] + [rng.randrange(ec.Q) for _ in range(5)]
    xs = [ec.Fq([b.var(w) for w in ec.limbs_of(v)], 1 << 24) for v in vals]
    for i in range(len(xs) - 1):
        f.mul(xs[i], xs[i + 1])
        f.mul(f.sub(xs[i], xs[i + 1]), f.add(xs[i], xs[i + 1]))
    prog = b.program()
    inputs = [w for v in vals for w in ec.limbs_of(v)]
    got = prog.evaluate(poseidon_consts("small"), inputs, threads=1)
    assert np.array_equal(got, np.array(b.values, dtype=np.uint64))
    bad = list(inputs)
    bad[3] = 1 << 29                                                       # a limb far out of range: no hint exists
    with pytest.raises(ValueError):
        prog.evaluate(poseidon_consts("small"), bad, threads=1)


@pytest.mark.parametrize("msg", [b"", b"abc", bytes(range(111)), bytes(range(112)), bytes(range(200))])
def test_sha512_gadget_matches_hashlib(msg):
    ec, rec = _mods()
    b = rec.CircuitBuilder(object(), n_wires=144)
    g = ec.Sha512Gadget(b)
    out = g.hash_bytes([[g.one if (byte >> i) & 1 else g.zero for i in range(8)] for byte in msg])
    assert bytes(sum(b.value(bit) << i for i, bit in enumerate(byte)) for byte in out) == hashlib.sha512(msg).digest()
    if len(msg) <= 3:
        _gates_hold(b)


def test_curve_formulas_against_affine_arithmetic():
    ec, rec = _mods()
    b = rec.CircuitBuilder(object(), n_wires=144)
    f = ec.NNF(b)
    ed = ec.Edwards(f)
    aff = lambda Pt: (f.value(Pt[0]) * pow(f.value(Pt[2]), ec.Q - 2, ec.Q) % ec.Q, f.value(Pt[1]) * pow(f.value(Pt[2]), ec.Q - 2, ec.Q) % ec.Q)
    P1, P2 = ec._ed_mul(123456789, ec.BASE), ec._ed_mul(987654321987654321, ec.BASE)
    ext = lambda Pt: (f.witness(Pt[0]), f.witness(Pt[1]), f.const(1), f.witness(Pt[0] * Pt[1] % ec.Q))
    E1, E2 = ext(P1), ext(P2)
    assert aff(ed.double(E1)) == ec._ed_add(P1, P1)
    assert aff(ed.add_niels(E1, ed.to_niels(E2))) == ec._ed_add(P1, P2)
    assert aff(ed.add_niels(E1, ed.to_niels(ed.identity()))) == P1                      # complete: the identity is an ordinary operand
    assert aff(ed.add_niels(E1, ed.to_niels(E1))) == ec._ed_add(P1, P1)                 # ... and so is the point itself
    D2 = ed.double(ed.double(E1, need_t=False))
    assert aff(D2) == ec._ed_mul(4, P1) and f.value(D2[3]) * f.value(D2[2]) % ec.Q == f.value(D2[0]) * f.value(D2[1]) % ec.Q
    # scalar multiplications: a fixed-base one and a variable-base one, on 8-bit scalars padded to the 64 windows
    bits = lambda v: [[b.constant((v >> (4 * w + i)) & 1) for i in range(4)] for w in range(64)]
    s = 0x1D3F
    assert aff(ed.mul_base(bits(s))) == ec._ed_mul(s, ec.BASE)
    assert aff(ed.mul_var(E2, bits(s))) == ec._ed_mul(s, P2)
    assert ec.fixed_base_tables()[3][5][:2] == ((lambda Pt: ((Pt[1] + Pt[0]) % ec.Q, (Pt[1] - Pt[0]) % ec.Q))(ec._ed_mul(5 * 16**3, ec.BASE)))


def test_half_size_scalar_pairs():
    """ed25519_circuit.half_size_pair: u odd, u k = +-v (mod L), both well below the 144 bits the circuit gives them — for random, tiny and
    extreme k; and the statement WITHOUT the split (one 253-bit scalar multiplication, round 3's first form) still decides a valid signature"""
    ec, rec = _mods()
    rng = random.Random(12)
    for k in [0, 1, 2, 1 << 126, (1 << 252) + 5] + [rng.randrange(ec.ELL) for _ in range(300)]:
        u, v, neg = ec.half_size_pair(k)
        assert u & 1 and 0 < u < 1 << 140 and 0 <= v < 1 << 140 and (u * k - (-v if neg else v)) % (8 * ec.ELL) == 0
    with pytest.raises(ValueError):
        # k = 1/2 mod L: 2 k = L + 1, the lattice's short vectors all have an EVEN first coordinate ((8, 4 + 4 L mod 8L) ...) and every odd one is huge.  Such k (probability ~2^-36 even when
        # ground for; a validator who grinds its nonce for one only makes its own signature unprovable: its slot is then flagged 0) have no
        # split form; split_scalars=False remains for them
        ec.half_size_pair((ec.ELL + 1) // 2)
    with pytest.raises(ValueError):
        ec.half_size_pair(ec.ELL - 1)                                       # 8 k = -8 (mod 8L): again only even short vectors
    msg = b"the unsplit form"
    pub, sig = ec.keypair_and_sign(bytes(32), msg)
    b = rec.CircuitBuilder(object(), n_wires=144)
    st = ec.verify_statement(b, pub, sig, msg, split_scalars=False)
    assert 2900 < st["stats"]["field_products"] < 3000
    bad = bytearray(sig)
    bad[1] ^= 4
    with pytest.raises(ValueError):
        ec.verify_statement(rec.CircuitBuilder(object(), n_wires=144), pub, bytes(bad), msg, split_scalars=False)


def _cases():
    with open(os.path.join(G, "ed25519.json")) as fh:
        return json.load(fh)["cases"]


def test_fixtures_valid_signatures_lay_down_invalid_ones_cannot():
    """RFC 8032 §7.1 TEST 1-3 and the OpenSSL fixtures: the statement's verdict is the fixture's"""
    ec, rec = _mods()
    cases = _cases()
    rfc = [c for c in cases if c["src"].startswith("rfc")]
    # (a circuit is ~3 s of Python builder: two RFC vectors — the empty message and a one-byte one —, one OpenSSL signature, and one invalid case of each kind the fixtures hold: tampered message / R / S, wrong key, non-reduced S)
    invalid, kinds = [], set()
    for c in cases:
        if not c["valid"] and len(bytes.fromhex(c["sig"])) == 64 and c["src"] not in kinds:
            kinds.add(c["src"])
            invalid.append(c)
    picked = rfc[:2] + [c for c in cases if c["valid"] and not c["src"].startswith("rfc")][:1] + invalid
    assert any(c["valid"] for c in picked) and any(not c["valid"] for c in picked)
    for c in picked:
        pub, msg, sig = bytes.fromhex(c["pub"]), bytes.fromhex(c["msg"]), bytes.fromhex(c["sig"])
        if len(sig) != 64:
            continue                                                         # (over-long / truncated encodings are refused before any circuit)
        try:
            b, st = ec.ed25519_circuit(object(), pub, sig, msg)
            ok = True
        except ValueError:
            ok = False
        assert ok == c["valid"], c["src"]
        if ok:
            assert [b.value(v) for v in st["key_words"]] + [b.value(v) for v in st["msg_bytes"]] == ec.public_inputs(pub, msg)
            assert 2200 < st["stats"]["field_products"] < 2400


def test_recorded_circuit_replays_other_signatures_and_refuses_forgeries():
    ec, rec = _mods()
    msg1, msg2 = b"vote: block 4000000 round 0, validator 17".ljust(48, b"."), b"vote: block 4000001 round 0, validator 99".ljust(48, b".")
    pub1, sig1 = ec.keypair_and_sign(bytes(range(32)), msg1)
    pub2, sig2 = ec.keypair_and_sign(hashlib.sha256(b"other").digest(), msg2)
    b, st = ec.ed25519_circuit(object(), pub1, sig1, msg1)
    prog = b.program()
    assert prog.stats["rows"] == 1 << 16                                        # half-size scalars + every wire routed: one signature in 2^16 rows
    consts = poseidon_consts("small")
    vals = prog.evaluate(consts, ec.witness_inputs(pub1, sig1, msg1), threads=1)
    assert np.array_equal(vals, np.array(b.values, dtype=np.uint64))
    vals2 = prog.evaluate(consts, ec.witness_inputs(pub2, sig2, msg2), threads=1)
    assert [int(vals2[v]) for v in prog.public_vars] == ec.public_inputs(pub2, msg2)
    forged = bytearray(sig2)
    forged[40] ^= 1                                                          # another S
    with pytest.raises(ValueError):
        prog.evaluate(consts, ec.witness_inputs(pub2, bytes(forged), msg2), threads=1)
    with pytest.raises(ValueError):                                          # the right signature for another message
        prog.evaluate(consts, ec.witness_inputs(pub2, sig2, msg1), threads=1)
    with pytest.raises(ValueError):                                          # ... or under another key
        prog.evaluate(consts, ec.witness_inputs(pub1, sig2, msg2), threads=1)


@pytest.mark.gpu
def test_one_signature_proved_and_verified(prover, oracle, pkg):
    ec, rec = _mods()
    consts = poseidon_consts("small")
    prover.set_poseidon_constants(*consts)
    oracle.orc_poseidon_set_constants(*(ptr(a) for a in consts))
    msg = b"canonical vote sign bytes stand-in, 64 bytes long ..............."[:64]
    pub, sig = ec.keypair_and_sign(hashlib.sha256(b"validator 0").digest(), msg)
    b, st = ec.ed25519_circuit(prover, pub, sig, msg)
    ck, dw, public = b.build()
    assert ck.log_n == 16 and public == ec.public_inputs(pub, msg)
    proof = ck.prove_(dw, 28, 16, public=public)
    assert ck.verify(proof, 28, 16, public=public), prover.last_reject
    pref.verify_plonk(proof, oracle, pos_consts=consts, public=public)
    other = list(public)
    other[0] ^= 1                                                            # another key
    assert not ck.verify(proof, 28, 16, public=other)
    # the recorded program replays another validator's signature on the device; a forged one is refused before any proof
    prog = b.program()
    pub2, sig2 = ec.keypair_and_sign(hashlib.sha256(b"validator 1").digest(), msg)
    vals = prog.evaluate(consts, ec.witness_inputs(pub2, sig2, msg))
    dw2, public2 = prog.device_witness(prover, vals)
    proof2 = ck.prove_(dw2, 28, 16, public=public2)
    assert public2 == ec.public_inputs(pub2, msg) and ck.verify(proof2, 28, 16, public=public2)
    assert not ck.verify(proof2, 28, 16, public=public)
    bad = bytearray(sig2)
    bad[5] ^= 0x10
    with pytest.raises(ValueError):
        prog.evaluate(consts, ec.witness_inputs(pub2, bytes(bad), msg))
    dw.free()
    dw2.free()
    ck.free()
