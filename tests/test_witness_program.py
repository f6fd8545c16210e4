"""The recorded witness program (recursion.WitnessProgram + glp_witness_eval, host C++): a circuit laid down once by the Python builder is
re-evaluated for new inputs without the builder.  CPU: the C evaluator reproduces every variable the builder computed for the golden proofs'
verifier circuits (Poseidon through the library's host permutation vs the oracle's), refuses tampered inputs, and the SHA-256 gadget's program
hashes NEW messages.  GPU: a recursion circuit recorded from one batch of leaf proofs proves other batches."""
import hashlib
import importlib
import json
import os
import struct
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
    return tuple(importlib.import_module(graft.PKG_NAME + m) for m in (".recursion", ".verifier_circuit", ".gadgets", ".mapreduce"))


def _oracle_prover(oracle):
    class OracleProver:
        def poseidon_permute(self, states):
            s = np.ascontiguousarray(states, dtype=np.uint64).copy()
            for i in range(s.shape[0]):
                row = s[i].copy()
                oracle.orc_poseidon_permute(ptr(row))
                s[i] = row
            return s
    return OracleProver()


@pytest.mark.parametrize("which", ["plonk", "gates", "sha"])
def test_c_evaluator_reproduces_the_builders_witness(oracle, which):
    rec, vc, _, _ = _mods()
    consts = poseidon_consts("small")
    oracle.orc_poseidon_set_constants(*(ptr(a) for a in consts))
    with open(os.path.join(G, "proofs.json")) as f:
        g = json.load(f)[which]
    proof = bytes.fromhex(g["proof"])
    b = rec.CircuitBuilder(_oracle_prover(oracle))
    kw = dict(n_routed=g.get("R"), n_public=g.get("n_public", 0), poseidon_consts=consts if which != "plonk" else None, sha=which == "sha")
    vc.verify_in_circuit(b, proof, g["circuit_cap"], g["queries"], g["pow_bits"], g["W"], **kw)
    prog = b.program()
    inputs, ws = prog.inputs_from_words([proof])
    vals = prog.evaluate(consts, inputs)
    assert np.array_equal(vals, np.array(b.values, dtype=np.uint64)), "the C evaluator and the Python builder disagree on a variable"
    prog.check_words(vals, ws)
    # tampered proofs: a flipped input word breaks a copy constraint, a flipped constant word is caught before evaluation
    w = np.frombuffer(proof, dtype="<u8").copy()
    refused = 0
    targets = list(range(0, len(w), max(1, len(w) // 50)))
    for t in targets:
        bad = w.copy()
        bad[t] ^= np.uint64(1)
        try:
            i2, ws2 = prog.inputs_from_words([bad.tobytes()])
            prog.check_words(prog.evaluate(consts, i2), ws2)
        except ValueError:
            refused += 1
    assert refused == len(targets)
    assert prog.stats["inputs"] == len(inputs) and prog.stats["variables"] == len(vals)


@pytest.mark.parametrize("which", ["plonk", "sha"])
def test_verifier_circuit_on_extension_rows(oracle, which):
    """the same verifier laid down with extension-arithmetic rows (ext_gate=True): fewer arithmetic gates (about a tenth: the verifier's
    arithmetic is mostly additions and scalings, which an arithmetic gate already does in 8 wires per extension element), the C evaluator
    (op EXTMULADD) reproduces the builder variable for variable, tampered proofs are refused"""
    rec, vc, _, _ = _mods()
    consts = poseidon_consts("small")
    oracle.orc_poseidon_set_constants(*(ptr(a) for a in consts))
    with open(os.path.join(G, "proofs.json")) as f:
        g = json.load(f)[which]
    proof = bytes.fromhex(g["proof"])
    kw = dict(n_routed=g.get("R"), n_public=g.get("n_public", 0), poseidon_consts=consts if which != "plonk" else None, sha=which == "sha")
    plain = rec.CircuitBuilder(_oracle_prover(oracle))
    vc.verify_in_circuit(plain, proof, g["circuit_cap"], g["queries"], g["pow_bits"], g["W"], **kw)
    b = rec.CircuitBuilder(_oracle_prover(oracle), ext_gate=True)
    vc.verify_in_circuit(b, proof, g["circuit_cap"], g["queries"], g["pow_bits"], g["W"], **kw)
    prog = b.program()
    n_plain = sum(len(r) for rows in plain.arith_rows.values() for r in rows)
    assert prog.stats["ext_rows"] > 0 and prog.has_ext and prog.consts.shape[0] == 7
    assert prog.stats["arith_gates"] < 0.95 * n_plain
    inputs, ws = prog.inputs_from_words([proof])
    vals = prog.evaluate(consts, inputs)
    assert np.array_equal(vals, np.array(b.values, dtype=np.uint64))
    w = np.frombuffer(proof, dtype="<u8").copy()
    for t in range(9, len(w), max(1, len(w) // 25)):
        bad = w.copy()
        bad[t] ^= np.uint64(1)
        with pytest.raises(ValueError):
            i2, ws2 = prog.inputs_from_words([bad.tobytes()])
            prog.check_words(prog.evaluate(consts, i2), ws2)


def test_segments_evaluate_in_parallel_and_false_independence_is_refused(oracle, tmp_path):
    """two proofs' verifier sub-circuits recorded as independent segments: glp_witness_eval_mt on 4 threads reproduces the builder's values
    (the constants both segments read were hoisted into the prefix); a segment that reads the other segment's variable is refused"""
    rec, vc, _, _ = _mods()
    consts = poseidon_consts("small")
    oracle.orc_poseidon_set_constants(*(ptr(a) for a in consts))
    with open(os.path.join(G, "proofs.json")) as f:
        g = json.load(f)["plonk"]
    proof = bytes.fromhex(g["proof"])
    b = rec.CircuitBuilder(_oracle_prover(oracle))
    digests = []
    for k in range(3):
        b.begin_segment()
        digests.append(vc.verify_in_circuit(b, proof, g["circuit_cap"], g["queries"], g["pow_bits"], g["W"], proof_id=k)["digest"])
        b.end_segment()
    root = b.two_to_one(b.two_to_one(digests[0], digests[1])[:4], digests[2])            # the tail reads every segment
    prog = b.program()
    assert prog.seg_bounds is not None and prog.seg_bounds.size == 4
    inputs, ws = prog.inputs_from_words([proof] * 3)
    want = np.array(b.values, dtype=np.uint64)
    for threads in (1, 4):
        assert np.array_equal(prog.evaluate(consts, inputs, threads=threads), want)
    # the recording survives a round trip through its file form (plain arrays, loaded without pickle)
    prog.save(str(tmp_path / "rec.npz"))
    again = rec.WitnessProgram.load(str(tmp_path / "rec.npz"))
    i2, ws2 = again.inputs_from_words([proof] * 3)
    assert np.array_equal(i2, inputs) and np.array_equal(again.evaluate(consts, i2, threads=2), want)
    again.check_words(want, ws2)
    assert np.array_equal(again.cell_index, prog.cell_index) and np.array_equal(again.consts, prog.consts) and again.stats == prog.stats
    assert [int(want[v]) for v in root] == [b.value(v) for v in root]
    # a false claim: segment 2 multiplies a variable that segment 1 computed
    b2 = rec.CircuitBuilder(_oracle_prover(oracle))
    b2.begin_segment()
    x = b2.var(3, tag=(0, 0))
    y = b2.mul(x, x)
    b2.end_segment()
    b2.begin_segment()
    z = b2.var(5, tag=(0, 1))
    b2.mul(z, y)
    b2.end_segment()
    b2.begin_segment()
    b2.mul(b2.var(7, tag=(0, 2)), b2.constant(9))
    b2.end_segment()
    p2 = b2.program()
    with pytest.raises(ValueError):
        p2.evaluate(consts, [3, 5, 7], threads=2)
    assert int(p2.evaluate(consts, [3, 5, 7], threads=1)[y]) == 9                          # serially the same program is fine
    with pytest.raises(ValueError):
        b2.begin_segment(); b2.end_segment(); b2.var(1); b2.begin_segment()                 # a gap between segments


def test_sha256_program_hashes_new_messages(oracle):
    """the SHA-256 gadget recorded for one 64-byte message, replayed by the C evaluator for others"""
    rec, _, gd, _ = _mods()
    consts = poseidon_consts("small")
    b = rec.CircuitBuilder(object())
    g = gd.Sha256Gadget(b)
    first = bytes(range(64))
    words = [g.public_word(struct.unpack(">I", first[4 * k: 4 * k + 4])[0]) for k in range(16)]
    state = g.hash_bits([bit for w in words for bit in g.word_bits_be(w)])
    digest_vars = [w[1] for w in state]
    assert b"".join(struct.pack(">I", b.value(v)) for v in digest_vars) == hashlib.sha256(first).digest()
    prog = b.program()
    rng = np.random.default_rng(5)
    for _ in range(3):
        msg = rng.integers(0, 256, 64, dtype=np.uint8).tobytes()
        vals = prog.evaluate(consts, [struct.unpack(">I", msg[4 * k: 4 * k + 4])[0] for k in range(16)])
        assert b"".join(struct.pack(">I", int(vals[v])) for v in digest_vars) == hashlib.sha256(msg).digest()
    with pytest.raises(ValueError):
        prog.evaluate(consts, [1 << 33] + [0] * 15)                 # a "word" that is not 32 bits: the range check fails


def _witness_for(circ, rng):
    """another satisfying witness of a copy-free arithmetic circuit: fresh inputs, outputs by the gate equation"""
    q, c0, c1, c2 = ([int(v) for v in circ["consts"][k]] for k in range(4))
    W, n = circ["W"], 1 << circ["log_n"]
    w = [[int(rng.integers(0, 1 << 62)) * 4 % P for _ in range(n)] for _ in range(W)]
    for i in range(n):
        if q[i]:
            for gidx in range(W // 4):
                x, y, z = w[4 * gidx][i], w[4 * gidx + 1][i], w[4 * gidx + 2][i]
                w[4 * gidx + 3][i] = (c0[i] * x * y + c1[i] * z + c2[i]) % P
    return np.array(w, dtype=np.uint64)


@pytest.mark.gpu
def test_recursion_program_recorded_once_proves_other_batches(prover, oracle, pkg, tmp_path):
    """the recursion circuit is laid down ONCE (Python builder) from a sample batch of leaf proofs; other batches — proofs of the same leaf
    circuit for other witnesses — are proved through the recorded program (C evaluator + GPU), with the same key; a bad proof is refused"""
    rec, vc, _, mr = _mods()
    consts = poseidon_consts("small")
    prover.set_poseidon_constants(*consts)
    oracle.orc_poseidon_set_constants(*(ptr(a) for a in consts))
    rng = np.random.default_rng(77)
    circ = pref.build_circuit(rng, 9, 16, copy_prob=0.0)
    ck = pkg.PlonkCircuit(prover, circ["consts"], circ["sigmas"])
    nq, pw = 6, 4
    batches = [[ck.prove(_witness_for(circ, rng), nq, pw) for _ in range(2)] for _ in range(3)]
    assert len({p for bt in batches for p in bt}) == 6 and all(ck.verify(p, nq, pw) for bt in batches for p in bt)
    rp = vc.RecursionProgram(prover, batches[0], ck.cap(), nq, pw, 16, consts)
    for bt in batches:
        proof, public = rp.prove(bt, 8, 4)
        digests = [prover.proof_digest(p) for p in bt]
        assert public == [v for d in digests for v in d] + rec.merkle_root_host(prover, digests)
        assert prover.plonk_verify(proof, rp.key(), 8, 4, public=public), prover.last_reject
    pref.verify_plonk(proof, oracle, pos_consts=consts, public=public)
    # the recorded circuit is the circuit a fresh build would make
    ck2, dw2, pub2, _ = vc.recursive_aggregation_circuit(prover, batches[2], ck.cap(), nq, pw, 16)
    assert np.array_equal(ck2.cap(), rp.key()) and pub2 == public
    dw2.free()
    ck2.free()
    bad = list(batches[1])
    ww = np.frombuffer(bad[1], dtype="<u8").copy()
    ww[len(ww) // 3] ^= np.uint64(1)
    bad[1] = ww.tobytes()
    with pytest.raises(ValueError):
        rp.prove(bad, 8, 4)
    with pytest.raises(ValueError):
        rp.prove(batches[1][:1] + [batches[1][1][:-8]], 8, 4)
    # build once, load elsewhere: the saved recording commits to the same key and proves
    rp.save(str(tmp_path / "rp.npz"))
    rp2 = vc.RecursionProgram.load(prover, str(tmp_path / "rp.npz"), consts)
    assert np.array_equal(rp2.key(), rp.key())
    proof2, public2 = rp2.prove(batches[1], 8, 4)
    assert prover.plonk_verify(proof2, rp.key(), 8, 4, public=public2), prover.last_reject
    rp2.free()
    rp.free()
    ck.free()


@pytest.mark.parametrize("which,ext", [("plonk", False), ("gates", False), ("sha", True)])
def test_cloned_verifier_segments_equal_direct_ones(oracle, which, ext):
    """CircuitBuilder.clone_segment (what RecursionProgram records its 2nd..Nth child with): three verifier sub-circuits laid down through the gadget
    code vs ONE laid down and cloned twice — the same constants, the same rows, the same copy classes over the same cells (hence the same circuit
    key), the same wire values for new inputs, the same refusals."""
    rec, vc, _, _ = _mods()
    consts = poseidon_consts("small")
    oracle.orc_poseidon_set_constants(*(ptr(a) for a in consts))
    with open(os.path.join(G, "proofs.json")) as f:
        g = json.load(f)[which]
    proof = bytes.fromhex(g["proof"])
    kw = dict(n_routed=g.get("R"), n_public=g.get("n_public", 0), poseidon_consts=consts if which != "plonk" else None, sha=which == "sha")
    args = (g["circuit_cap"], g["queries"], g["pow_bits"], g["W"])
    direct = rec.CircuitBuilder(_oracle_prover(oracle), ext_gate=ext)
    outs_d = []
    for k in range(3):
        direct.begin_segment()
        outs_d.append(vc.verify_in_circuit(direct, proof, *args, proof_id=k, **kw))
        direct.end_segment()
    cloned = rec.CircuitBuilder(_oracle_prover(oracle), ext_gate=ext)
    cloned.begin_segment()
    m0 = cloned.mark()
    out0 = vc.verify_in_circuit(cloned, proof, *args, proof_id=0, **kw)
    m1 = cloned.mark()
    cloned.end_segment()
    words = lambda lid: np.frombuffer(proof, dtype="<u8")
    got = cloned.clone_segment(m0, m1, out0, [({0: 1}, words), ({0: 2}, words)])
    assert got is not None, "the verifier sub-circuit uses something clone_segment does not copy"
    cloned.fill_values(consts)
    outs_c = [out0] + got
    for od, oc in zip(outs_d, outs_c):
        for key in ("public", "digest"):
            assert [direct.value(v) for v in od[key]] == [cloned.value(v) for v in oc[key]]
    for b in (direct, cloned):                       # a statement after the segments (reads the values fill_values wrote)
        for v in b.two_to_one(outs_d[0]["digest"] if b is direct else outs_c[0]["digest"], outs_d[2]["digest"] if b is direct else outs_c[2]["digest"]):
            b.public_input(v)
    pd, pc = direct.program(), cloned.program()
    assert pd.log_n == pc.log_n and {k: v for k, v in pd.stats.items() if k != "variables"} == {k: v for k, v in pc.stats.items() if k != "variables"}
    assert np.array_equal(pd.consts, pc.consts)
    assert np.array_equal(pd.cell_index >= pd.n_values, pc.cell_index >= pc.n_values)      # empty cells and cells holding a fixed constant
    assert np.array_equal(pd.fixed, pc.fixed)
    used_d, used_c = pd.cell_index < pd.n_values, pc.cell_index < pc.n_values
    cls_d, cls_c = pd.roots[pd.cell_index[used_d].astype(np.int64)], pc.roots[pc.cell_index[used_c].astype(np.int64)]
    pairs = np.unique(np.stack([cls_d, cls_c], axis=1), axis=0)
    assert np.unique(cls_d).size == np.unique(cls_c).size == pairs.shape[0], "the copy classes over the cells differ"
    assert np.array_equal(pd.input_tags, pc.input_tags) and np.array_equal(pd.wc_const, pc.wc_const) and np.array_equal(pd.wc_bits, pc.wc_bits)
    # new inputs through both programs: the same wire matrix
    wires = []
    for p in (pd, pc):
        inputs, ws = p.inputs_from_words([proof] * 3)
        vals = p.evaluate(consts, inputs)
        p.check_words(vals, ws)
        full = np.concatenate([vals, p.fixed_values, np.zeros(1, dtype=np.uint64)])
        idx = np.where(p.cell_index == 0xFFFFFFFF, full.size - 1, p.cell_index).astype(np.int64)
        wires.append(full[idx])
    assert np.array_equal(wires[0], wires[1])
    bad = np.frombuffer(proof, dtype="<u8").copy()
    bad[len(bad) // 2] ^= np.uint64(1)
    for p in (pd, pc):
        with pytest.raises(ValueError):
            i2, ws2 = p.inputs_from_words([proof, proof, bad.tobytes()])
            p.check_words(p.evaluate(consts, i2), ws2)
    # a clone whose inputs do not verify: refused by fill_values, as the gadget code refuses to lay such a proof down
    c2 = rec.CircuitBuilder(_oracle_prover(oracle), ext_gate=ext)
    c2.begin_segment()
    m0 = c2.mark()
    o0 = vc.verify_in_circuit(c2, proof, *args, proof_id=0, **kw)
    m1 = c2.mark()
    c2.end_segment()
    assert c2.clone_segment(m0, m1, o0, [({0: 1}, lambda lid: bad)]) is not None
    with pytest.raises(ValueError):
        c2.fill_values(consts)
