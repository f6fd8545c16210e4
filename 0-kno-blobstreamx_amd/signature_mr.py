"""The Ed25519 half of a light-client statement as a MapReduce of proofs (VERDICT r2 "missing" 1;
upstream names recalled, unverified — reference file:line NONE, the mount is empty: tendermintx ``verify_signatures`` over curta's EdDSA
accelerator).

Map      one leaf per validator SLOT of the target set: "flag = 1  =>  the slot's key signed these vote bytes" (ed25519_circuit.verify_statement
         with a flag, half-size scalars, all 144 wires routed: 2 312 non-native field products, 61.6k rows of a 2^16-row circuit; a slot with flag 0
         verifies a fixed dummy triple).
         Public inputs: the key's 8 big-endian words, the flag, and the 8 words of the BLOCK HASH the vote bytes carry at `hash_offset`.
Reduce   a node verifies `fan_in` children in-circuit, requires that all of them vote for the same block hash, and folds their signer-digest
         leaves (gadgets.signer_leaf: Poseidon over key words and flag) into the binary Poseidon tree gadgets._signer_digest defines.
         Public inputs of every node: block hash (8 words), subtree digest (4 words).
The root therefore says: "every slot flagged in the signer digest D carries a valid Ed25519 signature, under its key, of vote bytes naming block
hash H".  combined_skip_mr.py verifies that root inside the CombinedSkip outer circuit and equates (H, D) with the target header hash and the
signer digest its voting-power rules were computed from — the signatures are then part of the proof, not a native side check.
Slots are padded to a power of two with all-zero keys and flag 0 (the padding gadgets._signer_digest uses).  Vote bytes are opaque except for
the block hash (build-defined stand-in for the canonical vote encoding: fixed length, hash at a fixed offset).  Everything here is build-defined."""
import importlib
import struct
import time

import numpy as np

from . import SHA_GATE_WIRES
from .data_commitment_mr import DataCommitmentMapReduce
from .ed25519_circuit import keypair_and_sign, verify_statement, witness_inputs
from .gadgets import signer_leaf, signer_tree
from .recursion import CircuitBuilder

LEAF_PUBLIC = 17


class SignatureSetMapReduce(DataCommitmentMapReduce):
    N_PUBLIC = 12                                   # nodes: block hash (8), signer-digest subtree (4)

    def __init__(self, prover, poseidon_consts, msg_len=112, hash_offset=16, fan_in=8, num_queries=28, pow_bits=16, map_provers=()):
        if hash_offset + 32 > msg_len or hash_offset % 4 or msg_len % 4:
            raise ValueError("the vote bytes must hold a 32-byte block hash at a word-aligned offset")
        super().__init__(prover, poseidon_consts, leaf_blocks=1, fan_in=fan_in, num_queries=num_queries, pow_bits=pow_bits, map_provers=map_provers)
        self.msg_len, self.hash_offset = int(msg_len), int(hash_offset)

    def _child_n_public(self, level):
        return LEAF_PUBLIC if level == 1 else self.N_PUBLIC

    def _child_has_poseidon_rows(self, level):
        return level > 1                             # a signature leaf is arithmetic gates and ADD rows only

    def _child_n_routed(self, level):
        return 144 if level == 1 else 80             # the signature leaf routes all 144 wires (36 gate slots per row): 61.6k rows, a 2^16-row circuit

    def _child_has_sha_rows(self, level):
        return level == 1                            # ... and the nodes above have Poseidon rows and arithmetic only

    def _combine_for(self, span):
        leaf_children = span == 1

        def combine(b, outs):
            if leaf_children:
                hashes = [o["public"][9:17] for o in outs]
                leaves = [signer_leaf(b, o["public"][:8], o["public"][8]) for o in outs]
            else:
                hashes = [o["public"][:8] for o in outs]
                leaves = [o["public"][8:12] for o in outs]
            for other in hashes[1:]:
                for x, y in zip(hashes[0], other):
                    b.assert_equal(x, y)                                      # every slot's vote names the SAME block
            return hashes[0] + signer_tree(b, leaves)
        return combine

    # ---- Map ------------------------------------------------------------------------------------------------------------------------------
    def vote_bytes(self, block_hash, slot=0):
        """build-defined stand-in for a validator's canonical vote sign-bytes: fixed length, the block hash at hash_offset, the rest slot-specific"""
        body = bytearray((17 * slot + 3 * k) & 0xFF for k in range(self.msg_len))
        body[self.hash_offset:self.hash_offset + 32] = bytes(block_hash)
        return bytes(body)

    def _record_leaf(self):
        t0 = time.perf_counter()
        msg = self.vote_bytes(bytes(32))
        pub, sig = keypair_and_sign(bytes(32), msg)
        b = CircuitBuilder(self.prover, n_wires=SHA_GATE_WIRES, n_routed=SHA_GATE_WIRES)
        st = verify_statement(b, pub, sig, msg, flag=True)
        one = b.constant(1)
        hb = st["msg_bytes"][self.hash_offset:self.hash_offset + 32]
        words = []
        for k in range(0, 32, 4):                                              # big-endian words of the block hash bytes
            hi = b.arith(1 << 24, 1, 0, hb[k], one, b.arith(1 << 16, 0, 0, hb[k + 1], one, hb[k + 1]))
            words.append(b.arith(1 << 8, 1, 0, hb[k + 2], one, b.arith(1, 1, 0, hi, one, hb[k + 3])))
        for v in st["key_words"] + [st["flag"]] + words:
            b.public_input(v)
        self.leaf_program = b.program()
        self.leaf_circuit = self.leaf_program.setup(self.prover)
        self.map_circuits = [self.leaf_program.setup(p) for p in self.map_provers]
        self.leaf_stats = dict(self.leaf_program.stats, field_products=st["stats"]["field_products"])
        self.record_seconds["leaf"] = round(time.perf_counter() - t0, 3)

    def prove_leaf(self, pubkey, signature, msg, flag, which=0):
        """(proof, public) for one slot; a slot with flag 0 needs no signature (None)"""
        if self.leaf_program is None:
            self._record_leaf()
        if len(bytes(msg)) != self.msg_len or len(bytes(pubkey)) != 32:
            raise ValueError("vote bytes / key of another length than this circuit was recorded for")
        prover, circuit = (self.prover, self.leaf_circuit) if which == 0 else (self.map_provers[which - 1], self.map_circuits[which - 1])
        inputs = witness_inputs(pubkey, signature if flag else bytes(64), msg, bool(flag))
        vals = self.leaf_program.evaluate(self.consts, inputs, threads=1)
        dw, public = self.leaf_program.device_witness(prover, vals, reuse=True)
        return circuit.prove_(dw, self.nq, self.pw, public=public), public

    def _slots(self, pubkeys, signatures, msgs, flags, total=None):
        n = len(pubkeys)
        if not (len(signatures) == len(msgs) == len(flags) == n) or n < 1:
            raise ValueError("one signature (or None), vote and flag per validator")
        total = (1 << max(0, (n - 1).bit_length())) if total is None else total
        pad = total - n
        return (list(pubkeys) + [bytes(32)] * pad, list(signatures) + [None] * pad, list(msgs) + [msgs[0]] * pad, [bool(f) for f in flags] + [False] * pad)

    def _map(self, slots, lo, hi):
        """leaf proofs of slots [lo, hi).  The per-signature hints of the circuit witness — the decoded x coordinates of A and R and
        k = SHA-512(R || A || M) mod L — come from ONE launch of the GPU witness kernel over the slots' verified triples (glp_ed25519_witness, the
        kernel north_star names for this job); a flagged slot the kernel rejects is refused here, before any witness program runs."""
        from .ed25519_circuit import dummy_signature
        for i in range(lo, hi):
            if len(bytes(slots[2][i])) != self.msg_len or len(bytes(slots[0][i])) != 32:
                raise ValueError("vote bytes / key of another length than this circuit was recorded for")
        d_pub, d_sig, d_msg = dummy_signature(self.msg_len)
        triples = [(slots[0][i], slots[1][i], slots[2][i]) if slots[3][i] else (d_pub, d_sig, d_msg) for i in range(lo, hi)]
        if any(t[1] is None or len(bytes(t[1])) != 64 for t in triples):
            raise ValueError("a flagged slot has no 64-byte signature")
        recs = self.prover.ed25519_witness([bytes(t[0]) for t in triples], [bytes(t[1]) for t in triples], [bytes(t[2]) for t in triples])
        bad = [lo + j for j in range(len(triples)) if not int(recs[j][0])]
        if bad:
            raise ValueError(f"the signatures of slots {bad[:8]} do not verify (GPU witness kernel)")
        return self._map_inputs([witness_inputs(slots[0][i], slots[1][i] if slots[3][i] else bytes(64), slots[2][i], slots[3][i], record=recs[i - lo])
                                 for i in range(lo, hi)])

    def prove_set(self, pubkeys, signatures, msgs, flags):
        """one proof for a validator set's signatures: public = block hash (8 words), signer digest (4 words).  signatures[i] may be None where
        flags[i] is false.  ValueError when a flagged slot's signature does not verify or the votes name different blocks."""
        if self.leaf_program is None:
            self._record_leaf()
        n, F = len(pubkeys), self.fan_in
        groups, full = -(-n // F), 1 << max(0, (n - 1).bit_length())
        # Padding to a power of two can cost a lot of whole leaves (100 validators -> 128 slots: 28 dummy verifications).  When the level-1
        # groups fit ONE root (<= 16), only the last group is padded and the root takes the real groups' node proofs plus CONSTANT digests for
        # the all-padding groups: the signer digest is the same 128-slot tree, 24 leaves and 3 nodes are never proved.
        trimmed = groups * F < full and 2 <= groups <= 16 and full % F == 0 and n > F
        slots = self._slots(pubkeys, signatures, msgs, flags, total=groups * F if trimmed else None)
        t0 = time.perf_counter()
        leaves = self._map(slots, 0, len(slots[0]))
        t1 = time.perf_counter()
        levels = []
        if len(leaves) == 1:
            raise ValueError("a set of one slot has no node to fold it: use at least two validators")
        if trimmed:
            nodes, _, key1, lvl = self.reduce(leaves, levels, max_levels=1)
            root, public, key = self._padded_root(nodes, key1, lvl, full // F, levels)
        else:
            root, public, key, _ = self.reduce(leaves, levels)
        t2 = time.perf_counter()
        return {"root_proof": root, "public": public, "key": key, "slots": len(leaves), "map_seconds": round(t1 - t0, 4),
                "reduce_seconds": round(t2 - t1, 4), "levels": levels, "record_seconds": dict(self.record_seconds),
                "block_hash": b"".join(struct.pack(">I", v) for v in public[:8]), "signer_digest": public[8:12]}

    def _padded_root(self, nodes, child_key, level, total_groups, timings=None):
        """ONE root over len(nodes) level-1 node proofs and (total_groups - len(nodes)) all-padding groups, whose subtree digest is a constant of
        the circuit (computed in it from the padding leaf): the same statement and digest as the power-of-two tree"""
        vc = importlib.import_module(__package__ + ".verifier_circuit")
        k = ("padded_root", len(nodes), total_groups, bytes(np.ascontiguousarray(child_key, dtype=np.uint64)))
        t0 = time.perf_counter()
        if k not in self.nodes:
            F = self.fan_in

            def combine(b, outs):
                hashes = [o["public"][:8] for o in outs]
                for other in hashes[1:]:
                    for x, y in zip(hashes[0], other):
                        b.assert_equal(x, y)
                zero = b.constant(0)
                pad_group = signer_tree(b, [signer_leaf(b, [zero] * 8, zero)] * F)
                return hashes[0] + signer_tree(b, [o["public"][8:12] for o in outs] + [pad_group] * (total_groups - len(outs)))
            spec = dict(leaf_key=child_key, n_public=self.N_PUBLIC, child_is_recursion=True, child_sha=False)
            self.nodes[k] = vc.RecursionProgram(self.prover, nodes, child_key, self.nq, self.pw, SHA_GATE_WIRES, self.consts, n_routed=80,
                                                n_public=self.N_PUBLIC, cap_height=1, child_is_recursion=True, child_sha=False, combine=combine,
                                                builder_wires=SHA_GATE_WIRES, specs=[spec] * len(nodes))
            self.record_seconds[f"padded_root_{len(nodes)}_of_{total_groups}"] = round(time.perf_counter() - t0, 3)
        rp = self.nodes[k]
        proof, public = rp.prove(nodes, self.nq, self.pw)
        if timings is not None:
            timings.append({"level": level, "nodes": 1, "fan_in": len(nodes), "constant_groups": total_groups - len(nodes), "rows": rp.stats["rows"],
                            "seconds_including_first_recording": round(time.perf_counter() - t0, 4)})
        return proof, public, rp.key()

    def prove_set_distributed(self, pubkeys, signatures, msgs, flags, device=None, comm=None):
        """prove_set with the slots spread over the ranks: rank r proves and folds the r-th contiguous part, ONE all-gather of node proofs, rank 0
        folds the root (mapreduce.reduce_tree_distributed).  Every rank passes the whole set.  Returns the dict on rank 0 (root_proof None elsewhere)."""
        mrm = importlib.import_module(__package__ + ".mapreduce")
        rank, world = mrm._world(comm)
        if self.leaf_program is None:
            self._record_leaf()
        slots = self._slots(pubkeys, signatures, msgs, flags)
        total = len(slots[0])
        if total % world or total // world < 2:
            raise ValueError("the padded slot count does not split into at least two slots per rank")
        per = total // world
        state, levels = {}, []

        def fold_local(_):
            t0 = time.perf_counter()
            leaves = self._map(slots, rank * per, (rank + 1) * per)
            state["map_seconds"] = round(time.perf_counter() - t0, 4)
            proof, public, key, level = self.reduce(leaves, levels)
            state.update(key=key, level=level, public=public, span=self.last_span)
            return proof

        def fold_root(nodes):
            proof, public, key, _ = self.reduce(nodes, levels, child_key=state["key"], level=state["level"], span=state["span"])
            state.update(key=key, public=public)
            return proof
        t0 = time.perf_counter()
        out = mrm.reduce_tree_distributed(fold_local, fold_root, [b""] * per, 1 << 18, device=device, comm=comm)
        root, public = out["root_proof"], state["public"]
        return {"root_proof": root, "public": public if root is not None else None, "key": state["key"] if root is not None else None, "slots": total,
                "ranks": world, "map_seconds": state["map_seconds"], "seconds": round(time.perf_counter() - t0, 4), "levels": levels,
                "record_seconds": dict(self.record_seconds),
                "block_hash": b"".join(struct.pack(">I", v) for v in public[:8]) if root is not None else None,
                "signer_digest": public[8:12] if root is not None else None}

    def expected_key(self, n_validators):
        """the verifier's own setup: the root circuit's key for sets padded to the same slot count, from a synthetic set (all flags 0)"""
        h = bytes(32)
        msgs = [self.vote_bytes(h, i) for i in range(n_validators)]
        return self.prove_set([bytes(32)] * n_validators, [None] * n_validators, msgs, [False] * n_validators)["key"]

    def verify_set(self, root_proof, key, block_hash, signer_digest):
        public = list(struct.unpack(">8I", bytes(block_hash))) + [int(v) for v in signer_digest]
        return bool(self.prover.plonk_verify(root_proof, key, self.nq, self.pw, public=public))
