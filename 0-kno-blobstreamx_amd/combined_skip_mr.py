"""CombinedSkip as a MapReduce of proofs (SURVEY.md §3 call stack (B), §8a row a11, BASELINE configs[2] "skip=128, MapReduce batch=8" and configs[3]
"skip=1024"; upstream names recalled, unverified — reference file:line NONE, the mount is empty: blobstreamx ``CombinedSkipCircuit`` = tendermintx
``verify_skip`` + the data commitment of the skipped range proved with plonky2x ``mapreduce`` over header batches).

Shape (build-defined statement, NOT upstream's circuit):
  Map      a leaf takes `batch` = 8 consecutive headers after a header whose hash it is given (data_commitment_mr.HeaderChainMapReduce): every
           last_block_id link, every height field and every data_hash constrained; public = (start hash, end hash, subtree root R, first height).
  Reduce   nodes verify `fan_in` children in-circuit, check adjacency (hash and height) and combine the R's with SHA-256 inner nodes.
  Outer    ONE more circuit verifies the chain's root proof in-circuit AND lays down the light-client skip statement (gadgets.skip_statement:
           both validator sets hashed from keys and powers, the trusted header's next_validators_hash and the target header's validators_hash bound
           to them, > 2/3 of the target power and > 1/3 of the trusted power flagged, both block numbers), and ties the two together: the chain
           starts at the trusted header's hash (as computed by the skip half), ends at the target header's hash, its first height is
           trusted block + 1 and the target block is trusted block + skip.
Public inputs of the final proof (30 words, as gadgets.combined_skip_circuit's): trusted header hash (8), target header hash (8), signer digest
(4), trusted block, target block, data commitment (8).  The Ed25519 half stays outside the circuit: the signer digest names who was flagged and
``blobstream.verify_signers`` checks exactly those signatures natively — until the curve rows exist.
Which target validator is which trusted validator (`trusted_index`) and every length (set sizes, varint groups of powers / heights, field
lengths) are CONSTANTS of the outer circuit: another shape is another circuit (and key).  Multi-GPU: rank r proves and folds the r-th contiguous
part of the chain, ONE all-gather of node proofs, rank 0 folds the root and proves the outer circuit."""
import importlib
import struct
import time

import numpy as np

from . import SHA_GATE_WIRES
from .data_commitment_mr import HeaderChainMapReduce
from .gadgets import Sha256Rows, skip_statement, skip_statement_inputs

N_PUBLIC = 30


class CombinedSkipMapReduce:
    def __init__(self, prover, poseidon_consts, skip, batch=8, fan_in=8, num_queries=28, pow_bits=16, map_provers=(), height_varint_bytes=4,
                 field_lengths=(4, 12, 5, 13, 72, 34, 34, 34, 34, 34, 34, 34, 34, 22), max_skip=1 << 20, chain=None, signatures=None):
        """chain: a HeaderChainMapReduce of the same batch / fan-in / parameters to SHARE (its leaf and node recordings serve every skip length;
        it is then not freed by free()).
        signatures: a signature_mr.SignatureSetMapReduce (same query / PoW parameters): prove_skip(..., votes=(signatures, vote bytes)) then
        ALSO proves the target validators' Ed25519 signatures (one leaf per slot, folded to a root) and the outer circuit verifies that root and
        equates its block hash with the target header hash and its signer digest with the one the power rules were computed from — the
        statement is then complete, signatures included.  Not freed by free()."""
        if skip % batch or skip < batch:
            raise ValueError("skip must be a whole number of batches")
        self.prover, self.consts = prover, tuple(np.ascontiguousarray(a, dtype=np.uint64) for a in poseidon_consts)
        self.skip, self.batch, self.nq, self.pw, self.max_skip = int(skip), int(batch), num_queries, pow_bits, int(max_skip)
        self._own_chain = chain is None
        self.chain = chain if chain is not None else HeaderChainMapReduce(
            prover, poseidon_consts, leaf_headers=batch, fan_in=fan_in, num_queries=num_queries, pow_bits=pow_bits, map_provers=map_provers,
            height_varint_bytes=height_varint_bytes, field_lengths=field_lengths, defer_commitment=(batch >= 8 and skip >= 2 * batch))
        if (self.chain.leaf_blocks, self.chain.nq, self.chain.pw) != (batch, num_queries, pow_bits):
            raise ValueError("the shared chain object has other parameters")
        self.outer = {}                  # (chain root key, trusted_index, ..., signature root key) -> RecursionProgram
        self.record_seconds = {}
        self.sigs = signatures
        if signatures is not None and (signatures.nq, signatures.pw) != (num_queries, pow_bits):
            raise ValueError("the signature MapReduce has other parameters")

    # ---- the outer circuit ------------------------------------------------------------------------------------------------------------
    def _outer(self, chain_root, chain_key, chain_is_node, sample, sig_root=None, sig_key=None):
        vc = importlib.import_module(__package__ + ".verifier_circuit")
        tf, trusted, vf, target, signed, idx, h0 = sample
        k = (bytes(np.ascontiguousarray(chain_key, dtype=np.uint64)), tuple(idx), len(trusted[0]), len(target[0]),
             None if sig_key is None else bytes(np.ascontiguousarray(sig_key, dtype=np.uint64)))
        if k in self.outer:
            return self.outer[k]
        t0 = time.perf_counter()
        skip = self.skip

        def combine(b, outs):
            g = Sha256Rows(b)
            b.auto_tag_list, b._auto_pos = (1 if sig_root is None else 2), 0     # the skip statement's free inputs: the word list after the proof(s)
            ht, hv, sd, blocks = skip_statement(b, g, tf, trusted, vf, target, signed, idx, heights=(h0, h0 + skip), max_skip=self.max_skip)
            b.auto_tag_list = None
            pub = outs[0]["public"]                                   # the chain root's statement: start (8), end (8), R (8), first height
            for x, y in zip(ht + hv, pub[:16]):
                b.assert_equal(x, y)                                  # the chain walks from THE trusted header to THE target header
            b.assert_equal(b.arith(0, 1, 1, blocks[0], blocks[0], blocks[0]), pub[24])            # first chain height = trusted block + 1
            b.assert_equal(b.arith(0, 1, skip, blocks[0], blocks[0], blocks[0]), blocks[1])       # target block = trusted block + skip
            if sig_root is not None:
                ps = outs[1]["public"]                                # the signature set's root: block hash (8), signer digest (4)
                for x, y in zip(hv + sd, ps[:12]):
                    b.assert_equal(x, y)                              # the votes name THE target header, the flags are the verified slots
            return ht + hv + sd + blocks + pub[16:24]
        chain_spec = dict(leaf_key=chain_key, n_public=HeaderChainMapReduce.N_PUBLIC, child_is_recursion=chain_is_node)
        proofs, specs = [chain_root], [chain_spec]
        if sig_root is not None:
            proofs.append(sig_root)
            specs.append(dict(leaf_key=sig_key, n_public=self.sigs.N_PUBLIC, child_is_recursion=True, child_sha=False))
        rp = vc.RecursionProgram(self.prover, proofs, chain_key, self.nq, self.pw, SHA_GATE_WIRES, self.consts, n_routed=80,
                                 n_public=HeaderChainMapReduce.N_PUBLIC, cap_height=1, child_is_recursion=chain_is_node, child_sha=True,
                                 combine=combine, builder_wires=SHA_GATE_WIRES, specs=specs)
        self.record_seconds["outer"] = round(time.perf_counter() - t0, 3)
        self.outer[k] = rp
        return rp

    def _check_shapes(self, trusted_fields, chain_headers, trusted_height):
        if len(chain_headers) != self.skip:
            raise ValueError(f"this object proves skips of {self.skip} headers")
        if tuple(len(bytes(f)) for f in trusted_fields) != self.chain.field_lengths:
            raise ValueError("the trusted header's field encodings do not have the recorded lengths")

    def _finish(self, root_chain, chain_key, chain_is_node, case, t_chain, sig_out=None):
        tf, trusted, chain_headers, target, signed, idx, h0 = case
        vf = chain_headers[-1]
        sig_root, sig_key = (sig_out["root_proof"], sig_out["key"]) if sig_out is not None else (None, None)
        rp = self._outer(root_chain, chain_key, chain_is_node, (tf, trusted, vf, target, signed, idx, h0), sig_root, sig_key)
        t0 = time.perf_counter()
        words = np.array(skip_statement_inputs(tf, trusted, vf, target, signed, heights=(h0, h0 + self.skip)), dtype=np.uint64)
        # word lists in the order the outer circuit tagged them: the verified proofs first, the skip statement's witness last
        proof, public = rp.prove([root_chain] + ([sig_root] if sig_root is not None else []) + [words], self.nq, self.pw)
        t1 = time.perf_counter()
        be = lambda ws: b"".join(struct.pack(">I", v) for v in ws)
        return {"root_proof": proof, "public": public, "key": rp.key(), "outer_seconds": round(t1 - t0, 4), "chain_seconds": round(t_chain, 4),
                "signatures_in_circuit": sig_root is not None,
                "trusted_hash": be(public[:8]), "target_hash": be(public[8:16]), "signer_digest": public[16:20], "trusted_block": public[20],
                "target_block": public[21], "commitment": be(public[22:30]), "outer_rows": rp.stats["rows"]}

    # ---- proving ----------------------------------------------------------------------------------------------------------------------
    def _disjoint_provers(self):
        mine = {id(p) for p in [self.chain.prover] + list(self.chain.map_provers)}
        return self.sigs is not None and not (mine & {id(p) for p in [self.sigs.prover] + list(self.sigs.map_provers)})

    def _prove_votes(self, target, signed, votes, chain_headers, distributed=None):
        if votes is None:
            return None
        if self.sigs is None:
            raise ValueError("this object was made without a signature MapReduce")
        sigs, msgs = votes
        if distributed is None:
            return self.sigs.prove_set(target[0], sigs, msgs, signed)
        return self.sigs.prove_set_distributed(target[0], sigs, msgs, signed, **distributed)

    def prove_skip(self, trusted_fields, trusted, chain_headers, target, signed, trusted_index, trusted_height, votes=None):
        """trusted_fields: the trusted header's 14 field encodings (its field 2 = height, field 8 = BytesValue(hash of `trusted`));
        chain_headers: the `skip` headers after it, the last one being the target (its field 7 = BytesValue(hash of `target`)), each linked to its
        predecessor through field 4 and carrying its height in field 2; trusted / target = (pubkeys, voting_powers); signed = the target
        validators' flags; trusted_index[i] = position of target validator i in the trusted set, or None.  ValueError when a premise fails."""
        self._check_shapes(trusted_fields, chain_headers, trusted_height)
        start = HeaderChainMapReduce.header_hash(trusted_fields)

        def chain_part():
            t0 = time.perf_counter()
            o = self.chain.prove_chain(start, trusted_height + 1, chain_headers)
            return o, time.perf_counter() - t0

        def vote_part():
            t0 = time.perf_counter()
            o = self._prove_votes(target, signed, votes, chain_headers)
            return o, time.perf_counter() - t0
        if votes is not None and self._disjoint_provers():
            # the two MapReduces have provers (ctxs) of their own: run them side by side — the latency-bound tails of one Reduce (a few node
            # proofs, then one root) overlap the throughput-bound Map of the other
            from concurrent.futures import ThreadPoolExecutor
            with ThreadPoolExecutor(2) as ex:
                fc, fv = ex.submit(chain_part), ex.submit(vote_part)
                (out, t_chain), (sig_out, t_sig) = fc.result(), fv.result()
        else:
            (out, t_chain), (sig_out, t_sig) = chain_part(), vote_part()
        res = self._finish(out["root_proof"], out["key"], out["leaves"] > 1, (trusted_fields, trusted, chain_headers, target, signed, trusted_index,
                                                                               trusted_height), t_chain, sig_out)
        res.update(leaves=out["leaves"], map_seconds=out["map_seconds"], reduce_seconds=out["reduce_seconds"], levels=out["levels"],
                   record_seconds=dict(self.chain.record_seconds, **self.record_seconds))
        if sig_out is not None:
            res.update(signature_seconds=round(t_sig, 4), signature_slots=sig_out["slots"], signature_map_seconds=sig_out["map_seconds"],
                       signature_levels=sig_out["levels"], signature_record_seconds=sig_out["record_seconds"])
        return res

    def prove_skip_distributed(self, trusted_fields, trusted, chain_headers, target, signed, trusted_index, trusted_height, device=None, comm=None,
                               votes=None):
        """prove_skip with the chain spread over the ranks (HeaderChainMapReduce.prove_chain_distributed); the outer circuit is proved on rank 0.
        Every rank passes the whole case.  Returns the prove_skip dict on rank 0, {"root_proof": None, ...} elsewhere."""
        self._check_shapes(trusted_fields, chain_headers, trusted_height)
        mrm = importlib.import_module(__package__ + ".mapreduce")
        if mrm._world(comm)[1] == 1:
            # one rank: no exchange to order — the plain form, which may run the two MapReduces side by side (at N > 1 they stay sequential: their
            # collectives must be entered in the same order on every rank)
            res = self.prove_skip(trusted_fields, trusted, chain_headers, target, signed, trusted_index, trusted_height, votes=votes)
            res["ranks"] = 1
            return res
        start = HeaderChainMapReduce.header_hash(trusted_fields)
        t0 = time.perf_counter()
        out = self.chain.prove_chain_distributed(start, trusted_height + 1, chain_headers, device=device, comm=comm)
        t_chain = time.perf_counter() - t0
        t0 = time.perf_counter()
        sig_out = self._prove_votes(target, signed, votes, chain_headers, distributed=dict(device=device, comm=comm))       # every rank takes part
        t_sig = time.perf_counter() - t0
        if out["root_proof"] is None:
            return {"root_proof": None, "map_seconds": out["map_seconds"], "chain_seconds": round(t_chain, 4), "leaves": out["leaves"], "ranks": out["ranks"],
                    "signature_seconds": round(t_sig, 4), "signature_map_seconds": sig_out["map_seconds"] if sig_out else None}
        res = self._finish(out["root_proof"], out["key"], out["leaves"] > 1, (trusted_fields, trusted, chain_headers, target, signed, trusted_index,
                                                                               trusted_height), t_chain, sig_out)
        res.update(leaves=out["leaves"], ranks=out["ranks"], map_seconds=out["map_seconds"], levels=out["levels"],
                   record_seconds=dict(self.chain.record_seconds, **self.record_seconds))
        if sig_out is not None:
            res.update(signature_seconds=round(t_sig, 4), signature_slots=sig_out["slots"], signature_map_seconds=sig_out["map_seconds"],
                       signature_levels=sig_out["levels"], signature_record_seconds=sig_out["record_seconds"])
        return res

    # ---- the consumer -----------------------------------------------------------------------------------------------------------------
    def synthetic_votes(self, case, seeds):
        """the target validators' vote bytes and Ed25519 signatures for a case made with real_keys=True: (signatures, vote bytes) for prove_skip's
        `votes` — every flagged validator signs the vote naming the target header's hash"""
        from .ed25519_circuit import keypair_and_sign
        _, _, chain, (vk, _), signed, _, _ = case
        target_hash = HeaderChainMapReduce.header_hash(chain[-1])
        msgs = [self.sigs.vote_bytes(target_hash, i) for i in range(len(vk))]
        sigs = []
        for i, (key, sg) in enumerate(zip(vk, signed)):
            if not sg:
                sigs.append(None)
                continue
            pub, sig = keypair_and_sign(seeds[i], msgs[i])
            assert pub == key
            sigs.append(sig)
        return sigs, msgs

    def synthetic_case(self, n_trusted, n_target, trusted_index, trusted_height=None, power_groups=6, seed=0, real_keys=False):
        """a well-formed case of this object's shape (used by expected_key, tests and the bench): random keys, powers with `power_groups` varint
        groups, every target validator flagged, headers with random opaque fields and real links, heights, validator-set hashes.
        real_keys: the validators' keys are real Ed25519 public keys (derived from per-validator seeds, returned as an eighth element) so that
        synthetic_votes can sign for them.  Returns the prove_skip argument tuple (+ the target validators' seeds with real_keys)."""
        bs = importlib.import_module(__package__ + ".blobstream")
        rng = np.random.default_rng(seed)
        h0 = (1 << (7 * (self.chain.n_groups - 1))) + 17 if trusted_height is None else int(trusted_height)
        seeds_by_key = {}

        def key():
            if not real_keys:
                return rng.integers(0, 256, 32, dtype=np.uint8).tobytes()
            from .ed25519_circuit import keypair_and_sign
            sd = rng.integers(0, 256, 32, dtype=np.uint8).tobytes()
            pub, _ = keypair_and_sign(sd, b"")
            seeds_by_key[pub] = sd
            return pub
        power = lambda: int(rng.integers(1 << (7 * (power_groups - 1)), 1 << (7 * power_groups - 1)))
        tk, tp = [key() for _ in range(n_trusted)], [power() for _ in range(n_trusted)]
        vk = [tk[t] if t is not None else key() for t in trusted_index]
        vp = [power() for _ in range(n_target)]
        lens = self.chain.field_lengths
        fields = lambda: [rng.integers(0, 256, L, dtype=np.uint8).tobytes() for L in lens]
        tf = fields()
        tf[2] = b"\x08" + bs.encode_varint(h0)
        tf[8] = b"\x0a\x20" + bs.validator_set_hash(self.prover, tk, tp)
        prev, chain = HeaderChainMapReduce.header_hash(tf), []
        for k in range(self.skip):
            f = fields()
            f[2] = b"\x08" + bs.encode_varint(h0 + 1 + k)
            f[4] = b"\x0a\x20" + prev + f[4][34:]
            f[6] = b"\x0a\x20" + f[6][2:]
            if k == self.skip - 1:
                f[7] = b"\x0a\x20" + bs.validator_set_hash(self.prover, vk, vp)
            chain.append(f)
            prev = HeaderChainMapReduce.header_hash(f)
        case = (tf, (tk, tp), chain, (vk, vp), [True] * n_target, list(trusted_index), h0)
        return case + ([seeds_by_key[k] for k in vk],) if real_keys else case

    def expected_key(self, n_trusted, n_target, trusted_index, power_groups=6):
        """the VERIFIER's own setup: the outer circuit's verifying key for this shape, from a synthetic case proved on this object (see
        DataCommitmentMapReduce.expected_key: a key must never be taken from the prover)"""
        if self.sigs is None:
            return self.prove_skip(*self.synthetic_case(n_trusted, n_target, trusted_index, power_groups=power_groups))["key"]
        *case, seeds = self.synthetic_case(n_trusted, n_target, trusted_index, power_groups=power_groups, real_keys=True)
        return self.prove_skip(*case, votes=self.synthetic_votes(case, seeds))["key"]

    @staticmethod
    def evm_values(public):
        """the final proof's statement in the packed form a contract reads / writes ([RECALLED] plonky2x evm_read / evm_write of the skip circuit:
        input = (trusted_block u64, trusted_header_hash bytes32, target_block u64), output = (target_header_hash bytes32, data_commitment
        bytes32), big-endian, blobstream.pack_skip_inputs / pack_outputs) — and the check that the proof's public inputs ARE those bytes' 32-bit
        words (block numbers as single field elements, the signer digest in between).  Returns (input bytes, output bytes)."""
        from .blobstream import pack_outputs, pack_skip_inputs, public_words
        be = lambda ws: b"".join(struct.pack(">I", int(v)) for v in ws)
        trusted_hash, target_hash, commitment = be(public[:8]), be(public[8:16]), be(public[22:30])
        inp = pack_skip_inputs(int(public[20]), trusted_hash, int(public[21]))
        outp = pack_outputs(target_hash, commitment)
        assert public_words(outp) == [int(v) for v in public[8:16]] + [int(v) for v in public[22:30]]
        assert public_words(trusted_hash) == [int(v) for v in public[:8]]
        return inp, outp

    def verify(self, root_proof, key, trusted_hash, target_hash, signer_digest, trusted_block, target_block, commitment):
        public = list(struct.unpack(">8I", bytes(trusted_hash))) + list(struct.unpack(">8I", bytes(target_hash))) + [int(v) for v in signer_digest] + \
            [int(trusted_block), int(target_block)] + list(struct.unpack(">8I", bytes(commitment)))
        return bool(self.prover.plonk_verify(root_proof, key, self.nq, self.pw, public=public))

    def free(self):
        for rp in self.outer.values():
            rp.free()
        self.outer = {}
        if self._own_chain:
            self.chain.free()
