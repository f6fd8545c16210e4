"""DataCommitment over a LARGE block range as a MapReduce of proofs (BASELINE configs[4] shape: 4096 blocks; upstream
names recalled, unverified — reference file:line NONE, the mount is empty: blobstreamx ``DataCommitmentCircuit`` built on plonky2x ``mapreduce``).

Map: the range is cut into leaves of ``leaf_blocks`` blocks; a leaf proof states, for tuples it knows (height, dataRoot):
    public = [ R: the 8 words of the RFC 6962 SHA-256 subtree root over abi.encode(height, dataRoot) ]
           + [ D: hash_no_pad (Poseidon) of the subrange's 16 * leaf_blocks tuple words ]
with every SHA-256 compression on the SHA row gates (gadgets.Sha256Rows) and the sponge on Poseidon rows.  One circuit for every leaf: it is
recorded once (recursion.WitnessProgram) and replayed per leaf — witness evaluation on the host, placement and row filling on the GPU.
Reduce: a node verifies ``fan_in`` child proofs completely IN-CIRCUIT (verifier_circuit.verify_in_circuit, SHA-row and Poseidon-row constraints
of the child included) and states the same thing one level up:
    R' = SHA-256 inner nodes (0x01 || left || right) over the children's R,   D' = Poseidon two_to_one tree over the children's D
so every level has the same 12 public inputs and the root proof says: "R is the data commitment of tuples whose digest tree is D".  Whoever
knows the tuples recomputes D on the host (`tuples_digest`) and needs nothing else — no leaf proof, no node proof.
Everything here is build-defined (NOT upstream's circuit, statement or proof format)."""
import importlib
import struct
import time

import numpy as np

from . import SHA_GATE_WIRES, poseidon_permute_host
from .gadgets import Sha256Rows
from .recursion import CircuitBuilder


def tuple_words(height, data_root):
    """the 16 big-endian 32-bit words of abi.encode(height, dataRoot)"""
    return list(struct.unpack(">16I", int(height).to_bytes(32, "big") + bytes(data_root)))


def _hash_no_pad_host(consts, words):
    state = np.zeros(12, dtype=np.uint64)
    for off in range(0, len(words), 8):
        chunk = words[off:off + 8]
        state[:len(chunk)] = chunk
        state = poseidon_permute_host(consts, state)[0]
    return [int(v) for v in state[:4]]


def tuples_digest(consts, heights, data_roots, leaf_blocks):
    """D of a range: per leaf hash_no_pad of its tuple words, then the binary Poseidon two_to_one tree over the leaves (what the proofs expose)"""
    level = []
    for k in range(0, len(heights), leaf_blocks):
        words = [w for h, r in zip(heights[k:k + leaf_blocks], data_roots[k:k + leaf_blocks]) for w in tuple_words(h, r)]
        level.append(_hash_no_pad_host(consts, words))
    while len(level) > 1:
        st = np.array([level[2 * k] + level[2 * k + 1] + [0, 0, 0, 0] for k in range(len(level) // 2)], dtype=np.uint64)
        level = [[int(v) for v in row[:4]] for row in poseidon_permute_host(consts, st)]
    return level[0]


def _combine(b, outs):
    """a node's statement from its children's: SHA-256 inner nodes over the R's, Poseidon two_to_one over the D's"""
    g = Sha256Rows(b)
    roots = [o["public"][:8] for o in outs]
    digs = [o["public"][8:12] for o in outs]
    while len(roots) > 1:
        roots = [g.hash_prefixed_64(0x01, roots[k] + roots[k + 1]) for k in range(0, len(roots), 2)]
        digs = [b.two_to_one(digs[k], digs[k + 1]) for k in range(0, len(digs), 2)]
    return roots[0] + digs[0]


class DataCommitmentMapReduce:
    """prove_range(heights, data_roots) -> one proof for the whole range; verify(...) checks it against the tuples.  leaf_blocks and fan_in are
    powers of two; the number of leaves must be fan_in^k * m with the last level's fan-in m a power of two <= fan_in (e.g. 64 leaves, fan_in 16:
    4 nodes of 16, then a root of 4)."""

    N_PUBLIC = 12                       # public inputs of every proof of the tree: R (8 words) then D (4 words)

    def _combine_for(self, span):
        """the node's statement-combining hook; span = how many leaf units (blocks) each child covers at this level (span == leaf_blocks:
        the children are leaf proofs)"""
        return _combine

    def _child_has_poseidon_rows(self, level):
        """the circuit flags of the proofs a level-`level` node verifies: here leaves hash their tuples with Poseidon rows, nodes always have them"""
        return True

    def _child_has_sha_rows(self, level):
        """do the proofs a level-`level` node verifies come from circuits with SHA-256 / ADD rows (here: always — leaves hash, nodes combine roots)"""
        return True

    def _child_n_public(self, level):
        """public inputs of the proofs a level-`level` node verifies (the same at every level here)"""
        return self.N_PUBLIC

    def _child_n_routed(self, level):
        """routed wires of the circuits a level-`level` node verifies (80 of 144 everywhere here)"""
        return 80

    def __init__(self, prover, poseidon_consts, leaf_blocks=64, fan_in=8, num_queries=28, pow_bits=16, map_provers=()):
        """map_provers: further Provers on the same GPU (their Poseidon constants set): the Map step then proves leaves on all of them at once,
        one host thread each (the latency-bound phases of one leaf proof overlap the throughput-bound phases of another, as in mapreduce.py)"""
        assert leaf_blocks >= 1 and leaf_blocks & (leaf_blocks - 1) == 0 and fan_in >= 2 and fan_in & (fan_in - 1) == 0
        self.prover, self.consts = prover, tuple(np.ascontiguousarray(a, dtype=np.uint64) for a in poseidon_consts)
        self.map_provers, self.map_circuits = list(map_provers), []
        self.leaf_blocks, self.fan_in, self.nq, self.pw = leaf_blocks, fan_in, num_queries, pow_bits
        self.leaf_program = self.leaf_circuit = None
        self.nodes = {}                 # (level, fan-in, span, child key) -> RecursionProgram
        self.node_replicas = {}         # id(node program) -> [its replicas on each map prover]
        self.record_seconds = {}

    # ---- Map ----------------------------------------------------------------------------------------------------------------------
    def _record_leaf(self):
        t0 = time.perf_counter()
        b = CircuitBuilder(self.prover, n_wires=SHA_GATE_WIRES)
        g = Sha256Rows(b)
        words_all, level = [], []
        for k in range(self.leaf_blocks):
            words = [b.range32(b.var(0 if j else k)) for j in range(16)]               # sample values; the program's inputs, in tuple order
            words_all += words
            level.append(g.hash_prefixed_64(0x00, words))
        while len(level) > 1:
            level = [g.hash_prefixed_64(0x01, level[k] + level[k + 1]) for k in range(0, len(level), 2)]
        digest = b.hash_no_pad(words_all)
        for v in level[0] + digest:
            b.public_input(v)
        self.leaf_program = b.program()
        self.leaf_circuit = self.leaf_program.setup(self.prover)
        self.map_circuits = [self.leaf_program.setup(p) for p in self.map_provers]          # the same circuit (same key) committed on each ctx
        self.record_seconds["leaf"] = round(time.perf_counter() - t0, 3)

    def prove_leaf(self, heights, data_roots, which=0):
        """(proof, public) for one leaf's subrange, through the recorded leaf program; which: 0 = the main prover, k = map_provers[k-1]"""
        assert len(heights) == len(data_roots) == self.leaf_blocks
        if self.leaf_program is None:
            self._record_leaf()
        prover, circuit = (self.prover, self.leaf_circuit) if which == 0 else (self.map_provers[which - 1], self.map_circuits[which - 1])
        inputs = [w for h, r in zip(heights, data_roots) for w in tuple_words(h, r)]
        vals = self.leaf_program.evaluate(self.consts, inputs, threads=1)
        dw, public = self.leaf_program.device_witness(prover, vals, reuse=True)          # the program's own buffer: no allocation per leaf
        return circuit.prove_(dw, self.nq, self.pw, public=public), public

    def _map_inputs(self, inputs_list):
        """leaf proofs for a list of input vectors of the recorded leaf program, in order.  The witness programs are evaluated on a pool of host
        threads of their own (glp_witness_eval releases the GIL; a 1.3 M-variable signature leaf takes ~25 ms on one core) while the provers —
        this object's prover and its map_provers, one host thread each — place the witnesses and prove: the GPU never waits for a witness after the
        first ones.  A witness that does not satisfy the circuit raises ValueError from here."""
        import os
        from concurrent.futures import ThreadPoolExecutor
        n_workers = 1 + len(self.map_provers)
        n_eval = max(1, min(len(inputs_list), (len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else os.cpu_count() or 2) - n_workers, 12))
        evaluate = lambda inp: self.leaf_program.evaluate(self.consts, inp, threads=1)
        with ThreadPoolExecutor(n_eval) as epool:
            futs = [epool.submit(evaluate, inp) for inp in inputs_list]

            def work(w):
                if w:
                    self.map_provers[w - 1].bind_thread()
                prover, circuit = (self.prover, self.leaf_circuit) if w == 0 else (self.map_provers[w - 1], self.map_circuits[w - 1])
                out = []
                for i in range(w, len(inputs_list), n_workers):
                    dw, public = self.leaf_program.device_witness(prover, futs[i].result(), reuse=True)
                    out.append((i, circuit.prove_(dw, self.nq, self.pw, public=public)))
                return out
            try:
                if n_workers == 1 or len(inputs_list) == 1:
                    done = work(0)
                else:
                    with ThreadPoolExecutor(n_workers) as ex:
                        done = sorted((p for f in [ex.submit(work, w) for w in range(n_workers)] for p in f.result()), key=lambda t: t[0])
            finally:
                for f in futs:
                    f.cancel()
        return [p for _, p in done]

    def prove_leaves(self, heights, data_roots):
        """the Map step of a (sub)range: its leaf proofs in order, on every prover this object has"""
        if self.leaf_program is None:
            self._record_leaf()
        B = self.leaf_blocks
        return self._map_inputs([[w for h, r in zip(heights[k:k + B], data_roots[k:k + B]) for w in tuple_words(h, r)] for k in range(0, len(heights), B)])

    # ---- Reduce -------------------------------------------------------------------------------------------------------------------
    def _node(self, level, proofs, child_key, span=0):
        vc = importlib.import_module(__package__ + ".verifier_circuit")
        k = (level, len(proofs), span, bytes(np.ascontiguousarray(child_key, dtype=np.uint64)))      # the child circuit's key is a CONSTANT of the node circuit
        if k not in self.nodes:
            t0 = time.perf_counter()
            self.nodes[k] = vc.RecursionProgram(self.prover, proofs, child_key, self.nq, self.pw, SHA_GATE_WIRES, self.consts, n_routed=self._child_n_routed(level),
                                                n_public=self._child_n_public(level), cap_height=1, child_is_recursion=self._child_has_poseidon_rows(level), child_sha=self._child_has_sha_rows(level),
                                                combine=self._combine_for(span), builder_wires=SHA_GATE_WIRES)
            self.record_seconds[f"node_level{level}_fan{len(proofs)}"] = round(time.perf_counter() - t0, 3)
        return self.nodes[k]

    def reduce(self, proofs, timings=None, child_key=None, level=1, span=None, max_levels=None):
        """fold child proofs level by level; returns (root proof, its public inputs, its verifying key, the next level number).  child_key /
        level / span: where the fold starts (default: leaf proofs at level 1, each covering leaf_blocks units; a later start folds node proofs made
        elsewhere, e.g. on other ranks, each covering `span` units).  max_levels: stop after that many levels and return the LIST of node proofs
        of the last one in place of the root"""
        levels_done = 0
        cur, key, public = list(proofs), (self.leaf_circuit.cap() if child_key is None else child_key), None
        span = self.leaf_blocks if span is None else span
        while True:
            fan = min(self.fan_in, len(cur))
            if len(cur) % fan or fan & (fan - 1):
                raise ValueError("the number of proofs at a level is not a multiple of a power-of-two fan-in")
            t0 = time.perf_counter()
            rp = self._node(level, cur[:fan], key, span)
            groups = [cur[k:k + fan] for k in range(0, len(cur), fan)]
            if len(groups) > 1 and self.map_provers:
                # several nodes of one level: one host thread per prover, each with its own commitment of the level's (shared) recording
                from concurrent.futures import ThreadPoolExecutor
                if id(rp) not in self.node_replicas:
                    self.node_replicas[id(rp)] = [rp.replicate(p) for p in self.map_provers]
                workers = [rp] + self.node_replicas[id(rp)]

                def work(w):
                    if w:
                        self.map_provers[w - 1].bind_thread()
                    return [(g, workers[w].prove(groups[g], self.nq, self.pw)) for g in range(w, len(groups), len(workers))]
                with ThreadPoolExecutor(len(workers)) as ex:
                    done = sorted((r for f in [ex.submit(work, w) for w in range(len(workers))] for r in f.result()), key=lambda t: t[0])
                nxt, public = [pr_[0] for _, pr_ in done], done[-1][1][1]
            else:
                nxt = []
                for grp in groups:
                    proof, public = rp.prove(grp, self.nq, self.pw)
                    nxt.append(proof)
            if timings is not None:
                timings.append({"level": level, "nodes": len(nxt), "fan_in": fan, "rows": rp.stats["rows"],
                                "seconds_including_first_recording": round(time.perf_counter() - t0, 4)})
            key, level, span = rp.key(), level + 1, span * fan
            levels_done += 1
            if max_levels is not None and levels_done >= max_levels:
                self.last_span = span
                return nxt, public, key, level
            if len(nxt) == 1:
                self.last_span = span
                return nxt[0], public, key, level
            cur = nxt

    def prove_range_distributed(self, heights, data_roots, device=None, comm=None):
        """the same proof with the work spread over the ranks (mapreduce.reduce_tree_distributed): rank r proves the leaves of the r-th CONTIGUOUS
        part of the range and folds them into one node proof on its own GPU, ONE all-gather carries the `world` node proofs, rank 0 folds them into
        the root.  Every rank passes the whole range (it only touches its part).  Returns the prove_range dict on rank 0 (root_proof None elsewhere)."""
        mrm = importlib.import_module(__package__ + ".mapreduce")
        rank, world = mrm._world(comm)
        n = len(heights)
        n_leaves = n // self.leaf_blocks
        if n % self.leaf_blocks or n_leaves % world or len(data_roots) != n:
            raise ValueError("the range is not a whole number of leaves per rank")
        per = n_leaves // world
        lo = rank * per * self.leaf_blocks
        state, levels = {}, []

        def fold_local(_):
            t0 = time.perf_counter()
            hi = lo + per * self.leaf_blocks
            leaves = self.prove_leaves(heights[lo:hi], data_roots[lo:hi])
            state["map_seconds"] = round(time.perf_counter() - t0, 4)
            if per == 1:
                state.update(key=self.leaf_circuit.cap(), level=1, public=None, span=self.leaf_blocks)
                return leaves[0]
            proof, public, key, level = self.reduce(leaves, levels)
            state.update(key=key, level=level, public=public, span=self.last_span)
            return proof

        def fold_root(nodes):
            proof, public, key, _ = self.reduce(nodes, levels, child_key=state["key"], level=state["level"], span=state["span"])
            state.update(key=key, public=public)
            return proof
        t0 = time.perf_counter()
        out = mrm.reduce_tree_distributed(fold_local, fold_root, [b""] * per, 1 << 18, device=device, comm=comm)
        root = out["root_proof"]
        public = state["public"]
        if root is not None and public is None:
            public = [int(v) for v in importlib.import_module(__package__).proof_public_inputs(root)]
        return {"root_proof": root, "public": public, "key": state["key"] if root is not None else None, "leaves": n_leaves, "ranks": world,
                "map_seconds": state["map_seconds"], "seconds": round(time.perf_counter() - t0, 4), "levels": levels,
                "record_seconds": dict(self.record_seconds),
                "commitment": b"".join(struct.pack(">I", v) for v in public[:8]) if root is not None else None}

    def prove_range(self, heights, data_roots):
        n = len(heights)
        if n % self.leaf_blocks or len(data_roots) != n:
            raise ValueError("the range is not a whole number of leaves")
        t0 = time.perf_counter()
        leaves = self.prove_leaves(heights, data_roots)
        t1 = time.perf_counter()
        levels = []
        if len(leaves) == 1:
            root_proof, public, key = leaves[0], [int(v) for v in importlib.import_module(__package__).proof_public_inputs(leaves[0])], self.leaf_circuit.cap()
        else:
            root_proof, public, key, _ = self.reduce(leaves, levels)
        t2 = time.perf_counter()
        return {"root_proof": root_proof, "public": public, "key": key, "leaves": len(leaves), "map_seconds": round(t1 - t0, 4),
                "reduce_seconds": round(t2 - t1, 4), "levels": levels, "record_seconds": dict(self.record_seconds),
                "commitment": b"".join(struct.pack(">I", v) for v in public[:8])}

    def expected_key(self, n_blocks):
        """The VERIFIER's own setup: the verifying key of the root circuit for a range of n_blocks, derived on THIS object from a synthetic range
        of the same shape (the circuits — leaf, every node level — depend on the shape and the parameters only, never on the tuples).  `verify`
        must be given a key obtained this way (or from a recording the verifier trusts), never one handed over with the proof: a key names the
        circuit that was run, and a prover-chosen key could name one without, say, the adjacency checks.  Costs one proof of the synthetic range
        (the circuits are recorded on the way and stay cached for real ranges)."""
        return self.prove_range(list(range(1, n_blocks + 1)), [bytes(32)] * n_blocks)["key"]

    def verify(self, root_proof, key, heights, data_roots, commitment):
        """the consumer: `key` comes from the verifier's own setup (expected_key), not from the prover. `commitment` (32 bytes) is the data commitment of exactly these tuples — the proof's public inputs must be the
        commitment's words followed by the tuples' digest tree, and the proof must verify for `key` (the root circuit's verifying key)"""
        public = list(struct.unpack(">8I", bytes(commitment))) + tuples_digest(self.consts, heights, data_roots, self.leaf_blocks)
        return bool(self.prover.plonk_verify(root_proof, key, self.nq, self.pw, public=public))

    def free(self):
        for reps in self.node_replicas.values():
            for rp in reps:
                rp.circuit.free()
        self.node_replicas = {}
        for rp in self.nodes.values():
            rp.free()
        self.nodes = {}
        if self.leaf_circuit is not None:
            self.leaf_program.release()
            for c in self.map_circuits:
                c.free()
            self.map_circuits = []
            self.leaf_circuit.free()
            self.leaf_circuit = self.leaf_program = None


# ---- the header-chain form as a MapReduce ([RECALLED] blobstreamx: the data commitment is proved over HEADERS walked from a trusted hash) --------
def _varint_groups(value, n_groups):
    groups = [(int(value) >> (7 * j)) & 0x7F for j in range(n_groups)]
    if int(value) >> (7 * n_groups) or (n_groups > 1 and groups[-1] == 0):
        raise ValueError(f"height {value} does not encode in exactly {n_groups} varint bytes")
    return groups


def _tuple_leaf(b, g, hk, data_words, c2_15):
    """the RFC 6962 leaf hash of abi.encode(height, data_hash) for a height VARIABLE hk (< 2^49) and the data hash's 8 word variables: the height is
    the low two 32-bit words of a uint256, split with hi < 2^17 shown (without that bound hi * 2^32 + lo == hk has a second solution mod p)"""
    lo, hi = b.bit_field(hk, 0, 32), b.bit_field(hk, 32, 17)
    b.range32(lo)
    b.range32(hi)
    b.range32(b.arith(1, 0, 0, hi, c2_15, hi))                                          # hi < 2^17: hi * 2^32 + lo < 2^49 cannot wrap mod p, so
    b.assert_equal(b.arith(1, 1, 0, hi, g.c2_32, lo), hk)                                # (hi, lo) is THE split of hk (no hi = 2^32 - 1, lo = hk + 1 alias)
    zero = b.constant(0)
    return g.hash_prefixed_64(0x00, [zero] * 6 + [hi, lo] + list(data_words))


def _chain_leaf_statement(b, g, start_hash_words, first_height, headers, n_groups, defer_commitment=False):
    """one leaf of the chain MapReduce on builder b: `headers` (field encodings) follow a header whose hash is start_hash (INPUT words); header k
    sits at height first_height + k (a variable + constant; its Int64Value field encoding is built in-circuit from range-checked 7-bit groups, whose
    number n_groups is a constant of the circuit), links to its predecessor's hash through last_block_id, and its data_hash feeds tuple k.
    Returns the leaf's public inputs: start hash (8), end hash (8), subtree root R (8), first height (1).
    defer_commitment: the leaf does NOT hash its tuples; it exposes every header's data hash instead — start hash (8), end hash (8), first
    height (1), then 8 words per header — and the level-1 NODE hashes the tuples of its children (30 of an 8-header leaf's 368 compressions are
    tuple hashing: without them the leaf fits 2^16 rows instead of 2^17, and the node circuit has the room)."""
    from .gadgets import header_hash_statement
    start = [b.range32(b.var(v)) for v in start_hash_words]
    first = b.var(first_height)
    wrap = lambda ws: [b.constant(0x0a), b.constant(0x20)] + [x for w in ws for x in g.bytes_of_word(w)]
    c128, c2_15 = b.constant(128), b.constant(1 << 15)
    prev, leaves = start, []
    for k, fields in enumerate(headers):
        if len(bytes(fields[4])) < 34 or len(bytes(fields[6])) != 34:
            raise ValueError("header fields 4 (last_block_id) / 6 (data_hash) do not have the expected encodings")
        hk = first if k == 0 else b.arith(0, 1, k, first, first, first)                  # first + k
        gv = []
        for gval in _varint_groups(b.value(hk), n_groups):
            v = b.var(gval)
            b.range32(v)
            b.range32(b.arith(1, 0, 0, v, b.constant(1 << 25), v))                       # v < 2^7
            gv.append(v)
        acc = gv[-1]
        for v in reversed(gv[:-1]):
            acc = b.arith(1, 1, 0, acc, c128, v)
        b.assert_equal(acc, hk)                                                           # the groups spell THIS height
        # (a canonical varint needs a non-zero top group; a zero one would be a different byte string, hence a different header hash)
        hfield = [b.constant(0x08)] + [b.arith(0, 1, 0x80, v, v, v) for v in gv[:-1]] + [gv[-1]]
        data_hash = [g.byte(b.var(v)) for v in bytes(fields[6])[2:]]
        block_id = wrap(prev) + [g.byte(b.var(v)) for v in bytes(fields[4])[34:]]
        prev = header_hash_statement(b, g, fields, bound={2: hfield, 4: block_id, 6: [b.constant(0x0a), b.constant(0x20)] + data_hash})
        root_words = [g.word_from_bytes(data_hash[j:j + 4]) for j in range(0, 32, 4)]
        leaves.append(root_words if defer_commitment else _tuple_leaf(b, g, hk, root_words, c2_15))
    if defer_commitment:
        return start + prev + [first] + [w for ws in leaves for w in ws]
    while len(leaves) > 1:
        leaves = [g.hash_prefixed_64(0x01, leaves[j] + leaves[j + 1]) for j in range(0, len(leaves), 2)]
    return start + prev + leaves[0] + [first]


def _chain_leaf_inputs(start_hash, first_height, headers, n_groups):
    """the recorded leaf program's input vector, in the order _chain_leaf_statement creates its free variables"""
    out = list(struct.unpack(">8I", bytes(start_hash))) + [int(first_height)]
    for k, fields in enumerate(headers):
        out += _varint_groups(first_height + k, n_groups)
        out += list(bytes(fields[6])[2:])
        out += list(bytes(fields[4])[34:])
        for j, fb in enumerate(fields):
            if j not in (2, 4, 6):
                out += list(bytes(fb))
    return out


class HeaderChainMapReduce(DataCommitmentMapReduce):
    """The data commitment of a CHAIN of headers as a MapReduce of proofs: a leaf takes `leaf_headers` consecutive headers (14 field encodings each)
    after a header whose hash it is given, constrains every link (last_block_id = the predecessor's hash, computed in-circuit), every height field
    (first_height + k) and hashes the (height, data_hash) tuples into its subtree root; its public inputs are (start hash, end hash, R, first
    height).  A node verifies its children in-circuit and checks that they are ADJACENT — child k+1 starts at child k's end hash and at
    first_height + span — before combining the R's with SHA-256 inner nodes: (start of the first, end of the last, R', first height of the first).
    The root proof therefore says: "walking the headers from start_hash, at heights first_height.., one arrives at end_hash, and their data hashes
    commit to R".  About 45 compressions per header.  Heights must all encode in `height_varint_bytes` varint bytes (a constant of the circuits)."""
    N_PUBLIC = 25

    def __init__(self, prover, poseidon_consts, leaf_headers=8, fan_in=8, num_queries=28, pow_bits=16, map_provers=(), height_varint_bytes=4,
                 field_lengths=(4, 12, 5, 13, 72, 34, 34, 34, 34, 34, 34, 34, 34, 22), defer_commitment=None):
        """defer_commitment (default: leaves of 8 or more headers): the leaves expose their headers' data hashes and the LEVEL-1 NODES hash the
        (height, data_hash) tuples and the leaf subtrees — round 3: an 8-header leaf is 368 compressions x 178 rows = 2.7k rows past 2^16; without
        its 30 tuple / subtree compressions it fits 2^16 rows (half the leaf proof), and the node circuit (75k of 131k rows used) has the room.
        The statement of every node — hence of the root — is unchanged; a chain of ONE leaf has no node and is refused in this mode."""
        super().__init__(prover, poseidon_consts, leaf_blocks=leaf_headers, fan_in=fan_in, num_queries=num_queries, pow_bits=pow_bits,
                         map_provers=map_provers)
        self.defer = (leaf_headers >= 8) if defer_commitment is None else bool(defer_commitment)
        if self.defer and leaf_headers & (leaf_headers - 1):
            raise ValueError("deferred tuple hashing needs a power-of-two number of headers per leaf")
        if not 1 <= height_varint_bytes <= 7:
            raise ValueError("heights are below 2^49: at most 7 varint bytes")
        self.n_groups, self.field_lengths = height_varint_bytes, tuple(field_lengths)

    def _child_n_public(self, level):
        return 17 + 8 * self.leaf_blocks if (level == 1 and self.defer) else self.N_PUBLIC

    def _combine_for(self, span):
        deferred_leaves = self.defer and span == self.leaf_blocks                    # the children are leaves that expose data hashes, not R

        def combine(b, outs):
            g = Sha256Rows(b)
            first_of = (lambda o: o["public"][16]) if deferred_leaves else (lambda o: o["public"][24])
            for left, right in zip(outs, outs[1:]):
                for x, y in zip(left["public"][8:16], right["public"][:8]):
                    b.assert_equal(x, y)                                              # right starts where left ended
                lf = first_of(left)
                b.assert_equal(b.arith(0, 1, span, lf, lf, lf), first_of(right))      # ... and span headers later
            if deferred_leaves:
                c2_15 = b.constant(1 << 15)
                roots = []
                for o in outs:                                                        # the child's tuples, hashed HERE: heights first .. first + B - 1
                    first, words = o["public"][16], o["public"][17:]
                    lv = []
                    for k in range(self.leaf_blocks):
                        hk = first if k == 0 else b.arith(0, 1, k, first, first, first)
                        lv.append(_tuple_leaf(b, g, hk, words[8 * k: 8 * k + 8], c2_15))
                    while len(lv) > 1:
                        lv = [g.hash_prefixed_64(0x01, lv[j] + lv[j + 1]) for j in range(0, len(lv), 2)]
                    roots.append(lv[0])
            else:
                roots = [o["public"][16:24] for o in outs]
            while len(roots) > 1:
                roots = [g.hash_prefixed_64(0x01, roots[k] + roots[k + 1]) for k in range(0, len(roots), 2)]
            return outs[0]["public"][:8] + outs[-1]["public"][8:16] + roots[0] + [first_of(outs[0])]
        return combine

    def _child_has_poseidon_rows(self, level):
        return level > 1                                 # chain leaves are SHA rows and arithmetic only; nodes carry the verifier's Poseidon rows

    def _record_leaf(self):
        t0 = time.perf_counter()
        b = CircuitBuilder(self.prover, n_wires=SHA_GATE_WIRES)
        g = Sha256Rows(b)
        sample_height = 1 << (7 * (self.n_groups - 1))                                   # the smallest height with this many varint bytes
        fields = [bytes(n) for n in self.field_lengths]
        fields[6] = b"\x0a\x20" + bytes(32)
        for v in _chain_leaf_statement(b, g, [0] * 8, sample_height, [fields] * self.leaf_blocks, self.n_groups, self.defer):
            b.public_input(v)
        self.leaf_program = b.program()
        self.leaf_circuit = self.leaf_program.setup(self.prover)
        self.map_circuits = [self.leaf_program.setup(p) for p in self.map_provers]
        self.record_seconds["leaf"] = round(time.perf_counter() - t0, 3)

    def prove_leaf(self, start_hash, first_height, headers, which=0):
        """(proof, public) for ONE leaf: the statement is about the chain that starts at start_hash / first_height and carries these headers' OTHER
        bytes — the hash inside each last_block_id, the height fields and the data-hash prefixes are built in-circuit, the caller's copies of them are
        not read here (prove_chain / _map_chain compare them with what the circuit builds and refuse a mismatch; a leaf proved at the wrong place
        does not connect in the nodes)"""
        if self.leaf_program is None:
            self._record_leaf()
        if len(headers) != self.leaf_blocks or any(tuple(len(bytes(f)) for f in h) != self.field_lengths for h in headers):
            raise ValueError("a leaf takes leaf_headers headers whose field encodings have the recorded lengths")
        prover, circuit = (self.prover, self.leaf_circuit) if which == 0 else (self.map_provers[which - 1], self.map_circuits[which - 1])
        vals = self.leaf_program.evaluate(self.consts, _chain_leaf_inputs(start_hash, first_height, headers, self.n_groups), threads=1)
        dw, public = self.leaf_program.device_witness(prover, vals, reuse=True)
        return circuit.prove_(dw, self.nq, self.pw, public=public), public

    @staticmethod
    def header_hash(fields):
        """host restatement of a header hash (RFC 6962 root over the field encodings): what links the leaves on the host side"""
        import hashlib

        def tree(xs):
            if len(xs) == 1:
                return hashlib.sha256(b"\x00" + xs[0]).digest()
            k = 1 << ((len(xs) - 1).bit_length() - 1)
            return hashlib.sha256(b"\x01" + tree(xs[:k]) + tree(xs[k:])).digest()
        return tree([bytes(f) for f in fields])

    def prove_chain(self, start_hash, first_height, headers):
        """headers: the chain after the header with hash start_hash (each header's fields 2 and 4 must already be the real ones: height encoding
        and last_block_id naming its predecessor — what a node serves).  Returns the prove_range-style dict; public = start hash, end hash,
        commitment (8 words each), first height."""
        n, B = len(headers), self.leaf_blocks
        if n % B:
            raise ValueError("the chain is not a whole number of leaves")
        if self.leaf_program is None:
            self._record_leaf()
        hashes = [bytes(start_hash)] + [self.header_hash(h) for h in headers]           # host side: the start hash of every leaf
        t0 = time.perf_counter()
        leaves = self._map_chain(hashes, first_height, headers, 0, n)
        t1 = time.perf_counter()
        levels = []
        if len(leaves) == 1:
            if self.defer:
                raise ValueError("with deferred tuple hashing a chain needs at least two leaves (the level-1 node hashes the tuples)")
            root_proof, key = leaves[0], self.leaf_circuit.cap()
            public = [int(v) for v in importlib.import_module(__package__).proof_public_inputs(root_proof)]
        else:
            root_proof, public, key, _ = self.reduce(leaves, levels)
        t2 = time.perf_counter()
        return {"root_proof": root_proof, "public": public, "key": key, "leaves": len(leaves), "map_seconds": round(t1 - t0, 4),
                "reduce_seconds": round(t2 - t1, 4), "levels": levels, "record_seconds": dict(self.record_seconds),
                "end_hash": b"".join(struct.pack(">I", v) for v in public[8:16]), "commitment": b"".join(struct.pack(">I", v) for v in public[16:24])}

    def _map_chain(self, hashes, first_height, headers, lo, hi):
        """leaf proofs of headers[lo:hi] (a whole number of leaves), on every prover this object has; hashes[k] = hash of the header BEFORE header k"""
        B = self.leaf_blocks
        for k in range(lo, hi, B):
            if any(tuple(len(bytes(f)) for f in h) != self.field_lengths for h in headers[k:k + B]):
                raise ValueError("a leaf takes leaf_headers headers whose field encodings have the recorded lengths")
        # The leaf circuit BUILDS each header's height field, the prefix of its data-hash field and the hash inside its last_block_id from (first
        # height, the predecessor's hash as computed in-circuit): those bytes of the caller's headers are not inputs.  Headers that disagree with
        # what the circuit will hash are refused here — otherwise the proof would silently be about the chain the circuit implies, not the one given
        # (found by profiles/soak_combined_skip.py: a flipped bit in the first header of a leaf was ignored).
        from .blobstream import encode_varint
        for k in range(lo, hi):
            f = headers[k]
            if bytes(f[4])[:34] != b"\x0a\x20" + hashes[k]:
                raise ValueError(f"header {k} does not name its predecessor's hash in its last_block_id")
            if bytes(f[2]) != b"\x08" + encode_varint(first_height + k) or bytes(f[6])[:2] != b"\x0a\x20":
                raise ValueError(f"header {k}: the height field is not the encoding of {first_height + k}, or the data-hash field is malformed")
        return self._map_inputs([_chain_leaf_inputs(hashes[k], first_height + k, headers[k:k + B], self.n_groups) for k in range(lo, hi, B)])

    def prove_chain_distributed(self, start_hash, first_height, headers, device=None, comm=None):
        """prove_chain with the work spread over the ranks (mapreduce.reduce_tree_distributed): rank r proves and folds the r-th contiguous part of
        the chain on its own GPU, ONE all-gather of the node proofs, rank 0 folds the root (its nodes check that the parts are adjacent).  Every rank
        passes the whole chain.  Returns the prove_chain dict on rank 0 (root_proof None elsewhere)."""
        mrm = importlib.import_module(__package__ + ".mapreduce")
        rank, world = mrm._world(comm)
        n, B = len(headers), self.leaf_blocks
        if n % (B * world):
            raise ValueError("the chain is not a whole number of leaves per rank")
        if self.leaf_program is None:
            self._record_leaf()
        hashes = [bytes(start_hash)] + [self.header_hash(h) for h in headers]
        per = n // world
        state, levels = {}, []

        def fold_local(_):
            t0 = time.perf_counter()
            leaves = self._map_chain(hashes, first_height, headers, rank * per, (rank + 1) * per)
            state["map_seconds"] = round(time.perf_counter() - t0, 4)
            if len(leaves) == 1:
                if self.defer:
                    raise ValueError("with deferred tuple hashing every rank needs at least two leaves")
                state.update(key=self.leaf_circuit.cap(), level=1, public=None, span=B)
                return leaves[0]
            proof, public, key, level = self.reduce(leaves, levels)
            state.update(key=key, level=level, public=public, span=self.last_span)
            return proof

        def fold_root(nodes):
            proof, public, key, _ = self.reduce(nodes, levels, child_key=state["key"], level=state["level"], span=state["span"])
            state.update(key=key, public=public)
            return proof
        t0 = time.perf_counter()
        out = mrm.reduce_tree_distributed(fold_local, fold_root, [b""] * (per // B), 1 << 18, device=device, comm=comm)
        root, public = out["root_proof"], state["public"]
        if root is not None and public is None:
            public = [int(v) for v in importlib.import_module(__package__).proof_public_inputs(root)]
        return {"root_proof": root, "public": public, "key": state["key"] if root is not None else None, "leaves": n // B, "ranks": world,
                "map_seconds": state["map_seconds"], "seconds": round(time.perf_counter() - t0, 4), "levels": levels,
                "record_seconds": dict(self.record_seconds),
                "end_hash": b"".join(struct.pack(">I", v) for v in public[8:16]) if root is not None else None,
                "commitment": b"".join(struct.pack(">I", v) for v in public[16:24]) if root is not None else None}

    def synthetic_chain(self, n_headers, first_height=None, start_hash=bytes(32)):
        """a well-formed chain of the recorded shape (all-zero opaque fields, real height encodings and links): (headers, end hash)"""
        from .blobstream import encode_varint
        first = (1 << (7 * (self.n_groups - 1))) if first_height is None else int(first_height)
        prev, out = bytes(start_hash), []
        for k in range(n_headers):
            f = [bytes(n) for n in self.field_lengths]
            f[2] = b"\x08" + encode_varint(first + k)
            f[4] = b"\x0a\x20" + prev + bytes(self.field_lengths[4] - 34)
            f[6] = b"\x0a\x20" + bytes(32)
            if len(f[2]) != self.field_lengths[2]:
                raise ValueError("the recorded height field length does not fit these heights")
            out.append(f)
            prev = self.header_hash(f)
        return out, prev

    def expected_key(self, n_headers):
        """the verifier's own setup (see DataCommitmentMapReduce.expected_key): the root circuit's key for a chain of n_headers, from a synthetic
        chain of the recorded shape"""
        first = 1 << (7 * (self.n_groups - 1))
        headers, _ = self.synthetic_chain(n_headers, first)
        return self.prove_chain(bytes(32), first, headers)["key"]

    def verify_chain(self, root_proof, key, start_hash, end_hash, commitment, first_height):
        """the consumer (key: from the verifier's own expected_key, never from the prover): the proof says that the headers walked from start_hash (at heights first_height, first_height + 1, ...) end at end_hash and
        commit their data hashes to `commitment`; key = the root circuit's verifying key (it fixes the number of headers)"""
        public = list(struct.unpack(">8I", bytes(start_hash))) + list(struct.unpack(">8I", bytes(end_hash))) + \
            list(struct.unpack(">8I", bytes(commitment))) + [int(first_height)]
        return bool(self.prover.plonk_verify(root_proof, key, self.nq, self.pw, public=public))
