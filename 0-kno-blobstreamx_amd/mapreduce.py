"""MapReduce over header batches across GPUs (upstream name
recalled: plonky2x ``mapreduce`` generator — reference file:line NONE, the mount is empty).

Map: leaf i of a skip/data-commitment circuit is proved on rank ``i % world`` (one process
per GPU, independent leaves, no data-path collective while proving).  Exchange: ONE all-gather
of the fixed-size, zero-padded leaf-proof blobs over RCCL/xGMI (``torch.distributed``
backend "nccl" on GPUs, "gloo" in the CPU tests).  Reduce: every rank (or rank 0) then holds
all leaf proofs in leaf order for the recursive aggregation.

Payloads are O(100 KiB) per leaf: the all-gather is latency-bound, so it is issued once per
MapReduce level with all of a rank's leaves packed into one tensor, not once per leaf.
"""
import struct

import torch
import torch.distributed as dist

HEADER = struct.Struct("<QQ")  # (leaf index, payload length)


def leaves_of_rank(n_leaves, rank, world):
    """leaf i -> rank i % world (round-robin keeps ranks within one leaf of each other)"""
    return list(range(rank, n_leaves, world))


def pack_leaves(blobs, padded_len, n_rows=None):
    """[(leaf_index, bytes)] -> uint8 tensor [n_rows or n_local, HEADER + padded_len]; unused rows carry
    leaf index 2^64-1 (so they can be told from leaf 0) and every record is zero padded.  Built with numpy:
    torch's host fill/copy kernels spin up an OpenMP pool on a 128-core box, 100 ms for 4 MiB."""
    import numpy as np
    rec = HEADER.size + padded_len
    rows = len(blobs) if n_rows is None else n_rows
    out = np.zeros((rows, rec), dtype=np.uint8)
    out[:, : HEADER.size] = np.frombuffer(HEADER.pack(2**64 - 1, 0), dtype=np.uint8)
    for r, (idx, b) in enumerate(blobs):
        if len(b) > padded_len:
            raise ValueError(f"leaf {idx}: proof of {len(b)} bytes exceeds padded_len {padded_len}")
        out[r, : HEADER.size] = np.frombuffer(HEADER.pack(idx, len(b)), dtype=np.uint8)
        out[r, HEADER.size: HEADER.size + len(b)] = np.frombuffer(b, dtype=np.uint8)
    return torch.from_numpy(out)


def _world(comm):
    """(rank, world): from the prover's own communicator (glp_comm_init) when one is given, else torch.distributed"""
    if comm is not None:
        return comm.comm_rank, comm.comm_size
    if dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def allgather_leaf_proofs(local_blobs, n_leaves, padded_len, device=None, comm=None):
    """Every rank passes its own [(leaf_index, proof_bytes)]; returns the list of all
    ``n_leaves`` proofs in leaf order on every rank.  Ranks may own different leaf counts
    (n_leaves % world != 0): shorter ranks pad with empty records.
    comm: a Prover whose ctx owns an RCCL communicator (``Prover.comm_init``) — the exchange then runs behind the C ABI
    (``glp_allgather_proofs``), exactly what a Rust/C++ host would call, instead of through torch.distributed."""
    _, world = _world(comm)
    per_rank = (n_leaves + world - 1) // world
    if len(local_blobs) > per_rank:
        raise ValueError(f"{len(local_blobs)} local leaves but at most {per_rank} per rank")
    rec = HEADER.size + padded_len
    buf = pack_leaves(local_blobs, padded_len, n_rows=per_rank)
    if comm is not None:
        # staging first (no collective inside), then agree: no rank may fail locally inside the exchange while its peers wait in RCCL
        ok = 1
        try:
            comm.comm_reserve(per_rank * rec)
        except Exception:  # noqa: BLE001 — reported below, once every rank knows
            ok = 0
        if world > 1:
            ok = int(comm.allreduce_min([ok])[0])
        if not ok:
            raise RuntimeError("the exchange's staging could not be allocated on some rank: no leaf proofs were exchanged")
        rows = comm.allgather_bytes(buf.numpy().tobytes()).reshape(world * per_rank, rec)
    else:
        if device is not None:
            buf = buf.to(device)
        if world > 1:
            out = torch.empty((world * per_rank, rec), dtype=torch.uint8, device=buf.device)
            dist.all_gather_into_tensor(out, buf)
        else:
            out = buf
        rows = out.cpu().numpy()                  # one d2h copy; slicing below is memcpy, not per-byte Python
    proofs = [None] * n_leaves
    for row in rows:
        idx, ln = HEADER.unpack(row[: HEADER.size].tobytes())
        if idx == 2**64 - 1:
            continue
        if idx >= n_leaves or proofs[idx] is not None or ln > rec - HEADER.size:
            raise ValueError(f"bad or duplicate leaf index {idx}")
        proofs[idx] = row[HEADER.size: HEADER.size + ln].tobytes()
    missing = [i for i, p in enumerate(proofs) if p is None]
    if missing:
        raise ValueError(f"leaf proofs missing after all-gather: {missing[:8]}")
    return proofs


def map_prove_gather(prove_leaf, n_leaves, padded_len, device=None, comm=None):
    """The Map + exchange steps of a MapReduce proof: this rank proves leaves
    ``leaves_of_rank(n_leaves, rank, world)`` with ``prove_leaf(i) -> bytes`` — or a LIST of such callables,
    one per concurrent prover of this rank, each run by its own host thread — (e.g.
    ``PlonkCircuit.prove`` of the leaf circuit on leaf i's witness), then every rank receives all
    proofs in leaf order through ONE all-gather.  What is returned is the ordered proof list; the Reduce step
    consumes it: ``reduce_verify`` (native verification), ``reduce_aggregate`` (digest tree proof), ``reduce_recursive`` /
    ``reduce_tree`` (the leaf proofs verified in-circuit: verifier_circuit.py)."""
    rank, world = _world(comm)
    ids = leaves_of_rank(n_leaves, rank, world)
    mine, failure = None, None
    try:
        mine = _prove_local(prove_leaf, ids)
    except Exception as e:  # noqa: BLE001 — re-raised below, after every rank knows
        failure = e
    # a rank whose map step failed must not leave its peers waiting in the all-gather: agree on success first
    if world > 1:
        if comm is not None:
            everyone_ok = int(comm.allreduce_min([0 if failure is not None else 1])[0])
        else:
            flag = torch.tensor([0 if failure is not None else 1], dtype=torch.int32, device=device if device is not None else "cpu")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            everyone_ok = int(flag.item())
        if everyone_ok == 0 and failure is None:
            raise RuntimeError("map step failed on another rank: no leaf proofs were exchanged")
    if failure is not None:
        raise failure
    return allgather_leaf_proofs(mine, n_leaves, padded_len, device=device, comm=comm)


def _prove_local(prove_leaf, ids):
    """this rank's leaf proofs, [(leaf index, bytes)] in leaf order"""
    if callable(prove_leaf):
        return [(i, prove_leaf(i)) for i in ids]
    else:
        # several provers on this GPU (one ctx = one stream each, driven by one host thread each): the
        # launch- and latency-bound phases of one leaf proof (transcript round trips, the top Merkle levels,
        # FRI queries) overlap with the throughput-bound phases of another.  ctypes releases the GIL.
        from concurrent.futures import ThreadPoolExecutor
        workers = list(prove_leaf)
        with ThreadPoolExecutor(len(workers)) as ex:
            futs = [ex.submit(lambda w=w, sub=ids[k::len(workers)]: [(i, w(i)) for i in sub]) for k, w in enumerate(workers)]
            return sorted((p for f in futs for p in f.result()), key=lambda t: t[0])


def reduce_verify(verify_leaf, proofs, device=None, comm=None):
    """The Reduce step in its NATIVE form (the recursive forms are reduce_recursive / reduce_tree / reduce_tree_distributed): every gathered leaf
    proof is checked by the native verifier (``verify_leaf(proof_bytes) -> bool``, e.g. ``PlonkCircuit.verify``, or a list of them — one per
    ctx — run by one host thread each), the work split across
    ranks — rank r checks the leaves PROVED BY rank r+1 — and the verdicts are combined with one
    all-reduce(MIN).  The result is a boolean ("all leaves verify"), not a succinct proof: that is what the recursive forms produce."""
    rank, world = _world(comm)
    owner = (rank + 1) % world
    ids = list(range(owner, len(proofs), world))
    if callable(verify_leaf):
        ok = int(all([verify_leaf(proofs[i]) for i in ids]))
    else:
        # a list of verifiers (one ctx each): host threads, the native verifier runs without the GIL
        from concurrent.futures import ThreadPoolExecutor
        vs = list(verify_leaf)
        with ThreadPoolExecutor(len(vs)) as ex:
            futs = [ex.submit(lambda v=v, sub=ids[k::len(vs)]: all([v(proofs[i]) for i in sub])) for k, v in enumerate(vs)]
            ok = int(all([f.result() for f in futs]))
    if world > 1:
        if comm is not None:
            ok = int(comm.allreduce_min([ok])[0])
        else:
            t = torch.tensor([ok], dtype=torch.int32, device=device if device is not None else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            ok = int(t.item())
    return bool(ok)


ZERO_DIGEST = [0, 0, 0, 0]      # pads the leaf list to a power of two in the aggregation tree


def reduce_aggregate(prover, verify_leaf, proofs, device=None, comm=None, num_queries=28, pow_bits=16):
    """The Reduce step with its first in-circuit part (recursion.py).  (1) every gathered leaf proof is verified natively, split
    across ranks, verdicts all-reduced (``reduce_verify``: host arithmetic — the part that is NOT in-circuit yet);
    (2) rank 0 hashes each leaf proof to its 4-word digest (statement + commitments) and PROVES, on the GPU, the binary Poseidon
    tree over the digests: one succinct proof whose public inputs are the digests and the root.  Returns a dict: ``ok`` (every leaf
    verifies), and on rank 0 ``root_proof``, ``root``, ``digests``, ``key`` (the aggregation circuit's verifying key)."""
    import importlib
    rank, _ = _world(comm)
    ok = reduce_verify(verify_leaf, proofs, device=device, comm=comm)
    out = {"ok": ok}
    if rank == 0 and ok:
        rec = importlib.import_module(__package__ + ".recursion")
        digests = [prover.proof_digest(p) for p in proofs]
        size = 1 << max(1, (len(digests) - 1).bit_length())
        digests += [ZERO_DIGEST] * (size - len(digests))
        root_proof, root, key = rec.aggregate(prover, digests, num_queries, pow_bits)
        out.update(root_proof=root_proof, root=root, digests=digests, key=key)
    return out


def verify_aggregate(prover, root_proof, key, digests, root, min_queries=28, min_pow_bits=16):
    """the consumer of a Reduce: the root proof is about exactly these leaf digests and this root, for the n-leaf aggregation
    circuit ``key`` (recursion.aggregation_key).  The leaf proofs behind the digests are checked by whoever holds them."""
    public = [int(v) for d in digests for v in d] + [int(v) for v in root]
    return prover.plonk_verify(root_proof, key, min_queries, min_pow_bits, public=public)


def reduce_recursive(prover, proofs, leaf_key, num_queries, pow_bits, n_wires, n_routed=None, n_public=0, cap_height=4,
                     root_queries=28, root_pow_bits=16):
    """The Reduce step as a RECURSION (verifier_circuit.py): rank 0 proves, on the GPU, one circuit that verifies every gathered leaf proof
    in-circuit (transcript, proof of work, every Merkle opening, FRI combination / folds / final polynomial, PLONK identity) and hashes the
    leaf digests into a root.  Returns {"root_proof", "public" (leaf public inputs + digests + root), "key" (the recursion circuit's verifying
    key), "stats"}; whoever checks `root_proof` against (`key`, `public`) needs none of the leaf proofs.  A leaf proof that does not verify
    makes the circuit impossible to lay down (ValueError)."""
    import importlib
    vc = importlib.import_module(__package__ + ".verifier_circuit")
    size = 1 << max(0, (len(proofs) - 1).bit_length())
    if size != len(proofs):
        raise ValueError("reduce_recursive takes a power-of-two number of leaf proofs")
    ck, dw, public, stats = vc.recursive_aggregation_circuit(prover, proofs, leaf_key, num_queries, pow_bits, n_wires, n_routed, n_public, cap_height)
    try:
        root_proof = ck.prove_(dw, root_queries, root_pow_bits, public=public)
        return {"root_proof": root_proof, "public": public, "key": ck.cap(), "stats": stats}
    finally:
        dw.free()
        ck.free()


def reduce_tree(prover, proofs, leaf, poseidon_consts, fan_in=2, node_queries=28, node_pow_bits=16):
    """The Reduce step as a TREE of recursions (upstream's shape: plonky2x mapreduce folds leaf proofs pairwise).  Level 1 nodes verify
    `fan_in` leaf proofs each in-circuit; every later level verifies `fan_in` proofs OF THE PREVIOUS LEVEL — recursion proofs, i.e. Poseidon-row
    circuits, whose 118 row constraints are then part of the in-circuit identity — until one proof is left.
    leaf = {"key", "num_queries", "pow_bits", "n_wires", ["n_routed", "n_public", "cap_height"]}: the leaf circuit and its parameters.
    Returns {"root_proof", "public", "key", "levels": [per level: nodes, rows, prove seconds, build seconds]}.  Every node of one level shares
    one circuit (and key); a verifier of the root needs the root proof, its public inputs and the last level's key only."""
    import importlib
    import time
    vc = importlib.import_module(__package__ + ".verifier_circuit")
    params = dict(leaf_key=leaf["key"], num_queries=leaf["num_queries"], pow_bits=leaf["pow_bits"], n_wires=leaf["n_wires"],
                  n_routed=leaf.get("n_routed"), n_public=leaf.get("n_public", 0), cap_height=leaf.get("cap_height", 4), poseidon_consts=None)
    cur, levels = list(proofs), []
    if len(cur) < fan_in or fan_in < 1 or (fan_in & (fan_in - 1)):
        raise ValueError("reduce_tree: fan_in must be a power of two and at most the number of proofs")
    while True:
        if len(cur) % fan_in:
            raise ValueError("reduce_tree: the number of proofs at a level is not a multiple of fan_in")
        nxt, key, pub_len, t_build, t_prove, rows = [], None, None, 0.0, 0.0, 0
        for k in range(0, len(cur), fan_in):
            t0 = time.perf_counter()
            ck, dw, public, stats = vc.recursive_aggregation_circuit(prover, cur[k:k + fan_in], **params)
            t1 = time.perf_counter()
            try:
                proof = ck.prove_(dw, node_queries, node_pow_bits, public=public)
                t2 = time.perf_counter()
                k_here = ck.cap()
                if key is not None and (len(public) != pub_len or not (k_here == key).all()):
                    raise RuntimeError("nodes of one level built different circuits")
                key, pub_len, rows = k_here, len(public), stats["rows"]
                nxt.append((proof, public))
                t_build += t1 - t0
                t_prove += t2 - t1
            finally:
                dw.free()
                ck.free()
        levels.append({"nodes": len(nxt), "verifies": "leaf proofs" if params["poseidon_consts"] is None else "recursion proofs", "rows": rows,
                       "build_circuit_seconds": round(t_build, 3), "prove_seconds": round(t_prove, 4)})
        if len(nxt) == 1:
            return {"root_proof": nxt[0][0], "public": nxt[0][1], "key": key, "levels": levels}
        cur = [p for p, _ in nxt]
        params = dict(leaf_key=key, num_queries=node_queries, pow_bits=node_pow_bits, n_wires=136, n_routed=80, n_public=pub_len, cap_height=1,
                      poseidon_consts=poseidon_consts)


def reduce_tree_distributed(fold_local, fold_root, my_proofs, padded_len, device=None, comm=None):
    """The Reduce step as a two-level recursion SPREAD OVER THE RANKS.  (1) every rank folds ITS OWN leaf proofs into one node proof
    (``fold_local(proofs) -> bytes``: a recursion proof over the leaves this GPU proved — they never leave it); (2) ONE all-gather of the
    `world` node proofs (a second MapReduce level whose leaf r is rank r's node: the same agreement-then-gather as the Map exchange, so a rank
    whose fold fails takes every rank out before the collective); (3) rank 0 folds the node proofs into the root
    (``fold_root(node_proofs) -> bytes``).  The exchange carries one proof per rank instead of every leaf proof.  world must be a power of two
    and every rank must hold the same number of leaves (one level-1 circuit).  Returns {"nodes": all node proofs in rank order, "root_proof":
    bytes on rank 0 (the node itself when world == 1), None elsewhere}."""
    rank, world = _world(comm)
    if world & (world - 1):
        raise ValueError("reduce_tree_distributed: the number of ranks must be a power of two")
    counts = [len(my_proofs)]
    if world > 1:
        if comm is not None:
            lo, hi = int(comm.allreduce_min([len(my_proofs)])[0]), -int(comm.allreduce_min([2**32 - len(my_proofs)])[0]) + 2**32
        else:
            t = torch.tensor([len(my_proofs), -len(my_proofs)], dtype=torch.int64, device=device if device is not None else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            lo, hi = int(t[0].item()), -int(t[1].item())
        counts = [lo, hi]
    if min(counts) != max(counts) or not my_proofs:
        raise ValueError("reduce_tree_distributed: every rank must hold the same (non-zero) number of leaf proofs")
    nodes = map_prove_gather(lambda i: fold_local(my_proofs), world, padded_len, device=device, comm=comm)      # leaf i of this level = rank i
    if world == 1:
        return {"nodes": nodes, "root_proof": nodes[0]}
    return {"nodes": nodes, "root_proof": fold_root(nodes) if rank == 0 else None}


class RecursionFolders:
    """fold_local / fold_root for reduce_tree_distributed, as real recursions (verifier_circuit.RecursionProgram): level 1 verifies leaf proofs
    of the circuit `leaf` = {"key", "num_queries", "pow_bits", "n_wires", ["n_routed", "n_public", "cap_height", "sha"]} in-circuit, level 2
    verifies level-1 proofs (Poseidon-row circuits) in-circuit.  The circuits are recorded once per fan-in and kept (record_seconds says what
    that cost); every later fold is witness evaluation + proving.  After fold_local / fold_root, `public` holds the public inputs of the
    proof just made and `key` the verifying key it belongs to."""

    def __init__(self, prover, leaf, poseidon_consts, node_queries=28, node_pow_bits=16, compact=False):
        """compact: every node states ONLY the Poseidon Merkle root of the LEAF digests below it (4 public words at every level: the root proof's
        statement does not grow with the number of leaves) instead of its children's public inputs, their digests and the root.  A leaf's digest
        binds its header, public inputs and caps, so the 4 words are checked against the leaf proofs' digests alone (recursion.merkle_root_host);
        a level-2 node combines its children's 4-word roots — the tree over all leaves, when every rank folds the same power-of-two count."""
        self.prover, self.leaf, self.consts = prover, dict(leaf), poseidon_consts
        self.nq, self.pw, self.compact = node_queries, node_pow_bits, bool(compact)
        self.programs, self.record_seconds = {}, {}
        self.public = self.key = None
        self.local_key = self.local_public_len = None

    def _program(self, level, proofs, **kw):
        import importlib
        import time
        vc = importlib.import_module(__package__ + ".verifier_circuit")
        k = (level, len(proofs))
        if k not in self.programs:
            t0 = time.perf_counter()
            self.programs[k] = vc.RecursionProgram(self.prover, proofs, poseidon_values=self.consts, **kw)
            self.record_seconds[k] = round(time.perf_counter() - t0, 3)
        return self.programs[k]

    def fold_local(self, proofs):
        lf = self.leaf
        rp = self._program(1, proofs, leaf_key=lf["key"], num_queries=lf["num_queries"], pow_bits=lf["pow_bits"], n_wires=lf["n_wires"],
                           n_routed=lf.get("n_routed"), n_public=lf.get("n_public", 0), cap_height=lf.get("cap_height", 4),
                           child_is_recursion=bool(lf.get("poseidon", False)), child_sha=bool(lf.get("sha", False)),
                           combine=(lambda b, outs: _root_of(b, [o["digest"] for o in outs])) if self.compact else None)
        proof, self.public = rp.prove(proofs, self.nq, self.pw)
        self.key = self.local_key = rp.key()
        self.local_public_len = len(self.public)
        return proof

    def fold_root(self, node_proofs):
        if self.local_key is None:
            raise RuntimeError("fold_root before fold_local: the level-1 circuit (its key, its public-input count) is not known yet")
        rp = self._program(2, node_proofs, leaf_key=self.local_key, num_queries=self.nq, pow_bits=self.pw, n_wires=rp_wires(self.programs),
                           n_routed=80, n_public=self.local_public_len, cap_height=1, child_is_recursion=True,
                           combine=(lambda b, outs: _root_of(b, [o["public"] for o in outs])) if self.compact else None)
        proof, self.public = rp.prove(node_proofs, self.nq, self.pw)
        self.key = rp.key()
        return proof

    def free(self):
        for rp in self.programs.values():
            rp.free()
        self.programs = {}


def _root_of(b, digests):
    """Poseidon Merkle root (two_to_one levels) of a power-of-two list of 4-variable digests, on builder b"""
    level = [list(d) for d in digests]
    assert all(len(d) == 4 for d in level) and len(level) & (len(level) - 1) == 0
    while len(level) > 1:
        level = [b.two_to_one(level[2 * k], level[2 * k + 1]) for k in range(len(level) // 2)]
    return level[0]


def rp_wires(programs):
    """wire count of the level-1 recursion circuits (what the level-2 verifier is told its children have)"""
    return next(rp.circuit.n_wires for (level, _), rp in programs.items() if level == 1)
