"""The N>1 path on CPU: two processes over gloo.  Covers the MapReduce leaf sharding and the
all-gather of padded leaf proofs (row a11), and bench.py's rank/time reduction arithmetic."""
import hashlib
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def fake_proof(i):
    """deterministic stand-in payload of leaf-dependent length"""
    return hashlib.sha256(b"leaf%d" % i).digest() * (1 + i % 5) + bytes([i % 256])


def _worker(rank, world, port, n_leaves, q):
    import importlib
    import __graft_entry__ as graft
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mr = importlib.import_module(graft.PKG_NAME + ".mapreduce")
    mine = mr.leaves_of_rank(n_leaves, rank, world)
    local = [(i, fake_proof(i)) for i in mine]
    proofs = mr.allgather_leaf_proofs(local, n_leaves, padded_len=200)
    ok = proofs == [fake_proof(i) for i in range(n_leaves)]
    # the map + exchange wrapper used by bench.py --mapreduce (prove_leaf stands in for PlonkCircuit.prove)
    ok = ok and mr.map_prove_gather(fake_proof, n_leaves, padded_len=200) == proofs
    # Reduce: each rank checks the leaves proved by the next rank, verdicts combined by all-reduce(MIN);
    # one bad leaf anywhere must flip the verdict on EVERY rank
    checked = []
    ok = ok and mr.reduce_verify(lambda p: (checked.append(p) or True), proofs) is True
    ok = ok and checked == [fake_proof(i) for i in range((rank + 1) % world, n_leaves, world)]
    bad_leaf = n_leaves - 1
    ok = ok and mr.reduce_verify(lambda p: p != fake_proof(bad_leaf), proofs) is False
    # a map step that fails on ONE rank must raise on EVERY rank before the all-gather (no peer left waiting in the collective)
    def flaky(i):
        if rank == 1:
            raise ValueError("leaf prover failed")
        return fake_proof(i)
    try:
        mr.map_prove_gather(flaky, n_leaves, padded_len=200)
        ok = ok and (world == 1 or n_leaves < 2)
    except ValueError:
        ok = ok and rank == 1
    except RuntimeError as e:
        ok = ok and rank != 1 and "another rank" in str(e)
    # Reduce as a two-level tree over the ranks: each rank folds its own leaves, ONE gather of the node blobs, rank 0 folds the root
    if n_leaves % world == 0 and n_leaves >= world:
        fold = lambda ps: hashlib.sha256(b"".join(ps)).digest()
        out = mr.reduce_tree_distributed(fold, fold, [fake_proof(i) for i in mine], padded_len=64)
        want_nodes = [fold([fake_proof(i) for i in mr.leaves_of_rank(n_leaves, r, world)]) for r in range(world)]
        ok = ok and out["nodes"] == want_nodes and out["root_proof"] == ((fold(want_nodes) if world > 1 else want_nodes[0]) if rank == 0 else None)
        def bad_fold(ps):
            if rank == world - 1:
                raise ValueError("a leaf proof does not verify")
            return fold(ps)
        try:
            mr.reduce_tree_distributed(bad_fold, fold, [fake_proof(i) for i in mine], padded_len=64)
            ok = False
        except ValueError:
            ok = ok and rank == world - 1
        except RuntimeError as e:
            ok = ok and rank != world - 1 and "another rank" in str(e)
    elif world > 1:
        try:
            mr.reduce_tree_distributed(lambda ps: b"x", lambda ps: b"y", [fake_proof(i) for i in mine], padded_len=64)
            ok = False
        except ValueError as e:
            ok = ok and "same" in str(e)
    # the bench's max-over-ranks time reduction
    t = torch.tensor([1.0 + rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    q.put((rank, ok, mine, float(t.item())))
    dist.barrier()
    dist.destroy_process_group()


def _golden_leaf_proofs(n_leaves):
    """real-format leaf proofs for the exchange: the committed golden circuit proof (made by the GPU prover, tests/golden/
    proofs.json) on the even leaves and a tampered copy (one flipped word) on leaf 3 when it exists"""
    import json
    import numpy as np
    with open(os.path.join(ROOT, "tests", "golden", "proofs.json")) as f:
        g = json.load(f)["plonk"]
    good = bytes.fromhex(g["proof"])
    w = np.frombuffer(good, dtype="<u8").copy()
    w[len(w) // 2] ^= np.uint64(2)
    return g, [w.tobytes() if i == 3 else good for i in range(n_leaves)]


def _worker_real_proofs(rank, world, port, n_leaves, q):
    """the Map output is REAL proof bytes: every rank contributes its leaves' proofs, the all-gather reassembles them, and
    Reduce runs the product's native HOST verifier (no GPU) on the gathered blobs, split across ranks"""
    import importlib
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import __graft_entry__ as graft
    from conftest import poseidon_consts
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pkg = graft.load_package()
    mr = importlib.import_module(graft.PKG_NAME + ".mapreduce")
    g, blobs = _golden_leaf_proofs(n_leaves)
    consts = poseidon_consts("small")
    cap = np.array(g["circuit_cap"], dtype=np.uint64)
    mine = mr.leaves_of_rank(n_leaves, rank, world)
    proofs = mr.map_prove_gather(lambda i: blobs[i], n_leaves, padded_len=len(blobs[0]) + 64)
    ok = proofs == blobs
    verify = lambda p: pkg.plonk_verify_host(consts, p, cap, g["queries"], g["pow_bits"])[0]
    verdict_all = mr.reduce_verify(verify, proofs)                         # leaf 3 is tampered (when n_leaves > 3)
    clean = [blobs[0]] * n_leaves
    verdict_clean = mr.reduce_verify(verify, clean)
    q.put((rank, ok, verdict_all, verdict_clean, mine))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_leaves", [6, 3])
def test_mapreduce_real_proof_blobs_two_ranks(n_leaves):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=_worker_real_proofs, args=(r, world, port, n_leaves, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, ok, verdict_all, verdict_clean, mine in res:
        assert ok, f"rank {rank}: gathered proofs differ from the leaves' proofs"
        assert verdict_clean is True, f"rank {rank}: the golden proofs were not accepted"
        assert verdict_all is (n_leaves <= 3), f"rank {rank}: tampered leaf 3 must flip the verdict on every rank"


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("n_leaves", [8, 7, 1])
def test_mapreduce_allgather_two_ranks(n_leaves):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_leaves, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    owned = []
    for rank, ok, mine, tmax in res:
        assert ok, f"rank {rank} did not reassemble the leaf proofs"
        assert tmax == 2.0
        owned += mine
    assert sorted(owned) == list(range(n_leaves))


def test_single_process_gather_and_errors():
    import importlib
    import __graft_entry__ as graft
    mr = importlib.import_module(graft.PKG_NAME + ".mapreduce")
    blobs = [(i, fake_proof(i)) for i in range(5)]
    assert mr.allgather_leaf_proofs(blobs, 5, 200) == [b for _, b in blobs]
    with pytest.raises(ValueError):
        mr.pack_leaves([(0, b"x" * 300)], 200)
    with pytest.raises(ValueError):
        mr.allgather_leaf_proofs(blobs[:4], 5, 200)       # a leaf is missing
