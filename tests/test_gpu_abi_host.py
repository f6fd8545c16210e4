"""A compiled C++ host (g++, no hipcc, no Python in the data path) drives the hot path through the
C ABI on the GPU — the shape of the Rust/C++ integration INTEGRATION.md describes."""
import os
import subprocess
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import __graft_entry__ as graft  # noqa: E402

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build_and_run(pkg, tmp_path, name):
    pkg.load_library()
    exe = tmp_path / name
    libdir = os.path.dirname(pkg.LIB_PATH)
    subprocess.run(["g++", "-std=c++17", "-O2", "-Wall", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "cpp", name + ".cpp"), "-o", str(exe), "-L", libdir, "-lglprover",
                    "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"], check=True)
    r = subprocess.run([str(exe)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), r.stdout


def test_cpp_host_roundtrip(pkg, tmp_path):
    _build_and_run(pkg, tmp_path, "host_roundtrip")


def test_cpp_host_mapreduce_path(pkg, tmp_path):
    """circuit setup, leaf proofs with public inputs, RCCL exchange, statement-bound verification, verdict all-reduce, digests — through the
    C ABI only, from a compiled host: the MapReduce path a Rust host would run"""
    _build_and_run(pkg, tmp_path, "host_mapreduce")


def test_cpp_host_replays_a_recorded_circuit(pkg, prover, tmp_path):
    """Python as the offline circuit compiler only: a DataCommitment leaf (SHA rows + Poseidon rows, 2 blocks) is recorded and exported as raw
    arrays; a g++-compiled host commits it, generates the witness for the shipped inputs (threads + device placement + row fillers), proves and
    verifies — C ABI only — and arrives at the key and public inputs the Python side computes"""
    import hashlib
    import importlib
    import struct
    import numpy as np
    from conftest import poseidon_consts
    dm = importlib.import_module(graft.PKG_NAME + ".data_commitment_mr")
    consts = poseidon_consts("small")
    prover.set_poseidon_constants(*consts)
    mr = dm.DataCommitmentMapReduce(prover, consts, leaf_blocks=2, fan_in=2, num_queries=10, pow_bits=6)
    mr._record_leaf()
    rng = np.random.default_rng(5)
    hs = [77, 78]
    rs = [rng.integers(0, 256, 32, dtype=np.uint8).tobytes() for _ in hs]
    inputs = [w for h, r in zip(hs, rs) for w in dm.tuple_words(h, r)]
    out_dir = tmp_path / "rec"
    mr.leaf_program.export_raw(str(out_dir), sample_inputs=inputs)
    np.concatenate([np.asarray(a, dtype=np.uint64) for a in consts]).astype("<u8").tofile(str(tmp_path / "pc.bin"))
    exe = tmp_path / "host_replay"
    libdir = os.path.dirname(pkg.LIB_PATH)
    subprocess.run(["g++", "-std=c++17", "-O2", "-Wall", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "host_replay.cpp"),
                    "-o", str(exe), "-L", libdir, "-lglprover", "-pthread", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"], check=True)
    r = subprocess.run([str(exe), str(out_dir), str(tmp_path / "pc.bin")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), r.stdout
    lines = dict(ln.split(" ", 1) for ln in r.stdout.strip().splitlines() if " " in ln)
    assert int(lines["key0"]) == int(mr.leaf_circuit.cap()[0])
    public = [int(v) for v in lines["public"].split()]
    lvl = [hashlib.sha256(b"\x00" + int(h).to_bytes(32, "big") + rr).digest() for h, rr in zip(hs, rs)]
    root = hashlib.sha256(b"\x01" + lvl[0] + lvl[1]).digest()
    assert public[:8] == list(struct.unpack(">8I", root)) and public[8:] == dm.tuples_digest(consts, hs, rs, 2)
    assert "out-of-range input -> -7" in r.stdout
    mr.free()


def test_cpp_host_proves_a_recursion_node(pkg, prover, tmp_path):
    """a recursion node from a compiled host: the verifier circuit of two leaf proofs is recorded and exported by Python; the g++ host reads the two
    PROOF FILES, checks the recorded facts about them, builds the inputs from the recorded tags, evaluates the verifier's witness on threads,
    proves and verifies — and its public inputs are the leaf digests and their root, as the Python RecursionProgram computes them"""
    import importlib
    import numpy as np
    import bench
    from conftest import poseidon_consts
    vc = importlib.import_module(graft.PKG_NAME + ".verifier_circuit")
    rec = importlib.import_module(graft.PKG_NAME + ".recursion")
    consts = poseidon_consts("small")
    prover.set_poseidon_constants(*consts)
    c, s_, w = bench.synthetic_circuit(prover, 9, 16)
    ck = pkg.PlonkCircuit(prover, c, s_)
    dw = prover.to_device(w)
    proofs = [ck.prove_(dw, 6, 4) for _ in range(2)]
    rp = vc.RecursionProgram(prover, proofs, ck.cap(), 6, 4, 16, consts)
    _, want_public = rp.prove(proofs, 10, 6)
    out_dir = tmp_path / "node"
    rp.program.export_raw(str(out_dir))
    np.concatenate([np.asarray(a, dtype=np.uint64) for a in consts]).astype("<u8").tofile(str(tmp_path / "pc.bin"))
    files = []
    for k, p in enumerate(proofs):
        f = tmp_path / f"proof_{k}.bin"
        f.write_bytes(p)
        files.append(str(f))
    exe = tmp_path / "host_replay"
    libdir = os.path.dirname(pkg.LIB_PATH)
    subprocess.run(["g++", "-std=c++17", "-O2", "-Wall", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "host_replay.cpp"),
                    "-o", str(exe), "-L", libdir, "-lglprover", "-pthread", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"], check=True)
    r = subprocess.run([str(exe), str(out_dir), str(tmp_path / "pc.bin")] + files, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), r.stdout
    lines = dict(ln.split(" ", 1) for ln in r.stdout.strip().splitlines() if " " in ln)
    assert int(lines["key0"]) == int(rp.key()[0]) and [int(v) for v in lines["public"].split()] == want_public
    digests = [prover.proof_digest(p) for p in proofs]
    assert want_public == [v for d in digests for v in d] + rec.merkle_root_host(prover, digests)
    # a tampered leaf proof: the evaluator refuses (a copy constraint of the verifier circuit fails) and the host reports it
    bad = np.frombuffer(proofs[1], dtype="<u8").copy()
    bad[len(bad) // 2] ^= np.uint64(1)
    (tmp_path / "bad.bin").write_bytes(bad.tobytes())
    r2 = subprocess.run([str(exe), str(out_dir), str(tmp_path / "pc.bin"), files[0], str(tmp_path / "bad.bin")], stdout=subprocess.PIPE,
                        stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r2.returncode != 0 and "FAIL" in r2.stdout
    rp.free()
    dw.free()
    ck.free()


def test_cpp_host_runs_a_whole_range_mapreduce(pkg, prover, tmp_path):
    """the statement-level loop from a compiled host (tests/cpp/host_range.cpp; C ABI only): the data commitment of 8 blocks as a MapReduce of proofs —
    4 leaves proved from their input vectors, level 1 folded locally, the node proofs exchanged through the ctx's RCCL communicator
    (glp_allgather_proofs, one rank), the root folded — from the leaf recording and one node recording per level that Python exported.  The host
    arrives at the root key and the public inputs (commitment = hashlib's, tuples digest) that DataCommitmentMapReduce.prove_range computes, and its
    root proof is accepted by the Python side's verify()."""
    import hashlib
    import importlib
    import struct
    import numpy as np
    from conftest import poseidon_consts
    dm = importlib.import_module(graft.PKG_NAME + ".data_commitment_mr")
    consts = poseidon_consts("small")
    prover.set_poseidon_constants(*consts)
    nq, pw = 6, 4
    mr = dm.DataCommitmentMapReduce(prover, consts, leaf_blocks=2, fan_in=2, num_queries=nq, pow_bits=pw)
    rng = np.random.default_rng(88)
    heights = [5_000_000 + k for k in range(8)]
    roots = [rng.integers(0, 256, 32, dtype=np.uint8).tobytes() for _ in heights]
    out = mr.prove_range(heights, roots)                                   # records the leaf and the two node circuits; the reference result
    assert [lv["nodes"] for lv in out["levels"]] == [2, 1]
    mr.leaf_program.export_raw(str(tmp_path / "leaf"))
    node_dirs = []
    for (level, n_children, span, _), node in sorted(mr.nodes.items(), key=lambda kv: kv[0][0]):
        assert n_children == 2
        d = tmp_path / f"node{level}"
        node.program.export_raw(str(d))
        node_dirs.append(str(d))
    assert len(node_dirs) == 2
    inputs = np.array([[w for h, r in zip(heights[k:k + 2], roots[k:k + 2]) for w in dm.tuple_words(h, r)] for k in range(0, 8, 2)], dtype="<u8")
    inputs.tofile(str(tmp_path / "leaf_inputs.bin"))
    np.concatenate([np.asarray(a, dtype=np.uint64) for a in consts]).astype("<u8").tofile(str(tmp_path / "pc.bin"))
    exe = tmp_path / "host_range"
    libdir = os.path.dirname(pkg.LIB_PATH)
    subprocess.run(["g++", "-std=c++17", "-O2", "-Wall", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "host_range.cpp"),
                    "-o", str(exe), "-L", libdir, "-lglprover", "-pthread", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"], check=True)
    root_file = tmp_path / "root.bin"
    r = subprocess.run([str(exe), str(tmp_path / "pc.bin"), str(nq), str(pw), "2", str(tmp_path / "leaf"), "4", str(tmp_path / "leaf_inputs.bin"), str(root_file)]
                       + node_dirs, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), r.stdout
    assert "exchanged 2 node proofs" in r.stdout and "level 2 nodes 1" in r.stdout
    lines = dict(ln.split(" ", 1) for ln in r.stdout.strip().splitlines() if " " in ln)
    public = [int(v) for v in lines["public"].split()]
    assert int(lines["key0"]) == int(out["key"][0]) and public == out["public"]
    lvl = [hashlib.sha256(b"\x00" + int(h).to_bytes(32, "big") + rr).digest() for h, rr in zip(heights, roots)]
    while len(lvl) > 1:
        lvl = [hashlib.sha256(b"\x01" + lvl[i] + lvl[i + 1]).digest() for i in range(0, len(lvl), 2)]
    assert public[:8] == list(struct.unpack(">8I", lvl[0]))
    assert mr.verify(root_file.read_bytes(), out["key"], heights, roots, lvl[0]), prover.last_reject        # the compiled host's root, Python's verifier
    # inputs that do not fit the leaf circuit (a word above 32 bits): the host's witness evaluator refuses
    bad = inputs.copy()
    bad[2, 5] = 1 << 40
    bad.tofile(str(tmp_path / "bad_inputs.bin"))
    r2 = subprocess.run([str(exe), str(tmp_path / "pc.bin"), str(nq), str(pw), "2", str(tmp_path / "leaf"), "4", str(tmp_path / "bad_inputs.bin"), str(root_file)]
                        + node_dirs, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r2.returncode != 0 and "FAIL" in r2.stdout
    mr.free()
