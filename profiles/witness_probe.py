"""profiles/witness_probe.py — measurement aid: throughput of the witness kernels (rows a9/a10) on one MI355X:
Ed25519 verification witnesses, SHA-256 / SHA-512 round traces, the validator-set Merkle root and a 4096-block data commitment."""
import hashlib
import importlib
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as graft

pkg = graft.load_package()
bs = importlib.import_module(graft.PKG_NAME + ".blobstream")
pr = pkg.Prover(0)
with open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "ed25519.json")) as f:
    base = [c for c in json.load(f)["cases"] if c["valid"]]
for n in (128, 16384, 131072):
    pubs = [bytes.fromhex(base[i % len(base)]["pub"]) for i in range(n)]
    sigs = [bytes.fromhex(base[i % len(base)]["sig"]) for i in range(n)]
    msgs = [bytes.fromhex(base[i % len(base)]["msg"]) for i in range(n)]
    stride = max(1, max(len(m) for m in msgs))
    Pb = np.frombuffer(b"".join(pubs), dtype=np.uint8).reshape(n, 32)
    Sb = np.frombuffer(b"".join(sigs), dtype=np.uint8).reshape(n, 64)
    Mb = np.zeros((n, stride), dtype=np.uint8)
    Ln = np.zeros(n, dtype=np.uint32)
    for i, m in enumerate(msgs):
        Mb[i, :len(m)] = np.frombuffer(m, dtype=np.uint8)
        Ln[i] = len(m)
    dp, ds, dm, dl = (pr.to_device(a) for a in (Pb, Sb, Mb, Ln))
    do = pr.alloc(n * 37 * 8)
    for rep in range(3):
        pr.sync()
        t0 = time.perf_counter()
        pr._chk(pr.lib.glp_ed25519_witness(pr.ctx, dp.ptr, ds.ptr, dm.ptr, stride, dl.ptr, n, do.ptr), "ed25519")
        pr.sync()
        dt = time.perf_counter() - t0
    out = do.download((n, 37))
    assert out[:, 0].all()
    print(f"ed25519 witness: n={n}: {dt * 1e3:.2f} ms -> {n / dt / 1e3:.1f} k signatures/s", flush=True)
    for b in (dp, ds, dm, dl, do):
        b.free()
for n in (1 << 16, 1 << 20):
    for name, lib_fn, blk, words, dsz in (("sha256", pr.lib.glp_sha256_trace, 64, 576 * 4, 32), ("sha512", pr.lib.glp_sha512_trace, 128, 720 * 8, 64)):
        pad = np.frombuffer(pkg.sha_pad(b"x" * 50, blk) * n, dtype=np.uint8)
        d = pr.to_device(pad)
        dd = pr.alloc(n * dsz)
        dtr = pr.alloc(n * words)
        for rep in range(3):
            pr.sync()
            t0 = time.perf_counter()
            pr._chk(lib_fn(pr.ctx, d.ptr, n, 1, dd.ptr, dtr.ptr), name)
            pr.sync()
            dt = time.perf_counter() - t0
        print(f"{name} round trace: {n} one-block messages: {dt * 1e3:.2f} ms -> {n / dt / 1e6:.1f} M blocks/s, trace written at {n * words / dt / 1e9:.0f} GB/s", flush=True)
        for x in (d, dd, dtr):
            x.free()
rng = np.random.default_rng(1)
for n in (100, 150, 10000):
    keys = [rng.integers(0, 256, 32, dtype=np.uint8).tobytes() for _ in range(n)]
    powers = [int(rng.integers(1, 2**40)) for _ in range(n)]
    t0 = time.perf_counter()
    for _ in range(5):
        h = bs.validator_set_hash(pr, keys, powers)
    print(f"validator_set_hash n={n}: {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms per call (incl. host encoding + h2d)", flush=True)
heights = list(range(1, 4097))
roots = [rng.integers(0, 256, 32, dtype=np.uint8).tobytes() for _ in range(4096)]
t0 = time.perf_counter()
for _ in range(5):
    dc = bs.data_commitment(pr, heights, roots)
print(f"data_commitment of 4096 blocks: {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms per call", flush=True)
pr.close()
