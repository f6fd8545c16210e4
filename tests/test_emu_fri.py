"""Challenger and FRI kernels (eval at an extension point, combine, proof of work) run on the
CPU through tests/emu and checked against Python big-int restatements (tests/fri_verifier.py
holds the independent extension-field arithmetic and the Python challenger)."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import numpy as np
import pytest

import fri_verifier as fv
from conftest import P, poseidon_consts, ptr, rand_field


def consts384(kind):
    rc, circ, diag = poseidon_consts(kind)
    return np.concatenate([rc, circ, diag]).astype(np.uint64), (rc, circ, diag)


@pytest.mark.parametrize("kind", ["small", "big"])
def test_challenger_matches_python_restatement(emu, oracle, kind):
    c384, (rc, circ, diag) = consts384(kind)
    oracle.orc_poseidon_set_constants(ptr(rc), ptr(circ), ptr(diag))
    rng = np.random.default_rng(7)
    # a script mixing observations and challenges with every buffer fill level
    script, ref = [], []
    pych = fv.Challenger(fv.Hasher(oracle))
    vals = [int(v) for v in rand_field(rng, 200)]
    vi = 0
    for step in range(120):
        nobs = int(rng.integers(0, 20))
        for _ in range(nobs):
            v = vals[vi % 200]; vi += 1
            script.append((0, v)); pych.observe(v)
        nch = int(rng.integers(0, 12))
        for _ in range(nch):
            script.append((1, 0)); ref.append(pych.challenge())
    sc = np.array(script, dtype=np.uint64).reshape(-1)
    out = np.zeros(len(ref) + 1, dtype=np.uint64)
    k = emu.emu_challenger(ptr(c384), 1 if kind == "small" else 0, ptr(sc), len(script), ptr(out))
    assert k == len(ref)
    assert [int(v) for v in out[:k]] == ref


@pytest.mark.parametrize("log_n,n_polys", [(2, 1), (8, 3), (13, 2)])
def test_eval_at_extension_point(emu, log_n, n_polys):
    rng = np.random.default_rng(log_n)
    n = 1 << log_n
    co = rand_field(rng, (n_polys, n))
    z = rand_field(rng, 2)
    out = np.zeros(2 * n_polys, dtype=np.uint64)
    assert emu.emu_eval_at_ext(ptr(co), n, log_n, n_polys, ptr(z), ptr(out)) == 0
    zz = (int(z[0]), int(z[1]))
    for p in range(n_polys):
        acc = (0, 0)
        for cf in reversed([int(v) for v in co[p]]):
            acc = fv.eadd(fv.emul(acc, zz), (cf, 0))
        assert (int(out[2 * p]), int(out[2 * p + 1])) == acc


def python_combine(ldes, log_N, alpha, openings, zeta, shift):
    N = 1 << log_N
    w = fv.root(log_N)
    apow = [(1, 0)]
    for _ in range(len(openings) - 1):
        apow.append(fv.emul(apow[-1], alpha))
    Y = (0, 0)
    for a, y in zip(apow, openings):
        Y = fv.eadd(Y, fv.emul(a, y))
    out = []
    for i in range(N):
        x = shift * pow(w, fv.rev(i, log_N), P) % P
        acc = (0, 0)
        for k, row in enumerate(ldes):
            acc = fv.eadd(acc, fv.escale(apow[k], int(row[i])))
        out.append(fv.emul(fv.esub(acc, Y), fv.einv(fv.esub((x, 0), zeta))))
    return out, apow, Y


def test_fri_combine_two_batches(emu):
    rng = np.random.default_rng(3)
    log_N = 7
    N = 1 << log_N
    b0, b1 = rand_field(rng, (3, N)), rand_field(rng, (2, N))
    alpha = tuple(int(v) for v in rand_field(rng, 2))
    zeta = tuple(int(v) for v in rand_field(rng, 2))
    openings = [tuple(int(v) for v in rand_field(rng, 2)) for _ in range(5)]
    want, apow, Y = python_combine(list(b0) + list(b1), log_N, alpha, openings, zeta, 7)
    ap = np.array([c for a in apow for c in a], dtype=np.uint64)
    acc = np.zeros(2 * N, dtype=np.uint64)
    Yv, zv = np.array(Y, dtype=np.uint64), np.array(zeta, dtype=np.uint64)
    assert emu.emu_fri_combine(ptr(b0), 3, log_N, ptr(ap[:6]), ptr(Yv), ptr(zv), 7, ptr(acc), 1, 0) == 0
    assert emu.emu_fri_combine(ptr(b1), 2, log_N, ptr(ap[6:]), ptr(Yv), ptr(zv), 7, ptr(acc), 0, 1) == 0
    got = [(int(acc[2 * i]), int(acc[2 * i + 1])) for i in range(N)]
    assert got == want


def test_proof_of_work_finds_smallest_nonce(emu, oracle):
    c384, (rc, circ, diag) = consts384("small")
    oracle.orc_poseidon_set_constants(ptr(rc), ptr(circ), ptr(diag))
    h = fv.Hasher(oracle)
    seed = [int(v) for v in rand_field(np.random.default_rng(1), 4)]
    pow_bits = 6
    want = next(nn for nn in range(10000) if (h.permute(seed + [nn] + [0] * 7)[0] >> (64 - pow_bits)) == 0)
    found = ctypes.c_ulonglong(0)
    sd = np.array(seed, dtype=np.uint64)
    assert emu.emu_pow(ptr(sd), 0, 512, pow_bits, ptr(c384), 1, ctypes.byref(found)) == 0
    assert found.value == want
