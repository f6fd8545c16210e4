"""The product's kernel bodies + host planning run on the CPU (tests/emu) and compared
bit-for-bit with the oracle.  This is the pre-GPU gate for index arithmetic, LDS sizing and
barrier placement; the GPU parity tests proper are in test_gpu_*.py."""
import json
import os

import numpy as np
import pytest

from conftest import P, ptr, rand_field

G = os.path.join(os.path.dirname(__file__), "golden")


def test_product_field_arithmetic_vs_golden(emu):
    with open(os.path.join(G, "field.json")) as f:
        fld = json.load(f)
    for a, b, s, d, m in fld["binary"]:
        a, b = int(a), int(b)
        assert emu.emu_gl_add(a, b) == int(s)
        assert emu.emu_gl_sub(a, b) == int(d)
        assert emu.emu_gl_mul(a, b) == int(m)
    for a, s, r in fld["pow2"]:
        assert emu.emu_gl_mul_pow2(int(a), s) == int(r), (a, s)
    # reduce128 on arbitrary (non-product) inputs
    rng = np.random.default_rng(3)
    for _ in range(2000):
        hi, lo = int(rng.integers(0, 2**63)) * 2 + 1, int(rng.integers(0, 2**63)) * 2
        assert emu.emu_gl_reduce128(hi, lo) == ((hi << 64) | lo) % P
    for hi, lo in ((2**64 - 1, 2**64 - 1), (2**64 - 1, 0), (0, 2**64 - 1), (2**32 - 1, P), (P, P)):
        assert emu.emu_gl_reduce128(hi, lo) == ((hi << 64) | lo) % P


CASES = [
    # log_n, batch, inverse, bitrev, plan, in_place
    (0, 3, 0, 0, None, True), (1, 3, 0, 0, None, True), (3, 5, 0, 0, None, True), (5, 3, 1, 1, None, True),
    (6, 3, 0, 0, None, True), (7, 40, 1, 0, None, True), (8, 2, 0, 0, None, False), (9, 9, 0, 1, None, True),
    (10, 2, 1, 1, None, True), (11, 3, 0, 0, None, True), (12, 1, 0, 0, None, True), (12, 3, 1, 1, None, False),
    (13, 1, 0, 0, None, True), (14, 2, 1, 0, None, False), (15, 1, 0, 1, None, True), (16, 2, 1, 1, None, True),
    (16, 1, 0, 0, "8:4,8:4", True), (16, 1, 0, 0, "10:2,6:4", False), (18, 1, 0, 0, "6:4,6:4,6:4", True),
    (18, 1, 1, 0, "6:4,6:4,6:4", False), (18, 1, 0, 1, "6:4,6:4,6:4", True), (17, 1, 0, 0, "6:3,11:3", True),
    (18, 1, 0, 0, "10:3,8:4", True), (18, 2, 1, 0, "10:2,8:4", False),
    # batched: per-element inter-pass twiddle tables + (tile position, polynomial) workgroup order
    (13, 9, 0, 0, None, True), (14, 8, 1, 1, None, False), (16, 8, 0, 0, "10:3,6:4", True), (17, 8, 0, 0, "6:3,11:3", True),
    (16, 12, 1, 0, "10:2,6:4", True),
    # radix-32 register steps (32 elements per work-item)
    (10, 5, 0, 0, "10:2:5", True), (9, 9, 1, 1, "9:3:5", True), (18, 1, 0, 0, "10:3:5,8:4", True),
    (19, 1, 1, 0, "10:2:5,9:3:5", False), (20, 1, 0, 1, "10:3:5,10:3:5", True), (18, 1, 0, 0, "9:3:5,9:3:5", True),
    # round 3: the split (32-bit halves) LDS exchange of the radix-32 work-items, also on the 2^11 / 2^12 tiles (the two-pass plans of 2^22 / 2^24),
    # natural order (strip + finalT, PLAIN instantiations), inverse with its scale, several polynomials, and FINAL_ROWS (64-bit restaging)
    (20, 1, 0, 0, "10:3:5,10:3:5", True), (18, 2, 1, 0, "12:2:5,6:4", False), (18, 1, 0, 0, "6:4,12:3:5", True), (17, 3, 0, 0, "11:3:5,6:3", True),
    (12, 2, 0, 0, "12:1:5", True), (11, 3, 1, 1, "11:2:5", False), (18, 1, 0, 1, "12:2:5,6:4", True),
    # round 3: radix-64 work-items (2^11 = 32 * 64, 2^12 = 64 * 64: two register steps, one split exchange) — strip and finalT, both tile widths,
    # forward and inverse, in place and out of place, more than one polynomial (the two-pass plans of 2^22 / 2^23 themselves ran here once, bit-exact, and on
    # the GPU against the default plans: profiles/r03_ntt_e6_probe.jsonl)
    # 8- and 4-element work-items on the 2^10 tile (register steps 2 * 8 * 8 * 8 and 4^5: the defaults of 2^20 x 2 and 2^20 x 1): strip, finalT,
    # rows / bit-reversed, inverse (the 2^20 defaults themselves: bit-exact here once, 50-60 s each under emulation, and in tests/test_gpu_ntt.py)
    (16, 2, 0, 0, "10:2:3,6:4", True), (16, 1, 1, 0, "6:4,10:3:3", False), (10, 3, 1, 1, "10:1:3", True),
    (16, 2, 1, 0, "10:2:2,6:4", False), (16, 1, 0, 0, "6:4,10:2:2", True), (10, 2, 0, 1, "10:1:2", True),
    (17, 2, 0, 0, "11:2:6,6:3", True), (17, 1, 1, 0, "6:3,11:3:6", False), (18, 1, 0, 0, "12:2:6,6:4", True), (18, 3, 1, 0, "6:4,12:3:6", True),
]


if os.environ.get("GLP_EMU_CASE_STRIDE"):                     # the alternative-generator child run takes every k-th case
    CASES = CASES[::int(os.environ["GLP_EMU_CASE_STRIDE"])]


@pytest.mark.parametrize("log_n,batch,inv,rev,plan,in_place", CASES)
def test_emulated_ntt_vs_oracle(emu, oracle, log_n, batch, inv, rev, plan, in_place):
    rng = np.random.default_rng(1000 + log_n * 7 + batch)
    n = 1 << log_n
    x = rand_field(rng, (batch, n))
    if batch > 1:
        x[0, :] = P - 1          # extreme values
        x[1, :] = 0
    ref = x.copy()
    if log_n > 0:
        oracle.orc_ntt(ptr(ref), log_n, batch, inv)
        if rev:
            oracle.orc_bitrev_rows(ptr(ref), log_n, batch)
    src = x.copy()
    dst = src if in_place else np.zeros_like(x)
    if log_n == 0:
        pytest.skip("size-1 transform is a copy handled on the host side of the C ABI")
    rc = emu.emu_ntt(ptr(src), ptr(dst), n, n, log_n, batch, inv, rev, plan.encode() if plan else None)
    assert rc == 0
    assert np.array_equal(dst, ref)
    if not in_place:
        assert np.array_equal(src, x), "out-of-place transform must not modify its source"


LDE_CASES = [
    # log_n, rate_bits, batch, plan (of the size-n transforms)
    (6, 1, 3, None), (7, 3, 2, None), (10, 3, 3, None), (12, 2, 1, None), (13, 3, 2, None), (14, 1, 3, None),
    (16, 3, 1, "8:4,8:4"), (16, 2, 2, "10:3,6:4"), (18, 1, 1, "6:4,6:4,6:4"), (15, 4, 1, None),
]


@pytest.mark.parametrize("log_n,rb,batch,plan", LDE_CASES)
def test_emulated_coset_lde_bitrev_vs_oracle(emu, oracle, log_n, rb, batch, plan):
    """LDE by cosets (2^rb size-n transforms per polynomial, input scale fused into the first pass,
    each coset written to its block of the bit-reversed output) == padded size-N transform + bit reversal"""
    rng = np.random.default_rng(4242 + log_n * 5 + rb)
    n, N = 1 << log_n, 1 << (log_n + rb)
    coeffs = rand_field(rng, (batch, n))
    coeffs[0, :] = P - 1
    shift = 7 if rb != 2 else int(rng.integers(2, 2**63))
    ref = np.zeros((batch, N), dtype=np.uint64)
    oracle.orc_lde_coset(ptr(coeffs), ptr(ref), log_n, rb, batch, shift)
    oracle.orc_bitrev_rows(ptr(ref), log_n + rb, batch)
    out = np.full((batch, N), 0xABCD, dtype=np.uint64)
    src = coeffs.copy()
    rc = emu.emu_lde_coset_bitrev(ptr(src), ptr(out), log_n, rb, batch, shift, plan.encode() if plan else None)
    assert rc == 0
    assert np.array_equal(out, ref)
    assert np.array_equal(src, coeffs)


def test_emulated_coset_lde_bitrev_vs_golden(emu):
    """the by-cosets LDE under emulation against the committed big-int vectors (no oracle in the loop)"""
    if os.environ.get("GLP_EMU_ALTGEN") == "1":
        pytest.skip("the golden vectors are generator-7 DFTs")
    with open(os.path.join(G, "lde_bitrev.json")) as f:
        cases = json.load(f)["cases"]
    for c in cases:
        coeffs = np.array([int(v) for v in c["coeffs"]], dtype=np.uint64)
        want = np.array([int(v) for v in c["values_bitrev"]], dtype=np.uint64)
        out = np.zeros(len(want), dtype=np.uint64)
        assert emu.emu_lde_coset_bitrev(ptr(coeffs), ptr(out), c["log_n"], c["rate_bits"], 1, int(c["shift"]), None) == 0
        assert np.array_equal(out, want), (c["log_n"], c["rate_bits"])


def test_emulated_ntt_strided_batch(emu, oracle):
    """polynomials embedded in wider rows (stride > n) on both sides"""
    rng = np.random.default_rng(5)
    log_n, batch, n = 10, 3, 1024
    x = rand_field(rng, (batch, n))
    src = np.full((batch, n + 24), 0xDEAD, dtype=np.uint64)
    src[:, :n] = x
    dst = np.full((batch, n + 8), 0xBEEF, dtype=np.uint64)
    rc = emu.emu_ntt(ptr(src), ptr(dst), n + 24, n + 8, log_n, batch, 0, 0, None)
    assert rc == 0
    ref = x.copy()
    oracle.orc_ntt(ptr(ref), log_n, batch, 0)
    assert np.array_equal(dst[:, :n], ref)
    assert np.all(dst[:, n:] == 0xBEEF)


def test_emulated_kernels_under_the_alternative_two_adic_generator():
    """the two-adic generator is a BUILD parameter (csrc/gl_field.cuh): the kernel bodies compiled on 7277203076849721926 (w_64 = 2^3: other
    shift twiddles in every butterfly, other tables) against the oracle switched to the same generator — every fifth NTT case and the coset
    LDE cases, in a child process (the emulation library and the product library it reads the generator from are selected by environment)."""
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    alt = os.path.join(os.path.dirname(here), "0-kno-blobstreamx_amd", "lib", "libglprover_altgen.so")
    if os.environ.get("GLP_EMU_ALTGEN") == "1":
        pytest.skip("this IS the alternative-generator run")
    if not os.path.exists(alt):
        pytest.skip("lib/libglprover_altgen.so not built (__graft_entry__.build() makes it)")
    env = dict(os.environ, GLP_LIB=alt, GLP_EMU_ALTGEN="1", GLP_EMU_CASE_STRIDE="5")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-p", "no:cacheprovider", "-k", "emulated_ntt or lde"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, cwd=os.path.dirname(here))
    assert r.returncode == 0, r.stdout[-3000:]
    assert "passed" in r.stdout.strip().splitlines()[-1]
