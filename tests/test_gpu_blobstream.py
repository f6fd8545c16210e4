"""Circuit-level witness logic on the GPU (0-kno-blobstreamx_amd/blobstream.py over the C ABI): validator-set
hash, data commitment, signature checks and voting-power thresholds of a light-client skip, against
independent hashlib / fixture-based restatements.  The formats are public specs restated from memory
(the reference mount is empty): see the module's [SPEC] / [RECALLED] tags."""
import importlib

import numpy as np
import pytest

import __graft_entry__ as graft
from test_blobstream_host import tm_root
from test_emu_ed25519 import load_cases

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def bs():
    graft.load_package()
    return importlib.import_module(graft.PKG_NAME + ".blobstream")


@pytest.mark.parametrize("n", [0, 1, 2, 3, 7, 100, 150, 1000])
def test_validator_set_hash(prover, bs, n):
    rng = np.random.default_rng(n)
    keys = [rng.integers(0, 256, 32, dtype=np.uint8).tobytes() for _ in range(n)]
    powers = [int(rng.integers(0, 2**62)) >> int(rng.integers(0, 62)) for _ in range(n)]      # every varint length
    assert bs.validator_set_hash(prover, keys, powers) == tm_root([bs.encode_validator(k, p) for k, p in zip(keys, powers)])


def test_variable_length_leaves_general(prover):
    rng = np.random.default_rng(77)
    leaves = [rng.integers(0, 256, ln, dtype=np.uint8).tobytes() for ln in (0, 1, 54, 55, 56, 63, 64, 118, 119, 200, 1000, 0, 7)]
    assert prover.tm_merkle_root_var(leaves) == tm_root(leaves)
    assert prover.tm_merkle_root_var([b""]) == tm_root([b""])


@pytest.mark.parametrize("n", [1, 2, 400, 4096])
def test_data_commitment(prover, bs, n):
    """DataCommitmentCircuit's witness (BASELINE configs[4]: a 4096-block range)"""
    rng = np.random.default_rng(n + 5)
    heights = [1_000_000 + i for i in range(n)]
    roots = [rng.integers(0, 256, 32, dtype=np.uint8).tobytes() for _ in range(n)]
    assert bs.data_commitment(prover, heights, roots) == tm_root([bs.encode_data_root_tuple(h, r) for h, r in zip(heights, roots)])


def test_skip_witness(prover, bs):
    """validators = the signers of the valid OpenSSL / RFC 8032 fixtures, each signing its own bytes"""
    base = [c for c in load_cases() if c["valid"]]
    keys = [bytes.fromhex(c["pub"]) for c in base]
    msgs = [bytes.fromhex(c["msg"]) for c in base]
    sigs = [bytes.fromhex(c["sig"]) for c in base]
    n = len(base)
    powers = [10] * n
    # everyone signs: accepted, all thresholds met
    w = bs.skip_witness(prover, keys, powers, keys, powers, sigs, msgs)
    assert w["accept"] and w["signed_power"] == 10 * n == w["total_power"] and list(w["signature_valid"]) == [True] * n
    assert w["validators_hash"] == tm_root([bs.encode_validator(k, p) for k, p in zip(keys, powers)]) == w["trusted_validators_hash"]
    # exactly 2/3 of the power (6 of 9 equal validators) is not enough; a corrupted signature does not count
    assert n == 9
    some = [s if i < 6 else None for i, s in enumerate(sigs)]
    w = bs.skip_witness(prover, keys, powers, keys, powers, some, msgs)
    assert w["signed_power"] == 60 and not w["two_thirds_signed"] and not w["accept"]
    bad = list(sigs)
    bad[0] = bytes([bad[0][0] ^ 1]) + bad[0][1:]
    w = bs.skip_witness(prover, keys, powers, keys, powers, bad, msgs)
    assert not w["signature_valid"][0] and w["signed_power"] == 80 and w["accept"]
    # a target set that shares only 3 of 9 equal-power trusted validators: exactly 1/3 of the trusted power is not enough
    rng = np.random.default_rng(1)
    strangers = [rng.integers(0, 256, 32, dtype=np.uint8).tobytes() for _ in range(6)]
    trusted_keys = keys[:3] + strangers
    w = bs.skip_witness(prover, trusted_keys, [10] * 9, keys, powers, sigs, msgs)
    assert w["two_thirds_signed"] and w["trusted_power_signed"] == 30 and not w["one_third_of_trusted_signed"] and not w["accept"]
    w = bs.skip_witness(prover, trusted_keys, [11, 10, 10] + [10] * 6, keys, powers, sigs, msgs)
    assert w["trusted_power_signed"] == 31 and w["one_third_of_trusted_signed"] and w["accept"]


def test_variable_length_offsets_are_bounded(prover, pkg):
    """the variable-length leaf kernel trusts its offsets, so the entry point checks them — decreasing offsets
    or offsets past data_len are GLP_E_INVALID, and nothing is read out of range"""
    import ctypes
    data = prover.to_device(np.arange(64, dtype=np.uint8))
    out = ctypes.create_string_buffer(32)
    good = prover.to_device(np.array([0, 10, 10, 64], dtype=np.uint64))
    assert prover.lib.glp_tm_merkle_root_var(prover.ctx, data.ptr, 64, good.ptr, 3, out) == 0
    assert out.raw == tm_root([bytes(range(0, 10)), b"", bytes(range(10, 64))])
    for offs, dlen in (([0, 10, 5, 64], 64), ([0, 10, 20, 65], 64), ([0, 10, 20, 2**40], 64), ([0, 10, 20, 64], 63)):
        bad = prover.to_device(np.array(offs, dtype=np.uint64))
        assert prover.lib.glp_tm_merkle_root_var(prover.ctx, data.ptr, dlen, bad.ptr, 3, out) == -1, (offs, dlen)
        bad.free()
    data.free()
    good.free()


def test_ed25519_overlong_length_is_invalid(prover):
    """a message length beyond the row stride must not read the neighbour's bytes: the record is all zero"""
    base = [c for c in load_cases() if c["valid"]][:3]
    pubs = [bytes.fromhex(c["pub"]) for c in base]
    sigs = [bytes.fromhex(c["sig"]) for c in base]
    msgs = [bytes.fromhex(c["msg"]) for c in base]
    n = len(base)
    stride = max(1, max(len(m) for m in msgs))
    P_ = np.frombuffer(b"".join(pubs), dtype=np.uint8).reshape(n, 32)
    S_ = np.frombuffer(b"".join(sigs), dtype=np.uint8).reshape(n, 64)
    M_ = np.zeros((n, stride), dtype=np.uint8)
    for i, m in enumerate(msgs):
        M_[i, :len(m)] = np.frombuffer(m, dtype=np.uint8)
    lens = np.array([len(m) for m in msgs], dtype=np.uint32)
    lens[1] = stride + 1                      # over-long: would read into row 2
    lens[2] = 0xFFFFFFFF if n > 2 else lens[2]
    dp, ds, dm, dl = (prover.to_device(a) for a in (P_, S_, M_, lens))
    do = prover.alloc(n * 37 * 8)
    prover._chk(prover.lib.glp_ed25519_witness(prover.ctx, dp.ptr, ds.ptr, dm.ptr, stride, dl.ptr, n, do.ptr), "glp_ed25519_witness")
    rec = do.download((n, 37))
    assert rec[0, 0] == 1                     # the in-range row still verifies
    assert not rec[1].any() and not rec[2].any()
    for b in (dp, ds, dm, dl, do):
        b.free()
