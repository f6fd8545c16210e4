"""Row a10 on the GPU: Ed25519 witness records through the C ABI vs the Python big-int oracle on
the OpenSSL / RFC 8032 fixtures, plus a validator-set sized batch signed by OpenSSL-made keys
re-used across messages."""
import json
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from test_emu_ed25519 import check_records, load_cases  # noqa: E402

pytestmark = pytest.mark.gpu


def test_fixture_records(prover):
    cases = load_cases()
    out = prover.ed25519_witness([bytes.fromhex(c["pub"]) for c in cases], [bytes.fromhex(c["sig"]) for c in cases],
                                 [bytes.fromhex(c["msg"]) for c in cases])
    check_records(cases, out)


def test_batch_of_150_mixed(prover):
    """150 records built from the valid fixtures with every third one corrupted"""
    base = [c for c in load_cases() if c["valid"]]
    cases = []
    for i in range(150):
        c = dict(base[i % len(base)])
        if i % 3 == 2:
            m = bytearray(bytes.fromhex(c["msg"]) or b"\x00")
            m[0] ^= 1 + (i % 7)
            c = {"src": f"corrupt{i}", "pub": c["pub"], "sig": c["sig"], "msg": bytes(m).hex(), "valid": False}
        cases.append(c)
    out = prover.ed25519_witness([bytes.fromhex(c["pub"]) for c in cases], [bytes.fromhex(c["sig"]) for c in cases],
                                 [bytes.fromhex(c["msg"]) for c in cases])
    assert [bool(r[0]) for r in out] == [c["valid"] for c in cases]
    check_records(cases[:12], out[:12])
