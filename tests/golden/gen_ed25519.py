#!/usr/bin/env python3
"""Generates tests/golden/ed25519.json with the OpenSSL 3 CLI present in this image (keys,
signatures) — fixtures from a widely deployed independent implementation — plus RFC 8032 §7.1
TEST 1-3 typed in from the RFC.  Run: python3 tests/golden/gen_ed25519.py (needs `openssl`)."""
import json
import os
import subprocess
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))


def sh(*a, **k):
    return subprocess.run(a, check=True, stdout=subprocess.PIPE, **k).stdout


def main():
    cases = []
    with tempfile.TemporaryDirectory() as td:
        for i, msg in enumerate([b"x", b"a", b"tendermint vote sign bytes " * 4, bytes(range(200)), b"\x00" * 113, os.urandom(1)]):
            if i == 5:
                msg = bytes([0x5A]) * 64
            key, mf, sf = (os.path.join(td, n) for n in ("k.pem", "m.bin", "s.bin"))
            sh("openssl", "genpkey", "-algorithm", "ed25519", "-out", key)
            pub = sh("openssl", "pkey", "-in", key, "-pubout", "-outform", "DER")[-32:]
            open(mf, "wb").write(msg)
            sh("openssl", "pkeyutl", "-sign", "-rawin", "-in", mf, "-inkey", key, "-out", sf)
            sig = open(sf, "rb").read()
            assert len(sig) == 64
            cases.append({"src": "openssl", "pub": pub.hex(), "msg": msg.hex(), "sig": sig.hex(), "valid": True})
    rfc = [
        ("d75a980182b10ab7d54bfed3c964073a0ee172f3daa62325af021a68f707511a", "",
         "e5564300c360ac729086e2cc806e828a84877f1eb8e5d974d873e065224901555fb8821590a33bacc61e39701cf9b46bd25bf5f0595bbe24655141438e7a100b"),
        ("3d4017c3e843895a92b70aa74d1b7ebc9c982ccf2ec4968cc0cd55f12af4660c", "72",
         "92a009a9f0d4cab8720e820b5f642540a2b27b5416503f8fb3762223ebdb69da085ac1e43e15996e458f3613d0f11d8c387b2eaeb4302aeeb00d291612bb0c00"),
        ("fc51cd8e6218a1a38da47ed00230f0580816ed13ba3303ac5deb911548908025", "af82",
         "6291d657deec24024827e69c3abe01a30ce548a284743a445e3680d7db5ac3ac18ff9b538d16f290ae67f760984dc6594a7c15e9716ed28dc027beceea1ec40a"),
    ]
    for pub, msg, sig in rfc:
        cases.append({"src": "rfc8032-7.1", "pub": pub, "msg": msg, "sig": sig, "valid": True})
    # negative cases derived from the valid ones
    base = cases[2]
    flip = lambda hx, i: hx[:i] + format(int(hx[i], 16) ^ 1, "x") + hx[i + 1:]
    cases.append({"src": "tampered-msg", "pub": base["pub"], "msg": flip(base["msg"], 3), "sig": base["sig"], "valid": False})
    cases.append({"src": "tampered-R", "pub": base["pub"], "msg": base["msg"], "sig": flip(base["sig"], 5), "valid": False})
    cases.append({"src": "tampered-S", "pub": base["pub"], "msg": base["msg"], "sig": flip(base["sig"], 70), "valid": False})
    cases.append({"src": "wrong-key", "pub": cases[1]["pub"], "msg": base["msg"], "sig": base["sig"], "valid": False})
    L = 2**252 + 27742317777372353535851937790883648493
    s = int.from_bytes(bytes.fromhex(base["sig"][64:]), "little") + L
    cases.append({"src": "S-not-reduced", "pub": base["pub"], "msg": base["msg"], "sig": base["sig"][:64] + s.to_bytes(32, "little").hex(), "valid": False})
    with open(os.path.join(HERE, "ed25519.json"), "w") as f:
        json.dump({"cases": cases}, f, indent=0)
        f.write("\n")


if __name__ == "__main__":
    main()
