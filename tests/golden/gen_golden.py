#!/usr/bin/env python3
"""Generates the committed golden vectors in this directory from FIRST PRINCIPLES:
Python big-int arithmetic for the Goldilocks field / DFT definition, hashlib for SHA-2.
Nothing here comes from the reference (its mount holds no source, tests or fixtures:
SURVEY.md §0), so these vectors pin the oracle and the HIP path to the published
definitions, not to plonky2 ("parity unpinned" — DESIGN.md).

Run:  python3 tests/golden/gen_golden.py   (deterministic; rewrites the JSON files)
"""
import hashlib
import json
import os
import random

P = 2**64 - 2**32 + 1
HERE = os.path.dirname(os.path.abspath(__file__))


def root(k):
    return pow(7, (P - 1) >> k, P)


def dft(x, inverse=False):
    n = len(x)
    k = n.bit_length() - 1
    w = root(k)
    if inverse:
        w = pow(w, P - 2, P)
    out = []
    for i in range(n):
        wi = pow(w, i, P)
        acc, t = 0, 1
        for v in x:
            acc = (acc + v * t) % P
            t = t * wi % P
        out.append(acc)
    if inverse:
        ninv = pow(n, P - 2, P)
        out = [v * ninv % P for v in out]
    return out


def dump(name, obj):
    with open(os.path.join(HERE, name), "w") as f:
        json.dump(obj, f, separators=(",", ":"))
        f.write("\n")


def main():
    rnd = random.Random(0x9E3779B97F4A7C15)
    edge = [0, 1, 2, 7, P - 1, P - 2, 2**32 - 1, 2**32, 2**32 + 1, 2**63, 2**64 - 2**32, (P - 1) // 2, (P + 1) // 2,
            0xFFFFFFFF, 0xFFFFFFFE00000001, 0x00000000FFFFFFFF, 0xFFFFFFFF00000000]
    vals = edge + [rnd.randrange(P) for _ in range(24)]
    field = {"p": str(P), "binary": [], "pow2": [], "roots": [], "inv": []}
    for a in vals:
        for b in vals[:20]:
            field["binary"].append([str(a), str(b), str((a + b) % P), str((a - b) % P), str(a * b % P)])
    for a in edge + [rnd.randrange(P) for _ in range(6)]:
        for s in range(192):
            field["pow2"].append([str(a), s, str(a * pow(2, s, P) % P)])
    for k in range(33):
        field["roots"].append([k, str(root(k))])
    for a in vals:
        if a:
            field["inv"].append([str(a), str(pow(a, P - 2, P))])
    dump("field.json", field)

    ntt = {"cases": []}
    for log_n in (1, 2, 3, 4, 6, 8, 10):
        n = 1 << log_n
        x = [rnd.randrange(P) for _ in range(n)]
        if log_n == 3:
            x = [P - 1] * n           # all-max input
        if log_n == 4:
            x = [0] * n
            x[1] = 1                  # delta at 1 -> powers of w
        y = dft(x)
        ntt["cases"].append({"log_n": log_n, "x": [str(v) for v in x], "fwd": [str(v) for v in y],
                             "inv": [str(v) for v in dft(x, True)]})
    dump("ntt.json", ntt)

    lde = {"cases": []}
    for log_n, rate_bits, shift in ((3, 1, 7), (4, 3, 7), (6, 2, 7), (5, 3, 49)):
        n = 1 << log_n
        c = [rnd.randrange(P) for _ in range(n)]
        padded = [c[j] * pow(shift, j, P) % P for j in range(n)] + [0] * ((n << rate_bits) - n)
        lde["cases"].append({"log_n": log_n, "rate_bits": rate_bits, "shift": str(shift), "coeffs": [str(v) for v in c],
                             "values": [str(v) for v in dft(padded)]})
    dump("lde.json", lde)

    # SHA-2: FIPS 180-4 example messages + lengths around the padding boundaries
    msgs = [b"", b"abc", b"abcdbcdecdefdefgefghfghighijhijkijkljklmklmnlmnomnopnopq",
            b"abcdefghbcdefghicdefghijdefghijkefghijklfghijklmghijklmnhijklmnoijklmnopjklmnopqklmnopqrlmnopqrsmnopqrstnopqrstu",
            b"a" * 55, b"a" * 56, b"a" * 63, b"a" * 64, b"a" * 111, b"a" * 112, b"a" * 119, b"a" * 127, b"a" * 128,
            bytes(range(256)) * 3]
    sha = {"cases": [{"msg": m.hex(), "sha256": hashlib.sha256(m).hexdigest(), "sha512": hashlib.sha512(m).hexdigest()}
                     for m in msgs]}
    dump("sha2.json", sha)

    # Tendermint simple Merkle tree (RFC 6962 prefixes, split at largest power of two < n)
    def tm_root(items):
        if not items:
            return hashlib.sha256(b"").digest()
        if len(items) == 1:
            return hashlib.sha256(b"\x00" + items[0]).digest()
        k = 1
        while k * 2 < len(items):
            k *= 2
        return hashlib.sha256(b"\x01" + tm_root(items[:k]) + tm_root(items[k:])).digest()

    tm = {"leaf_len": 40, "cases": []}
    for n in (0, 1, 2, 3, 4, 5, 7, 8, 13, 100):
        leaves = [bytes(rnd.randrange(256) for _ in range(40)) for _ in range(n)]
        tm["cases"].append({"n": n, "leaves": b"".join(leaves).hex(), "root": tm_root(leaves).hex()})
    dump("tendermint_merkle.json", tm)

    # coset LDE with BIT-REVERSED output at the sizes the by-cosets path serves (log_n >= 6): values[i] =
    # f(shift * w_N^bitrev(i)).  Own generator so that the files above keep their bytes.
    rnd2 = random.Random(0xB17AE5)
    ldeb = {"cases": []}
    for log_n, rate_bits, shift in ((6, 1, 7), (7, 3, 7), (8, 3, 7), (8, 2, 0x123456789ABCDEF)):
        n, log_N = 1 << log_n, log_n + rate_bits
        c = [rnd2.randrange(P) for _ in range(n)]
        padded = [c[j] * pow(shift, j, P) % P for j in range(n)] + [0] * ((n << rate_bits) - n)
        nat = dft(padded)
        rev = [nat[int(format(i, f"0{log_N}b")[::-1], 2)] for i in range(1 << log_N)]
        ldeb["cases"].append({"log_n": log_n, "rate_bits": rate_bits, "shift": str(shift), "coeffs": [str(v) for v in c],
                              "values_bitrev": [str(v) for v in rev]})
    dump("lde_bitrev.json", ldeb)


if __name__ == "__main__":
    main()
