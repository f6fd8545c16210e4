"""oracle/ed25519_oracle.py — CPU restatement of Ed25519 verification (TEST INFRASTRUCTURE ONLY).

Follows RFC 8032 §5.1 (decoding §5.1.3, verification §5.1.7, cofactorless equation
[S]B = R + [k]A as in the RFC's reference code §6) in plain Python big-int arithmetic.
Reference file:line it follows: NONE — /root/reference holds no source (the upstream path
is curta's Ed25519 gadget, recalled name only).  Pinned by tests/golden/ed25519.json: keys and
signatures produced in this container by OpenSSL 3.0's EVP Ed25519 (generating script
committed) plus RFC 8032 §7.1 TEST 1-3.  Parity with curta's trace layout: UNPINNED.

witness(pub, msg, sig) returns the advice values a verification circuit needs:
  k = SHA512(R || A || M) mod L, the decoded points A and R (affine), P1 = [S]B, P2 = [k]A
  (affine) and the verdict.  The GPU kernel (ed25519_kernels.cuh) emits the same record.
"""
import hashlib

p = 2**255 - 19
L = 2**252 + 27742317777372353535851937790883648493
d = -121665 * pow(121666, p - 2, p) % p
SQRT_M1 = pow(2, (p - 1) // 4, p)
By = 4 * pow(5, p - 2, p) % p


def _recover_x(y, sign):
    if y >= p:
        return None
    x2 = (y * y - 1) * pow(d * y * y + 1, p - 2, p) % p
    if x2 == 0:
        return None if sign else 0
    x = pow(x2, (p + 3) // 8, p)
    if (x * x - x2) % p != 0:
        x = x * SQRT_M1 % p
    if (x * x - x2) % p != 0:
        return None
    if (x & 1) != sign:
        x = p - x
    return x


Bx = _recover_x(By, 0)


def decode_point(b32):
    y = int.from_bytes(b32, "little")
    sign = y >> 255
    y &= (1 << 255) - 1
    x = _recover_x(y, sign)
    return None if x is None else (x, y)


def _ext(P):
    return (P[0], P[1], 1, P[0] * P[1] % p)


def _add(P, Q):
    A = (P[1] - P[0]) * (Q[1] - Q[0]) % p
    B = (P[1] + P[0]) * (Q[1] + Q[0]) % p
    C = 2 * P[3] * Q[3] * d % p
    D = 2 * P[2] * Q[2] % p
    E, F, G, H = B - A, D - C, D + C, B + A
    return (E * F % p, G * H % p, F * G % p, E * H % p)


def _mul(s, P):
    Q = (0, 1, 1, 0)
    while s > 0:
        if s & 1:
            Q = _add(Q, P)
        P = _add(P, P)
        s >>= 1
    return Q


def _affine(P):
    zi = pow(P[2], p - 2, p)
    return (P[0] * zi % p, P[1] * zi % p)


def witness(pub32, msg, sig64):
    """dict(valid, k, s, A, R, P1, P2) with points affine (x, y) or None when decoding fails"""
    out = {"valid": False, "k": 0, "s": 0, "A": None, "R": None, "P1": None, "P2": None}
    if len(pub32) != 32 or len(sig64) != 64:
        return out
    A = decode_point(pub32)
    R = decode_point(sig64[:32])
    s = int.from_bytes(sig64[32:], "little")
    out.update(A=A, R=R, s=s)
    if A is None or R is None or s >= L:
        return out
    k = int.from_bytes(hashlib.sha512(sig64[:32] + pub32 + msg).digest(), "little") % L
    P1 = _affine(_mul(s, _ext((Bx, By))))
    P2 = _affine(_mul(k, _ext(A)))
    rhs = _affine(_add(_ext(R), _ext(P2)))
    out.update(k=k, P1=P1, P2=P2, valid=(P1 == rhs))
    return out


def verify(pub32, msg, sig64):
    return witness(pub32, msg, sig64)["valid"]
