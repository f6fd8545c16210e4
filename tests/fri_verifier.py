"""Independent verifier for the build-defined FRI opening proof — TEST
INFRASTRUCTURE.  Written in Python big-int arithmetic from the protocol description, sharing no
code with the prover; Poseidon goes through the CPU oracle (oracle/gl_oracle.c) with the same
injected constants.  Raises VerifyError on any inconsistency."""
import ctypes

import numpy as np

P = 2**64 - 2**32 + 1
W = 7
TAG = 0x32304952464C4747
u64p = ctypes.POINTER(ctypes.c_uint64)


class VerifyError(Exception):
    pass


# ---- extension field F_p[X]/(X^2 - 7) ------------------------------------------------------
def eadd(x, y):
    return ((x[0] + y[0]) % P, (x[1] + y[1]) % P)


def esub(x, y):
    return ((x[0] - y[0]) % P, (x[1] - y[1]) % P)


def emul(x, y):
    return ((x[0] * y[0] + W * x[1] * y[1]) % P, (x[0] * y[1] + x[1] * y[0]) % P)


def escale(x, s):
    return (x[0] * s % P, x[1] * s % P)


def einv(x):
    n = (x[0] * x[0] - W * x[1] * x[1]) % P
    ni = pow(n, P - 2, P)
    return (x[0] * ni % P, (-x[1]) * ni % P)


def rev(i, bits):
    return int(format(i, f"0{bits}b")[::-1], 2) if bits else 0


def root(k):
    return pow(7, (P - 1) >> k, P)


class Hasher:
    """Poseidon through the oracle (constants must already be set on it)"""

    def __init__(self, oracle):
        self.o = oracle

    def permute(self, state):
        a = np.array(state, dtype=np.uint64)
        self.o.orc_poseidon_permute(a.ctypes.data_as(u64p))
        return [int(v) for v in a]

    def hash_or_noop(self, elems):
        if len(elems) <= 4:
            return list(elems) + [0] * (4 - len(elems))
        s = [0] * 12
        for off in range(0, len(elems), 8):
            chunk = elems[off:off + 8]
            s[:len(chunk)] = chunk
            s = self.permute(s)
        return s[:4]

    def two_to_one(self, left, right):
        return self.permute(list(left) + list(right) + [0] * 4)[:4]


class Challenger:
    def __init__(self, hasher):
        self.h = hasher
        self.state = [0] * 12
        self.inp = []
        self.out = []

    def _duplex(self):
        self.state[:len(self.inp)] = self.inp
        self.inp = []
        self.state = self.h.permute(self.state)
        self.out = self.state[:8]

    def observe(self, x):
        self.out = []
        self.inp.append(x % P)
        if len(self.inp) == 8:
            self._duplex()

    def challenge(self):
        if self.inp or not self.out:
            self._duplex()
        return self.out.pop()

    def ext_challenge(self):
        a = self.challenge()
        b = self.challenge()
        return (a, b)


def merkle_check(h, leaf_digest, index, path, cap):
    cur = leaf_digest
    for lvl, sib in enumerate(path):
        cur = h.two_to_one(cur, sib) if ((index >> lvl) & 1) == 0 else h.two_to_one(sib, cur)
    ci = index >> len(path)
    if ci >= len(cap) or cur != cap[ci]:
        raise VerifyError("Merkle path does not lead to the cap")


def parse_and_verify(proof_bytes, oracle, challenger=None, words=None, pos=0, allow_trailing=False, min_rate_bits=1):
    """verify a stand-alone FRI proof, or (challenger/words/pos given) the FRI part embedded in a
    larger proof, continuing that proof's transcript"""
    if words is None:
        words = np.frombuffer(proof_bytes, dtype="<u8")

    def take(n=1):
        nonlocal pos
        if pos + n > len(words):
            raise VerifyError("proof truncated")
        out = [int(v) for v in words[pos:pos + n]]
        pos += n
        return out

    h = Hasher(oracle)
    ch = challenger if challenger is not None else Challenger(h)
    tag, log_n, rb, cap0, a, fb, nq, pow_bits, shift, nb, n_pts = take(11)
    if tag != TAG:
        raise VerifyError("bad tag")
    if not (1 <= n_pts <= 4) or nb == 0 or nb > 64:
        raise VerifyError("bad header")
    # the remaining header fields are untrusted sizes and shift counts: bound them before they are used as such
    if not (1 <= log_n <= 32) or rb > 8 or not (1 <= a <= 8) or fb > 32 or not (1 <= nq <= 1 << 12) or pow_bits > 63 or not (0 < shift < P) \
            or log_n + rb > 40:
        raise VerifyError("bad header")
    # rate 1 proves nothing (every word is a codeword): min_rate_bits=0 exists only so a test can show that a forged
    # rate-1 proof is otherwise well formed
    if rb < max(min_rate_bits, 0) or (rb == 0 and min_rate_bits > 0):
        raise VerifyError("rate_bits below the required minimum (rate 1 proves nothing)")
    mults = take(n_pts)
    pm = take(2 * nb)
    n_polys, masks = pm[0::2], pm[1::2]
    if any(m == 0 or (m >> n_pts) for m in masks) or not any(m & 1 for m in masks):
        raise VerifyError("bad open masks")
    for w in [tag, log_n, rb, cap0, a, fb, nq, pow_bits, shift, nb, n_pts] + mults + pm:
        ch.observe(w)
    log_N = log_n + rb
    N = 1 << log_N
    if cap0 != min(cap0, log_N):
        raise VerifyError("cap height")
    L = (log_n - fb) // a if log_n > fb else 0
    final_bits = log_n - a * L
    caps = []
    for _ in range(nb):
        c = take(4 << cap0)
        if any(v >= P for v in c):
            raise VerifyError("non-canonical cap")
        for v in c:
            ch.observe(v)
        caps.append([c[4 * i:4 * i + 4] for i in range(1 << cap0)])
    zeta = ch.ext_challenge()
    # openings in (point, batch, polynomial) order
    order = [(p, b) for p in range(n_pts) for b in range(nb) if (masks[b] >> p) & 1]
    total = sum(n_polys[b] for _, b in order)
    op = take(2 * total)
    if any(v >= P for v in op):
        raise VerifyError("non-canonical opening")
    for v in op:
        ch.observe(v)
    openings = [(op[2 * k], op[2 * k + 1]) for k in range(total)]
    alpha = ch.ext_challenge()
    apow = [(1, 0)]
    for _ in range(total - 1):
        apow.append(emul(apow[-1], alpha))
    zs_pt = [escale(zeta, m) for m in mults]
    Ys = [(0, 0)] * n_pts
    kk = 0
    for p, b in order:
        for _ in range(n_polys[b]):
            Ys[p] = eadd(Ys[p], emul(apow[kk], openings[kk]))
            kk += 1
    layer_caps, betas, layer_log, layer_caph = [], [], [], []
    log_len = log_N
    for _ in range(L):
        log_leaves = log_len - a
        chh = min(cap0, log_leaves)
        c = take(4 << chh)
        for v in c:
            ch.observe(v)
        layer_caps.append([c[4 * i:4 * i + 4] for i in range(1 << chh)])
        betas.append(ch.ext_challenge())
        layer_log.append(log_len)
        layer_caph.append(chh)
        log_len -= a
    fin = take(2 << final_bits)
    for v in fin:
        ch.observe(v)
    final_poly = [(fin[2 * j], fin[2 * j + 1]) for j in range(1 << final_bits)]
    seed = [ch.challenge() for _ in range(4)]
    (nonce,) = take(1)
    if pow_bits:
        out0 = h.permute(seed + [nonce % P] + [0] * 7)[0]
        if out0 >> (64 - pow_bits):
            raise VerifyError("proof of work failed")
    ch.observe(nonce % P)
    idxs = [ch.challenge() & (N - 1) for _ in range(nq)]

    w_N = root(log_N)
    for q in range(nq):
        (idx,) = take(1)
        if idx != idxs[q]:
            raise VerifyError("query index does not match the transcript")
        x = shift * pow(w_N, rev(idx, log_N), P) % P
        leaves = []
        for b in range(nb):
            leaf = take(n_polys[b])
            path = take(4 * (log_N - cap0))
            path = [path[4 * i:4 * i + 4] for i in range(log_N - cap0)]
            merkle_check(h, h.hash_or_noop(leaf), idx, path, caps[b])
            leaves.append(leaf)
        cur = (0, 0)
        k = 0
        accs = [(0, 0)] * n_pts
        for p, b in order:
            for v in leaves[b]:
                accs[p] = eadd(accs[p], escale(apow[k], v))
                k += 1
        for p in range(n_pts):
            if any(pp == p for pp, _ in order):
                cur = eadd(cur, emul(esub(accs[p], Ys[p]), einv(esub((x, 0), zs_pt[p]))))
        sh = shift
        for l in range(L):
            ll = layer_log[l]
            leaf = take(2 << a)
            vals = [(leaf[2 * j], leaf[2 * j + 1]) for j in range(1 << a)]
            log_leaves = ll - a
            path = take(4 * (log_leaves - layer_caph[l]))
            path = [path[4 * i:4 * i + 4] for i in range(log_leaves - layer_caph[l])]
            p_l = idx >> (a * l)
            leaf_idx = p_l >> a
            if vals[p_l & ((1 << a) - 1)] != cur:
                raise VerifyError(f"query {q}: layer {l} value does not continue the fold")
            merkle_check(h, h.hash_or_noop(leaf), leaf_idx, path, layer_caps[l])
            beta = betas[l]
            base = leaf_idx << a
            cl = ll
            for _ in range(a):
                wl = root(cl)
                nxt = []
                for i in range(len(vals) // 2):
                    xi = sh * pow(wl, rev(base + 2 * i, cl), P) % P
                    f0, f1 = vals[2 * i], vals[2 * i + 1]
                    inv2 = pow(2, P - 2, P)
                    s = escale(eadd(f0, f1), inv2)
                    d = escale(esub(f0, f1), inv2 * pow(xi, P - 2, P) % P)
                    nxt.append(eadd(s, emul(beta, d)))
                vals = nxt
                base >>= 1
                cl -= 1
                beta = emul(beta, beta)
                sh = sh * sh % P
            cur = vals[0]
        # final polynomial at the remaining point
        fl = log_N - a * L
        xf = sh * pow(root(fl), rev(idx >> (a * L), fl), P) % P
        ev = (0, 0)
        for cf in reversed(final_poly):
            ev = eadd(escale(ev, xf), cf)
        if ev != cur:
            raise VerifyError(f"query {q}: final polynomial mismatch")
    if pos != len(words) and not allow_trailing:
        raise VerifyError("trailing data in proof")
    by_point = {}
    k = 0
    for p, b in order:
        by_point[(p, b)] = openings[k:k + n_polys[b]]
        k += n_polys[b]
    return {"log_n": log_n, "rate_bits": rb, "layers": L, "final_bits": final_bits, "zeta": zeta, "openings": openings,
            "n_polys": n_polys, "queries": idxs, "points": zs_pt, "openings_at": by_point, "caps": caps, "end": pos}
