"""TEST INFRASTRUCTURE for the build-defined circuit of DESIGN.md §3.6: a random satisfiable circuit generator
(arithmetic / constant gates, public inputs, Poseidon rows, copy constraints over the routed wires), Python big-int
restatements of the permutation products (row a6) and of the quotient on the LDE domain (row a7), a big-int Poseidon row
(witness + the 118 constraints), and an independent verifier of the whole proof (transcript replay + the PLONK identity
at zeta on top of tests/fri_verifier.py).  Shares no code with the product."""
import numpy as np

import fri_verifier as fv

P = fv.P
CHUNK = 8
NCHAL = 2
NCONST = 6                     # q_arith, c0, c1, c2, q_pi, q_pos
TAG = 0x32304B4C504C4747       # "GGLPLK02"
FLAG_POSEIDON = 1
POS_WIRES = 135
POS_CONSTRAINTS = 123
POS_SWAP, POS_ADV0, POS_DELTA0 = 24, 25, 131
FLAG_SHA = 2
NCONST_SHA = 10                # ... + q_she, q_sha, q_shw, q_add
SHA_WIRES = 144
SHA_CONSTRAINTS = 140
SHA_E, SHA_A, SHA_W, SHA_ADD = 0, 1, 2, 3
M32 = 0xFFFFFFFF


FLAG_EXT = 4                   # one more constant column, q_ext, LAST: rows whose 8-wire chunks are multiply-adds in F_p[X]/(X^2 - 7)


def n_const(flags):
    return NCONST + (4 if flags & FLAG_SHA else 0) + (1 if flags & FLAG_EXT else 0)


def ks_of(W):
    out, t = [], 1
    for _ in range(W):
        out.append(t)
        t = t * 7 % P
    return out


# ---- field helpers: the same formulas over ints (base field) or pairs (quadratic extension) ------------------------------
class Base:
    zero = 0
    add = staticmethod(lambda a, b: (a + b) % P)
    sub = staticmethod(lambda a, b: (a - b) % P)
    mul = staticmethod(lambda a, b: a * b % P)
    scale = staticmethod(lambda a, k: a * k % P)
    addc = staticmethod(lambda a, k: (a + k) % P)
    emb = staticmethod(lambda v: v % P)


class Ext:
    zero = (0, 0)
    add = staticmethod(fv.eadd)
    sub = staticmethod(fv.esub)
    mul = staticmethod(fv.emul)
    scale = staticmethod(fv.escale)
    addc = staticmethod(lambda a, k: ((a[0] + k) % P, a[1]))
    emb = staticmethod(lambda v: (v % P, 0))


# ---- Poseidon row (width 12, x^7, 4 + 22 + 4 rounds; round = add constants, S-box, MDS) ------------------------------------
def _mds(F, s, circ, diag, rc_next):
    out = []
    for r in range(12):
        acc = F.scale(s[r], diag[r])
        for i in range(12):
            acc = F.add(acc, F.scale(s[(i + r) % 12], circ[i]))
        out.append(F.addc(acc, rc_next[r]) if rc_next is not None else acc)
    return out


def _sbox(F, x):
    x2 = F.mul(x, x)
    x3 = F.mul(x2, x)
    x4 = F.mul(x2, x2)
    return F.mul(x3, x4)


def poseidon_row(inputs, consts, swap=0):
    """the 135 wire values of a Poseidon row from its 12 inputs and swap bit: in, out, swap, then the S-box inputs of every round after the
    first (3 x 12 full, 22 partial lane-0 values, 4 x 12 full), then the 4 deltas swap * (in[4+i] - in[i]).  swap = 1 exchanges in[0..4) and
    in[4..8) before the permutation.  consts = (rc[360], circ[12], diag[12]) as ints."""
    rc, circ, diag = consts
    F = Base
    deltas = [swap * (inputs[4 + i] - inputs[i]) % P for i in range(4)]
    st = list(inputs)
    for i in range(4):
        st[i] = (st[i] + deltas[i]) % P
        st[4 + i] = (st[4 + i] - deltas[i]) % P
    s = [F.addc(st[i], rc[i]) for i in range(12)]
    adv = []
    rnd = 0
    for r in range(4):
        if r > 0:
            adv += s
        s = [_sbox(F, v) for v in s]
        s = _mds(F, s, circ, diag, rc[(rnd + 1) * 12:(rnd + 2) * 12])
        rnd += 1
    for r in range(22):
        adv.append(s[0])
        s[0] = _sbox(F, s[0])
        s = _mds(F, s, circ, diag, rc[(rnd + 1) * 12:(rnd + 2) * 12])
        rnd += 1
    for r in range(4):
        adv += s
        s = [_sbox(F, v) for v in s]
        s = _mds(F, s, circ, diag, rc[(rnd + 1) * 12:(rnd + 2) * 12] if rnd + 1 < 30 else None)
        rnd += 1
    return list(inputs) + s + [swap] + adv + deltas


def poseidon_constraints(F, wires, consts):
    """the 123 constraint values of a Poseidon row at one point; wires = the row's first 135 wire values (F elements)"""
    rc, circ, diag = consts
    sw = wires[POS_SWAP]
    out = [F.sub(F.mul(sw, sw), sw)]
    st = list(wires[:12])
    for i in range(4):
        d = wires[POS_DELTA0 + i]
        out.append(F.sub(d, F.mul(sw, F.sub(st[4 + i], st[i]))))
        st[i], st[4 + i] = F.add(st[i], d), F.sub(st[4 + i], d)
    s = [F.addc(st[i], rc[i]) for i in range(12)]
    rnd, aw = 0, POS_ADV0
    for r in range(4):
        if r > 0:
            for i in range(12):
                out.append(F.sub(wires[aw + i], s[i]))
            s = list(wires[aw:aw + 12])
            aw += 12
        s = [_sbox(F, v) for v in s]
        s = _mds(F, s, circ, diag, rc[(rnd + 1) * 12:(rnd + 2) * 12])
        rnd += 1
    for r in range(22):
        p = wires[aw]
        aw += 1
        out.append(F.sub(p, s[0]))
        s[0] = _sbox(F, p)
        s = _mds(F, s, circ, diag, rc[(rnd + 1) * 12:(rnd + 2) * 12])
        rnd += 1
    for r in range(4):
        for i in range(12):
            out.append(F.sub(wires[aw + i], s[i]))
        s = [_sbox(F, v) for v in wires[aw:aw + 12]]
        aw += 12
        s = _mds(F, s, circ, diag, rc[(rnd + 1) * 12:(rnd + 2) * 12] if rnd + 1 < 30 else None)
        rnd += 1
    for i in range(12):
        out.append(F.sub(wires[12 + i], s[i]))
    assert len(out) == POS_CONSTRAINTS and aw == POS_DELTA0
    return out


# ---- SHA-256 rows: words on wires 0..11, four 32-bit groups of bit wires at 12/44/76/108, carries at 140..143 ------
def _rotr(x, r):
    return ((x >> r) | (x << (32 - r))) & M32


def sha_row(kind, words, k_const=0):
    """the 144 wire values of a SHA row from its INPUT words (ints < 2^32; T1 may reach 2^35): outputs, bit groups and carries computed here.
    E: words = (e, f, g, h, d, w);  A: (a, b, c, T1);  W: (w16, w15, w7, w2);  ADD: up to four (x, y) pairs, flattened."""
    r = [0] * 12
    if kind == SHA_E:
        e, f, g, h, d, w = words
        t1 = h + (_rotr(e, 6) ^ _rotr(e, 11) ^ _rotr(e, 25)) + ((e & f) ^ (~e & g & M32)) + k_const + w
        e_new = (d + t1) & M32
        r[:8] = [e, f, g, h, d, w, t1, e_new]
        groups, carry = [e, f, g, e_new], (d + t1) >> 32
    elif kind == SHA_A:
        a, b, c, t1 = words
        tot = t1 + (_rotr(a, 2) ^ _rotr(a, 13) ^ _rotr(a, 22)) + ((a & b) ^ (a & c) ^ (b & c))
        r[:5] = [a, b, c, t1, tot & M32]
        groups, carry = [a, b, c, tot & M32], tot >> 32
    elif kind == SHA_W:
        w16, w15, w7, w2 = words
        tot = w16 + (_rotr(w15, 7) ^ _rotr(w15, 18) ^ (w15 >> 3)) + w7 + (_rotr(w2, 17) ^ _rotr(w2, 19) ^ (w2 >> 10))
        r[:5] = [w16, w15, w7, w2, tot & M32]
        groups, carry = [w15, w2, 0, tot & M32], tot >> 32
    else:
        groups, carry = [0, 0, 0, 0], 0
        for k in range(len(words) // 2):
            x, y = words[2 * k], words[2 * k + 1]
            r[3 * k:3 * k + 3] = [x, y, (x + y) & M32]
            groups[k] = (x + y) & M32
            carry |= ((x + y) >> 32) << k
    bits = [(g >> i) & 1 for g in groups for i in range(32)] + [(carry >> k) & 1 for k in range(4)]
    return r + bits


def sha_constraints(F, wires, q4, c2):
    """the 140 selector-weighted constraint values of the SHA block at one point; wires = the row's first 144 wire values, q4 = (q_she, q_sha,
    q_shw, q_add), c2 = the row's c2 constant — all F elements"""
    qE, qA, qW, qD = q4
    q_any = F.add(F.add(qE, qA), F.add(qW, qD))
    out = [F.mul(q_any, F.sub(F.mul(b, b), b)) for b in wires[12:144]]
    X, Y, Z, N, cw = wires[12:44], wires[44:76], wires[76:108], wires[108:140], wires[140:144]

    def pack(bits):
        acc = F.zero
        for i, b in enumerate(bits):
            acc = F.add(acc, F.scale(b, 1 << i))
        return acc

    xor = lambda a, b: F.sub(F.add(a, b), F.scale(F.mul(a, b), 2))
    px, py, pz, pn = pack(X), pack(Y), pack(Z), pack(N)
    ch = pack([F.add(z, F.mul(x, F.sub(y, z))) for x, y, z in zip(X, Y, Z)])
    maj = pack([F.add(F.mul(x, y), F.mul(z, xor(x, y))) for x, y, z in zip(X, Y, Z)])
    S1 = pack([xor(xor(X[(i + 6) % 32], X[(i + 11) % 32]), X[(i + 25) % 32]) for i in range(32)])
    S0 = pack([xor(xor(X[(i + 2) % 32], X[(i + 13) % 32]), X[(i + 22) % 32]) for i in range(32)])
    s0 = pack([xor(xor(X[(i + 7) % 32], X[(i + 18) % 32]), X[i + 3]) if i + 3 < 32 else xor(X[(i + 7) % 32], X[(i + 18) % 32]) for i in range(32)])
    s1 = pack([xor(xor(Y[(i + 17) % 32], Y[(i + 19) % 32]), Y[i + 10]) if i + 10 < 32 else xor(Y[(i + 17) % 32], Y[(i + 19) % 32]) for i in range(32)])
    w = wires
    two32 = 1 << 32
    car2 = F.add(cw[0], F.scale(cw[1], 2))
    car3 = F.add(car2, F.scale(cw[2], 4))
    mix = lambda e, a, ww, d: F.add(F.add(F.mul(qE, e), F.mul(qA, a)), F.add(F.mul(qW, ww), F.mul(qD, d)))
    zero = F.zero
    out.append(mix(F.sub(px, w[0]), F.sub(px, w[0]), F.sub(px, w[1]), F.sub(px, w[2])))
    out.append(mix(F.sub(py, w[1]), F.sub(py, w[1]), F.sub(py, w[3]), F.sub(py, w[5])))
    out.append(mix(F.sub(pz, w[2]), F.sub(pz, w[2]), zero, F.sub(pz, w[8])))
    out.append(mix(F.sub(pn, w[7]), F.sub(pn, w[4]), F.sub(pn, w[4]), F.sub(pn, w[11])))
    e4 = F.sub(w[6], F.add(F.add(F.add(w[3], S1), F.add(ch, c2)), w[5]))
    a4 = F.sub(F.add(w[4], F.scale(car3, two32)), F.add(F.add(w[3], S0), maj))
    w4 = F.sub(F.add(w[4], F.scale(car2, two32)), F.add(F.add(w[0], s0), F.add(w[2], s1)))
    d4 = F.sub(F.add(w[2], F.scale(cw[0], two32)), F.add(w[0], w[1]))
    out.append(mix(e4, a4, w4, d4))
    e5 = F.sub(F.add(w[7], F.scale(car3, two32)), F.add(w[4], w[6]))
    d5 = F.sub(F.add(w[5], F.scale(cw[1], two32)), F.add(w[3], w[4]))
    out.append(mix(e5, zero, zero, d5))
    out.append(F.mul(qD, F.sub(F.add(w[8], F.scale(cw[2], two32)), F.add(w[6], w[7]))))
    out.append(F.mul(qD, F.sub(F.add(w[11], F.scale(cw[3], two32)), F.add(w[9], w[10]))))
    assert len(out) == SHA_CONSTRAINTS
    return out


def int_consts(consts):
    return tuple([int(v) for v in a] for a in consts)


# ---- circuit generator -------------------------------------------------------------------------------------------------------
def build_circuit(rng, log_n, W, copy_prob=0.5, n_routed=None, n_public=0, poseidon_rows=(), consts=None, public_values=None, sha_rows=(),
                  ext_rows=()):
    """random satisfiable instance.  Returns a dict: consts [6][n], sigmas [R][n], wires [W][n] (uint64), public (list of ints),
    shape fields.  poseidon_rows: row indices that carry a permutation (needs W >= 130, R >= 24 and the Poseidon constants);
    n_public: rows 0..n_public-1 expose wire 0 as a public input (public_values: what those cells must hold; default random)."""
    n = 1 << log_n
    R = W if n_routed is None else n_routed
    G = R // 4
    pos = set(int(r) for r in poseidon_rows)
    sha = {int(r): int(rng.integers(0, 4)) for r in sha_rows}          # row -> kind
    ext = set(int(r) for r in ext_rows)
    assert not (pos & set(sha)) and (not sha or (W >= SHA_WIRES and R >= 16)) and not (ext & (pos | set(sha)))
    if pos:
        assert W >= POS_WIRES and R >= 24 and consts is not None
        consts = int_consts(consts)
    rnd = lambda: int(rng.integers(0, 1 << 62)) * 4 % P
    q = [0 if (i in pos or i in sha or i in ext) else (1 if rng.random() < 0.8 else 0) for i in range(n)]
    c0 = [rnd() for _ in range(n)]
    c1 = [rnd() for _ in range(n)]
    c2 = [rnd() if rng.random() < 0.5 else 0 for _ in range(n)]
    q_pi = [1 if i < n_public else 0 for i in range(n)]
    q_pos = [1 if i in pos else 0 for i in range(n)]
    q_sha = [[1 if sha.get(i) == kind else 0 for i in range(n)] for kind in range(4)]
    for i in sha:
        c2[i] = int(rng.integers(0, 1 << 32))                          # K_t of an E row (ignored by the other kinds)
    wires = [[0] * n for _ in range(W)]
    parent = {}

    def find(a):
        while parent.get(a, a) != a:
            parent[a] = parent.get(parent[a], parent[a])
            a = parent[a]
        return a

    def union(a, b):
        ra, rb = find(a), find(b)
        if ra != rb:
            parent[ra] = rb

    cells = []

    def fresh_or_copy(j, i):
        if j == 0 and i < n_public and public_values is not None:
            wires[j][i] = int(public_values[i]) % P
            cells.append((j, i))
            return
        if cells and rng.random() < copy_prob:
            src = cells[int(rng.integers(0, len(cells)))]
            wires[j][i] = wires[src[0]][src[1]]
            union((j, i), src)
        else:
            wires[j][i] = rnd()
        cells.append((j, i))

    r32 = lambda: int(rng.integers(0, 1 << 32))
    for i in range(n):
        if i in ext:
            for c in range(R // 8):
                for k in range(6):              # x0, x1, y0, y1, z0, z1: free cells or copies
                    fresh_or_copy(8 * c + k, i)
                x0, x1, y0, y1, z0, z1 = (wires[8 * c + k][i] for k in range(6))
                wires[8 * c + 6][i] = (x0 * y0 + 7 * x1 * y1 + z0) % P
                wires[8 * c + 7][i] = (x0 * y1 + x1 * y0 + z1) % P
                cells.append((8 * c + 6, i))
                cells.append((8 * c + 7, i))
            for j in range(R, W):
                wires[j][i] = rnd()
            continue
        if i in sha:
            kind = sha[i]
            if kind == SHA_E:
                row = sha_row(kind, [r32() for _ in range(6)], c2[i])
            elif kind == SHA_A:
                row = sha_row(kind, [r32(), r32(), r32(), int(rng.integers(0, 5 << 32))])
            elif kind == SHA_W:
                row = sha_row(kind, [r32() for _ in range(4)])
            else:
                row = sha_row(kind, [r32() for _ in range(2 * int(rng.integers(0, 5)))])
            for j in range(W):
                wires[j][i] = row[j] if j < SHA_WIRES else rnd()
                if j < min(R, 12):
                    cells.append((j, i))                  # words: may be copied FROM
            continue
        if i in pos:
            for j in range(12):
                fresh_or_copy(j, i)                       # inputs: free cells (copies of earlier outputs chain permutations)
            row = poseidon_row([wires[j][i] for j in range(12)], consts, swap=int(rng.integers(0, 2)))
            for j in range(12, POS_WIRES):
                wires[j][i] = row[j]
                if j < R:
                    cells.append((j, i))                  # determined cells: may be copied FROM
            for j in range(POS_WIRES, W):
                wires[j][i] = rnd()
                if j < R:
                    cells.append((j, i))
            continue
        for g in range(G):
            for k in range(3):          # inputs x, y, z
                fresh_or_copy(4 * g + k, i)
            x, y, z = wires[4 * g][i], wires[4 * g + 1][i], wires[4 * g + 2][i]
            wires[4 * g + 3][i] = (c0[i] * x * y + c1[i] * z + c2[i]) % P if q[i] else rnd()
            cells.append((4 * g + 3, i))
        for j in range(R, W):
            wires[j][i] = rnd()                           # advice wires: unconstrained outside Poseidon rows
    # sigma: one cycle per equivalence class (routed wires only)
    classes = {}
    for j in range(R):
        for i in range(n):
            classes.setdefault(find((j, i)), []).append((j, i))
    ks = ks_of(R)
    w = fv.root(log_n)
    wp = [1] * n
    for i in range(1, n):
        wp[i] = wp[i - 1] * w % P
    sigma = [[0] * n for _ in range(R)]
    for members in classes.values():
        for a, (j, i) in enumerate(members):
            jj, ii = members[(a + 1) % len(members)]
            sigma[j][i] = ks[jj] * wp[ii] % P
    to_np = lambda rows: np.array(rows, dtype=np.uint64)
    q_ext = [1 if i in ext else 0 for i in range(n)]
    return {"log_n": log_n, "W": W, "R": R, "n_public": n_public,
            "flags": (FLAG_POSEIDON if pos else 0) | (FLAG_SHA if sha else 0) | (FLAG_EXT if ext else 0),
            "consts": to_np([q, c0, c1, c2, q_pi, q_pos] + (q_sha if sha else []) + ([q_ext] if ext else [])), "sigmas": to_np(sigma), "wires": to_np(wires),
            "public": [wires[0][i] for i in range(n_public)], "pos_consts": consts}


def ref_zs(circ, beta, gamma):
    """[NCHAL*M][n]: per challenge Z then pi_0 .. pi_{M-2}, by the definition (routed wires only)"""
    n, R = 1 << circ["log_n"], circ["R"]
    M = R // CHUNK
    ks = ks_of(R)
    w = fv.root(circ["log_n"])
    wires = [[int(v) for v in r] for r in circ["wires"][:R]]
    sig = [[int(v) for v in r] for r in circ["sigmas"]]
    out = []
    for t in range(NCHAL):
        Z = [1] * n
        parts = [[0] * n for _ in range(M - 1)]
        x = 1
        for i in range(n):
            run = Z[i]
            for c in range(M):
                num = den = 1
                for j in range(c * CHUNK, (c + 1) * CHUNK):
                    num = num * (wires[j][i] + beta[t] * ks[j] % P * x + gamma[t]) % P
                    den = den * (wires[j][i] + beta[t] * sig[j][i] + gamma[t]) % P
                run = run * num % P * pow(den, P - 2, P) % P
                if c < M - 1:
                    parts[c][i] = run
            if i + 1 < n:
                Z[i + 1] = run
            else:
                assert run == 1, "permutation product does not close: the circuit's copy constraints are violated"
            x = x * w % P
        out += [Z] + parts
    return np.array(out, dtype=np.uint64)


def public_input_poly_at(F, public, x, log_n):
    """PI(x) = sum_i pi_i L_i(x), L_i = w^i (x^n - 1) / (n (x - w^i)): pi_i on row i < len(public), 0 on every other row"""
    n = 1 << log_n
    w = fv.root(log_n)
    xn = x
    for _ in range(log_n):
        xn = F.mul(xn, xn)
    zh_over_n = F.scale(F.sub(xn, F.emb(1)), pow(n, P - 2, P))
    inv = (lambda v: pow(v, P - 2, P)) if F is Base else fv.einv
    acc, wi = F.zero, 1
    for pv in public:
        li = F.mul(F.scale(zh_over_n, wi), inv(F.sub(x, F.emb(wi))))
        acc = F.add(acc, F.scale(li, pv % P))
        wi = wi * w % P
    return acc


def constraint_sum(t, x, n, R, ks, beta, gamma, alpha, consts, sigmas, wires, zs, z_next, pi_at_x, ext=False, pos_consts=None):
    """sum_idx alpha_t^idx * C_idx at a point; values are ints (base field) or ext pairs.  Constraint order: 0 L_1 (Z - 1),
    1 public inputs, then per chunk (perm, gate, gate), then (Poseidon circuits) the 118 row constraints times q_pos."""
    F = Ext if ext else Base
    M = R // CHUNK
    one = F.emb(1)
    xn = x
    for _ in range(n.bit_length() - 1):
        xn = F.mul(xn, xn)
    inv = fv.einv if ext else (lambda v: pow(v, P - 2, P))
    l1 = F.mul(F.sub(xn, one), inv(F.scale(F.sub(x, one), n % P)))
    q, c0, c1, c2, q_pi, q_pos = consts[:6]
    q_ext = consts[-1] if len(consts) in (NCONST + 1, NCONST_SHA + 1) else None
    acc = F.mul(l1, F.sub(zs[t * M], one))
    ap = alpha[t]
    acc = F.add(acc, F.scale(F.sub(F.mul(q_pi, wires[0]), pi_at_x), ap))
    prev = zs[t * M]
    bx = F.scale(x, beta[t])
    for c in range(M):
        num, den = one, one
        for j in range(c * CHUNK, (c + 1) * CHUNK):
            wg = F.addc(wires[j], gamma[t])
            num = F.mul(num, F.add(wg, F.scale(bx, ks[j])))
            den = F.mul(den, F.add(wg, F.scale(sigmas[j], beta[t])))
        nxt = zs[t * M + 1 + c] if c + 1 < M else z_next[t]
        perm = F.sub(F.mul(prev, num), F.mul(nxt, den))
        w8 = wires[c * CHUNK:(c + 1) * CHUNK]
        gate = lambda xx, yy, zz, ww: F.mul(q, F.sub(F.add(F.add(F.mul(c0, F.mul(xx, yy)), F.mul(c1, zz)), c2), ww))
        g0, g1 = gate(*w8[0:4]), gate(*w8[4:8])
        if q_ext is not None:                  # the chunk as w = x * y + z in the quadratic extension, in the same two slots
            x0, x1, y0, y1, z0, z1, w0, w1 = w8
            e0 = F.sub(F.add(F.add(F.mul(x0, y0), F.scale(F.mul(x1, y1), 7)), z0), w0)
            e1 = F.sub(F.add(F.add(F.mul(x0, y1), F.mul(x1, y0)), z1), w1)
            g0, g1 = F.add(g0, F.mul(q_ext, e0)), F.add(g1, F.mul(q_ext, e1))
        for con in (perm, g0, g1):
            ap = ap * alpha[t] % P
            acc = F.add(acc, F.scale(con, ap))
        prev = nxt
    if pos_consts is not None:
        pacc = F.zero
        for con in poseidon_constraints(F, wires, pos_consts):
            ap = ap * alpha[t] % P
            pacc = F.add(pacc, F.scale(con, ap))
        acc = F.add(acc, F.mul(q_pos, pacc))
    if len(consts) >= NCONST_SHA:
        for con in sha_constraints(F, wires, consts[6:10], c2):
            ap = ap * alpha[t] % P
            acc = F.add(acc, F.scale(con, ap))
    return acc


def ref_quotient(circ, lde, beta, gamma, alpha, rate_bits=3):
    """quotient values on the LDE domain, bit-reversed order: lde = dict of bit-reversed LDE value matrices (consts, sigmas,
    wires, zs) as lists of int rows.  The public-input polynomial is evaluated by its definition.  Returns [NCHAL][N] ints."""
    log_n, R = circ["log_n"], circ["R"]
    n, log_N = 1 << log_n, log_n + rate_bits
    N = 1 << log_N
    M = R // CHUNK
    ks = ks_of(R)
    wN = fv.root(log_N)
    pos_consts = circ["pos_consts"] if circ["flags"] & FLAG_POSEIDON else None
    out = [[0] * N for _ in range(NCHAL)]
    for i in range(N):
        e = fv.rev(i, log_N)
        x = 7 * pow(wN, e, P) % P
        inext = fv.rev((e + (1 << rate_bits)) % N, log_N)
        zh_inv = pow(pow(x, n, P) - 1, P - 2, P)
        col = lambda mat, idx=i: [r[idx] for r in mat]
        zs_i = col(lde["zs"])
        z_next = [lde["zs"][t * M][inext] for t in range(NCHAL)]
        pi_x = public_input_poly_at(Base, circ["public"], x, log_n)
        for t in range(NCHAL):
            v = constraint_sum(t, x, n, R, ks, beta, gamma, alpha, col(lde["consts"]), col(lde["sigmas"]), col(lde["wires"]), zs_i, z_next, pi_x,
                               pos_consts=pos_consts)
            out[t][i] = v * zh_inv % P
    return out


def verify_plonk(proof_bytes, oracle, pos_consts=None, public=None):
    """independent verifier of glp_plonk_prove's output; raises fv.VerifyError.  pos_consts = (rc, circ, diag): needed for
    Poseidon-gate circuits (the constants the gate is about).  public: the expected public inputs (None = take the proof's)."""
    words = np.frombuffer(proof_bytes, dtype="<u8")
    h = fv.Hasher(oracle)
    ch = fv.Challenger(h)
    pos = 0

    def take(k):
        nonlocal pos
        if pos + k > len(words):
            raise fv.VerifyError("proof truncated")
        out = [int(v) for v in words[pos:pos + k]]
        pos += k
        for v in out:
            ch.observe(v)
        return out

    tag, log_n, W, R, rb, cap_h, n_pub, flags = take(8)
    if tag != TAG or rb != 3 or W % 8 or not (8 <= W <= 160) or R % 8 or not (8 <= R <= W) or not (3 <= log_n <= 24) or n_pub > (1 << log_n) \
            or flags & ~(FLAG_POSEIDON | FLAG_SHA | FLAG_EXT):
        raise fv.VerifyError("bad plonk header")
    if flags & FLAG_SHA and (W < SHA_WIRES or R < 16):
        raise fv.VerifyError("SHA-row circuit: bad shape")
    poseidon = bool(flags & FLAG_POSEIDON)
    if poseidon and (W < POS_WIRES or R < 24 or pos_consts is None):
        raise fv.VerifyError("Poseidon-gate circuit: bad shape or constants not supplied")
    pub = take(n_pub)
    if any(v >= P for v in pub):
        raise fv.VerifyError("non-canonical public input")
    if public is not None and [int(v) for v in public] != pub:
        raise fv.VerifyError("public inputs differ from the expected statement")
    n, log_N = 1 << log_n, log_n + rb
    M = R // CHUNK
    capw = 4 << min(cap_h, log_N)
    cap_pre = take(capw)
    cap_wires = take(capw)
    beta = [ch.challenge() for _ in range(NCHAL)]
    gamma = [ch.challenge() for _ in range(NCHAL)]
    cap_zs = take(capw)
    alpha = [ch.challenge() for _ in range(NCHAL)]
    cap_q = take(capw)
    info = fv.parse_and_verify(None, oracle, challenger=ch, words=words, pos=pos)
    # the FRI part must be about exactly these commitments, shapes and points
    flat = lambda cap: [v for d in cap for v in d]
    if info["n_polys"] != [n_const(flags) + R, W, NCHAL * M, NCHAL << rb] or info["log_n"] != log_n or info["rate_bits"] != rb:
        raise fv.VerifyError("FRI statement does not match the circuit shape")
    if [flat(c) for c in info["caps"]] != [cap_pre, cap_wires, cap_zs, cap_q]:
        raise fv.VerifyError("FRI caps differ from the committed caps")
    g = fv.root(log_n)
    if info["points"] != [info["zeta"], fv.escale(info["zeta"], g)] or sorted(info["openings_at"]) != [(0, 0), (0, 1), (0, 2), (0, 3), (1, 2)]:
        raise fv.VerifyError("wrong opening points")
    zeta = info["zeta"]
    pre, wires, zs, quot = (info["openings_at"][(0, b)] for b in range(4))
    zs_next = info["openings_at"][(1, 2)]
    consts, sigmas = pre[:n_const(flags)], pre[n_const(flags):]
    ks = ks_of(R)
    zn = zeta
    for _ in range(log_n):
        zn = fv.emul(zn, zn)
    zh = fv.esub(zn, (1, 0))
    pi_z = public_input_poly_at(Ext, pub, zeta, log_n)
    pc = int_consts(pos_consts) if poseidon else None
    for t in range(NCHAL):
        lhs = constraint_sum(t, zeta, n, R, ks, beta, gamma, alpha, consts, sigmas, wires, zs, [zs_next[tt * M] for tt in range(NCHAL)], pi_z,
                             ext=True, pos_consts=pc)
        tz, zp = (0, 0), (1, 0)
        for c in range(1 << rb):
            tz = fv.eadd(tz, fv.emul(zp, quot[t * (1 << rb) + c]))
            zp = fv.emul(zp, zn)
        if lhs != fv.emul(zh, tz):
            raise fv.VerifyError(f"PLONK identity fails for challenge {t}")
    return {"log_n": log_n, "W": W, "R": R, "public": pub, "flags": flags, "beta": beta, "gamma": gamma, "alpha": alpha, "zeta": zeta,
            "fri": info}
