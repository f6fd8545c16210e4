"""TEST INFRASTRUCTURE for the build-defined circuit of DESIGN.md §3.6: a random satisfiable
circuit generator, Python big-int restatements of the permutation products (row a6) and of
the quotient on the LDE domain (row a7), and an independent verifier of the whole proof
(transcript replay + the PLONK identity at zeta on top of tests/fri_verifier.py)."""
import numpy as np

import fri_verifier as fv

P = fv.P
CHUNK = 8
NCHAL = 2
TAG = 0x31304B4C504C4747


def ks_of(W):
    out, t = [], 1
    for _ in range(W):
        out.append(t)
        t = t * 7 % P
    return out


def build_circuit(rng, log_n, W, copy_prob=0.5):
    """random satisfiable instance: returns consts [3][n], sigma_vals [W][n], wires [W][n] (uint64)"""
    n = 1 << log_n
    G = W // 4
    rnd = lambda: int(rng.integers(0, 1 << 62)) * 4 % P
    q = [1 if rng.random() < 0.8 else 0 for _ in range(n)]
    c0 = [rnd() for _ in range(n)]
    c1 = [rnd() for _ in range(n)]
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
    for i in range(n):
        for g in range(G):
            for k in range(3):          # inputs x, y, z
                j = 4 * g + k
                if cells and rng.random() < copy_prob:
                    src = cells[int(rng.integers(0, len(cells)))]
                    wires[j][i] = wires[src[0]][src[1]]
                    union((j, i), src)
                else:
                    wires[j][i] = rnd()
                cells.append((j, i))
            x, y, z = wires[4 * g][i], wires[4 * g + 1][i], wires[4 * g + 2][i]
            wires[4 * g + 3][i] = (c0[i] * x * y + c1[i] * z) % P if q[i] else rnd()
            cells.append((4 * g + 3, i))
    # sigma: one cycle per equivalence class
    classes = {}
    for j in range(W):
        for i in range(n):
            classes.setdefault(find((j, i)), []).append((j, i))
    ks = ks_of(W)
    w = fv.root(log_n)
    wp = [1] * n
    for i in range(1, n):
        wp[i] = wp[i - 1] * w % P
    sigma = [[0] * n for _ in range(W)]
    for members in classes.values():
        for a, (j, i) in enumerate(members):
            jj, ii = members[(a + 1) % len(members)]
            sigma[j][i] = ks[jj] * wp[ii] % P
    to_np = lambda rows: np.array(rows, dtype=np.uint64)
    return {"log_n": log_n, "W": W, "consts": to_np([q, c0, c1]), "sigmas": to_np(sigma), "wires": to_np(wires)}


def ref_zs(circ, beta, gamma):
    """[NCHAL*M][n]: per challenge Z then pi_0 .. pi_{M-2}, by the definition"""
    n, W = 1 << circ["log_n"], circ["W"]
    M = W // CHUNK
    ks = ks_of(W)
    w = fv.root(circ["log_n"])
    wires = [[int(v) for v in r] for r in circ["wires"]]
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


def constraint_sum(t, x, n, W, ks, beta, gamma, alpha, consts, sigmas, wires, zs, z_next, ext=False):
    """sum_idx alpha_t^idx * C_idx at a point; values are ints (base field) or ext pairs"""
    M = W // CHUNK
    if ext:
        add, sub, mul = fv.eadd, fv.esub, fv.emul
        emb = lambda v: (v % P, 0)
        xn = x
        for _ in range(n.bit_length() - 1):
            xn = fv.emul(xn, xn)
        l1 = fv.emul(fv.esub(xn, (1, 0)), fv.einv(fv.escale(fv.esub(x, (1, 0)), n % P)))
    else:
        add = lambda a, b: (a + b) % P
        sub = lambda a, b: (a - b) % P
        mul = lambda a, b: a * b % P
        emb = lambda v: v % P
        l1 = (pow(x, n, P) - 1) * pow(n * (x - 1) % P, P - 2, P) % P
    one = emb(1)
    q, c0, c1 = consts
    acc = mul(l1, sub(zs[t * M], one))
    ap = 1
    prev = zs[t * M]
    bx = mul(emb(beta[t]), x)
    for c in range(M):
        num, den = one, one
        for j in range(c * CHUNK, (c + 1) * CHUNK):
            wg = add(wires[j], emb(gamma[t]))
            num = mul(num, add(wg, mul(bx, emb(ks[j]))))
            den = mul(den, add(wg, mul(emb(beta[t]), sigmas[j])))
        nxt = zs[t * M + 1 + c] if c + 1 < M else z_next[t]
        perm = sub(mul(prev, num), mul(nxt, den))
        w8 = wires[c * CHUNK:(c + 1) * CHUNK]
        g0 = mul(q, sub(add(mul(c0, mul(w8[0], w8[1])), mul(c1, w8[2])), w8[3]))
        g1 = mul(q, sub(add(mul(c0, mul(w8[4], w8[5])), mul(c1, w8[6])), w8[7]))
        for con in (perm, g0, g1):
            ap = ap * alpha[t] % P
            acc = add(acc, mul(emb(ap), con))
        prev = nxt
    return acc


def ref_quotient(circ, lde, beta, gamma, alpha, rate_bits=3):
    """quotient values on the LDE domain, bit-reversed order: lde = dict of bit-reversed LDE value
    matrices (consts, sigmas, wires, zs) as lists of int rows.  Returns [NCHAL][N] ints."""
    log_n, W = circ["log_n"], circ["W"]
    n, log_N = 1 << log_n, log_n + rate_bits
    N = 1 << log_N
    M = W // CHUNK
    ks = ks_of(W)
    wN = fv.root(log_N)
    out = [[0] * N for _ in range(NCHAL)]
    for i in range(N):
        e = fv.rev(i, log_N)
        x = 7 * pow(wN, e, P) % P
        inext = fv.rev((e + (1 << rate_bits)) % N, log_N)
        zh_inv = pow(pow(x, n, P) - 1, P - 2, P)
        col = lambda mat, idx=i: [r[idx] for r in mat]
        zs_i = col(lde["zs"])
        z_next = [lde["zs"][t * M][inext] for t in range(NCHAL)]
        for t in range(NCHAL):
            v = constraint_sum(t, x, n, W, ks, beta, gamma, alpha, col(lde["consts"]), col(lde["sigmas"]), col(lde["wires"]), zs_i, z_next)
            out[t][i] = v * zh_inv % P
    return out


def verify_plonk(proof_bytes, oracle):
    """independent verifier of glp_plonk_prove's output; raises fv.VerifyError"""
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

    tag, log_n, W, rb, cap_h = take(5)
    if tag != TAG or rb != 3 or W % 8 or not (8 <= W <= 128) or not (3 <= log_n <= 24):
        raise fv.VerifyError("bad plonk header")
    n, log_N = 1 << log_n, log_n + rb
    M = W // CHUNK
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
    if info["n_polys"] != [3 + W, W, NCHAL * M, NCHAL << rb] or info["log_n"] != log_n or info["rate_bits"] != rb:
        raise fv.VerifyError("FRI statement does not match the circuit shape")
    if [flat(c) for c in info["caps"]] != [cap_pre, cap_wires, cap_zs, cap_q]:
        raise fv.VerifyError("FRI caps differ from the committed caps")
    g = fv.root(log_n)
    if info["points"] != [info["zeta"], fv.escale(info["zeta"], g)] or sorted(info["openings_at"]) != [(0, 0), (0, 1), (0, 2), (0, 3), (1, 2)]:
        raise fv.VerifyError("wrong opening points")
    zeta = info["zeta"]
    pre, wires, zs, quot = (info["openings_at"][(0, b)] for b in range(4))
    zs_next = info["openings_at"][(1, 2)]
    consts, sigmas = pre[:3], pre[3:]
    ks = ks_of(W)
    zn = zeta
    for _ in range(log_n):
        zn = fv.emul(zn, zn)
    zh = fv.esub(zn, (1, 0))
    for t in range(NCHAL):
        lhs = constraint_sum(t, zeta, n, W, ks, beta, gamma, alpha, consts, sigmas, wires, zs, [zs_next[tt * M] for tt in range(NCHAL)], ext=True)
        tz, zp = (0, 0), (1, 0)
        for c in range(1 << rb):
            tz = fv.eadd(tz, fv.emul(zp, quot[t * (1 << rb) + c]))
            zp = fv.emul(zp, zn)
        if lhs != fv.emul(zh, tz):
            raise fv.VerifyError(f"PLONK identity fails for challenge {t}")
    return {"log_n": log_n, "W": W, "beta": beta, "gamma": gamma, "alpha": alpha, "zeta": zeta, "fri": info}
