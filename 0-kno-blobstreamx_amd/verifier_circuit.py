"""The in-circuit verifier of this library's circuit proofs: the recursion step of the MapReduce Reduce (SURVEY.md §8a row a11, §8f item 2;
upstream names recalled, unverified — reference file:line NONE, the mount is empty: plonky2 ``recursion::recursive_verifier::
verify_proof``, ``fri::recursive_verifier``, ``iop::challenger::RecursiveChallenger``).

``verify_in_circuit(builder, proof, leaf_key, ...)`` lays down, on ``recursion.CircuitBuilder``, EVERYTHING the native verifier
(csrc/verify.hip) checks for a proof of this library's circuits (the MapReduce leaf circuit, and recursion proofs themselves):
  * the Fiat-Shamir transcript — a duplex sponge of Poseidon rows absorbing the statement, the caps, the openings, the fold-layer caps, the
    final polynomial and the nonce, squeezing beta, gamma, alpha, zeta, the FRI alpha, the fold betas, the proof-of-work seed and the query
    indices — so every challenge below is a circuit variable derived from the proof's own words;
  * the proof of work (one Poseidon row, canonical bit decomposition, top bits zero);
  * per query: index = low bits of a transcript challenge (canonical decomposition); every Merkle opening (leaf sponge, path with the index
    bits choosing sides, cap entry chosen by the remaining bits); the batch combination sum alpha^k (f_k(x) - y_k)/(x - z_p) in the
    quadratic extension; every fold layer (the layer value continues the fold, arity-2^a folding with beta, beta^2, ...); the final polynomial;
  * the PLONK identity at zeta (L_1, public inputs, permutation argument chunks, arithmetic gates, and — for a child that is itself a
    Poseidon-row circuit, e.g. a recursion proof — the 123 Poseidon-row constraints) against the quotient chunks.
Bound as constants of the verifier circuit: the shape (header words, FRI parameters) and the leaf circuit's verifying key.  Returned: the
variables of the child's public inputs and of its 4-word digest, for the caller to expose or constrain.

A proof that the native verifier rejects cannot be laid down: some ``assert_equal`` meets two different values and the builder raises.
"""
import os

import numpy as np

from . import P
from .recursion import FRI_TAG, PLONK_TAG

W_EXT = 7
NCONST, CHUNK, NCHAL = 6, 8, 2


def _inv(x):
    return pow(x % P, P - 2, P)


def _root(k):
    return pow(7, (P - 1) >> k, P)


def _rev(i, bits):
    return int(format(i, f"0{bits}b")[::-1], 2) if bits else 0


class _G:
    """base- and extension-field helpers over a CircuitBuilder (extension elements are pairs of variables)"""

    def __init__(self, b):
        self.b = b
        self.one, self.zero = b.constant(1), b.constant(0)

    def k(self, v):
        return self.b.constant(v)

    def lin(self, c1, z, c2=0):
        """c1 * z + c2"""
        return self.b.arith(0, c1, c2, z, z, z)

    def mul(self, x, y):
        return self.b.arith(1, 0, 0, x, y, x)

    def add(self, x, y):
        return self.b.arith(1, 1, 0, x, self.one, y)

    def sub(self, x, y):
        return self.b.arith(1, P - 1, 0, x, self.one, y)

    def inv(self, x):
        """1/x: a witness, checked by one multiplication"""
        v = self.b.inverse(x)
        self.b.assert_equal(self.mul(x, v), self.one)
        return v

    def bit_select_const(self, bit, if1, if0=1):
        """bit ? if1 : if0 for constants"""
        return self.lin((if1 - if0) % P, bit, if0)

    def select(self, bit, t, f):
        return self.b.arith(1, 1, 0, bit, self.sub(t, f), f)

    def bits_canonical(self, x):
        """the 64 bits (LSB first) of the CANONICAL representative of x: booleans whose packed value is x, and not both
        (high 32 bits all ones) and (low 32 bits non-zero) — the one pattern of a 64-bit word >= p"""
        bits = [self.b.bit(x, i) for i in range(64)]
        for bit in bits:
            self.b.assert_bool(bit)
        two = self.k(2)
        lo = bits[31]
        for bit in reversed(bits[:31]):
            lo = self.b.arith(1, 1, 0, lo, two, bit)
        hi = bits[63]
        for bit in reversed(bits[32:63]):
            hi = self.b.arith(1, 1, 0, hi, two, bit)
        self.b.assert_equal(self.b.arith(1, 1, 0, hi, self.k(1 << 32), lo), x)
        hi_all = bits[32]
        for bit in bits[33:]:
            hi_all = self.mul(hi_all, bit)
        lo_zero = self.lin(P - 1, bits[0], 1)                        # prod (1 - b_i) over the low half
        for bit in bits[1:32]:
            lo_zero = self.mul(lo_zero, self.lin(P - 1, bit, 1))
        lo_nz = self.lin(P - 1, lo_zero, 1)
        self.b.assert_equal(self.mul(hi_all, lo_nz), self.zero)
        return bits

    # ---- quadratic extension F_p[X]/(X^2 - 7) --------------------------------------------------------------------------------
    def e_const(self, v):
        return (self.k(v[0]), self.k(v[1]))

    def e_val(self, x):
        return (self.b.value(x[0]), self.b.value(x[1]))

    def e_add(self, x, y):
        return (self.add(x[0], y[0]), self.add(x[1], y[1]))

    def e_sub(self, x, y):
        return (self.sub(x[0], y[0]), self.sub(x[1], y[1]))

    def e_mul(self, x, y):
        if self.b.ext_gate:                                    # one chunk of an extension row instead of four arithmetic gates
            return self.b.ext_mul_add(x, y, (self.zero, self.zero))
        t = self.mul(x[1], y[1])
        c0 = self.b.arith(1, W_EXT, 0, x[0], y[0], t)
        u = self.mul(x[1], y[0])
        return (c0, self.b.arith(1, 1, 0, x[0], y[1], u))

    def e_muladd(self, x, y, z):
        """x * y + z"""
        if self.b.ext_gate:
            return self.b.ext_mul_add(x, y, z)
        return self.e_add(z, self.e_mul(x, y))

    def e_scale(self, x, s):
        return (self.mul(x[0], s), self.mul(x[1], s))

    def e_scale_const(self, x, c):
        return (self.lin(c, x[0]), self.lin(c, x[1]))

    def e_muladd_base(self, acc, e, v):
        """acc + e * v, v in the base field"""
        return (self.b.arith(1, 1, 0, e[0], v, acc[0]), self.b.arith(1, 1, 0, e[1], v, acc[1]))

    def e_inv(self, x):
        w = self.b.ext_inverse(x[0], x[1])
        prod = self.e_mul(x, w)
        self.b.assert_equal(prod[0], self.one)
        self.b.assert_equal(prod[1], self.zero)
        return w

    def e_eq(self, x, y):
        self.b.assert_equal(x[0], y[0])
        self.b.assert_equal(x[1], y[1])

    def e_select(self, bit, t, f):
        return (self.select(bit, t[0], f[0]), self.select(bit, t[1], f[1]))

    def e_from_base(self, v):
        return (v, self.zero)


class _Challenger:
    """the library's duplex sponge (challenger.h), every permutation a constrained Poseidon row"""

    def __init__(self, g):
        self.g = g
        self.state = [g.zero] * 12
        self.inp, self.out = [], []

    def _duplex(self):
        self.state = self.g.b.poseidon(self.inp + self.state[len(self.inp):])
        self.inp = []
        self.out = list(self.state[:8])

    def observe(self, v):
        self.out = []
        self.inp.append(v)
        if len(self.inp) == 8:
            self._duplex()

    def challenge(self):
        if self.inp or not self.out:
            self._duplex()
        return self.out.pop()

    def ext_challenge(self):
        a = self.challenge()
        return (a, self.challenge())


def _poseidon_row_constraints(g, wires, consts):
    """the 123 constraint values of a Poseidon row (csrc/plonk_gates.h; swap bit in wire 24, deltas in 131..134) on extension-field wire values: the same walk as the gate, S-boxes
    as extension products, an MDS term as one arithmetic gate per component (acc + const * x)"""
    rc, circ, diag = consts
    kc = [g.k(c) for c in circ]
    kd = [g.k(d) if d else None for d in diag]

    def sbox(x):
        x2 = g.e_mul(x, x)
        x3 = g.e_mul(x2, x)
        x4 = g.e_mul(x2, x2)
        return g.e_mul(x3, x4)

    def mds(s, rc_next):
        out = []
        for r in range(12):
            acc = g.e_scale(s[r], kd[r]) if kd[r] is not None else g.e_from_base(g.zero)
            for i in range(12):
                x = s[(i + r) % 12]
                acc = (g.b.arith(1, 1, 0, x[0], kc[i], acc[0]), g.b.arith(1, 1, 0, x[1], kc[i], acc[1]))
            out.append((g.lin(1, acc[0], rc_next[r]), acc[1]) if rc_next is not None else acc)
        return out

    sw = wires[24]
    out = [g.e_sub(g.e_mul(sw, sw), sw)]
    st = list(wires[:12])
    for i in range(4):                                                     # delta_i = s * (in[4+i] - in[i]); the permuted state swaps by it
        d = wires[131 + i]
        out.append(g.e_sub(d, g.e_mul(sw, g.e_sub(st[4 + i], st[i]))))
        st[i], st[4 + i] = g.e_add(st[i], d), g.e_sub(st[4 + i], d)
    s = [(g.lin(1, st[i][0], rc[i]), st[i][1]) for i in range(12)]
    rnd, aw = 0, 25
    for r in range(4):
        if r > 0:
            out += [g.e_sub(wires[aw + i], s[i]) for i in range(12)]
            s = list(wires[aw:aw + 12])
            aw += 12
        s = mds([sbox(v) for v in s], rc[(rnd + 1) * 12:(rnd + 2) * 12])
        rnd += 1
    for r in range(22):
        p = wires[aw]
        aw += 1
        out.append(g.e_sub(p, s[0]))
        s[0] = sbox(p)
        s = mds(s, rc[(rnd + 1) * 12:(rnd + 2) * 12])
        rnd += 1
    for r in range(4):
        out += [g.e_sub(wires[aw + i], s[i]) for i in range(12)]
        s = mds([sbox(v) for v in wires[aw:aw + 12]], rc[(rnd + 1) * 12:(rnd + 2) * 12] if rnd + 1 < 30 else None)
        aw += 12
        rnd += 1
    out += [g.e_sub(wires[12 + i], s[i]) for i in range(12)]
    assert len(out) == 123 and aw == 131
    return out


def _sha_row_constraints(g, wires, q4, c2):
    """the 140 selector-weighted constraint values of the SHA-row block (csrc/plonk_gates.h) on extension-field wire values"""
    qE, qA, qW, qD = q4
    q_any = g.e_add(g.e_add(qE, qA), g.e_add(qW, qD))
    two, zero_e = g.k(2), g.e_from_base(g.zero)
    out = [g.e_mul(q_any, g.e_sub(g.e_mul(v, v), v)) for v in wires[12:144]]
    X, Y, Z, N, cw = wires[12:44], wires[44:76], wires[76:108], wires[108:140], wires[140:144]

    def pack(bits):
        acc = bits[-1]
        for v in reversed(bits[:-1]):                                      # Horner from the top bit: 2 * acc + bit
            acc = (g.b.arith(1, 1, 0, acc[0], two, v[0]), g.b.arith(1, 1, 0, acc[1], two, v[1]))
        return acc

    def xor(x, y):                                                         # x + y - 2xy
        m = g.e_mul(x, y)
        return g.e_sub(g.e_add(x, y), g.e_scale_const(m, 2))

    px, py, pz, pn = pack(X), pack(Y), pack(Z), pack(N)
    ch = pack([g.e_add(z, g.e_mul(x, g.e_sub(y, z))) for x, y, z in zip(X, Y, Z)])
    maj = pack([g.e_add(g.e_mul(x, y), g.e_mul(z, xor(x, y))) for x, y, z in zip(X, Y, Z)])
    S1 = pack([xor(xor(X[(i + 6) % 32], X[(i + 11) % 32]), X[(i + 25) % 32]) for i in range(32)])
    S0 = pack([xor(xor(X[(i + 2) % 32], X[(i + 13) % 32]), X[(i + 22) % 32]) for i in range(32)])
    s0 = pack([xor(xor(X[(i + 7) % 32], X[(i + 18) % 32]), X[i + 3]) if i + 3 < 32 else xor(X[(i + 7) % 32], X[(i + 18) % 32]) for i in range(32)])
    s1 = pack([xor(xor(Y[(i + 17) % 32], Y[(i + 19) % 32]), Y[i + 10]) if i + 10 < 32 else xor(Y[(i + 17) % 32], Y[(i + 19) % 32]) for i in range(32)])
    w = wires
    two32 = 1 << 32
    car2 = g.e_add(cw[0], g.e_scale_const(cw[1], 2))
    car3 = g.e_add(car2, g.e_scale_const(cw[2], 4))

    def mix(e, a, ww, d):
        acc = None
        for q, v in ((qE, e), (qA, a), (qW, ww), (qD, d)):
            if v is not None:
                t = g.e_mul(q, v)
                acc = t if acc is None else g.e_add(acc, t)
        return acc

    out.append(mix(g.e_sub(px, w[0]), g.e_sub(px, w[0]), g.e_sub(px, w[1]), g.e_sub(px, w[2])))
    out.append(mix(g.e_sub(py, w[1]), g.e_sub(py, w[1]), g.e_sub(py, w[3]), g.e_sub(py, w[5])))
    out.append(mix(g.e_sub(pz, w[2]), g.e_sub(pz, w[2]), None, g.e_sub(pz, w[8])))
    out.append(mix(g.e_sub(pn, w[7]), g.e_sub(pn, w[4]), g.e_sub(pn, w[4]), g.e_sub(pn, w[11])))
    e4 = g.e_sub(w[6], g.e_add(g.e_add(g.e_add(w[3], S1), g.e_add(ch, c2)), w[5]))
    a4 = g.e_sub(g.e_add(w[4], g.e_scale_const(car3, two32)), g.e_add(g.e_add(w[3], S0), maj))
    w4 = g.e_sub(g.e_add(w[4], g.e_scale_const(car2, two32)), g.e_add(g.e_add(w[0], s0), g.e_add(w[2], s1)))
    d4 = g.e_sub(g.e_add(w[2], g.e_scale_const(cw[0], two32)), g.e_add(w[0], w[1]))
    out.append(mix(e4, a4, w4, d4))
    e5 = g.e_sub(g.e_add(w[7], g.e_scale_const(car3, two32)), g.e_add(w[4], w[6]))
    d5 = g.e_sub(g.e_add(w[5], g.e_scale_const(cw[1], two32)), g.e_add(w[3], w[4]))
    out.append(mix(e5, None, None, d5))
    out.append(g.e_mul(qD, g.e_sub(g.e_add(w[8], g.e_scale_const(cw[2], two32)), g.e_add(w[6], w[7]))))
    out.append(g.e_mul(qD, g.e_sub(g.e_add(w[11], g.e_scale_const(cw[3], two32)), g.e_add(w[9], w[10]))))
    assert len(out) == 140 and zero_e is not None
    return out


def verify_in_circuit(b, proof, leaf_key, num_queries, pow_bits, n_wires, n_routed=None, n_public=0, cap_height=4, poseidon_consts=None,
                      proof_id=0, sha=False, ext=False):
    """lay the whole verification of `proof` down on builder `b` (see the module docstring).  The expected statement shape and the leaf
    circuit's key are CONSTANTS of the resulting circuit.  poseidon_consts = (rc, circ, diag): the child is a Poseidon-row circuit (flags = 1,
    e.g. a proof made by this very function's circuit: recursion on recursion); its 123 row constraints are then part of the identity.
    sha: the child has SHA-256 rows (flag 2, ten constant columns): the 140 constraints of that block join the identity too.
    ext: the child has extension-arithmetic rows (flag 4, one more constant column, last): its chunks' multiply-add equations share the gate slots.
    Returns {"public": [vars], "digest": [4 vars]}."""
    g = _G(b)
    words = [int(v) for v in np.frombuffer(bytes(proof), dtype="<u8")]
    pos = 0
    R = n_wires if n_routed is None else n_routed
    M = R // CHUNK

    def take(n):
        nonlocal pos
        if pos + n > len(words):
            raise ValueError("proof truncated")
        out = words[pos:pos + n]
        pos += n
        return out

    def take_const(expected, what):
        """words that must equal constants of this circuit (shape, key): checked now, and recorded for replays of the program"""
        p0 = pos
        if take(len(expected)) != list(expected):
            raise ValueError(what)
        b.word_checks += [("const", (proof_id, p0 + i), int(v)) for i, v in enumerate(expected)]

    def take_vars(n):
        p0 = pos
        vs = take(n)
        if any(v >= P for v in vs):
            raise ValueError("non-canonical proof word")
        return [b.var(v, tag=(proof_id, p0 + i)) for i, v in enumerate(vs)]

    ch = _Challenger(g)
    # ---- statement: header (constants), public inputs, the circuit's key (constants), then the prover's caps ----------------------
    log_n = words[1] if len(words) > 1 else 0
    rb = 3
    flags = (1 if poseidon_consts is not None else 0) | (2 if sha else 0) | (4 if ext else 0)
    nconst = NCONST + (4 if sha else 0) + (1 if ext else 0)
    if poseidon_consts is not None:
        poseidon_consts = tuple([int(v) for v in a] for a in poseidon_consts)
    if not 3 <= log_n <= 24:
        raise ValueError("the proof's header is not the expected statement shape")
    hdr = [PLONK_TAG, log_n, n_wires, R, rb, cap_height, n_public, flags]
    take_const(hdr, "the proof's header is not the expected statement shape")
    n, log_N = 1 << log_n, log_n + rb
    N = 1 << log_N
    cap0 = min(cap_height, log_N)
    capw = 4 << cap0
    stmt = [g.k(v % P) for v in hdr]
    pub = take_vars(n_public)
    stmt += pub
    key = [int(v) for v in leaf_key]
    take_const(key, "the proof is about another circuit (preprocessed cap differs from the key)")
    cap_pre = [g.k(v) for v in key]
    for v in stmt + cap_pre:
        ch.observe(v)
    cap_wires = take_vars(capw)
    for v in cap_wires:
        ch.observe(v)
    beta = [ch.challenge() for _ in range(NCHAL)]
    gamma = [ch.challenge() for _ in range(NCHAL)]
    cap_zs = take_vars(capw)
    for v in cap_zs:
        ch.observe(v)
    alpha = [ch.challenge() for _ in range(NCHAL)]
    cap_q = take_vars(capw)
    for v in cap_q:
        ch.observe(v)
    digest = b.hash_no_pad(stmt + cap_pre + cap_wires + cap_zs + cap_q)          # = glp_plonk_proof_digest of the child
    caps = [cap_pre, cap_wires, cap_zs, cap_q]

    # ---- FRI part: parameters are constants of this circuit ---------------------------------------------------------------------
    a_bits, fb = 4, min(5, log_n)
    gen = _root(log_n)
    n_polys = [nconst + R, n_wires, NCHAL * M, NCHAL << rb]
    masks = [1, 1, 3, 1]
    fhdr = [FRI_TAG, log_n, rb, cap0, a_bits, fb, num_queries, pow_bits, 7, 4, 2, 1, gen]
    for npk, mk in zip(n_polys, masks):
        fhdr += [npk, mk]
    take_const(fhdr, "the proof's FRI parameters are not the expected ones")
    for v in fhdr:
        ch.observe(g.k(v % P))
    for bi in range(4):
        p0 = pos
        if take(capw) != [b.value(v) for v in caps[bi]]:
            raise ValueError("FRI caps differ from the committed caps")
        if bi:
            b.word_checks += [("var", (proof_id, p0 + i), v) for i, v in enumerate(caps[bi])]
        else:
            b.word_checks += [("const", (proof_id, p0 + i), kv) for i, kv in enumerate(key)]
        for v in caps[bi]:
            ch.observe(v)
    zeta = ch.ext_challenge()
    order = [(p, bi) for p in range(2) for bi in range(4) if (masks[bi] >> p) & 1]
    total = sum(n_polys[bi] for _, bi in order)
    op = take_vars(2 * total)
    for v in op:
        ch.observe(v)
    openings = [(op[2 * k], op[2 * k + 1]) for k in range(total)]
    alpha_f = ch.ext_challenge()
    apow = [g.e_from_base(g.one)]
    for _ in range(total - 1):
        apow.append(g.e_mul(apow[-1], alpha_f))
    z_pts = [zeta, g.e_scale_const(zeta, gen)]
    Ys = [g.e_from_base(g.zero), g.e_from_base(g.zero)]
    kk = 0
    for p, bi in order:
        for _ in range(n_polys[bi]):
            Ys[p] = g.e_muladd(apow[kk], openings[kk], Ys[p])
            kk += 1
    L = (log_n - fb) // a_bits if log_n > fb else 0
    final_bits = log_n - a_bits * L
    layer_caps, betas, layer_log, layer_caph = [], [], [], []
    log_len = log_N
    for _ in range(L):
        chh = min(cap0, log_len - a_bits)
        c = take_vars(4 << chh)
        for v in c:
            ch.observe(v)
        layer_caps.append(c)
        betas.append(ch.ext_challenge())
        layer_log.append(log_len)
        layer_caph.append(chh)
        log_len -= a_bits
    fin = take_vars(2 << final_bits)
    for v in fin:
        ch.observe(v)
    final_poly = [(fin[2 * j], fin[2 * j + 1]) for j in range(1 << final_bits)]
    seed = [ch.challenge() for _ in range(4)]
    nonce = take_vars(1)[0]
    if pow_bits:
        out0 = b.poseidon(seed + [nonce] + [g.zero] * 7)[0]
        for bit in g.bits_canonical(out0)[64 - pow_bits:]:
            b.assert_equal(bit, g.zero)
    ch.observe(nonce)
    idx_bits = [g.bits_canonical(ch.challenge())[:log_N] for _ in range(num_queries)]

    minterm_cache = {}

    def minterms(sel_bits):
        """one-hot indicators of the index spelled by sel_bits (LSB first): m[e] = prod_k (bit_k if e has bit k else 1 - bit_k).  Shared by
        every tree opened at the same query index (the four initial trees select the same cap entry)."""
        key = tuple(sel_bits)
        if key not in minterm_cache:
            m = [g.one]
            for k, bit in enumerate(sel_bits):
                nb = g.lin(P - 1, bit, 1)                                  # 1 - bit
                if k == 0:
                    m = [nb, bit]
                else:
                    m = [g.mul(x, nb) for x in m] + [g.mul(x, bit) for x in m]
            minterm_cache[key] = m
        return minterm_cache[key]

    def merkle_to_cap(leaf_vars, bits, path_vars, cap_vars, cap_log):
        plen = len(path_vars)
        top = b.merkle_root_from_path(b.hash_no_pad(leaf_vars), bits[:plen], path_vars)
        if cap_log == 0:
            entry = cap_vars[:4]
        else:
            m = minterms(bits[plen:plen + cap_log])
            entry = []
            for w in range(4):                                             # word w of the selected entry = sum_e m[e] * cap[e][w]
                acc = g.mul(m[0], cap_vars[w])
                for e in range(1, 1 << cap_log):
                    acc = b.arith(1, 1, 0, m[e], cap_vars[4 * e + w], acc)
                entry.append(acc)
        for x, y in zip(top, entry):
            b.assert_equal(x, y)

    def point_from_bits(bits, nbits, shift):
        """shift * w_{2^nbits}^{rev(index)} for an index given by its bits (LSB first): bit k selects the factor w^(2^(nbits-1-k))"""
        x = g.k(shift)
        w = _root(nbits)
        for k in range(nbits):
            x = g.mul(x, g.bit_select_const(bits[k], pow(w, 1 << (nbits - 1 - k), P)))
        return x

    inv2 = _inv(2)
    for q in range(num_queries):
        bits = idx_bits[q]
        idx_val = sum(b.value(bit) << k for k, bit in enumerate(bits))
        b.word_checks.append(("bits", (proof_id, pos), list(bits)))
        if take(1) != [idx_val]:
            raise ValueError("query index does not match the transcript")
        x = point_from_bits(bits, log_N, 7)
        leaves = []
        for bi in range(4):
            leaf = take_vars(n_polys[bi])
            path = [take_vars(4) for _ in range(log_N - cap0)]
            merkle_to_cap(leaf, bits, path, caps[bi], cap0)
            leaves.append(leaf)
        accs = [g.e_from_base(g.zero), g.e_from_base(g.zero)]
        k = 0
        for p, bi in order:
            for v in leaves[bi]:
                accs[p] = g.e_muladd_base(accs[p], apow[k], v)
                k += 1
        cur = g.e_from_base(g.zero)
        for p in range(2):
            den = g.e_sub(g.e_from_base(x), z_pts[p])
            cur = g.e_muladd(g.e_sub(accs[p], Ys[p]), g.e_inv(den), cur)
        sh = 7
        for l in range(L):
            ll = layer_log[l]
            leaf = take_vars(2 << a_bits)
            vals = [(leaf[2 * j], leaf[2 * j + 1]) for j in range(1 << a_bits)]
            log_leaves = ll - a_bits
            path = [take_vars(4) for _ in range(log_leaves - layer_caph[l])]
            lbits = bits[a_bits * l:]                                            # bits of p_l = idx >> (a*l)
            # the layer value at position p_l & (2^a - 1) continues the fold
            sel = vals
            for kbit in range(a_bits):
                sel = [g.e_select(lbits[kbit], sel[2 * e + 1], sel[2 * e]) for e in range(len(sel) // 2)]
            g.e_eq(sel[0], cur)
            merkle_to_cap(leaf, lbits[a_bits:], path, layer_caps[l], layer_caph[l])
            # T = sh * w_ll^{rev_{ll-a}(leaf_idx)}: the variable part of every coset point; U = 1/T
            leaf_bits = lbits[a_bits:a_bits + log_leaves]
            T = g.k(sh)
            wl = _root(ll)
            for kbit in range(log_leaves):
                T = g.mul(T, g.bit_select_const(leaf_bits[kbit], pow(wl, 1 << (log_leaves - 1 - kbit), P)))
            U = g.inv(T)
            bt = betas[l]
            cl = ll
            for s in range(a_bits):
                wls = _root(cl)
                nxt = []
                for i in range(len(vals) // 2):
                    cconst = pow(wls, _rev(2 * i, a_bits - s) << log_leaves, P)      # x_i = T_s * cconst
                    m = g.lin(inv2 * _inv(cconst) % P, U)                            # 1 / (2 x_i)
                    f0, f1 = vals[2 * i], vals[2 * i + 1]
                    sm = g.e_scale_const(g.e_add(f0, f1), inv2)
                    d = g.e_scale(g.e_sub(f0, f1), m)
                    nxt.append(g.e_muladd(bt, d, sm))
                vals = nxt
                cl -= 1
                bt = g.e_mul(bt, bt)
                U = g.mul(U, U)
                sh = sh * sh % P
            cur = vals[0]
        fl = log_N - a_bits * L
        xf = point_from_bits(bits[a_bits * L:], fl, sh)
        ev = g.e_from_base(g.zero)
        for cf in reversed(final_poly):
            ev = g.e_add(g.e_scale(ev, xf), cf)
        g.e_eq(ev, cur)
    if pos != len(words):
        raise ValueError("trailing data in proof")
    b.word_checks.append(("const", (proof_id, 1), log_n))           # (also pins the proof length through the shape)

    # ---- the PLONK identity at zeta ------------------------------------------------------------------------------------------------
    offs, o = {}, 0
    for p, bi in order:
        offs[(p, bi)] = o
        o += n_polys[bi]
    pre = openings[offs[(0, 0)]: offs[(0, 0)] + n_polys[0]]
    wires = openings[offs[(0, 1)]: offs[(0, 1)] + n_polys[1]]
    zs = openings[offs[(0, 2)]: offs[(0, 2)] + n_polys[2]]
    quot = openings[offs[(0, 3)]: offs[(0, 3)] + n_polys[3]]
    zs_next = openings[offs[(1, 2)]: offs[(1, 2)] + n_polys[2]]
    consts, sigmas = pre[:nconst], pre[nconst:]
    ks = [pow(7, j, P) for j in range(R)]
    one_e = g.e_from_base(g.one)
    zn = zeta
    for _ in range(log_n):
        zn = g.e_mul(zn, zn)
    zh = g.e_sub(zn, one_e)
    # PI(zeta) = sum_i pi_i * w^i (zeta^n - 1) / (n (zeta - w^i))
    pi_z = g.e_from_base(g.zero)
    zh_over_n = g.e_scale_const(zh, _inv(n))
    wi = 1
    for pv in pub:
        li = g.e_mul(g.e_scale_const(zh_over_n, wi), g.e_inv(g.e_sub(zeta, g.e_const((wi, 0)))))
        pi_z = g.e_add(pi_z, g.e_scale(li, pv))
        wi = wi * gen % P
    l1 = g.e_mul(zh, g.e_inv(g.e_scale_const(g.e_sub(zeta, one_e), n % P)))
    q_ar, c0, c1, c2, q_pi, q_pos = consts[:6]
    pos_cons = _poseidon_row_constraints(g, wires, poseidon_consts) if flags & 1 else []
    sha_cons = _sha_row_constraints(g, wires, consts[6:10], c2) if sha else []
    for t in range(NCHAL):
        acc = g.e_mul(l1, g.e_sub(zs[t * M], one_e))
        ap = alpha[t]
        acc = g.e_add(acc, g.e_scale(g.e_sub(g.e_mul(q_pi, wires[0]), pi_z), ap))
        prev = zs[t * M]
        bx = g.e_scale(zeta, beta[t])
        for c in range(M):
            num, den = one_e, one_e
            for j in range(c * CHUNK, (c + 1) * CHUNK):
                wg = (g.add(wires[j][0], gamma[t]), wires[j][1])
                num = g.e_mul(num, g.e_add(wg, g.e_scale_const(bx, ks[j])))
                den = g.e_mul(den, g.e_add(wg, g.e_scale(sigmas[j], beta[t])))
            nxt = zs[t * M + 1 + c] if c + 1 < M else zs_next[t * M]
            perm = g.e_sub(g.e_mul(prev, num), g.e_mul(nxt, den))
            w8 = wires[c * CHUNK:(c + 1) * CHUNK]

            def gate(xx, yy, zz, ww):
                return g.e_mul(q_ar, g.e_sub(g.e_add(g.e_add(g.e_mul(c0, g.e_mul(xx, yy)), g.e_mul(c1, zz)), c2), ww))
            g0, g1 = gate(*w8[0:4]), gate(*w8[4:8])
            if ext:                                            # the chunk as w = x * y + z in the child's extension rows, same two slots
                x0, x1, y0, y1, z0, z1, w0, w1 = w8
                e0 = g.e_sub(g.e_add(g.e_muladd(x0, y0, g.e_scale_const(g.e_mul(x1, y1), W_EXT)), z0), w0)
                e1 = g.e_sub(g.e_add(g.e_muladd(x0, y1, g.e_mul(x1, y0)), z1), w1)
                g0, g1 = g.e_muladd(consts[-1], e0, g0), g.e_muladd(consts[-1], e1, g1)
            for con in (perm, g0, g1):
                ap = g.mul(ap, alpha[t])
                acc = g.e_add(acc, g.e_scale(con, ap))
            prev = nxt
        if flags & 1:
            pacc = g.e_from_base(g.zero)
            for con in pos_cons:
                ap = g.mul(ap, alpha[t])
                pacc = g.e_add(pacc, g.e_scale(con, ap))
            acc = g.e_muladd(q_pos, pacc, acc)
        for con in sha_cons:
            ap = g.mul(ap, alpha[t])
            acc = g.e_add(acc, g.e_scale(con, ap))
        tz, zp = g.e_from_base(g.zero), one_e
        for c in range(1 << rb):
            tz = g.e_muladd(zp, quot[t * (1 << rb) + c], tz)
            zp = g.e_mul(zp, zn)
        g.e_eq(acc, g.e_mul(zh, tz))
    return {"public": pub, "digest": digest}


def recursive_aggregation_circuit(prover, proofs, leaf_key, num_queries, pow_bits, n_wires, n_routed=None, n_public=0, cap_height=4,
                                  poseidon_consts=None, sha=False):
    """ONE circuit that verifies every proof in `proofs` (all of the same leaf circuit `leaf_key`, same parameters) and folds their digests into a
    Poseidon Merkle root: the Reduce step as a real recursion, fan-in len(proofs).  Public inputs: each leaf's public inputs, each leaf's
    4-word digest (leaf order), then the 4-word root.  A verifier of the resulting proof needs no leaf proof: the leaf circuit's key and the
    leaf parameters are constants of this circuit (part of ITS verifying key)."""
    from .recursion import CircuitBuilder
    n = len(proofs)
    assert n >= 1 and n & (n - 1) == 0, "a power-of-two number of leaves"
    b = CircuitBuilder(prover)
    level = []
    for k, proof in enumerate(proofs):
        out = verify_in_circuit(b, proof, leaf_key, num_queries, pow_bits, n_wires, n_routed, n_public, cap_height, poseidon_consts, proof_id=k,
                                sha=sha)
        for v in out["public"] + out["digest"]:
            b.public_input(v)
        level.append(out["digest"])
    while len(level) > 1:
        level = [b.two_to_one(level[2 * k], level[2 * k + 1]) for k in range(len(level) // 2)]
    for v in level[0]:
        b.public_input(v)
    stats = {"leaves": n, "poseidon_rows": len(b.pos_rows), "arith_gates": sum(len(r) for rows in b.arith_rows.values() for r in rows)}
    ck, dw, public = b.build()
    stats["rows"] = 1 << ck.log_n
    return ck, dw, public, stats


class RecursionProgram:
    """The recursion circuit for N proofs of ONE leaf circuit, recorded once and replayed for every later batch: `__init__` lays the circuit down
    from a sample batch (Python builder, seconds) and commits it; `prove(proofs)` evaluates the witness program on new proofs (glp_witness_eval,
    milliseconds of host C++), uploads the wires and proves on the GPU.  A batch containing a proof that does not verify is refused
    (ValueError): some copy constraint of the verifier circuit fails on its witness."""

    def __init__(self, prover, sample_proofs, leaf_key, num_queries, pow_bits, n_wires, poseidon_values, n_routed=None, n_public=0, cap_height=4,
                 child_is_recursion=False, child_sha=False, combine=None, builder_wires=136, ext_gate=False, child_ext=False, specs=None):
        """combine(b, outs) -> [public input variables]: what the node states about its children, laid down after their verification
        (outs[k] = {"public": child k's public-input variables, "digest": its 4 digest variables}).  Default: every child's public inputs and
        digest, then the Poseidon root of the digests.  builder_wires: wire count of THIS circuit (144 when combine uses SHA rows).
        ext_gate: lay THIS circuit down with extension-arithmetic rows (its proofs then carry flag 4: whoever verifies them in-circuit passes
        child_ext=True).
        specs: one dict per proof overriding (leaf_key, n_wires, n_routed, n_public, cap_height, child_is_recursion, child_sha, child_ext) — the
        children may then be proofs of DIFFERENT circuits (e.g. a header-chain root and a signature-set root under one outer statement); with
        specs any number of proofs is accepted."""
        from .recursion import CircuitBuilder
        n = len(sample_proofs)
        assert n >= 1 and (specs is not None or n & (n - 1) == 0), "a power-of-two number of proofs"
        self.prover, self.consts = prover, poseidon_values
        b = CircuitBuilder(prover, n_wires=builder_wires, ext_gate=ext_gate)
        outs = []

        def spec_of(k):
            sp = dict(leaf_key=leaf_key, n_wires=n_wires, n_routed=n_routed, n_public=n_public, cap_height=cap_height,
                      child_is_recursion=child_is_recursion, child_sha=child_sha, child_ext=child_ext)
            if specs is not None:
                sp.update(specs[k])
            return sp

        def same_circuit(a, c):
            return (all(np.array_equal(np.asarray(a[f]), np.asarray(c[f])) if f == "leaf_key" else a[f] == c[f] for f in a))

        clone = os.environ.get("GLP_RECORD_CLONE", "1") != "0"
        k = 0
        while k < n:
            proof, sp = sample_proofs[k], spec_of(k)
            b.begin_segment()            # one proof's verifier depends on constants and on itself: the witness evaluator runs them in parallel
            m0 = b.mark()
            outs.append(verify_in_circuit(b, proof, sp["leaf_key"], num_queries, pow_bits, sp["n_wires"], sp["n_routed"], sp["n_public"], sp["cap_height"],
                                          poseidon_values if sp["child_is_recursion"] else None, proof_id=k, sha=sp["child_sha"], ext=sp["child_ext"]))
            m1 = b.mark()
            b.end_segment()
            k += 1
            # the following proofs of the SAME circuit: copies of the sub-circuit just recorded (CircuitBuilder.clone_segment) instead of
            # another pass through the gadget code
            j = k
            while clone and j < n and len(sample_proofs[j]) == len(proof) and same_circuit(spec_of(j), sp):
                j += 1
            if j > k:
                got = b.clone_segment(m0, m1, outs[-1], [({k - 1: jj}, (lambda lid, jj=jj: np.frombuffer(bytes(sample_proofs[jj]), dtype="<u8"))) for jj in range(k, j)])
                if got is not None:
                    outs += got
                    k = j
        b.fill_values(poseidon_values)   # the clones' values, from one run of the witness evaluator (ValueError: one of the proofs does not verify)
        if combine is None:
            level = []
            for out in outs:
                for v in out["public"] + out["digest"]:
                    b.public_input(v)
                level.append(out["digest"])
            while len(level) > 1:
                level = [b.two_to_one(level[2 * k], level[2 * k + 1]) for k in range(len(level) // 2)]
            for v in level[0]:
                b.public_input(v)
        else:
            for v in combine(b, outs):
                b.public_input(v)
        self.program = b.program()
        self.circuit = self.program.setup(prover)
        self.stats = dict(self.program.stats, leaves=n)

    def key(self):
        return self.circuit.cap()

    def save(self, path):
        """the recorded circuit as an .npz of plain arrays (WitnessProgram.save): build once, load in every prover process"""
        self.program.save(path)

    @classmethod
    def load(cls, prover, path, poseidon_values):
        """a RecursionProgram from a saved recording: no builder run; the circuit is committed again on this prover (its key is a function of
        the recording alone, so it equals the key of the process that recorded it)"""
        from .recursion import WitnessProgram
        self = object.__new__(cls)
        self.prover, self.consts = prover, poseidon_values
        self.program = WitnessProgram.load(path)
        self.circuit = self.program.setup(prover)
        self.stats = dict(self.program.stats)
        return self

    def replicate(self, prover):
        """the same recorded circuit committed on ANOTHER prover (ctx) of the same GPU: same key; lets several nodes of one level be proved at
        once, one host thread per prover (the recording itself is shared)"""
        other = object.__new__(type(self))
        other.prover, other.consts, other.program, other.stats = prover, self.consts, self.program, self.stats
        other.circuit = self.program.setup(prover)
        return other

    def witness(self, proofs, reuse=False):
        """(device wires, public inputs) for a batch; reuse=True: the wires live in the program's own per-prover buffer (do not free)"""
        inputs, ws = self.program.inputs_from_words(proofs)
        vals = self.program.evaluate(self.consts, inputs)
        self.program.check_words(vals, ws)
        return self.program.device_witness(self.prover, vals, reuse=reuse)

    def prove(self, proofs, num_queries=28, pow_bits=16):
        """(root proof, public inputs) for a batch of proofs of the leaf circuit"""
        dw, public = self.witness(proofs, reuse=True)
        return self.circuit.prove_(dw, num_queries, pow_bits, public=public), public

    def free(self):
        self.program.release(self.prover)
        self.circuit.free()
