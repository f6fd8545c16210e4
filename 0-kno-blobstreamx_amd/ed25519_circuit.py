"""Ed25519 signature verification IN-CIRCUIT (SURVEY.md §8a row a10 seen from the circuit side, §8f item 4, VERDICT r2 "missing" 1; upstream
names recalled, unverified — reference file:line NONE, the mount is empty: curta's Ed25519 / SHA-512 chips behind plonky2x
``curta_eddsa_verify``).  Upstream proves the curve arithmetic in a separate STARK and verifies that proof in-circuit; here the same JOB is done
inside the one proof system with the gates it already has — no new row kinds, so prover kernels, verifiers and the recursion circuit are untouched:

  * ``NNF``      the NON-NATIVE field F_q, q = 2^255 - 19, on eleven 24-bit limbs.  A product a * b = k * q + r is shown over the INTEGERS column by
                 column of the limb convolution: 121 multiply-add gates, r / k / carries supplied by the prover (witness op 14, csrc/nnf25519.h)
                 and range-checked on ADD rows (r tight, k and the signed carries 32-bit): 55 range slots + ~230 gates = about 25 rows.
                 Additions / subtractions are limb-wise gates on "loose" limbs; every element carries a static bound and a product whose columns
                 could leave the carry range is refused at build time.
  * ``Edwards``  extended twisted-Edwards points (a = -1): doubling, complete addition with (projective or affine) Niels operands, 16-way one-hot
                 table look-ups; [S]B by 64 fixed-base windows (constant tables, no doublings), [k]A by a 16-entry table of A and 4-bit windows.
  * ``Sha512Gadget``  SHA-512 by bit decomposition over the arithmetic gate, 64-bit words as two 32-bit halves (Goldilocks cannot hold a 64-bit
                 sum): ~166k gates per block.
  * ``verify_statement``  RFC 8032 §5.1.7 with the cofactorless equation [S]B = R + [k]A: canonical decoding of A and R (the x coordinates are
                 witnesses checked against the curve equation and the sign bit), S < L, k = SHA-512(R || A || M) mod L (quotient witnessed,
                 k < L), the two scalar multiplications, projective comparison.  A signature that does not verify cannot be laid down.
One signature = 2 312 field products (half-size scalars), 61.6k rows of 144 routed wires: a 2^16-row circuit (bench leg ``combined_skip``).
Everything here is build-defined (NOT curta's AIR); formats are RFC 8032 / FIPS 180-4 restated from memory and pinned by the RFC 8032 §7.1
vectors and OpenSSL-made fixtures (tests/golden/ed25519.json) and hashlib."""
import hashlib
import struct

from . import P

NL, LB = 11, 24
MASK = (1 << LB) - 1
Q = (1 << 255) - 19
ELL = (1 << 252) + 27742317777372353535851937790883648493
D = (-121665 * pow(121666, Q - 2, Q)) % Q
CARRY_OFF = 1 << 28
# |column| of a product must stay below 2^24 * (2^32 - CARRY_OFF) (positive carries) with room for the k, r and carry-in terms
PRODUCT_LIMIT = (1 << LB) * ((1 << 32) - CARRY_OFF) - (1 << 40)


def limbs_of(v, n=NL):
    return [(v >> (LB * i)) & MASK for i in range(n)]


class Fq:
    """an element of F_q as limb variables; bound = strict upper bound of every limb's integer value"""
    __slots__ = ("limbs", "bound")

    def __init__(self, limbs, bound):
        self.limbs, self.bound = list(limbs), int(bound)


class NNF:
    def __init__(self, builder):
        self.b = builder
        self.one, self.zero = builder.constant(1), builder.constant(0)
        self.c2_8 = builder.constant(1 << 8)
        self._bias = {}
        self._c24 = {}
        self.n_mul = 0
        self.ONE = self.const(1)

    def cvar24(self, v):
        """a variable holding the constant v < 2^24, assembled from BYTE constants with two gates: the builder types rows by their gate
        constants, so a constant of its own costs a row — 256 byte constants plus two shared row types serve every table entry of the 64
        fixed-base tables (33 792 distinct limbs) instead"""
        if v not in self._c24:
            b = self.b
            b0, b1, b2 = (b.constant((v >> s) & 0xFF) for s in (0, 8, 16))
            self._c24[v] = b.arith(1 << 16, 1, 0, b2, self.one, b.arith(1 << 8, 1, 0, b1, self.one, b0))
        return self._c24[v]

    # ---- values ------------------------------------------------------------------------------------------------------------------------
    def value(self, x):
        return sum(self.b.value(v) << (LB * i) for i, v in enumerate(x.limbs))

    def const(self, v):
        return Fq([self.cvar24(w) for w in limbs_of(v % Q)], 1 << LB)

    def check_tight(self, v):
        """v < 2^24: v and v * 2^8 are 32-bit words (v * 2^8 < 2^40 cannot wrap)"""
        self.b.range32(v)
        self.b.range32(self.b.arith(1, 0, 0, v, self.c2_8, v))

    def witness(self, value):
        """a free input element (11 tight limbs, range-checked); the caller constrains what it is"""
        vs = [self.b.var(w) for w in limbs_of(value)]
        for v in vs:
            self.check_tight(v)
        return Fq(vs, 1 << LB)

    # ---- linear operations on loose limbs ------------------------------------------------------------------------------------------------
    def _bias_limbs(self, level):
        """a multiple of q whose limbs all lie in [level, level + 2^24): added to a difference it keeps every limb non-negative"""
        if level not in self._bias:
            base = sum(level << (LB * i) for i in range(NL))
            adj = limbs_of((-base) % Q)
            self._bias[level] = [level + a for a in adj]
        return self._bias[level]

    def lincomb(self, pos, neg=()):
        """sum(pos) - sum(neg) (mod q) on loose limbs: one gate per term and limb; bias = a multiple of q that dominates the subtracted limbs"""
        b = self.b
        level = sum(y.bound for y in neg)
        bias = self._bias_limbs(level) if neg else [0] * NL
        out = []
        for i in range(NL):
            acc, first = None, True
            for x in pos:
                acc = x.limbs[i] if acc is None else b.arith(1, 1, 0, x.limbs[i], self.one, acc)
            for y in neg:
                if acc is None:
                    acc = b.arith(P - 1, 0, bias[i], y.limbs[i], self.one, y.limbs[i])
                    first = False
                else:
                    acc = b.arith(P - 1, 1, bias[i] if first else 0, y.limbs[i], self.one, acc)
                    first = False
            out.append(acc)
        bound = sum(x.bound for x in pos) + (level + (1 << LB) if neg else 0)
        return Fq(out, bound)

    def add(self, x, y):
        return self.lincomb([x, y])

    def sub(self, x, y):
        return self.lincomb([x], [y])

    def scale(self, x, c):
        """x * c for a small positive integer c"""
        return Fq([self.b.arith(c, 0, 0, v, self.one, v) for v in x.limbs], x.bound * c)

    # ---- multiplication -------------------------------------------------------------------------------------------------------------------
    def mul(self, x, y):
        b = self.b
        if NL * (x.bound - 1) * (y.bound - 1) >= PRODUCT_LIMIT:
            raise AssertionError(f"product of limbs bounded by {x.bound} and {y.bound} could leave the carry range: reduce an operand first")
        r, k, c = b.nnf_mul_hints(x.limbs, y.limbs)
        self.n_mul += 1
        for v in r:
            self.check_tight(v)
        for v in k:
            b.range32(v)
        for v in c:
            b.range32(b.arith(0, 1, CARRY_OFF, v, v, v))                      # -2^28 <= c < 2^32 - 2^28
        lin = lambda acc, coef, v: b.arith(coef % P, 0, 0, v, self.one, v) if acc is None else b.arith(coef % P, 1, 0, v, self.one, acc)
        for t in range(2 * NL):
            acc = None
            for i in range(NL):
                j = t - i
                if 0 <= j < NL:
                    acc = b.arith(1, 0, 0, x.limbs[i], y.limbs[j], x.limbs[i]) if acc is None else b.arith(1, 1, 0, x.limbs[i], y.limbs[j], acc)
            if t < NL + 1:
                acc = lin(acc, 19, k[t])
            if 0 <= t - 10 < NL + 1:
                acc = lin(acc, -(1 << 15), k[t - 10])
            if t < NL:
                acc = lin(acc, -1, r[t])
            if t > 0:
                acc = lin(acc, 1, c[t - 1])
            if t < 2 * NL - 1:
                acc = lin(acc, -(1 << LB), c[t])
            b.assert_equal(acc, self.zero)
        return Fq(r, 1 << LB)

    def sqr(self, x):
        return self.mul(x, x)

    def assert_zero(self, x):
        """x = 0 (mod q): the canonical remainder of x * 1 is zero"""
        z = self.mul(x, self.ONE)
        for v in z.limbs:
            self.b.assert_equal(v, self.zero)

    def assert_equal(self, x, y):
        self.assert_zero(self.sub(x, y))

    # ---- selection ------------------------------------------------------------------------------------------------------------------------
    def mux(self, onehot, table):
        """sum_j onehot[j] * table[j] limb by limb (onehot: boolean variables with exactly one set)"""
        b = self.b
        out = []
        for i in range(NL):
            acc = None
            for s, e in zip(onehot, table):
                acc = b.arith(1, 0, 0, s, e.limbs[i], s) if acc is None else b.arith(1, 1, 0, s, e.limbs[i], acc)
            out.append(acc)
        return Fq(out, max(e.bound for e in table))

    def mux_const(self, onehot, values):
        """the same for constant table entries (integers mod q), through constant variables (cvar24)"""
        return self.mux(onehot, [self.const(v) for v in values])

    # ---- canonical form ---------------------------------------------------------------------------------------------------------------------
    def assert_le_const(self, limbs, c):
        """the integer with these TIGHT limbs is <= c: a witness d >= 0 with x + d = c, limb by limb with boolean borrows"""
        b = self.b
        cl = limbs_of(c, len(limbs))
        assert c >> (LB * len(limbs)) == 0
        borrow = None                                                            # 1 = a borrow went into this limb
        for i, x in enumerate(limbs):
            s = b.arith(P - 1, 0, cl[i] + (1 << LB), x, self.one, x)              # c_i + 2^24 - x_i
            if borrow is not None:
                s = b.arith(P - 1, 1, 0, borrow, self.one, s)
            d = b.bit_field(s, 0, LB)
            nb = b.bit(s, LB)                                                    # 1 = no borrow out
            b.assert_bool(nb)
            self.check_tight(d)
            b.assert_equal(b.arith(1 << LB, 1, 0, nb, self.one, d), s)
            borrow = b.arith(P - 1, 0, 1, nb, self.one, nb)                       # 1 - nb
        b.assert_equal(borrow, self.zero)

    def parity(self, x_limb0):
        """the low bit of a tight limb, as a boolean variable"""
        b = self.b
        bit, rest = b.bit(x_limb0, 0), b.bit_field(x_limb0, 1, LB - 1)
        b.assert_bool(bit)
        b.range32(rest)
        b.assert_equal(b.arith(2, 1, 0, rest, self.one, bit), x_limb0)
        return bit


# ---- the curve ---------------------------------------------------------------------------------------------------------------------------
def _ed_add(P1, P2):
    x1, y1, x2, y2 = *P1, *P2
    t = D * x1 * x2 * y1 * y2 % Q
    return ((x1 * y2 + x2 * y1) * pow(1 + t, Q - 2, Q) % Q, (y1 * y2 + x1 * x2) * pow(1 - t, Q - 2, Q) % Q)


def _ed_mul(s, Pt):
    """[s]Pt for an affine point, in extended coordinates (one inversion at the end): host-side helper for tables, tests and synthetic keys"""
    def add(A, B):
        a = (A[1] - A[0]) * (B[1] - B[0]) % Q
        b = (A[1] + A[0]) * (B[1] + B[0]) % Q
        c = 2 * D * A[3] * B[3] % Q
        d = 2 * A[2] * B[2] % Q
        e, f, g, h = b - a, d - c, d + c, b + a
        return (e * f % Q, g * h % Q, f * g % Q, e * h % Q)
    acc, cur = (0, 1, 1, 0), (Pt[0], Pt[1], 1, Pt[0] * Pt[1] % Q)
    while s:
        if s & 1:
            acc = add(acc, cur)
        cur = add(cur, cur)
        s >>= 1
    zi = pow(acc[2], Q - 2, Q)
    return (acc[0] * zi % Q, acc[1] * zi % Q)


def recover_x(y, sign):
    """RFC 8032 §5.1.3; None when y does not decode"""
    if y >= Q:
        return None
    x2 = (y * y - 1) * pow(D * y * y + 1, Q - 2, Q) % Q
    if x2 == 0:
        return None if sign else 0
    x = pow(x2, (Q + 3) // 8, Q)
    if (x * x - x2) % Q:
        x = x * pow(2, (Q - 1) // 4, Q) % Q
    if (x * x - x2) % Q:
        return None
    return Q - x if (x & 1) != sign else x


BY = 4 * pow(5, Q - 2, Q) % Q
BASE = (recover_x(BY, 0), BY)
_FIXED_TABLES = None


def fixed_base_tables():
    """64 tables of the multiples 0..15 of 16^w * B in affine Niels form (y + x, y - x, 2 d x y): constants of the circuit"""
    global _FIXED_TABLES
    if _FIXED_TABLES is None:
        tabs, base = [], BASE
        for _ in range(64):
            row, acc = [], (0, 1)
            for _j in range(16):
                row.append(((acc[1] + acc[0]) % Q, (acc[1] - acc[0]) % Q, 2 * D * acc[0] * acc[1] % Q))
                acc = _ed_add(acc, base)
            tabs.append(row)
            base = acc                                                           # 16 * base
        _FIXED_TABLES = tabs
    return _FIXED_TABLES


class Edwards:
    """points in extended coordinates (X, Y, Z, T), x = X/Z, y = Y/Z, T = XY/Z, on -x^2 + y^2 = 1 + d x^2 y^2"""

    def __init__(self, nnf):
        self.f = nnf
        self.d2 = nnf.const(2 * D)

    def identity(self):
        f = self.f
        return (f.const(0), f.const(1), f.const(1), f.const(0))

    def double(self, Pt, need_t=True):
        f = self.f
        X, Y, Z, _ = Pt
        A, B = f.sqr(X), f.sqr(Y)
        C = f.mul(Z, f.scale(Z, 2))                                               # 2 Z^2
        E = f.scale(f.mul(X, Y), 2)                                               # 2 X Y
        G = f.sub(B, A)                                                           # D + B with D = -A
        F = f.lincomb([B], [A, C])                                                # G - C
        H = f.lincomb([], [A, B])                                                 # D - B
        return (f.mul(E, F), f.mul(G, H), f.mul(F, G), f.mul(E, H) if need_t else None)

    def to_niels(self, Pt):
        f = self.f
        X, Y, Z, T = Pt
        return (f.add(Y, X), f.sub(Y, X), f.scale(Z, 2), f.mul(T, self.d2))

    def add_niels(self, Pt, N, need_t=True):
        """Pt + N for N = (y+x, y-x, 2z or None for an affine operand, 2dt): the complete unified addition (a = -1)"""
        f = self.f
        X, Y, Z, T = Pt
        ypx, ymx, z2, t2d = N
        A = f.mul(f.sub(Y, X), ymx)
        B = f.mul(f.add(Y, X), ypx)
        C = f.mul(T, t2d)
        Dd = f.scale(Z, 2) if z2 is None else f.mul(Z, z2)
        E, F, G, H = f.sub(B, A), f.sub(Dd, C), f.add(Dd, C), f.add(B, A)
        return (f.mul(E, F), f.mul(G, H), f.mul(F, G), f.mul(E, H) if need_t else None)

    def onehot16(self, bits4):
        """16 boolean selectors from 4 boolean variables (LSB first): selector j is 1 iff the bits spell j"""
        b = self.f.b
        sel = [self.f.one]
        for bit in reversed(bits4):                                               # the MSB first, so that the final index is sum bit_i 2^i
            nxt = []
            for s in sel:
                s1 = b.arith(1, 0, 0, s, bit, s)
                s0 = b.arith(1, P - 1, 0, s, self.f.one, s1)
                nxt += [s0, s1]
            sel = nxt
        return sel

    def mul_base(self, nibbles):
        """[s]B for s = sum nibble_w 16^w: 64 mixed additions with constant tables, no doublings.  nibbles[w] = 4 boolean variables (LSB first)"""
        f = self.f
        acc = self.identity()
        tabs = fixed_base_tables()
        for w, bits in enumerate(nibbles):
            sel = self.onehot16(bits)
            N = (f.mux_const(sel, [e[0] for e in tabs[w]]), f.mux_const(sel, [e[1] for e in tabs[w]]), None,
                 f.mux_const(sel, [e[2] for e in tabs[w]]))
            acc = self.add_niels(acc, N)
        return acc

    def mul_var(self, Pt, nibbles):
        """[k]Pt by a 16-entry table of multiples and 4-bit windows (MSB window first)"""
        f = self.f
        ident = self.identity()
        n1 = self.to_niels(Pt)
        mults = [ident, Pt]
        for _ in range(14):
            mults.append(self.add_niels(mults[-1], n1))
        table = [(f.const(1), f.const(1), f.const(2), f.const(0)), n1] + [self.to_niels(m) for m in mults[2:]]
        acc = None
        for bits in reversed(nibbles):
            sel = self.onehot16(bits)
            N = tuple(f.mux(sel, [e[c] for e in table]) for c in range(4))
            if acc is None:
                acc = self.add_niels(ident, N)
                continue
            for step in range(4):
                acc = self.double(acc, need_t=(step == 3))
            acc = self.add_niels(acc, N)
        return acc

    def table16(self, Pt):
        """the Niels forms of [0]Pt .. [15]Pt"""
        f = self.f
        n1 = self.to_niels(Pt)
        mults = [self.identity(), Pt]
        for _ in range(14):
            mults.append(self.add_niels(mults[-1], n1))
        return [(f.const(1), f.const(1), f.const(2), f.const(0)), n1] + [self.to_niels(m) for m in mults[2:]]

    def mul_var2(self, P1, nibbles1, P2, nibbles2):
        """[a]P1 + [b]P2 on ONE doubling chain (Straus): per 4-bit window four doublings and two table additions"""
        f = self.f
        assert len(nibbles1) == len(nibbles2)
        t1, t2 = self.table16(P1), self.table16(P2)
        acc = None
        for bits1, bits2 in zip(reversed(nibbles1), reversed(nibbles2)):
            if acc is not None:
                for step in range(4):
                    acc = self.double(acc, need_t=(step == 3))
            s1, s2 = self.onehot16(bits1), self.onehot16(bits2)
            acc = self.add_niels(self.identity() if acc is None else acc, tuple(f.mux(s1, [e[c] for e in t1]) for c in range(4)))
            acc = self.add_niels(acc, tuple(f.mux(s2, [e[c] for e in t2]) for c in range(4)))
        return acc

    def decode(self, y, x, sign_bit):
        """(x, y) is on the curve, x and y are canonical and x's parity is the sign bit: RFC 8032 §5.1.3 with x as a checked witness"""
        f = self.f
        f.assert_le_const(y.limbs, Q - 1)
        f.assert_le_const(x.limbs, Q - 1)
        yy = f.sqr(y)
        u = f.lincomb([yy], [f.ONE])                                              # y^2 - 1
        v = f.add(f.mul(yy, f.const(D)), f.ONE)                                   # d y^2 + 1
        f.assert_equal(f.mul(v, f.sqr(x)), u)
        f.b.assert_equal(f.parity(x.limbs[0]), sign_bit)
        return (x, y, f.const(1), f.mul(x, y))


# ---- SHA-512 by bit decomposition ------------------------------------------------------------------------------------------------------------
K512 = [
    0x428a2f98d728ae22, 0x7137449123ef65cd, 0xb5c0fbcfec4d3b2f, 0xe9b5dba58189dbbc, 0x3956c25bf348b538, 0x59f111f1b605d019, 0x923f82a4af194f9b,
    0xab1c5ed5da6d8118, 0xd807aa98a3030242, 0x12835b0145706fbe, 0x243185be4ee4b28c, 0x550c7dc3d5ffb4e2, 0x72be5d74f27b896f, 0x80deb1fe3b1696b1,
    0x9bdc06a725c71235, 0xc19bf174cf692694, 0xe49b69c19ef14ad2, 0xefbe4786384f25e3, 0x0fc19dc68b8cd5b5, 0x240ca1cc77ac9c65, 0x2de92c6f592b0275,
    0x4a7484aa6ea6e483, 0x5cb0a9dcbd41fbd4, 0x76f988da831153b5, 0x983e5152ee66dfab, 0xa831c66d2db43210, 0xb00327c898fb213f, 0xbf597fc7beef0ee4,
    0xc6e00bf33da88fc2, 0xd5a79147930aa725, 0x06ca6351e003826f, 0x142929670a0e6e70, 0x27b70a8546d22ffc, 0x2e1b21385c26c926, 0x4d2c6dfc5ac42aed,
    0x53380d139d95b3df, 0x650a73548baf63de, 0x766a0abb3c77b2a8, 0x81c2c92e47edaee6, 0x92722c851482353b, 0xa2bfe8a14cf10364, 0xa81a664bbc423001,
    0xc24b8b70d0f89791, 0xc76c51a30654be30, 0xd192e819d6ef5218, 0xd69906245565a910, 0xf40e35855771202a, 0x106aa07032bbd1b8, 0x19a4c116b8d2d0c8,
    0x1e376c085141ab53, 0x2748774cdf8eeb99, 0x34b0bcb5e19b48a8, 0x391c0cb3c5c95a63, 0x4ed8aa4ae3418acb, 0x5b9cca4f7763e373, 0x682e6ff3d6b2b8a3,
    0x748f82ee5defb2fc, 0x78a5636f43172f60, 0x84c87814a1f0ab72, 0x8cc702081a6439ec, 0x90befffa23631e28, 0xa4506cebde82bde9, 0xbef9a3f7b2c67915,
    0xc67178f2e372532b, 0xca273eceea26619c, 0xd186b8c721c0c207, 0xeada7dd6cde0eb1e, 0xf57d4f7fee6ed178, 0x06f067aa72176fba, 0x0a637dc5a2c898a6,
    0x113f9804bef90dae, 0x1b710b35131c471b, 0x28db77f523047d84, 0x32caab7b40c72493, 0x3c9ebe0a15c9bebc, 0x431d67c49c100d4c, 0x4cc5d4becb3e42b6,
    0x597f299cfc657e2a, 0x5fcb6fab3ad6faec, 0x6c44198c4a475817]
IV512 = [0x6a09e667f3bcc908, 0xbb67ae8584caa73b, 0x3c6ef372fe94f82b, 0xa54ff53a5f1d36f1, 0x510e527fade682d1, 0x9b05688c2b3e6c1f, 0x1f83d9abfb41bd6b,
         0x5be0cd19137e2179]


class Sha512Gadget:
    """a word = (bits[64] LSB first, lo, hi): bit variables and the two packed 32-bit halves"""

    def __init__(self, builder):
        self.b = builder
        self.one, self.two, self.zero = builder.constant(1), builder.constant(2), builder.constant(0)
        self.c2_32 = builder.constant(1 << 32)

    def xor(self, x, y):
        s = self.b.arith(1, 1, 0, x, self.one, y)
        return self.b.arith(P - 2, 1, 0, x, y, s)

    def pack(self, bits):
        acc = bits[-1]
        for bit in reversed(bits[:-1]):
            acc = self.b.arith(1, 1, 0, acc, self.two, bit)
        return acc

    def word(self, bits):
        return (list(bits), self.pack(bits[:32]), self.pack(bits[32:]))

    def const_word(self, v):
        bits = [self.one if (v >> i) & 1 else self.zero for i in range(64)]
        return (bits, self.b.constant(v & 0xFFFFFFFF), self.b.constant(v >> 32))

    def _split(self, total, extra):
        """total < 2^(32 + extra): its low 32 bits (boolean variables) and the value above them"""
        b = self.b
        assert b.value(total) < (1 << (32 + extra))
        bits = [b.bit(total, i) for i in range(32 + extra)]
        for bit in bits:
            b.assert_bool(bit)
        low, high = self.pack(bits[:32]), self.pack(bits[32:])
        b.assert_equal(b.arith(1, 1, 0, high, self.c2_32, low), total)
        return bits[:32], low, high

    def add(self, words, const=0):
        """(sum of the words + const) mod 2^64, by halves with the carry between them"""
        b = self.b
        n = len(words) + (1 if const else 0)
        extra = max(1, (n).bit_length())
        lo = words[0][1]
        for w in words[1:]:
            lo = b.arith(1, 1, 0, lo, self.one, w[1])
        if const:
            lo = b.arith(1, 0, const & 0xFFFFFFFF, lo, self.one, lo)
        lo_bits, lo_w, carry = self._split(lo, extra)
        hi = b.arith(1, 1, 0, carry, self.one, words[0][2])
        for w in words[1:]:
            hi = b.arith(1, 1, 0, hi, self.one, w[2])
        if const:
            hi = b.arith(1, 0, const >> 32, hi, self.one, hi)
        hi_bits, hi_w, _ = self._split(hi, extra + 1)
        return (lo_bits + hi_bits, lo_w, hi_w)

    @staticmethod
    def rotr(bits, r):
        return [bits[(i + r) % 64] for i in range(64)]

    def shr(self, bits, r):
        return [bits[i + r] if i + r < 64 else self.zero for i in range(64)]

    def xor3w(self, x, y, z):
        return self.word([self.xor(self.xor(a, c), e) for a, c, e in zip(x, y, z)])

    def compress(self, state, block):
        b = self.b
        w = list(block)
        for t in range(16, 80):
            s0 = self.xor3w(self.rotr(w[t - 15][0], 1), self.rotr(w[t - 15][0], 8), self.shr(w[t - 15][0], 7))
            s1 = self.xor3w(self.rotr(w[t - 2][0], 19), self.rotr(w[t - 2][0], 61), self.shr(w[t - 2][0], 6))
            w.append(self.add([w[t - 16], s0, w[t - 7], s1]))
        a, bb, c, d, e, f, g, h = state
        for t in range(80):
            S1 = self.xor3w(self.rotr(e[0], 14), self.rotr(e[0], 18), self.rotr(e[0], 41))
            ch = self.word([b.arith(1, 1, 0, x, b.arith(1, P - 1, 0, y, self.one, z), z) for x, y, z in zip(e[0], f[0], g[0])])
            S0 = self.xor3w(self.rotr(a[0], 28), self.rotr(a[0], 34), self.rotr(a[0], 39))
            mj = self.word([b.arith(1, 1, 0, z, self.xor(x, y), b.arith(1, 0, 0, x, y, x)) for x, y, z in zip(a[0], bb[0], c[0])])
            new_e = self.add([d, h, S1, ch, w[t]], const=K512[t])
            new_a = self.add([h, S1, ch, w[t], S0, mj], const=K512[t])
            a, bb, c, d, e, f, g, h = new_a, a, bb, c, new_e, e, f, g
        return [self.add([x, y]) for x, y in zip(state, (a, bb, c, d, e, f, g, h))]

    def hash_bytes(self, byte_bits):
        """SHA-512 of a message given as bytes, each a list of 8 boolean variables LSB first.  Returns the 64 digest bytes in the same form."""
        n = len(byte_bits)
        const_byte = lambda v: [self.one if (v >> i) & 1 else self.zero for i in range(8)]
        msg = list(byte_bits) + [const_byte(0x80)] + [const_byte(0)] * ((111 - n) % 128) + [const_byte(v) for v in (8 * n).to_bytes(16, "big")]
        assert len(msg) % 128 == 0
        state = [self.const_word(v) for v in IV512]
        for off in range(0, len(msg), 128):
            words = []
            for k in range(16):
                bts = msg[off + 8 * k: off + 8 * k + 8]                           # big-endian word: first byte = bits 56..63
                words.append(self.word([bit for byte in reversed(bts) for bit in byte]))
            state = self.compress(state, words)
        out = []
        for wd in state:
            for j in range(8):                                                    # digest byte 8w + j = bits (7-j)*8 .. of word w
                out.append(wd[0][(7 - j) * 8: (7 - j) * 8 + 8])
        return out


# ---- the statement ---------------------------------------------------------------------------------------------------------------------------
def _byte_input(b, g, value):
    """a free input byte: (variable, its 8 boolean bit variables LSB first); the decomposition is the range check"""
    v = b.var(value)
    bits = [b.bit(v, i) for i in range(8)]
    for bit in bits:
        b.assert_bool(bit)
    b.assert_equal(g.pack(bits), v)
    return v, bits


def _limbs_from_byte_bits(g, byte_bits, n_limbs):
    """little-endian integer of the bytes as 24-bit limbs (three bytes each; a last limb may be shorter): packed from the bits"""
    flat = [bit for byte in byte_bits for bit in byte]
    return [g.pack(flat[LB * i: LB * i + LB]) for i in range(n_limbs) if flat[LB * i: LB * i + LB]]


HALF_NIBBLES = 36                     # windows of the half-size scalars: 144 bits (a reduced basis vector longer than that has probability ~2^-36)


def half_size_pair(k):
    """(u, v, neg): u ODD, 0 < u, 0 <= v, both below 2^144, with  u * k = (-v if neg else v)  (mod 8 L).  A short vector of the lattice
    {(t, r): r = t k mod 8L} (determinant 8L ~ 2^255.4) by Lagrange reduction; among the reduced basis and its sum / difference one vector has an
    odd first coordinate (the lattice contains (1, k)).  With them  [S]B = R + [k]A  <=>  [u S]B = [u]R + [+-v]A  — two 144-bit scalars on ONE shared
    doubling chain instead of a 253-bit one.  The modulus is the GROUP EXPONENT 8L, not L: [u k]A = [+-v]A then holds for EVERY curve point A, also
    one with a small-order component (a mod-L relation would be off by [q L]A, a point of order up to 8, for such keys); and u is odd and below L,
    so [u]X = O forces X = O: the split form is EXACTLY the cofactorless equation of RFC 8032, not an up-to-torsion variant.  ValueError in the
    ~2^-36 case that no such pair fits 144 bits."""
    k %= ELL
    b1, b2 = (1, k), (0, 8 * ELL)
    norm = lambda v: v[0] * v[0] + v[1] * v[1]
    if norm(b1) > norm(b2):
        b1, b2 = b2, b1
    while True:
        mu = (b1[0] * b2[0] + b1[1] * b2[1] + norm(b1) // 2) // norm(b1)         # nearest integer of <b1, b2> / <b1, b1>
        b2 = (b2[0] - mu * b1[0], b2[1] - mu * b1[1])
        if norm(b2) >= norm(b1):
            break
        b1, b2 = b2, b1
    cands = [c for c in (b1, b2, (b1[0] + b2[0], b1[1] + b2[1]), (b1[0] - b2[0], b1[1] - b2[1])) if c[0] & 1]
    t, r = min(cands, key=lambda c: max(abs(c[0]), abs(c[1])))
    if r < 0:
        t, r = -t, -r
    u, neg = abs(t), t < 0
    if u >> (4 * HALF_NIBBLES) or r >> (4 * HALF_NIBBLES):
        raise ValueError("no half-size scalar pair below 2^144 for this signature (probability ~2^-36)")
    assert (u * k - (-r if neg else r)) % (8 * ELL) == 0 and u & 1 and u < ELL
    return u, r, neg


_DUMMY = {}


def dummy_signature(msg_len):
    """a fixed valid (public key, signature, message) triple for messages of this length: what a slot whose `signed` flag is 0 verifies instead of
    its validator's signature (the circuit cannot skip constraints, it selects their inputs)"""
    if msg_len not in _DUMMY:
        m = bytes(msg_len)
        pub, sig = keypair_and_sign(hashlib.sha256(b"glprover unsigned slot").digest(), m)
        _DUMMY[msg_len] = (pub, sig, m)
    return _DUMMY[msg_len]


def witness_inputs(pub32, sig64, msg, flag=None, record=None, split_scalars=True):
    """the input vector of a program recorded from verify_statement for a message of this length, in the order the statement creates its free
    variables: [with a flag: the flag, the validator's key bytes, the message bytes, then for the VERIFIED triple — the validator's own when the
    flag is 1, dummy_signature's when it is 0 —] A bytes, R bytes, S bytes, message bytes [without a flag only], then the limbs of x_A, x_R, of
    the quotient t and of k = SHA-512(R || A || M) mod L.  ValueError when A or R does not decode (no witness exists).
    record: the 37-word record of the GPU witness kernel for the VERIFIED triple (glp_ed25519_witness: verdict, k, decoded A and R ...): x_A, x_R
    and k are then taken from the device's computation instead of being recomputed with Python integers (the circuit checks them either way)."""
    pub32, msg = bytes(pub32), bytes(msg)
    head = []
    if flag is not None:
        head = [1 if flag else 0] + list(pub32) + list(msg)
        if not flag:
            pub32, sig64, msg = dummy_signature(len(msg))
    sig64 = bytes(sig64)
    if len(sig64) != 64:
        raise ValueError("a signature is 64 bytes")
    h = int.from_bytes(hashlib.sha512(sig64[:32] + pub32 + msg).digest(), "little")
    if record is None:
        ya, yr = int.from_bytes(pub32, "little"), int.from_bytes(sig64[:32], "little")
        xa, xr = recover_x(ya & ((1 << 255) - 1), ya >> 255), recover_x(yr & ((1 << 255) - 1), yr >> 255)
        if xa is None or xr is None:
            raise ValueError("the public key or R does not decode to a curve point")
        t, k = divmod(h, ELL)
    else:
        word = lambda o: sum(int(record[o + j]) << (64 * j) for j in range(4))
        if not int(record[0]):
            raise ValueError("the GPU witness kernel rejects this signature: no witness exists")
        k, xa, xr = word(1), word(5), word(13)
        t = (h - k) // ELL                                                    # the circuit checks h = t * L + k: a wrong k from the device cannot pass
    body = (list(pub32) if flag is None else []) + list(sig64[:32]) + list(sig64[32:]) + (list(msg) if flag is None else [])
    out = head + body + limbs_of(xa) + limbs_of(xr) + limbs_of(t) + limbs_of(k)
    if not split_scalars:
        return out
    # the half-size form: u, v, the sign, and the quotients / remainder of  u * S = q1 * L + w  and  u * k -+ v = q2 * 8L
    S = int.from_bytes(sig64[32:], "little")
    u, v, neg = half_size_pair(k)
    q1, w = divmod(u * S, ELL)
    q2 = (u * k - (-v if neg else v)) // (8 * ELL)
    return out + limbs_of(u, 6) + limbs_of(v, 6) + [1 if neg else 0] + limbs_of(q1, 7) + limbs_of(w) + limbs_of(q2, 7)


def verify_statement(b, pub32, sig64, msg, flag=None, split_scalars=True):
    """Lay down, on builder b (144 wires: the range checks use ADD rows), the verification of ONE Ed25519 signature (RFC 8032 §5.1.7, equation
    [S]B = R + [k]A).  Free inputs in witness_inputs' order.  flag (None, or the slot's `signed` value): with a flag the statement is
    "flag = 1  =>  sig64 is the key's signature of msg" — the key and message that enter the verification are SELECTED by the flag between the
    slot's own and dummy_signature's (a slot that did not sign verifies the fixed dummy triple), so one circuit serves signers and non-signers.
    Returns {"key_words": 8 big-endian 32-bit word variables of the slot's public key, "msg_bytes": the message byte variables,
    "flag": the flag variable or None, "stats": {...}}.  ValueError when the signature does not verify (some constraint fails on its witness)."""
    pub32, msg = bytes(pub32), bytes(msg)
    vals = witness_inputs(pub32, sig64, msg, flag, split_scalars=split_scalars)
    it = iter(vals)
    f = NNF(b)
    g = Sha512Gadget(b)
    ed = Edwards(f)

    def decomposed(v):
        bits = [b.bit(v, i) for i in range(8)]
        for bit in bits:
            b.assert_bool(bit)
        b.assert_equal(g.pack(bits), v)
        return v, bits
    flag_var = None
    if flag is None:
        A_bytes = [_byte_input(b, g, next(it)) for _ in range(32)]
        own_key = A_bytes
    else:
        flag_var = b.var(next(it))
        b.assert_bool(flag_var)
        own_key = [_byte_input(b, g, next(it)) for _ in range(32)]
        own_msg = [_byte_input(b, g, next(it)) for _ in range(len(msg))]
        d_pub, _, d_msg = dummy_signature(len(msg))
        # flag ? own : dummy  =  dummy + flag * (own - dummy), byte by byte; the selected byte is decomposed again (it feeds the hash)
        pick = lambda own, dv: decomposed(b.arith(1, 0, dv, flag_var, b.arith(1, 0, P - dv, own, f.one, own), flag_var))
        A_bytes = [pick(v, dv) for (v, _), dv in zip(own_key, d_pub)]
    R_bytes = [_byte_input(b, g, next(it)) for _ in range(32)]
    S_bytes = [_byte_input(b, g, next(it)) for _ in range(32)]
    if flag is None:
        M_bytes = [_byte_input(b, g, next(it)) for _ in range(len(msg))]
        own_msg = M_bytes
    else:
        M_bytes = [pick(v, dv) for (v, _), dv in zip(own_msg, d_msg)]
    x_a, x_r = f.witness(sum(next(it) << (LB * i) for i in range(NL))), f.witness(sum(next(it) << (LB * i) for i in range(NL)))
    t_q = f.witness(sum(next(it) << (LB * i) for i in range(NL)))
    k_s = f.witness(sum(next(it) << (LB * i) for i in range(NL)))

    def y_and_sign(bts):
        bits = [bit for _, byte in bts for bit in byte]
        return Fq(_limbs_from_byte_bits(g, [bits[8 * i: 8 * i + 8] for i in range(31)] + [bits[248:255]], NL), 1 << LB), bits[255]
    y_a, sign_a = y_and_sign(A_bytes)
    y_r, sign_r = y_and_sign(R_bytes)
    PA = ed.decode(y_a, x_a, sign_a)
    PR = ed.decode(y_r, x_r, sign_r)
    # S < L, as limbs and as nibbles
    s_bits = [bit for _, byte in S_bytes for bit in byte]
    s_limbs = _limbs_from_byte_bits(g, [s_bits[8 * i: 8 * i + 8] for i in range(32)], NL)
    f.assert_le_const(s_limbs, ELL - 1)
    s_nibbles = [s_bits[4 * w: 4 * w + 4] for w in range(64)]
    # k = SHA-512(R || A || M) mod L:  h = t * L + k over the integers, k < L
    digest = g.hash_bytes([byte for _, byte in R_bytes] + [byte for _, byte in A_bytes] + [byte for _, byte in M_bytes])
    h_limbs = _limbs_from_byte_bits(g, digest, 22)
    f.assert_le_const(k_s.limbs, ELL - 1)
    ell = limbs_of(ELL)
    carry = None
    for col in range(22):
        acc = None
        for i in range(NL):
            j = col - i
            if 0 <= j < NL and ell[j]:
                acc = b.arith(ell[j], 0, 0, t_q.limbs[i], f.one, t_q.limbs[i]) if acc is None else b.arith(ell[j], 1, 0, t_q.limbs[i], f.one, acc)
        if col < NL:
            acc = k_s.limbs[col] if acc is None else b.arith(1, 1, 0, k_s.limbs[col], f.one, acc)
        if carry is not None:
            acc = carry if acc is None else b.arith(1, 1, 0, carry, f.one, acc)
        acc = b.arith(P - 1, 1, 0, h_limbs[col], f.one, acc) if acc is not None else b.arith(P - 1, 0, 0, h_limbs[col], f.one, h_limbs[col])
        if col < 21:
            carry = b.bit_field(acc, LB, 32)
            b.range32(carry)
            b.assert_equal(b.arith(1 << LB, 0, 0, carry, f.one, carry), acc)
        else:
            b.assert_equal(acc, f.zero)
    k_bits = []                                                                  # the bits of k: booleans that pack to its (tight, canonical) limbs
    for i, v in enumerate(k_s.limbs):
        lb = [b.bit(v, j) for j in range(LB if i < 10 else 16)]                    # limb 10 holds bits 240..: k < L < 2^253 keeps it below 2^13
        for bit in lb:
            b.assert_bool(bit)
        b.assert_equal(g.pack(lb), v)
        k_bits += lb
    if not split_scalars:
        # [S]B = R + [k]A, compared projectively
        k_nibbles = [k_bits[4 * w: 4 * w + 4] for w in range(64)]
        Q1 = ed.mul_base(s_nibbles)
        Q2 = ed.mul_var(PA, k_nibbles)
        Q3 = ed.add_niels(Q2, (f.add(PR[1], PR[0]), f.sub(PR[1], PR[0]), None, f.mul(PR[3], ed.d2)), need_t=False)
    else:
        # The same equation with HALF-SIZE scalars (half_size_pair): for an odd u and a v below 2^144 with u k = +-v (mod 8L),
        #     [S]B = R + [k]A   <=>   [u S mod L]B = [u]R + [v](+-A)
        # (multiply by u: it is odd and invertible mod L, so nothing of order 8 or L is lost; 8L is the group exponent, so [u k]A = [+-v]A for every A).  The two 144-bit scalars share ONE doubling chain:
        # 140 doublings and 72 additions instead of 252 and 64 — about 620 field products fewer per signature.
        def small(n_limbs, n_bits):
            vs = [b.var(next(it)) for _ in range(n_limbs)]
            bits = []
            for i, v in enumerate(vs):
                lb = [b.bit(v, j) for j in range(min(LB, n_bits - LB * i))]
                for bit in lb:
                    b.assert_bool(bit)
                b.assert_equal(g.pack(lb), v)                                      # the bits ARE the limb (so it is tight, and the top ones absent)
                bits += lb
            return vs, bits
        u_l, u_bits = small(6, 4 * HALF_NIBBLES)
        v_l, v_bits = small(6, 4 * HALF_NIBBLES)
        b.assert_equal(u_bits[0], f.one)                                           # u is odd (hence non-zero)
        neg = b.var(next(it))
        b.assert_bool(neg)
        q1 = [b.range32(b.var(next(it))) for _ in range(7)]
        w_l, w_bits = small(NL, 256)
        q2 = [b.range32(b.var(next(it))) for _ in range(7)]
        sgn = b.arith(P - 2, 0, 1, neg, f.one, neg)                                 # +1 or -1
        def relation(prod_a, prod_b, quot, tail, tail_coef, modulus):
            """sum_{i+j=t} a_i b_j - sum quot_i M_j - tail_coef * tail_t = 0 over the integers (M = the modulus' limbs), column by column with
            signed carries"""
            ell_l = limbs_of(modulus)
            carry = None
            n_cols = max(len(prod_a) + len(prod_b), len(quot) + NL) + 1
            for col in range(n_cols):
                acc = None
                for i, x in enumerate(prod_a):
                    j = col - i
                    if 0 <= j < len(prod_b):
                        acc = b.arith(1, 0, 0, x, prod_b[j], x) if acc is None else b.arith(1, 1, 0, x, prod_b[j], acc)
                for i, x in enumerate(quot):
                    j = col - i
                    if 0 <= j < NL and ell_l[j]:
                        acc = b.arith(P - ell_l[j], 0, 0, x, f.one, x) if acc is None else b.arith(P - ell_l[j], 1, 0, x, f.one, acc)
                if col < len(tail):
                    t = b.arith(P - 1, 0, 0, tail[col], tail_coef, tail[col])
                    acc = t if acc is None else b.arith(1, 1, 0, t, f.one, acc)
                if carry is not None:
                    acc = carry if acc is None else b.arith(1, 1, 0, carry, f.one, acc)
                if acc is None:
                    continue
                if col < n_cols - 1:
                    shifted = b.arith(1, 0, 1 << 54, acc, f.one, acc)               # |acc| < 2^53: acc + 2^54 is positive, its bits above 24 = carry + 2^30
                    cs = b.bit_field(shifted, LB, 32)
                    b.range32(cs)
                    carry = b.arith(1, 0, P - (1 << 30), cs, f.one, cs)
                    b.assert_equal(b.arith(1 << LB, 0, 0, carry, f.one, carry), acc)
                else:
                    b.assert_equal(acc, f.zero)
        relation(u_l, s_limbs, q1, w_l, f.one, ELL)                                 # u S = q1 L + w           (B has order exactly L)
        relation(u_l, k_s.limbs, q2, v_l, sgn, 8 * ELL)                             # u k = q2 8L +- v          (8L kills every curve point)
        nib = lambda bits, n: [bits[4 * i: 4 * i + 4] for i in range(n)]
        Q1 = ed.mul_base(nib(w_bits, 64))
        # +-A: the x coordinate (and T = x y) negated when neg = 1
        nx = f.lincomb([], [PA[0]])
        ax = Fq([b.arith(1, 1, 0, neg, b.arith(1, P - 1, 0, m, f.one, x), x) for x, m in zip(PA[0].limbs, nx.limbs)], max(PA[0].bound, nx.bound))
        As = (ax, PA[1], PA[2], f.mul(ax, PA[1]))
        Q3 = ed.mul_var2(PR, nib(u_bits, HALF_NIBBLES), As, nib(v_bits, HALF_NIBBLES))
    f.assert_equal(f.mul(Q1[0], Q3[2]), f.mul(Q3[0], Q1[2]))
    f.assert_equal(f.mul(Q1[1], Q3[2]), f.mul(Q3[1], Q1[2]))
    key_words = []
    for wd in range(8):                                                          # big-endian 32-bit words of the key bytes (the signer digest's form)
        bs = [own_key[4 * wd + j][0] for j in range(4)]
        hi = b.arith(1 << 24, 1, 0, bs[0], f.one, b.arith(1 << 16, 0, 0, bs[1], f.one, bs[1]))
        key_words.append(b.arith(1 << 8, 1, 0, bs[2], f.one, b.arith(1, 1, 0, hi, f.one, bs[3])))
    return {"key_words": key_words, "msg_bytes": [v for v, _ in own_msg], "flag": flag_var, "stats": {"field_products": f.n_mul}}


def ed25519_circuit(prover, pub32, sig64, msg):
    """the circuit of verify_statement: public inputs = the 8 key words then the message bytes.  Returns (builder, statement dict)."""
    from . import SHA_GATE_WIRES
    from .recursion import CircuitBuilder
    b = CircuitBuilder(prover, n_wires=SHA_GATE_WIRES, n_routed=SHA_GATE_WIRES)        # every wire routed: 36 gate slots per row
    st = verify_statement(b, pub32, sig64, msg)
    for v in st["key_words"] + st["msg_bytes"]:
        b.public_input(v)
    return b, st


def public_inputs(pub32, msg):
    return list(struct.unpack(">8I", bytes(pub32))) + list(bytes(msg))


# ---- synthetic inputs (bench / tests): RFC 8032 §5.1.5-5.1.6 key generation and signing in plain Python ---------------------------------------
def _encode_point(Pt):
    return (Pt[1] | ((Pt[0] & 1) << 255)).to_bytes(32, "little")


def keypair_and_sign(seed32, msg):
    """(public key, signature) of msg under the key derived from a 32-byte seed — for synthetic validators; there is no network or wallet here"""
    h = hashlib.sha512(bytes(seed32)).digest()
    a = (int.from_bytes(h[:32], "little") & ((1 << 254) - 8)) | (1 << 254)
    A = _encode_point(_ed_mul(a, BASE))
    r = int.from_bytes(hashlib.sha512(h[32:] + bytes(msg)).digest(), "little") % ELL
    R = _encode_point(_ed_mul(r, BASE))
    k = int.from_bytes(hashlib.sha512(R + A + bytes(msg)).digest(), "little") % ELL
    return A, R + ((r + k * a) % ELL).to_bytes(32, "little")
