"""In-circuit SHA-256 and the statements built on it (SURVEY.md §8a rows a9/a11 seen from the CIRCUIT side, VERDICT r1 "missing" item 3 / row g1;
upstream names recalled, unverified — reference file:line NONE, the mount is empty: curta's SHA-256 chip, blobstreamx ``DataCommitmentCircuit``,
tendermintx ``verify_step`` / ``verify_skip``, plonky2x ``evm_read`` / ``evm_write``).

Two layouts of one compression on ``recursion.CircuitBuilder``:
  * ``Sha256Rows`` — the SHA row gates of the proof system itself (csrc/plonk_gates.h): words are single variables, every row
    decomposes what it needs into bit wires of its own (filled on the device); 178 rows per compression.  What everything below uses.
  * ``Sha256Gadget`` — round 2's first form, by bit decomposition over the arithmetic gate ``w = c0*x*y + c1*z + c2`` alone (xor = a + b - 2ab,
    Ch = e(f - g) + g, Maj = ab + c(a xor b), word additions with a range-checked re-decomposition): ~66k gates = 3.3k rows per compression.
    Kept as the comparison (bench) and because it needs no circuit flag.
Upstream proves its SHA rounds in a separate STARK (curta) and verifies that proof in-circuit; custom rows inside the one proof system are the
substitute here.

Statements (all build-defined, NOT upstream's circuits; encodings are public specs restated from memory: RFC 6962, protobuf, Solidity ABI):
  ``data_commitment_rows_circuit`` / ``data_commitment_circuit``   the RFC 6962 root over abi.encode(height, dataRoot) tuples (public) = the data commitment
  ``validator_set_statement``     validators_hash from keys + powers (variable-length protobuf leaves) and the > 2/3 voting-power rule
  ``header_hash_statement``       a header hash over its 14 encoded fields, one field bound to bytes computed in-circuit
  ``data_commitment_chain_statement``   the header-chain form: headers linked through last_block_id, their data_hash fields feeding the commitment
  ``step_statement`` / ``skip_statement``   the light-client step (chain link through last_block_id, one set, > 2/3) and skip (two sets, > 2/3 of
                                  the target power, > 1/3 of the trusted power) statements, with a SIGNER DIGEST as public input
  ``combined_skip_circuit``       skip + header chain + data commitment in one circuit (CombinedSkip's shape minus Ed25519)
In THESE single-circuit statements the flags saying who signed are witnesses, exposed through the signer digest (a Poseidon tree over one leaf per
validator slot) so that ``blobstream.verify_signers`` can check exactly those signatures natively.  The MapReduce form verifies them IN-CIRCUIT:
signature_mr.py proves one ed25519_circuit leaf per slot and folds the slots into the same digest, combined_skip_mr.py equates the two (round 3).
data_commitment_mr.py builds the range MapReduce on top.
"""
import struct

from . import P
from .recursion import CircuitBuilder

K256 = [
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01,
    0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc,
    0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147,
    0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
    0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08,
    0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
    0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2]
IV256 = [0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19]


class Sha256Gadget:
    """words are (bits, word): bits = 32 boolean variables LSB first, word = the variable holding their value"""

    def __init__(self, builder):
        self.b = builder
        self.one, self.two, self.zero = builder.constant(1), builder.constant(2), builder.constant(0)
        self.c2_32 = builder.constant(1 << 32)

    # ---- bits -------------------------------------------------------------------------------------------------------------------
    def xor(self, x, y):
        s = self.b.arith(1, 1, 0, x, self.one, y)                      # x + y
        return self.b.arith(P - 2, 1, 0, x, y, s)                      # x + y - 2xy

    def xor3(self, x, y, z):
        return self.xor(self.xor(x, y), z)

    def ch(self, e, f, g):
        d = self.b.arith(1, P - 1, 0, f, self.one, g)                  # f - g
        return self.b.arith(1, 1, 0, e, d, g)                          # e(f - g) + g

    def maj(self, a, bb, c):
        x = self.xor(a, bb)
        m = self.b.arith(1, 0, 0, a, bb, a)                            # ab
        return self.b.arith(1, 1, 0, c, x, m)                          # c(a xor b) + ab

    # ---- words ------------------------------------------------------------------------------------------------------------------
    def pack(self, bits):
        """value of a little-endian bit list as one variable (Horner from the top bit)"""
        acc = bits[-1]
        for bit in reversed(bits[:-1]):
            acc = self.b.arith(1, 1, 0, acc, self.two, bit)            # 2*acc + bit
        return acc

    def word_from_bits(self, bits):
        return (list(bits), self.pack(bits))

    def const_word(self, v):
        bits = [self.one if (v >> i) & 1 else self.zero for i in range(32)]
        return (bits, self.b.constant(v))

    def reduce32(self, total, extra_bits):
        """total = a sum of words < 2^(32 + extra_bits): returns its low 32 bits as a word.  The decomposition is a range check:
        32 + extra_bits booleans whose packed value equals total."""
        v = self.b.value(total)
        assert v < (1 << (32 + extra_bits))
        bits = [self.b.bit(total, i) for i in range(32 + extra_bits)]
        for bit in bits:
            self.b.assert_bool(bit)
        low = self.pack(bits[:32])
        hi = self.pack(bits[32:])
        self.b.assert_equal(self.b.arith(1, 1, 0, hi, self.c2_32, low), total)     # hi * 2^32 + low == total
        return (bits[:32], low)

    def add_words(self, words, const=0):
        """(sum of the words + const) mod 2^32"""
        total = words[0][1]
        for k, w in enumerate(words[1:]):
            total = self.b.arith(1, 1, const if k == 0 else 0, total, self.one, w[1])
        if len(words) == 1 and const:
            total = self.b.arith(1, 0, const, total, self.one, total)
        n_terms = len(words) + (1 if const else 0)
        return self.reduce32(total, max(1, (n_terms - 1).bit_length()))

    @staticmethod
    def rotr(bits, r):
        return [bits[(i + r) % 32] for i in range(32)]

    def shr(self, bits, r):
        return [bits[i + r] if i + r < 32 else self.zero for i in range(32)]

    def xor3_words(self, x, y, z):
        return self.word_from_bits([self.xor3(a, bb, c) for a, bb, c in zip(x, y, z)])

    # ---- compression ------------------------------------------------------------------------------------------------------------
    def compress(self, state, block_words):
        """state: 8 words, block_words: 16 words -> 8 words (FIPS 180-4 section 6.2.2)"""
        w = list(block_words)
        for t in range(16, 64):
            s0 = self.xor3_words(self.rotr(w[t - 15][0], 7), self.rotr(w[t - 15][0], 18), self.shr(w[t - 15][0], 3))
            s1 = self.xor3_words(self.rotr(w[t - 2][0], 17), self.rotr(w[t - 2][0], 19), self.shr(w[t - 2][0], 10))
            w.append(self.add_words([w[t - 16], s0, w[t - 7], s1]))
        a, bb, c, d, e, f, g, h = state
        for t in range(64):
            S1 = self.xor3_words(self.rotr(e[0], 6), self.rotr(e[0], 11), self.rotr(e[0], 25))
            chw = self.word_from_bits([self.ch(x, y, z) for x, y, z in zip(e[0], f[0], g[0])])
            S0 = self.xor3_words(self.rotr(a[0], 2), self.rotr(a[0], 13), self.rotr(a[0], 22))
            mj = self.word_from_bits([self.maj(x, y, z) for x, y, z in zip(a[0], bb[0], c[0])])
            # T1 = h + S1 + ch + K[t] + w[t] (kept as an unreduced sum), new e = d + T1, new a = T1 + S0 + maj
            t1 = self.b.arith(1, 1, K256[t], h[1], self.one, S1[1])
            t1 = self.b.arith(1, 1, 0, t1, self.one, chw[1])
            t1 = self.b.arith(1, 1, 0, t1, self.one, w[t][1])                       # < 5 * 2^32
            new_e = self.reduce32(self.b.arith(1, 1, 0, t1, self.one, d[1]), 3)     # 6 terms
            t2 = self.b.arith(1, 1, 0, S0[1], self.one, mj[1])
            new_a = self.reduce32(self.b.arith(1, 1, 0, t1, self.one, t2), 3)       # 7 terms
            a, bb, c, d, e, f, g, h = new_a, a, bb, c, new_e, e, f, g
        return [self.add_words([x, y]) for x, y in zip(state, (a, bb, c, d, e, f, g, h))]

    def hash_bits(self, msg_bits):
        """SHA-256 of a message given as bit variables, MSB first inside each byte (len % 8 == 0): returns 8 words"""
        n = len(msg_bits)
        assert n % 8 == 0
        padded = list(msg_bits) + [self.one] + [self.zero] * ((447 - n) % 512)
        padded += [self.one if (n >> (63 - i)) & 1 else self.zero for i in range(64)]
        assert len(padded) % 512 == 0
        state = [self.const_word(v) for v in IV256]
        for off in range(0, len(padded), 512):
            words = []
            for k in range(16):
                be = padded[off + 32 * k: off + 32 * k + 32]                         # big-endian: first bit = bit 31
                words.append(self.word_from_bits(list(reversed(be))))
            state = self.compress(state, words)
        return state

    def word_bits_be(self, word):
        """the 32 bits of a word, most significant first (message order)"""
        return list(reversed(word[0]))

    def public_word(self, value):
        """a 32-bit public input: the packed word is the public cell, its bits are range-checked witnesses"""
        w = self.b.var(value)                               # the program's input; its bits are computed witnesses, range-checked here
        bits = [self.b.bit(w, i) for i in range(32)]
        for bit in bits:
            self.b.assert_bool(bit)
        self.b.assert_equal(self.pack(bits), w)
        self.b.public_input(w)
        return (bits, w)


class Sha256Rows:
    """SHA-256 on the SHA row gates (csrc/plonk_gates.h): words are single variables holding 32-bit values; one compression
    is 48 W + 64 E + 64 A + 2 ADD rows instead of ~66k arithmetic gates.  Inputs handed to `compress` must be range-checked words
    (outputs of rows are; free inputs go through `word`)."""

    def __init__(self, builder):
        self.b = builder
        self.zero = builder.constant(0)
        self.c2_8, self.c2_24, self.c2_32 = builder.constant(1 << 8), builder.constant(1 << 24), builder.constant(1 << 32)
        self.one = builder.constant(1)

    def word(self, v):
        """range-check a variable as a 32-bit word"""
        return self.b.range32(v)

    def public_word(self, value):
        w = self.b.var(value)
        self.b.range32(w)
        self.b.public_input(w)
        return w

    def compress(self, state, block):
        """state: 8 words, block: 16 words -> 8 words (FIPS 180-4 section 6.2.2)"""
        b = self.b
        w = list(block)
        for t in range(16, 64):
            w.append(b.sha_w(w[t - 16], w[t - 15], w[t - 7], w[t - 2]))
        a, bb, c, d, e, f, g, h = state
        for t in range(64):
            t1, e_new = b.sha_e(e, f, g, h, d, w[t], K256[t])
            a_new = b.sha_a(a, bb, c, t1)
            a, bb, c, d, e, f, g, h = a_new, a, bb, c, e_new, e, f, g
        return [b.add32(x, y) for x, y in zip(state, (a, bb, c, d, e, f, g, h))]

    def split_byte(self, w):
        """a word as (top 24 bits, low byte): w = hi * 2^8 + lo with hi < 2^24 and lo < 2^8 (each bound = two 32-bit range checks:
        v < 2^32 and v * 2^k < 2^32 as integers, since v * 2^k < 2^56 cannot wrap)"""
        b = self.b
        hi, lo = b.bit_field(w, 8, 24), b.bit_field(w, 0, 8)
        b.assert_equal(b.arith(1, 1, 0, hi, self.c2_8, lo), w)
        b.range32(hi)
        b.range32(lo)
        b.range32(b.arith(1, 0, 0, hi, self.c2_8, hi))
        b.range32(b.arith(1, 0, 0, lo, self.c2_24, lo))
        return hi, lo

    def hash_prefixed_64(self, prefix, words16):
        """SHA-256(prefix byte || the 64 bytes of 16 big-endian words): the RFC 6962 leaf (0x00) / inner-node (0x01) hashes.  The prefix shifts
        the data by one byte: message word j = low byte of data word j-1, then the top three bytes of data word j."""
        b = self.b
        parts = [self.split_byte(w) for w in words16]
        msg = [b.arith(0, 1, prefix << 24, self.zero, self.zero, parts[0][0])]                    # prefix * 2^24 + hi_0
        for j in range(1, 16):
            msg.append(b.arith(1, 1, 0, parts[j - 1][1], self.c2_24, parts[j][0]))               # lo_{j-1} * 2^24 + hi_j
        msg.append(b.arith(1, 0, 0x80 << 16, parts[15][1], self.c2_24, self.zero))               # lo_15 * 2^24 + 0x80 * 2^16
        msg += [self.zero] * 14 + [b.constant(65 * 8)]                                            # padding, 64-bit length
        state = [b.constant(v) for v in IV256]
        for off in (0, 16):
            state = self.compress(state, msg[off:off + 16])
        return state


    # ---- byte strings ---------------------------------------------------------------------------------------------------------------
    def byte(self, v):
        """range-check a variable as a byte: v < 2^32 and v * 2^24 < 2^32 as integers (v * 2^24 < 2^56 cannot wrap)"""
        b = self.b
        b.range32(v)
        b.range32(b.arith(1, 0, 0, v, self.c2_24, v))
        return v

    def word_from_bytes(self, b4):
        """big-endian word of four range-checked bytes (constants allowed)"""
        b = self.b
        c16, c8 = b.constant(1 << 16), self.c2_8
        hi = b.arith(1, 1, 0, b4[0], self.c2_24, b.arith(1, 0, 0, b4[1], c16, b4[1]))      # b0 * 2^24 + b1 * 2^16
        return b.arith(1, 1, 0, b4[2], c8, b.arith(1, 1, 0, hi, self.one, b4[3]))           # b2 * 2^8 + (hi + b3)

    def hash_bytes(self, byte_vars):
        """SHA-256 of a message given as range-checked byte variables (any length, fixed at circuit-build time): FIPS 180-4 padding as
        constants, one compression per 64-byte block.  Returns the 8 digest words."""
        b = self.b
        n = len(byte_vars)
        pad = [b.constant(0x80)] + [self.zero] * ((55 - n) % 64) + [b.constant(v) for v in (8 * n).to_bytes(8, "big")]
        msg = list(byte_vars) + pad
        assert len(msg) % 64 == 0
        words = [self.word_from_bytes(msg[k:k + 4]) for k in range(0, len(msg), 4)]
        state = [b.constant(v) for v in IV256]
        for off in range(0, len(words), 16):
            state = self.compress(state, words[off:off + 16])
        return state

    def bytes_of_word(self, w):
        """the four big-endian bytes of a range-checked word, as range-checked byte variables"""
        b = self.b
        bs = [b.bit_field(w, 24 - 8 * k, 8) for k in range(4)]
        for v in bs:
            self.byte(v)
        b.assert_equal(self.word_from_bytes(bs), w)
        return bs


def _validator_set(b, g, pubkeys, voting_powers):
    """hash a validator set in-circuit: returns {"root": 8 word variables (validators_hash), "keys": per validator its 32 byte variables,
    "powers": per validator its power variable, "total": the sum}.  Leaves: 0x00 || SimpleValidator{pub_key{ed25519}, voting_power} with the
    power's varint groups as range-checked 7-bit witnesses (their number is a constant of the circuit); powers < 2^49."""
    n = len(pubkeys)
    assert n >= 1 and len(voting_powers) == n
    c128 = b.constant(128)
    leaves, powers, keys = [], [], []
    for key, power in zip(pubkeys, voting_powers):
        power = int(power)
        if not 0 < power < (1 << 49) or len(key) != 32:
            raise ValueError("validator: 32-byte key and 0 < voting power < 2^49 expected")
        kb = [g.byte(b.var(v)) for v in bytes(key)]
        groups = []
        p = power
        while True:
            groups.append(p & 0x7F)
            p >>= 7
            if not p:
                break
        gv = []
        for gval in groups:
            v = b.var(gval)
            b.range32(v)
            b.range32(b.arith(1, 0, 0, v, b.constant(1 << 25), v))                       # v < 2^7
            gv.append(v)
        pw = gv[-1]
        for v in reversed(gv[:-1]):
            pw = b.arith(1, 1, 0, pw, c128, v)                                            # Horner: power = sum g_j * 128^j
        vbytes = [b.arith(0, 1, 0x80, v, v, v) for v in gv[:-1]] + [gv[-1]]               # continuation bit on all but the last group
        leaf = [b.constant(v) for v in (0x00, 0x0a, 0x22, 0x0a, 0x20)] + kb + [b.constant(0x10)] + vbytes
        leaves.append(g.hash_bytes(leaf))
        powers.append(pw)
        keys.append(kb)
    total = powers[0]
    for pw in powers[1:]:
        total = b.add(total, pw)
    return {"root": rfc6962_root(g, leaves), "keys": keys, "powers": powers, "total": total}


def _more_than(b, g, got, total, numerator, denominator):
    """denominator * got > numerator * total for sums below 2^57:  d = denominator * got - numerator * total - 1 is shown to be a non-negative
    60-bit number (d = hi * 2^32 + lo, lo < 2^32, hi < 2^28): a negative difference is p - k > 2^63 and has no such decomposition"""
    d = b.arith(0, denominator, P - 1, got, got, got)
    d = b.arith(1, 1, 0, total, b.constant(P - numerator), d)
    lo, hi = b.bit_field(d, 0, 32), b.bit_field(d, 32, 28)
    b.range32(lo)
    b.range32(hi)
    b.range32(b.arith(1, 0, 0, hi, b.constant(1 << 4), hi))
    b.assert_equal(b.arith(1, 1, 0, hi, g.c2_32, lo), d)


def _flags(b, signed):
    out = []
    for sg in signed:
        f = b.var(1 if sg else 0)
        b.assert_bool(f)
        out.append(f)
    return out


def validator_set_statement(b, g, pubkeys, voting_powers, signed, numerator=2, denominator=3):
    """Lay down, on builder b (gadget g = Sha256Rows(b)), the non-cryptographic half of a Tendermint commit check ([RECALLED] tendermintx's
    validator-set and voting-power logic; the encodings are [SPEC] protobuf / RFC 6962, as in blobstream.py):
      * every validator's Merkle leaf 0x00 || SimpleValidator{pub_key{ed25519 = pubkey}, voting_power} is hashed in-circuit from its 32 key bytes
        and its power's varint groups (7 bits each; the number of groups is a constant of the circuit, like the number of validators);
      * the RFC 6962 tree over the leaves (split at the largest power of two below n) gives validators_hash;
      * signed_power = sum of the powers whose `signed` flag is 1, total_power = the sum of all, and
        denominator * signed_power > numerator * total_power  (a 60-bit non-negative difference, shown by range checks).
    The flags are boolean WITNESSES: that a flagged validator's Ed25519 signature verifies is NOT constrained here (the GPU witness kernel checks it
    outside the circuit; the curve arithmetic in-circuit is what upstream's STARK is for).  Powers must be below 2^49 (7 varint groups), so no sum can
    wrap.  Returns (validators_hash: 8 word variables, signed_power, total_power).
    Input order of the recorded program: per validator its 32 key bytes then its varint groups; then the flags."""
    vs = _validator_set(b, g, pubkeys, voting_powers)
    flags = _flags(b, signed)
    got = b.mul(flags[0], vs["powers"][0])
    for pw, f in zip(vs["powers"][1:], flags[1:]):
        got = b.arith(1, 1, 0, f, pw, got)
    _more_than(b, g, got, vs["total"], numerator, denominator)
    return vs["root"], got, vs["total"]


def skip_statement(b, g, trusted_header_fields, trusted, target_header_fields, target, signed, trusted_index, heights=None, max_skip=1 << 20):
    """The non-cryptographic statement of a light-client SKIP ([RECALLED] tendermintx verify_skip; blobstream.skip_witness computes the same as a
    witness): from a trusted header to a target header that need not be its successor,
      1. the trusted header's next_validators_hash field (index 8) is BytesValue(hash of the trusted set),
      2. the target header's validators_hash field (index 7) is BytesValue(hash of the target set),
      3. the flagged target validators hold more than 2/3 of the target set's power,
      4. flagged target validators that are ALSO in the trusted set (trusted_index[i] = their position there, or None; same 32 key bytes, enforced
         by copy constraints) hold more than 1/3 of the TRUSTED set's power.
    trusted / target = (pubkeys, voting_powers); header fields = 14 opaque byte strings each (the bound field's own bytes are ignored).
    NOT constrained: that the flagged validators signed the target header (Ed25519) — but the proof exposes WHO was flagged (signer digest over
    the target set's keys and flags), so a consumer checks exactly those signatures natively (blobstream.verify_signers).
    heights = (trusted block, target block), optional: both headers' height fields (field 2) are then tied to variables, with
    trusted < target <= trusted + max_skip shown by range checks; returned as a fourth element [trusted_block, target_block].
    Returns (trusted header hash, target header hash, signer digest[, blocks])."""
    T = _validator_set(b, g, *trusted)
    V = _validator_set(b, g, *target)
    flags = _flags(b, signed)
    zero = b.constant(0)
    got, overlap = zero, zero
    for i, f in enumerate(flags):
        got = b.arith(1, 1, 0, f, V["powers"][i], got)
        t = trusted_index[i]
        if t is not None:
            for x, y in zip(V["keys"][i], T["keys"][t]):
                b.assert_equal(x, y)                                   # the same validator: the same 32 key bytes
            overlap = b.arith(1, 1, 0, f, T["powers"][t], overlap)
    _more_than(b, g, got, V["total"], 2, 3)
    _more_than(b, g, overlap, T["total"], 1, 3)
    wrap = lambda root: [b.constant(0x0a), b.constant(0x20)] + [x for w in root for x in g.bytes_of_word(w)]
    bound_t, bound_v, blocks = {8: wrap(T["root"])}, {7: wrap(V["root"])}, None
    if heights is not None:
        ht_var, bound_t[2] = _height_field(b, g, heights[0])
        hv_var, bound_v[2] = _height_field(b, g, heights[1])
        gap = b.sub(hv_var, ht_var)                                                     # 1 <= gap <= max_skip: gap - 1 and max_skip - gap are 32-bit
        b.range32(b.arith(0, 1, P - 1, gap, gap, gap))
        b.range32(b.arith(0, P - 1, int(max_skip), gap, gap, gap))
        blocks = [ht_var, hv_var]
    h_trusted = header_hash_statement(b, g, trusted_header_fields, bound=bound_t)
    h_target = header_hash_statement(b, g, target_header_fields, bound=bound_v)
    out = (h_trusted, h_target, _signer_digest(b, g, V["keys"], flags))
    return out if blocks is None else out + (blocks,)


def _groups7(value):
    out, p = [], int(value)
    while True:
        out.append(p & 0x7F)
        p >>= 7
        if not p:
            return out


def skip_statement_inputs(trusted_header_fields, trusted, target_header_fields, target, signed, heights=None):
    """the input vector of a program recorded from skip_statement, in the order the statement creates its free variables: per trusted validator
    its 32 key bytes and its power's 7-bit groups, the same per target validator, the flags, (with heights) the two heights' groups, then the
    bytes of every header field the statement does not bind (trusted: all but 8 and, with heights, 2; target: all but 7 and 2).  The number of
    groups per power / height and every field length are constants of the recorded circuit: the replay refuses a vector of another length."""
    out = []
    for keys, powers in (trusted, target):
        for key, power in zip(keys, powers):
            out += list(bytes(key)) + _groups7(power)
    out += [1 if sg else 0 for sg in signed]
    if heights is not None:
        out += _groups7(heights[0]) + _groups7(heights[1])
    for fields, bound in ((trusted_header_fields, {8}), (target_header_fields, {7})):
        skip = bound | ({2} if heights is not None else set())
        for k, fb in enumerate(fields):
            if k not in skip:
                out += list(bytes(fb))
    return out


def _height_field(b, g, height):
    """a header's height as a VARIABLE and its field encoding (Int64Value: 0x08 || varint) as byte variables tied to it: the varint's 7-bit groups
    are range-checked witnesses (their number is a constant of the circuit), height = sum g_j * 128^j.  Heights below 2^49."""
    height = int(height)
    if not 0 < height < (1 << 49):
        raise ValueError("height out of range")
    groups = []
    p = height
    while True:
        groups.append(p & 0x7F)
        p >>= 7
        if not p:
            break
    gv = []
    for gval in groups:
        v = b.var(gval)
        b.range32(v)
        b.range32(b.arith(1, 0, 0, v, b.constant(1 << 25), v))
        gv.append(v)
    hv = gv[-1]
    for v in reversed(gv[:-1]):
        hv = b.arith(1, 1, 0, hv, b.constant(128), v)
    return hv, [b.constant(0x08)] + [b.arith(0, 1, 0x80, v, v, v) for v in gv[:-1]] + [gv[-1]]


def signer_leaf(b, key_words, flag):
    """the 4-word leaf of the signer digest tree for one validator slot: Poseidon hash_no_pad over the 8 big-endian words of its key and its flag"""
    return b.hash_no_pad(list(key_words) + [flag])


def signer_tree(b, leaves):
    """binary Poseidon two_to_one tree over a power-of-two list of 4-word leaves"""
    level = list(leaves)
    while len(level) > 1:
        level = [b.two_to_one(level[k], level[k + 1]) for k in range(0, len(level), 2)]
    return level[0]


def _signer_digest(b, g, keys, flags):
    """The SIGNER DIGEST: a binary Poseidon tree over one leaf per validator slot — hash_no_pad(the 8 big-endian words of the key, the signed
    flag) — padded to a power of two with the leaf of an all-zero key and flag 0.  Exposed as public input it BINDS the proof to who was flagged.
    A TREE (round 3; it was one long sponge) so that the same digest can be assembled by a MapReduce over the slots: signature_mr.py proves, per
    slot, "flag = 1 => the slot's key signed the vote" and folds the slots' leaves into exactly this root, which the CombinedSkip outer circuit
    equates with the one computed here — the Ed25519 half of the statement, in-circuit.  (blobstream.verify_signers, the native check of the
    flagged signatures against this digest, remains for proofs without that half.)"""
    leaves = [signer_leaf(b, [g.word_from_bytes(kb[k:k + 4]) for k in range(0, 32, 4)], f) for kb, f in zip(keys, flags)]
    n = 1 << max(0, (len(leaves) - 1).bit_length())
    if n > len(leaves):
        zero = b.constant(0)
        pad = signer_leaf(b, [zero] * 8, zero)
        leaves += [pad] * (n - len(leaves))
    return signer_tree(b, leaves)


def signer_digest_host(poseidon_consts, pubkeys, signed, pad_to=None):
    """the same digest on the host (what a consumer recomputes from the keys and flags it was given); pad_to: number of slots (a power of two
    >= len(pubkeys), default the next power of two)"""
    from . import poseidon_permute_host
    import numpy as np

    def hash_no_pad(elems):
        state = np.zeros(12, dtype=np.uint64)
        for off in range(0, len(elems), 8):
            chunk = elems[off:off + 8]
            state[:len(chunk)] = chunk
            state = poseidon_permute_host(poseidon_consts, state)[0]
        return [int(v) for v in state[:4]]
    level = [hash_no_pad(list(struct.unpack(">8I", bytes(key))) + [1 if sg else 0]) for key, sg in zip(pubkeys, signed)]
    n = pad_to or (1 << max(0, (len(level) - 1).bit_length()))
    level += [hash_no_pad([0] * 9)] * (n - len(level))
    while len(level) > 1:
        st = np.array([level[2 * k] + level[2 * k + 1] + [0, 0, 0, 0] for k in range(len(level) // 2)], dtype=np.uint64)
        level = [[int(v) for v in row[:4]] for row in poseidon_permute_host(poseidon_consts, st)]
    return level[0]


def step_statement(b, g, trusted_header_fields, target_header_fields, validators, signed, trusted_height=None):
    """The non-cryptographic statement of a light-client STEP ([RECALLED] tendermintx verify_step): the target header is the trusted header's
    successor —
      1. the trusted header's next_validators_hash (field 8) and the target header's validators_hash (field 7) are BytesValue(hash of ONE set),
      2. the target header's last_block_id (field 4: 0x0a 0x20 || block hash || part-set header, 72 bytes) carries the TRUSTED header's hash —
         the bytes of the hash computed in this circuit, so the chain link is constrained, not asserted,
      3. the flagged validators hold more than 2/3 of the set's power.
    validators = (pubkeys, voting_powers).  The last 38 bytes of the target's field 4 (0x12 0x24 || part-set header) stay opaque witnesses.
    NOT constrained: the Ed25519 signatures — but the proof exposes WHO was flagged (signer digest, 4 words), so a consumer checks exactly those
    signatures natively.  trusted_height (optional): both headers' height fields (field 2) are then tied to variables — the trusted block number and
    its successor, trusted + 1 — returned as a fourth element [trusted_block, target_block] (upstream's public inputs name them).
    Returns (trusted header hash, target header hash, signer digest[, blocks])."""
    V = _validator_set(b, g, *validators)
    flags = _flags(b, signed)
    got = b.constant(0)
    for f, pw in zip(flags, V["powers"]):
        got = b.arith(1, 1, 0, f, pw, got)
    _more_than(b, g, got, V["total"], 2, 3)
    wrap = lambda ws: [b.constant(0x0a), b.constant(0x20)] + [x for w in ws for x in g.bytes_of_word(w)]
    bound_t, bound_v, blocks = {8: wrap(V["root"])}, {}, None
    if trusted_height is not None:
        ht_var, bound_t[2] = _height_field(b, g, trusted_height)
        hv_var, bound_v[2] = _height_field(b, g, int(trusted_height) + 1)
        b.assert_equal(b.arith(0, 1, 1, ht_var, ht_var, ht_var), hv_var)              # the successor: target = trusted + 1
        blocks = [ht_var, hv_var]
    h_trusted = header_hash_statement(b, g, trusted_header_fields, bound=bound_t)
    tail = bytes(target_header_fields[4])[34:]
    if len(bytes(target_header_fields[4])) < 34:
        raise ValueError("the target header's last_block_id field is shorter than 0x0a 0x20 || hash")
    block_id = wrap(h_trusted) + [g.byte(b.var(v)) for v in tail]
    bound_v.update({7: wrap(V["root"]), 4: block_id})
    h_target = header_hash_statement(b, g, target_header_fields, bound=bound_v)
    out = (h_trusted, h_target, _signer_digest(b, g, V["keys"], flags))
    return out if blocks is None else out + (blocks,)


def step_circuit(prover, trusted_header_fields, target_header_fields, validators, signed, trusted_height=None):
    """the circuit of step_statement: public inputs = the trusted header hash, the target header hash (8 words each), the signer digest (4) and,
    with trusted_height, the two block numbers (trusted, trusted + 1)"""
    from . import SHA_GATE_WIRES
    b = CircuitBuilder(prover, n_wires=SHA_GATE_WIRES)
    g = Sha256Rows(b)
    ht, hv, sd, *blocks = step_statement(b, g, trusted_header_fields, target_header_fields, validators, signed, trusted_height)
    for w in ht + hv + sd + (blocks[0] if blocks else []):
        b.public_input(w)
    to_bytes = lambda ws: b"".join(struct.pack(">I", b.value(w)) for w in ws)
    hb_t, hb_v = to_bytes(ht), to_bytes(hv)
    ck, dw, public = b.build()
    return ck, dw, public, hb_t, hb_v


def skip_circuit(prover, trusted_header_fields, trusted, target_header_fields, target, signed, trusted_index, heights=None, max_skip=1 << 20):
    """the circuit of skip_statement: public inputs = the trusted header hash, the target header hash (8 words each), the signer digest (4 words) and,
    with heights, the two block numbers.  Returns (circuit, device wires, public values, trusted header hash bytes, target header hash bytes);
    ValueError when a threshold (or the block gap) is not met."""
    from . import SHA_GATE_WIRES
    b = CircuitBuilder(prover, n_wires=SHA_GATE_WIRES)
    g = Sha256Rows(b)
    ht, hv, sd, *blocks = skip_statement(b, g, trusted_header_fields, trusted, target_header_fields, target, signed, trusted_index, heights, max_skip)
    for w in ht + hv + sd + (blocks[0] if blocks else []):
        b.public_input(w)
    to_bytes = lambda ws: b"".join(struct.pack(">I", b.value(w)) for w in ws)
    hb_t, hb_v = to_bytes(ht), to_bytes(hv)
    ck, dw, public = b.build()
    return ck, dw, public, hb_t, hb_v


def rfc6962_root(g, leaf_digests):
    """RFC 6962 / Tendermint simple Merkle root over already-hashed leaves (8 word variables each): split at the largest power of two below n"""
    if len(leaf_digests) == 1:
        return leaf_digests[0]
    k = 1 << ((len(leaf_digests) - 1).bit_length() - 1)
    return g.hash_prefixed_64(0x01, rfc6962_root(g, leaf_digests[:k]) + rfc6962_root(g, leaf_digests[k:]))


def header_hash_statement(b, g, fields, bound=None):
    """a Tendermint-style header hash in-circuit: the RFC 6962 root over the header's encoded fields ([RECALLED] cometbft Header.Hash(): 14 leaves,
    each the protobuf encoding of one field; here the encodings are opaque byte strings, which is all the hash depends on).  fields: list of bytes
    (witnesses, range-checked); bound: {field index: [byte variables]} replaces a field's bytes by variables computed elsewhere in the circuit —
    e.g. 0x0a 0x20 || validators_hash, which is what ties a header to a validator set.  Returns the 8 word variables of the header hash."""
    bound = bound or {}
    leaves = []
    for k, fb in enumerate(fields):
        bv = bound[k] if k in bound else [g.byte(b.var(v)) for v in bytes(fb)]
        leaves.append(g.hash_bytes([b.constant(0x00)] + list(bv)))
    return rfc6962_root(g, leaves)


def commit_check_circuit(prover, header_fields, validators_field, pubkeys, voting_powers, signed):
    """Skip/Step-shaped statement WITHOUT the signature half: public inputs = the 8 words of a header hash, then signed_power and total_power;
    constraints = that header's field `validators_field` is BytesValue(validators_hash) of a validator set (hashed in-circuit from keys and powers)
    whose flagged members hold more than 2/3 of the power.  That each flagged validator signed the header is NOT constrained (Ed25519 stays a GPU
    witness check).  Returns (circuit, device wires, public values, header hash bytes, validators_hash bytes)."""
    from . import SHA_GATE_WIRES
    b = CircuitBuilder(prover, n_wires=SHA_GATE_WIRES)
    g = Sha256Rows(b)
    vroot, got, total = validator_set_statement(b, g, pubkeys, voting_powers, signed)
    vbytes = [x for w in vroot for x in g.bytes_of_word(w)]
    field = [b.constant(0x0a), b.constant(0x20)] + vbytes
    hroot = header_hash_statement(b, g, header_fields, bound={validators_field: field})
    for w in hroot:
        b.public_input(w)
    b.public_input(got)
    b.public_input(total)
    hh = b"".join(struct.pack(">I", b.value(w)) for w in hroot)
    vh = b"".join(struct.pack(">I", b.value(w)) for w in vroot)
    ck, dw, public = b.build()
    return ck, dw, public, hh, vh


def data_commitment_chain_statement(b, g, start_header_fields, headers, first_height):
    """The header-chain form of the data commitment ([RECALLED] blobstreamx DataCommitmentCircuit: prove_data_commitment walks the headers between
    two trusted hashes): a start header and n following headers, each given as its 14 field encodings, with
      * header k's last_block_id (field 4: 0x0a 0x20 || hash || 38 opaque bytes) carrying the hash of header k-1 AS COMPUTED IN THIS CIRCUIT — the
        chain from the start header to the last one is constrained link by link;
      * header k's data_hash (field 6: BytesValue, 0x0a 0x20 || 32 bytes) feeding leaf k of the data commitment: the RFC 6962 root over
        abi.encode(first_height + k, data_hash_k) for the n headers AFTER the start header (n a power of two);
      * every header's height field (field 2: Int64Value, 0x08 || varint) is a CONSTANT of the circuit — first_height - 1 for the start header,
        first_height + k after it — so the heights in the commitment's tuples are the heights the hashed headers carry.
    Returns (start header hash, last header hash, data commitment root): 8 word variables each."""
    from .blobstream import encode_varint
    n = len(headers)
    assert n >= 1 and n & (n - 1) == 0 and first_height >= 1
    wrap = lambda ws: [b.constant(0x0a), b.constant(0x20)] + [x for w in ws for x in g.bytes_of_word(w)]
    height_field = lambda h: [b.constant(v) for v in b"\x08" + encode_varint(int(h))]
    prev = header_hash_statement(b, g, start_header_fields, bound={2: height_field(first_height - 1)})
    h_start = prev
    leaves = []
    for k, fields in enumerate(headers):
        if len(bytes(fields[4])) < 34 or len(bytes(fields[6])) != 34:
            raise ValueError("header fields 4 (last_block_id) / 6 (data_hash) do not have the expected encodings")
        data_hash = [g.byte(b.var(v)) for v in bytes(fields[6])[2:]]
        block_id = wrap(prev) + [g.byte(b.var(v)) for v in bytes(fields[4])[34:]]
        prev = header_hash_statement(b, g, fields, bound={2: height_field(first_height + k), 4: block_id,
                                                          6: [b.constant(0x0a), b.constant(0x20)] + data_hash})
        height_words = [b.constant(v) for v in struct.unpack(">8I", int(first_height + k).to_bytes(32, "big"))]
        root_words = [g.word_from_bytes(data_hash[j:j + 4]) for j in range(0, 32, 4)]
        leaves.append(g.hash_prefixed_64(0x00, height_words + root_words))
    while len(leaves) > 1:
        leaves = [g.hash_prefixed_64(0x01, leaves[j] + leaves[j + 1]) for j in range(0, len(leaves), 2)]
    return h_start, prev, leaves[0]


def data_commitment_chain_circuit(prover, start_header_fields, headers, first_height):
    """the circuit of data_commitment_chain_statement: public inputs = start header hash, end header hash, data commitment (8 words each).
    Returns (circuit, device wires, public values, start hash bytes, end hash bytes, commitment bytes)."""
    from . import SHA_GATE_WIRES
    b = CircuitBuilder(prover, n_wires=SHA_GATE_WIRES)
    g = Sha256Rows(b)
    hs, he, root = data_commitment_chain_statement(b, g, start_header_fields, headers, first_height)
    for w in hs + he + root:
        b.public_input(w)
    to_bytes = lambda ws: b"".join(struct.pack(">I", b.value(w)) for w in ws)
    out = (to_bytes(hs), to_bytes(he), to_bytes(root))
    ck, dw, public = b.build()
    return (ck, dw, public) + out


def combined_skip_circuit(prover, trusted_header_fields, trusted, chain_headers, target, signed, trusted_index, trusted_height, max_skip=1 << 20):
    """CombinedSkip's shape in ONE circuit ([RECALLED] blobstreamx CombinedSkipCircuit = skip + the data commitment of the skipped range), minus the
    Ed25519 half: chain_headers = the headers AFTER the trusted one up to and including the target (a power-of-two count), each as its 14 field
    encodings.  Constraints: the skip statement between the trusted header and the last chain header (validator sets, 2/3 and 1/3 power rules, block
    numbers) AND the header chain with its data commitment (data_commitment_chain_statement) — the two halves meet in the trusted and target header
    hashes, which each half computes from the fields and which are copy-constrained equal.  Public inputs: trusted header hash, target header
    hash (8 words each), signer digest (4), trusted block, target block, data commitment (8 words).
    Returns (circuit, device wires, public values, trusted hash bytes, target hash bytes, commitment bytes)."""
    from . import SHA_GATE_WIRES
    b = CircuitBuilder(prover, n_wires=SHA_GATE_WIRES)
    g = Sha256Rows(b)
    n = len(chain_headers)
    ht, hv, sd, blocks = skip_statement(b, g, trusted_header_fields, trusted, chain_headers[-1], target, signed, trusted_index,
                                        heights=(trusted_height, trusted_height + n), max_skip=max_skip)
    hs, he, root = data_commitment_chain_statement(b, g, trusted_header_fields, chain_headers, trusted_height + 1)
    for x, y in zip(ht + hv, hs + he):
        b.assert_equal(x, y)                       # both halves are about the SAME two headers
    for w in ht + hv + sd + blocks + root:
        b.public_input(w)
    to_bytes = lambda ws: b"".join(struct.pack(">I", b.value(w)) for w in ws)
    out = (to_bytes(ht), to_bytes(hv), to_bytes(root))
    ck, dw, public = b.build()
    return (ck, dw, public) + out


def validator_set_circuit(prover, pubkeys, voting_powers, signed, numerator=2, denominator=3):
    """the circuit of validator_set_statement with public inputs = the 8 words of validators_hash, then signed_power and total_power.
    Returns (circuit, device wires, public values, validators_hash bytes).  ValueError when the flagged validators do not hold more than
    numerator/denominator of the power (the witness cannot satisfy the circuit)."""
    from . import SHA_GATE_WIRES
    b = CircuitBuilder(prover, n_wires=SHA_GATE_WIRES)
    g = Sha256Rows(b)
    root, got, total = validator_set_statement(b, g, pubkeys, voting_powers, signed, numerator, denominator)
    for w in root:
        b.public_input(w)
    b.public_input(got)
    b.public_input(total)
    digest = b"".join(struct.pack(">I", b.value(w)) for w in root)
    ck, dw, public = b.build()
    return ck, dw, public, digest


def data_commitment_rows_circuit(prover, heights, data_roots):
    """DataCommitment over a power-of-two block range on the SHA row gates: the same statement as data_commitment_circuit (public inputs:
    per block the 16 big-endian words of abi.encode(height, dataRoot), then the 8 words of the commitment root) in ~370 rows per hash
    instead of ~6.6k.  Returns (circuit, device wires, public values, root bytes)."""
    from . import SHA_GATE_WIRES
    n = len(heights)
    assert n >= 1 and n & (n - 1) == 0 and len(data_roots) == n
    b = CircuitBuilder(prover, n_wires=SHA_GATE_WIRES)
    g = Sha256Rows(b)
    level = []
    for hgt, root in zip(heights, data_roots):
        tup = int(hgt).to_bytes(32, "big") + bytes(root)
        words = [g.public_word(struct.unpack(">I", tup[4 * k: 4 * k + 4])[0]) for k in range(16)]
        level.append(g.hash_prefixed_64(0x00, words))
    while len(level) > 1:
        level = [g.hash_prefixed_64(0x01, level[k] + level[k + 1]) for k in range(0, len(level), 2)]
    for w in level[0]:
        b.public_input(w)
    root = b"".join(struct.pack(">I", b.value(w)) for w in level[0])
    ck, dw, public = b.build()
    return ck, dw, public, root


def data_commitment_circuit(prover, heights, data_roots):
    """DataCommitment over a power-of-two block range, constrained in-circuit.  Returns (circuit, device wires, public values,
    root bytes).  Public values: per block 16 big-endian words of abi.encode(height, dataRoot) (32-byte big-endian height, 32-byte
    root), then the 8 words of the commitment root."""
    n = len(heights)
    assert n >= 1 and n & (n - 1) == 0 and len(data_roots) == n
    b = CircuitBuilder(prover)
    g = Sha256Gadget(b)
    byte_bits = lambda v: [g.one if (v >> (7 - i)) & 1 else g.zero for i in range(8)]
    level = []
    for hgt, root in zip(heights, data_roots):
        tup = int(hgt).to_bytes(32, "big") + bytes(root)
        words = [g.public_word(struct.unpack(">I", tup[4 * k: 4 * k + 4])[0]) for k in range(16)]
        msg = byte_bits(0x00) + [bit for w in words for bit in g.word_bits_be(w)]
        level.append(g.hash_bits(msg))
    while len(level) > 1:
        nxt = []
        for k in range(0, len(level), 2):
            msg = byte_bits(0x01) + [bit for w in level[k] for bit in g.word_bits_be(w)] + [bit for w in level[k + 1] for bit in g.word_bits_be(w)]
            nxt.append(g.hash_bits(msg))
        level = nxt
    for w in level[0]:
        b.public_input(w[1])
    root = b"".join(struct.pack(">I", b.value(w[1])) for w in level[0])
    ck, dw, public = b.build()
    return ck, dw, public, root
