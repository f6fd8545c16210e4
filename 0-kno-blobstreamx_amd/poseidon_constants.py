"""Poseidon-Goldilocks parameters for width 12, x^7, 8 full + 22 partial rounds.

The library ships NO built-in constants: they are injected through
``glp_set_poseidon_constants``.  plonky2's own table (360 round constants) is not in the
reference mount and is not reproduced from memory, so ``default_constants``
returns a documented, deterministic stand-in:

* round constants: the Grain-LFSR procedure of the Poseidon paper (parameters field=1,
  sbox=0, n=64, t=12, R_F=8, R_P=22; 80-bit state, taps 62/51/38/23/13/0, 160 warm-up
  steps, pair-wise bit filtering, rejection sampling below p).  NOT verified to equal
  plonky2's table (a quick check of the first constant did not match what is recalled
  of it), hence every digest produced with it is "self-consistent, not plonky2-compatible".
* MDS: circulant first row (17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20) plus diagonal
  (8, 0, ..., 0) — recalled, unverified; any small-integer MDS takes the same fast path.

Swap in the real table by passing it to ``Prover.set_poseidon_constants``.
"""
P = 2**64 - 2**32 + 1
MDS_CIRC = [17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20]
MDS_DIAG = [8] + [0] * 11


def grain_round_constants(n_bits=64, t=12, r_f=8, r_p=22, field=1, sbox=0):
    def bits_of(v, w):
        return [(v >> (w - 1 - i)) & 1 for i in range(w)]

    state = bits_of(field, 2) + bits_of(sbox, 4) + bits_of(n_bits, 12) + bits_of(t, 12) + bits_of(r_f, 10) + bits_of(r_p, 10) + [1] * 30

    def step():
        nonlocal state
        nb = state[62] ^ state[51] ^ state[38] ^ state[23] ^ state[13] ^ state[0]
        state = state[1:] + [nb]
        return nb

    for _ in range(160):
        step()

    def next_bit():
        while True:
            b1, b2 = step(), step()
            if b1:
                return b2

    out = []
    while len(out) < (r_f + r_p) * t:
        v = 0
        for _ in range(n_bits):
            v = (v << 1) | next_bit()
        if v < P:
            out.append(v)
    return out


_cache = None


def default_constants():
    """(rc[360], mds_circ[12], mds_diag[12]) — deterministic stand-in, see module docstring"""
    global _cache
    if _cache is None:
        _cache = (grain_round_constants(), list(MDS_CIRC), list(MDS_DIAG))
    return _cache
