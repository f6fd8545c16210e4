"""Circuits over the build-defined gate set (include/glprover.h: arithmetic/constant gates, public inputs, Poseidon rows, copy
constraints) and the first in-circuit pieces of the MapReduce Reduce step (upstream names
recalled, unverified — reference file:line NONE, the mount is empty: plonky2x ``CircuitBuilder``, plonky2
``CircuitBuilder::{mul_add, constant, connect, hash_n_to_hash_no_pad, verify_merkle_proof_to_cap}``).

``CircuitBuilder`` is the host-side ``define`` surface: variables, gates, copy constraints, public inputs; ``build`` lays the
gates out in rows (a row's constants are shared by its 20 gate slots, so rows are typed by their constants), derives sigma and
the witness matrix, and returns a ``PlonkCircuit`` ready to prove on the GPU.  Poseidon rows carry only their 12 inputs and 12
outputs as variables; their 106 S-box-input wires are filled on the device (``glp_poseidon_gate_fill_rows``).

What lives here: the builder and its row kinds (arithmetic gates, Poseidon rows with their swap input, SHA-256 rows, extension
rows), the recorded circuit (``WitnessProgram``: layout + straight-line witness program, save / load / raw export), the aggregation of leaf-proof
digests into a Poseidon Merkle root, in-circuit Merkle paths.  The in-circuit VERIFIER of leaf proofs is verifier_circuit.py; the SHA-256 gadgets and
the statements on them are gadgets.py.
"""
from array import array

import numpy as np

from . import (CIRCUIT_POSEIDON_GATE, P, PLONK_NCONST, PLONK_NCONST_SHA, POS_GATE_WIRES, SHA_GATE_WIRES, SHA_ROW_A, SHA_ROW_ADD, SHA_ROW_E,  # noqa: F401
               SHA_ROW_W, DeviceBuffer, PlonkCircuit)


_OP_WORDS = {0: 8, 1: 3, 2: 4, 3: 3, 4: 5, 5: 2, 6: 25, 7: 10, 8: 6, 9: 6, 10: 4, 11: 5, 12: 26, 13: 9, 14: 24}   # words per op (csrc/verify.hip)


class _Words(array):
    """the flat op words of a recording as a C array of u64 (`+=` a tuple of ints appends it): a signature leaf records 10 M words, and turning a
    Python list of that length into numpy was a second of every recording; this one is viewed by numpy in place"""
    __slots__ = ()

    def __iadd__(self, words):
        self.extend(words)
        return self


class CircuitBuilder:
    """variables are integer handles; every gate is added with its witness value computed on the spot (big-int arithmetic)"""

    def __init__(self, prover, n_wires=136, n_routed=80, ext_gate=False):
        assert n_routed % 8 == 0 and n_wires % 8 == 0 and 24 <= n_routed <= n_wires and n_wires >= POS_GATE_WIRES
        self.prover, self.W, self.R = prover, n_wires, n_routed
        self.G = n_routed // 4
        self.values = []                 # variable -> value
        self.parent = []                 # union-find over variables (copy constraints)
        self.arith_rows = {}             # (c0, c1, c2) -> list of rows, a row = list of (x, y, z, w) variable tuples
        self.pos_rows = []               # (in vars[12], out vars[12])
        self.sha_rows = []               # (kind, [12 variables or None], K): SHA-256 rows (csrc/plonk_gates.h); needs n_wires >= 144
        self.ext_gate = bool(ext_gate)   # extension-arithmetic rows allowed (one more constant column: a property of the circuit)
        self.ext_rows = [[]]             # rows of (x0, x1, y0, y1, z0, z1, w0, w1) tuples, n_routed / 8 per row
        self._add_open = None            # the ADD row still taking additions (4 per row)
        self.public = []
        self._consts = {}
        # the straight-line program that recomputes every variable from the free inputs (WitnessProgram / glp_witness_eval)
        self.prog = _Words("Q")         # flat op words (encoding: csrc/verify.hip)
        self._prefix = _Words("Q")      # ops of the constants: they run first, every segment may read them
        self.seg_bounds = []             # offsets into prog: [start_0, end_0 = start_1, ..., end_last] of the independent segments
        self.input_tags = []             # per free input: caller's tag (e.g. (proof number, word position)) or None
        self.eq_pairs = []               # copy constraints between different variables
        self.auto_tag_list = None        # when set: untagged free inputs are tagged (auto_tag_list, running position) — a statement laid down next to
        self._auto_pos = 0               # in-circuit verifiers reads its witness from ONE more word list after the proofs (combined_skip_mr.py)
        self.word_checks = []            # ("const", tag, value) / ("var", tag, variable) / ("bits", tag, [bit variables]): facts about input
        #                                  words that are checked outside the circuit when a recorded program is replayed

    # ---- variables and copy constraints ---------------------------------------------------------------------------------------
    def _new(self, value):
        self.values.append(int(value) % P)
        self.parent.append(len(self.parent))
        return len(self.values) - 1

    def var(self, value, tag=None):
        """a free witness value (an INPUT of the recorded program); tag says where a replay finds it"""
        v = self._new(value)
        if tag is None and self.auto_tag_list is not None:
            tag = (self.auto_tag_list, self._auto_pos)
            self._auto_pos += 1
        self.prog += (1, v, len(self.input_tags))
        self.input_tags.append(tag)
        return v

    def bit(self, x, k):
        """bit k of the canonical value of x (a computed witness: the caller still has to constrain it)"""
        v = self._new((self.values[x] >> k) & 1)
        self.prog += (2, v, x, k)
        return v

    def inverse(self, x):
        v = self._new(pow(self.values[x], P - 2, P))
        self.prog += (3, v, x)
        return v

    def ext_inverse(self, x0, x1):
        a, bb = self.values[x0], self.values[x1]
        n = pow((a * a - 7 * bb * bb) % P, P - 2, P)
        w0, w1 = self._new(a * n), self._new((-bb) * n)
        self.prog += (4, w0, w1, x0, x1)
        return w0, w1

    def begin_segment(self):
        """what is recorded until end_segment() depends only on constants and on itself (e.g. the verifier sub-circuit of ONE proof): the
        witness evaluator may run segments on different host threads (glp_witness_eval_mt checks the claim while it runs).  Segments are
        contiguous: nothing may be recorded between one segment's end and the next one's start."""
        if self.seg_bounds and self.seg_bounds[-1] != len(self.prog):
            raise ValueError("segments must follow each other directly")
        if not self.seg_bounds:
            self.seg_bounds.append(len(self.prog))
        self._open = True

    def end_segment(self):
        assert getattr(self, "_open", False)
        self._open = False
        self.seg_bounds.append(len(self.prog))

    def _find(self, a):
        while self.parent[a] != a:
            self.parent[a] = self.parent[self.parent[a]]
            a = self.parent[a]
        return a

    def assert_equal(self, a, b):
        """copy constraint: a and b are the same wire value (enforced by the permutation argument)"""
        if self.values[a] != self.values[b]:
            raise ValueError("assert_equal on different values: the witness does not satisfy the circuit")
        ra, rb = self._find(a), self._find(b)
        if ra != rb:
            self.parent[ra] = rb
            self.eq_pairs += (a, b)

    def value(self, v):
        return self.values[v]

    # ---- gates ----------------------------------------------------------------------------------------------------------------
    def arith(self, c0, c1, c2, x, y, z):
        """w = c0*x*y + c1*z + c2 (one slot of a row whose constants are (c0, c1, c2))"""
        k0, k1, k2 = int(c0) % P, int(c1) % P, int(c2) % P
        vals = self.values
        w = len(vals)                                            # (_new, inlined: this is the builder's hottest function — a million calls per signature)
        vals.append((k0 * vals[x] * vals[y] + k1 * vals[z] + k2) % P)
        self.parent.append(w)
        self.prog.extend((0, w, x, y, z, k0, k1, k2))
        key = (k0, k1, k2)
        rows = self.arith_rows.get(key)
        if rows is None:
            rows = self.arith_rows[key] = [[]]
        last = rows[-1]
        if len(last) == self.G:
            last = []
            rows.append(last)
        last.append((x, y, z, w))
        return w

    def constant(self, k):
        k = int(k) % P
        if k not in self._consts:
            main, self.prog = self.prog, self._prefix          # recorded into the prefix, wherever the constant is first used
            d = self._new(0)
            self.prog += (5, d)
            self._consts[k] = self.arith(0, 0, k, d, d, d)
            self.prog = main
        return self._consts[k]

    def mul(self, x, y):
        return self.arith(1, 0, 0, x, y, x)

    def add(self, x, y):
        return self.arith(1, 1, 0, x, self.constant(1), y)

    def sub(self, x, y):
        return self.arith(1, P - 1, 0, x, self.constant(1), y)

    def assert_bool(self, b):
        self.assert_equal(self.arith(1, P - 1, 0, b, b, b), self.constant(0))        # b*b - b = 0

    def select(self, b, t, f):
        """b ? t : f for a boolean b:  f + b*(t - f)"""
        return self.arith(1, 1, 0, b, self.sub(t, f), f)

    def public_input(self, v):
        self.public.append(v)

    def poseidon(self, ins, swap=None):
        """one permutation row; ins: 12 variables -> 12 output variables (values from the library's own permutation).  swap: a variable holding
        0 or 1 (the row constrains that) — 1 exchanges ins[0..4) and ins[4..8) before the permutation: the Merkle-path step without select gates"""
        assert len(ins) == 12
        permute = getattr(self.prover, "poseidon_permute_host", None) or self.prover.poseidon_permute      # host arithmetic: no device round trip
        vals = [self.values[v] for v in ins]
        if swap is not None:
            sv = self.values[swap]
            if sv > 1:
                raise ValueError("a Poseidon row's swap input is not a bit: the witness does not satisfy the circuit")
            if sv:
                vals = vals[4:8] + vals[:4] + vals[8:]
        out = permute(np.array([vals], dtype=np.uint64))[0]
        outs = [self._new(int(v)) for v in out]
        if swap is None:
            self.prog += (6, *outs, *ins)
            swap = self.constant(0)             # the row's swap cell is COPY-CONSTRAINED to zero: left free, a prover could set it (with matching
            #                                     deltas) and hash the blocks in the other order
        else:
            self.prog += (12, *outs, *ins, swap)
        self.pos_rows.append((list(ins), outs, swap))
        return outs

    # ---- SHA-256 rows: words (32-bit values) in, words out; the row's bit wires are filled on the device ------------------------------
    _M32 = 0xFFFFFFFF

    def _word(self, v):
        val = self.values[v]
        if val >> 32:
            raise ValueError("a SHA row input is not a 32-bit word: the witness does not satisfy the circuit")
        return val

    def sha_e(self, e, f, g, h, d, w, k_const):
        """e-half of round t: returns (T1, e_new) with T1 = h + Sigma1(e) + Ch(e,f,g) + K_t + w (unreduced) and e_new = (d + T1) mod 2^32.
        The row range-checks e, f, g and e_new; h, d, w must be range-checked words where they come from."""
        assert self.W >= SHA_GATE_WIRES
        ev, fv, gv, hv, dv, wv = (self._word(x) for x in (e, f, g, h, d, w))
        rot = lambda x, r: ((x >> r) | (x << (32 - r))) & self._M32
        t1v = hv + (rot(ev, 6) ^ rot(ev, 11) ^ rot(ev, 25)) + ((ev & fv) ^ (~ev & gv & self._M32)) + int(k_const) + wv
        t1, en = self._new(t1v), self._new((dv + t1v) & self._M32)
        self.prog += (7, t1, en, e, f, g, h, d, w, int(k_const))
        self.sha_rows.append((SHA_ROW_E, [e, f, g, h, d, w, t1, en], int(k_const)))
        return t1, en

    def sha_a(self, a, b, c, t1):
        """a-half of the round: a_new = (T1 + Sigma0(a) + Maj(a,b,c)) mod 2^32 (range-checks a, b, c, a_new; T1 comes from sha_e)"""
        av, bv, cv = (self._word(x) for x in (a, b, c))
        rot = lambda x, r: ((x >> r) | (x << (32 - r))) & self._M32
        an = self._new((self.values[t1] + (rot(av, 2) ^ rot(av, 13) ^ rot(av, 22)) + ((av & bv) ^ (av & cv) ^ (bv & cv))) & self._M32)
        self.prog += (8, an, a, b, c, t1)
        self.sha_rows.append((SHA_ROW_A, [a, b, c, t1, an], 0))
        return an

    def sha_w(self, w16, w15, w7, w2):
        """message schedule: (w16 + sigma0(w15) + w7 + sigma1(w2)) mod 2^32 (range-checks w15, w2 and the result)"""
        v16, v15, v7, v2 = (self._word(x) for x in (w16, w15, w7, w2))
        rot = lambda x, r: ((x >> r) | (x << (32 - r))) & self._M32
        wn = self._new((v16 + (rot(v15, 7) ^ rot(v15, 18) ^ (v15 >> 3)) + v7 + (rot(v2, 17) ^ rot(v2, 19) ^ (v2 >> 10))) & self._M32)
        self.prog += (9, wn, w16, w15, w7, w2)
        self.sha_rows.append((SHA_ROW_W, [w16, w15, w7, w2, wn], 0))
        return wn

    def add32(self, x, y):
        """(x + y) mod 2^32 for words x, y; the result is range-checked (four additions share one ADD row)"""
        assert self.W >= SHA_GATE_WIRES
        s = self._new((self._word(x) + self._word(y)) & self._M32)
        self.prog += (10, s, x, y)
        if self._add_open is None or len(self._add_open) == 12:
            self._add_open = []
            self.sha_rows.append((SHA_ROW_ADD, self._add_open, 0))
        self._add_open += [x, y, s]
        return s

    def range32(self, x):
        """x < 2^32: x + 0 through an ADD row, whose result (bits and all) is copy-constrained back to x"""
        self.assert_equal(self.add32(x, self.constant(0)), x)
        return x

    def bit_field(self, x, shift, bits):
        """(x >> shift) mod 2^bits of the canonical value (a computed witness: the caller constrains it)"""
        v = self._new((self.values[x] >> shift) & ((1 << bits) - 1))
        self.prog += (11, v, x, shift, bits)
        return v

    def nnf_mul_hints(self, a, b):
        """the witness values of a product in the NON-NATIVE field F_q, q = 2^255 - 19, on eleven 24-bit limbs (csrc/nnf25519.h, evaluator op 14):
        for limb variables a[11], b[11] (integer values A, B = sum limb * 2^(24 i); limbs may be loose, below 2^28) returns the 44 new variables
        (r[11], k[12], c[21]) with A * B = k * q + r, r canonical, and c the carries of the column identity
            col_t = sum_{i+j=t} a_i b_j + 19 k_t - 2^15 k_(t-10) - r_t,    col_t + c_(t-1) = c_t * 2^24,
        negative carries stored mod p.  Computed witnesses only: ed25519_circuit.NNF lays down the constraints (gates + range checks)."""
        assert len(a) == 11 and len(b) == 11
        av, bv = [self.values[v] for v in a], [self.values[v] for v in b]
        if any(v >> 28 for v in av + bv):
            raise ValueError("a non-native product operand limb is out of range: the witness does not satisfy the circuit")
        A, B = sum(v << (24 * i) for i, v in enumerate(av)), sum(v << (24 * i) for i, v in enumerate(bv))
        k, r = divmod(A * B, (1 << 255) - 19)
        rl = [(r >> (24 * i)) & 0xFFFFFF for i in range(11)]
        kl = [(k >> (24 * i)) & 0xFFFFFF for i in range(12)]
        assert k >> (24 * 12) == 0
        cs, carry = [], 0
        for t in range(22):
            col = sum(av[i] * bv[t - i] for i in range(11) if 0 <= t - i < 11) + carry
            col += (19 * kl[t] if t < 12 else 0) - ((kl[t - 10] << 15) if 0 <= t - 10 < 12 else 0) - (rl[t] if t < 11 else 0)
            assert col % (1 << 24) == 0
            carry = col >> 24
            if t < 21:
                cs.append(carry)
        assert carry == 0
        first = len(self.values)
        out = [self._new(v) for v in rl + kl + cs]
        self.prog += (14, first, *a, *b)
        return out[:11], out[11:23], out[23:]

    def ext_mul_add(self, x, y, z):
        """w = x * y + z in F_p[X]/(X^2 - 7) for pairs of variables: ONE chunk of an extension-arithmetic row (8 wires) instead of six
        arithmetic gates.  Needs ext_gate=True."""
        assert self.ext_gate, "this builder was made without extension rows"
        x0, x1, y0, y1, z0, z1 = (self.values[v] for v in (*x, *y, *z))
        w0 = self._new(x0 * y0 + 7 * x1 * y1 + z0)
        w1 = self._new(x0 * y1 + x1 * y0 + z1)
        self.prog += (13, w0, w1, *x, *y, *z)
        if len(self.ext_rows[-1]) == self.R // 8:
            self.ext_rows.append([])
        self.ext_rows[-1].append((*x, *y, *z, w0, w1))
        return (w0, w1)

    def two_to_one(self, left4, right4):
        """PoseidonHash::two_to_one: permute(left || right || 0 0 0 0)[0..4)"""
        zero = self.constant(0)
        return self.poseidon(list(left4) + list(right4) + [zero] * 4)[:4]

    def hash_no_pad(self, elems):
        """overwrite-mode sponge, rate 8 (the leaf hashing of the Merkle trees); <= 4 elements: the padded input itself"""
        zero = self.constant(0)
        if len(elems) <= 4:
            return list(elems) + [zero] * (4 - len(elems))
        state = [zero] * 12
        for off in range(0, len(elems), 8):
            chunk = list(elems[off:off + 8])
            state = self.poseidon(chunk + state[len(chunk):])
        return state[:4]

    def merkle_root_from_path(self, leaf_digest4, index_bits, siblings):
        """verify_merkle_proof: fold a digest up a path.  index_bits[l] (boolean variables, LSB first) says whether the node is the
        RIGHT child at level l; siblings[l] = 4 variables.  Returns the 4 variables of the node reached (compare with a cap entry).
        One Poseidon row per level: the row's swap input orders (node, sibling) and constrains the bit to be boolean."""
        cur = list(leaf_digest4)
        zero = self.constant(0)
        for b, sib in zip(index_bits, siblings):
            cur = self.poseidon(cur + list(sib) + [zero] * 4, swap=b)[:4]
        return cur

    # ---- cloning a recorded segment ---------------------------------------------------------------------------------------------
    # A recursion node verifies N proofs of ONE circuit: N copies of the same sub-circuit on different inputs.  Laying each copy down through the
    # gadget code costs ~0.4 s of Python per child; the copies differ from the first only in their variable numbers and input tags, so they are
    # made from the first one's recorded ops instead (numpy remapping), and every cloned variable's value comes from ONE run of the witness
    # evaluator over what has been recorded.  The result is the circuit a direct build lays down (same cells, same copy classes: same key).
    _OP_WORDS = _OP_WORDS
    _OP_LEN = {op: n for op, n in _OP_WORDS.items() if op in (0, 1, 2, 3, 4, 6, 11, 12, 13)}                        # the ops clone_segment copies
    _OP_VARS = {0: (1, 2, 3, 4), 1: (1,), 2: (1, 2), 3: (1, 2), 4: (1, 2, 3, 4), 6: tuple(range(1, 25)), 11: (1, 2), 12: tuple(range(1, 26)),
                13: tuple(range(1, 9))}

    def mark(self):
        """a position in the recording (take one right after begin_segment() and one right before end_segment())"""
        return {"vars": len(self.values), "prog": len(self.prog), "inputs": len(self.input_tags), "eq": len(self.eq_pairs), "wc": len(self.word_checks),
                "public": len(self.public), "sha": len(self.sha_rows)}

    def clone_segment(self, m0, m1, out, clones):
        """Repeat the segment recorded between marks m0 and m1 once per entry of `clones` = [(retag, words), ...]: retag maps the list numbers of the
        segment's input tags (and word checks) to the clone's, words(list number) -> that input list as an array of u64 words (the clone's free
        inputs take their values from it).  `out`: any nesting of lists / tuples / dicts of variables the segment produced; returns its image per
        clone.  The cloned variables' values are NOT valid until fill_values() has run.  Returns None (nothing recorded) when the segment uses ops
        or state this does not copy (SHA rows, non-native products, public inputs, untagged inputs): the caller then lays the copies down directly."""
        if m1["public"] != m0["public"] or m1["sha"] != m0["sha"] or getattr(self, "_open", False):
            return None
        v0, v1 = m0["vars"], m1["vars"]
        w = self.prog[m0["prog"]:m1["prog"]]
        offs = {op: [] for op in self._OP_LEN}
        pos, n = 0, len(w)
        while pos < n:
            op = w[pos]
            ln = self._OP_LEN.get(op)
            if ln is None:
                return None
            offs[op].append(pos)
            pos += ln
        if pos != n:
            return None
        tags = self.input_tags[m0["inputs"]:m1["inputs"]]
        if any(t is None for t in tags) or len(tags) != len(offs[1]):
            return None
        win = np.frombuffer(w, dtype=np.uint64)
        offs = {op: np.array(o, dtype=np.int64) for op, o in offs.items()}
        var_pos = np.concatenate([(offs[op][:, None] + np.array(self._OP_VARS[op], dtype=np.int64)[None, :]).ravel() for op in offs if offs[op].size])
        inp_pos = offs[1] + 2
        # constants first used inside the segment were recorded into the prefix: shared by every copy, never remapped
        shared = np.array(sorted(v for c in self._consts.values() if v0 <= c < v1 for v in (c - 1, c)), dtype=np.int64)
        # arithmetic gates by row constants, in recording order
        a_off = offs[0]
        if a_off.size:
            keys = np.stack([win[a_off + 5], win[a_off + 6], win[a_off + 7]], axis=1)
            uniq, inv = np.unique(keys, axis=0, return_inverse=True)
            inv = inv.ravel()
            groups = [(tuple(int(x) for x in uniq[g]), np.nonzero(inv == g)[0]) for g in range(uniq.shape[0])]
        else:
            groups = []
        # Poseidon rows of both kinds, in recording order
        p_off = np.concatenate([offs[6], offs[12]])
        p_swap = np.concatenate([np.zeros(offs[6].size, dtype=bool), np.ones(offs[12].size, dtype=bool)])
        order = np.argsort(p_off, kind="stable")
        p_off, p_swap = p_off[order], p_swap[order]
        e_off = offs[13]
        eq = np.array(self.eq_pairs[m0["eq"]:m1["eq"]], dtype=np.int64).reshape(-1, 2)
        wcs = self.word_checks[m0["wc"]:m1["wc"]]
        in_vars = win[offs[1] + 1].astype(np.int64)                   # the segment's input variables, in input order
        tag_pos = np.array([t[1] for t in tags], dtype=np.int64)
        tag_list = [t[0] for t in tags]
        zero = self.constant(0) if offs[6].size else None
        per_row = self.R // 8

        def image(x, lut):
            if isinstance(x, dict):
                return {k: image(v, lut) for k, v in x.items()}
            if isinstance(x, (list, tuple)):
                return type(x)(image(v, lut) for v in x)
            return int(lut[x]) if isinstance(x, (int, np.integer)) and 0 <= x < v1 else x

        results = []
        for retag, words in clones:
            base = len(self.values)
            lut = np.arange(v1, dtype=np.int64)
            lut[v0:v1] += base - v0
            if shared.size:
                lut[shared] = shared
            cl = win.copy()
            cl[var_pos] = lut[win[var_pos].astype(np.int64)].astype(np.uint64)
            cl[inp_pos] = (win[inp_pos].astype(np.int64) - m0["inputs"] + len(self.input_tags)).astype(np.uint64)
            self.values.extend([0] * (v1 - v0))
            self.parent.extend(range(base, base + v1 - v0))
            # free inputs: tags and values
            new_lists = [retag[t] for t in tag_list]
            self.input_tags.extend(zip(new_lists, tag_pos.tolist()))
            cache = {}
            for lid in set(new_lists):
                cache[lid] = np.asarray(words(lid), dtype=np.uint64)
            new_in = lut[in_vars]
            for v, lid, p in zip(new_in.tolist(), new_lists, tag_pos.tolist()):
                self.values[v] = int(cache[lid][p])
            self.begin_segment()
            self.prog.frombytes(cl.tobytes())
            self.end_segment()
            # gates
            for key, idx in groups:
                o = a_off[idx]
                slots = list(zip(cl[o + 2].tolist(), cl[o + 3].tolist(), cl[o + 4].tolist(), cl[o + 1].tolist()))
                rows = self.arith_rows.setdefault(key, [[]])
                room = self.G - len(rows[-1])
                rows[-1].extend(slots[:room])
                for i in range(room, len(slots), self.G):
                    rows.append(slots[i:i + self.G])
            if p_off.size:
                outs_m = cl[p_off[:, None] + np.arange(1, 13)[None, :]].tolist()
                ins_m = cl[p_off[:, None] + np.arange(13, 25)[None, :]].tolist()
                sw = np.where(p_swap, cl[np.minimum(p_off + 25, cl.size - 1)], 0).tolist()
                for ins_r, outs_r, has, sv in zip(ins_m, outs_m, p_swap.tolist(), sw):
                    self.pos_rows.append((ins_r, outs_r, sv if has else zero))
            if e_off.size:
                cols = cl[e_off[:, None] + np.array([3, 4, 5, 6, 7, 8, 1, 2])[None, :]].tolist()
                for t in cols:
                    if len(self.ext_rows[-1]) == per_row:
                        self.ext_rows.append([])
                    self.ext_rows[-1].append(tuple(t))
            # copy constraints, as assert_equal records them
            if eq.size:
                for a, b2 in lut[eq].tolist():
                    ra, rb = self._find(a), self._find(b2)
                    if ra != rb:
                        self.parent[ra] = rb
                        self.eq_pairs += (a, b2)
            for c in wcs:
                tag = (retag[c[1][0]], c[1][1])
                if c[0] == "const":
                    self.word_checks.append(("const", tag, c[2]))
                elif c[0] == "var":
                    self.word_checks.append(("var", tag, int(lut[c[2]])))
                else:
                    self.word_checks.append(("bits", tag, [int(lut[v]) for v in c[2]]))
            results.append(image(out, lut))
        self._stale_values = True
        return results

    def fill_values(self, poseidon_consts):
        """after clone_segment: every variable's value from ONE run of the witness evaluator over what has been recorded (the clones' inputs were
        set from their words); ValueError when a clone's inputs do not satisfy its copy constraints (e.g. a proof that does not verify)"""
        import ctypes
        import os
        from . import load_library
        if not getattr(self, "_stale_values", False):
            return
        lib = load_library()
        rc, circ, diag = (np.ascontiguousarray(a, dtype=np.uint64) for a in poseidon_consts)
        prog = np.frombuffer(self._prefix + self.prog, dtype=np.uint64)
        pos, w, n = 0, self.prog, len(self.prog)
        # the input vector: the value of each input variable (op 1: variable, input index), found by walking the ops
        inputs = np.zeros(len(self.input_tags), dtype=np.uint64)
        lens = self._OP_WORDS
        while pos < n:
            op = w[pos]
            if op == 1:
                inputs[w[pos + 2]] = self.values[w[pos + 1]]
            pos += lens[op]
        vals = np.zeros(len(self.values), dtype=np.uint64)
        bad = ctypes.c_size_t(0)
        sb = np.array([len(self._prefix) + o for o in self.seg_bounds], dtype=np.uint64) if len(self.seg_bounds) > 2 else None
        eqp = np.array(self.eq_pairs, dtype=np.uint64)
        rcode = lib.glp_witness_eval_mt(rc.ctypes.data, circ.ctypes.data, diag.ctypes.data, prog.ctypes.data, prog.size,
                                        inputs.ctypes.data if inputs.size else None, inputs.size, vals.ctypes.data, vals.size,
                                        eqp.ctypes.data if eqp.size else None, eqp.size // 2, ctypes.byref(bad),
                                        sb.ctypes.data if sb is not None else None, sb.size - 1 if sb is not None else 0, min(32, os.cpu_count() or 1))
        if rcode == -7:
            raise ValueError("a cloned segment's inputs do not satisfy the circuit (the witness evaluator refused them)")
        if rcode != 0:
            raise ValueError("witness program malformed after cloning")
        self.values = vals.tolist()
        self._stale_values = False

    # ---- layout ---------------------------------------------------------------------------------------------------------------
    def program(self):
        """the recorded circuit as a WitnessProgram: layout (constants, cells, sigma recipe) + the straight-line witness program"""
        return WitnessProgram(self)

    def build(self, cap_height=1):
        """rows: [public inputs][Poseidon rows][arithmetic rows by constants], padded to a power of two.
        Returns (PlonkCircuit, device wires buffer with the Poseidon rows filled, public values) for the values recorded while building."""
        prog = self.program()
        ck = prog.setup(self.prover, cap_height)
        dw, public = prog.device_witness(self.prover, np.array(self.values, dtype=np.uint64))
        return ck, dw, public


class WitnessProgram:
    """A circuit recorded by CircuitBuilder, separated from the witness it was recorded with: the layout (which variable sits in which cell,
    the constant columns, sigma) and the program that recomputes every variable from the free inputs.  `setup` commits the circuit once;
    `evaluate` + `device_witness` make the wire matrix for NEW inputs without touching the Python builder again (glp_witness_eval, host C++)."""

    def __init__(self, b):
        W, R, G = b.W, b.R, b.G
        self.W, self.R = W, R
        arith = [(key, row) for key, rows in sorted(b.arith_rows.items()) for row in rows if row]
        ext_rows = [r for r in b.ext_rows if r]
        self.has_sha, self.has_ext = bool(b.sha_rows), bool(b.ext_gate)
        n_rows = len(b.public) + len(b.pos_rows) + len(b.sha_rows) + len(ext_rows) + len(arith)
        self.log_n = max(3, (max(n_rows, 1) - 1).bit_length())
        n = 1 << self.log_n
        consts = np.zeros((PLONK_NCONST + (4 if self.has_sha else 0) + (1 if self.has_ext else 0), n), dtype=np.uint64)
        # placed cells (wire, row, variable), in the order public inputs, Poseidon rows, SHA rows, extension rows, arithmetic rows by constants —
        # row by row, cell by cell: setup() links the cells of a copy class in THIS order, so it is part of what the circuit's key is a function of.
        # Built with numpy (a signature leaf has 2.2 M gate slots, a recursion node 0.8 M: the per-slot Python loop was half of a recording).
        from itertools import chain
        I64 = np.int64
        cjs, cis, cvs = [], [], []

        def flat(it, count):
            return np.fromiter(it, dtype=I64, count=count)
        i = len(b.public)
        consts[4, :i] = 1
        cjs.append(np.zeros(i, dtype=I64)); cis.append(np.arange(i, dtype=I64)); cvs.append(flat(iter(b.public), i))
        npos = len(b.pos_rows)
        self.pos_row_ids = np.arange(i, i + npos, dtype=np.uint32)
        if npos:
            consts[5, i:i + npos] = 1
            cells = np.empty((npos, 25), dtype=I64)
            cells[:, :12] = flat(chain.from_iterable(r[0] for r in b.pos_rows), 12 * npos).reshape(npos, 12)
            cells[:, 12:24] = flat(chain.from_iterable(r[1] for r in b.pos_rows), 12 * npos).reshape(npos, 12)
            cells[:, 24] = flat((r[2] for r in b.pos_rows), npos)                 # GLP_POS_SWAP_WIRE: an index bit, or the constant 0 (never a free cell)
            cjs.append(np.tile(np.arange(25, dtype=I64), npos)); cis.append(np.repeat(np.arange(i, i + npos, dtype=I64), 25)); cvs.append(cells.ravel())
            i += npos
        nsha = len(b.sha_rows)
        self.sha_row_ids = np.arange(i, i + nsha, dtype=np.uint32)
        self.sha_kinds = np.fromiter((r[0] for r in b.sha_rows), dtype=np.uint32, count=nsha)
        if nsha:
            consts[6 + self.sha_kinds.astype(I64), np.arange(i, i + nsha)] = 1
            consts[3, i:i + nsha] = np.fromiter((r[2] for r in b.sha_rows), dtype=np.uint64, count=nsha)
            lens = np.fromiter((len(r[1]) for r in b.sha_rows), dtype=I64, count=nsha)
            tot = int(lens.sum())
            first = np.cumsum(lens) - lens
            cjs.append(np.arange(tot, dtype=I64) - np.repeat(first, lens)); cis.append(np.repeat(np.arange(i, i + nsha, dtype=I64), lens))
            cvs.append(flat(chain.from_iterable(r[1] for r in b.sha_rows), tot))
            i += nsha
        for row in ext_rows:                                             # q_ext is the LAST constant column
            consts[-1, i] = 1
            m = len(row)
            cjs.append(np.arange(8 * m, dtype=I64)); cis.append(np.full(8 * m, i, dtype=I64)); cvs.append(flat(chain.from_iterable(row), 8 * m))
            i += 1
        fixed = []                                   # (wire, row, value): cells of unused gate slots that must hold c2
        by_key = {}
        for key, row in arith:
            by_key.setdefault(key, []).append(row)
        for (c0, c1, c2), rows in by_key.items():     # (`arith` is sorted by key: so is by_key)
            nr = len(rows)
            consts[0, i:i + nr], consts[1, i:i + nr], consts[2, i:i + nr], consts[3, i:i + nr] = 1, c0, c1, c2
            lens = np.fromiter((len(r) for r in rows), dtype=I64, count=nr)
            ns = int(lens.sum())
            slots = flat(chain.from_iterable(chain.from_iterable(rows)), 4 * ns)
            srow = np.repeat(np.arange(i, i + nr, dtype=I64), lens)                        # row of each slot
            sg = np.arange(ns, dtype=I64) - np.repeat(np.cumsum(lens) - lens, lens)         # its position in the row
            cjs.append((4 * sg[:, None] + np.arange(4, dtype=I64)[None, :]).ravel()); cis.append(np.repeat(srow, 4)); cvs.append(slots)
            if c2:
                for r in np.nonzero(lens < G)[0].tolist():
                    fixed += [(4 * g + 3, i + r, c2) for g in range(int(lens[r]), G)]      # an unused slot must still satisfy its gate: w = c2
            i += nr
        self.consts = consts
        self.cj, self.ci, self.cv = np.concatenate(cjs), np.concatenate(cis), np.concatenate(cvs)
        if self.cj.size and int(self.cj.max()) >= R:
            raise ValueError("a variable sits on an unrouted wire")
        self.fixed = np.array(fixed, dtype=np.uint64).reshape(-1, 3)
        self.public_vars = np.array(b.public, dtype=np.int64)
        par = np.array(b.parent, dtype=np.int64)                          # union-find roots by pointer jumping (1.4 M variables: no Python loop)
        while True:
            nxt = par[par]
            if np.array_equal(nxt, par):
                break
            par = nxt
        self.roots = par
        self.n_values = len(b.values)
        self.prog = np.frombuffer(b._prefix + b.prog, dtype=np.uint64)       # (a view of the concatenation, which nothing else holds)
        self.seg_bounds = np.array([len(b._prefix) + o for o in b.seg_bounds], dtype=np.uint64) if len(b.seg_bounds) > 2 else None
        tags = b.input_tags
        if tags and all(t is not None for t in tags):
            self.input_tags = np.array(tags, dtype=np.int64).reshape(-1, 2)
        else:
            self.input_tags = None if tags else np.zeros((0, 2), dtype=np.int64)
        self.n_inputs = len(tags)
        self.eq_pairs = np.array(b.eq_pairs, dtype=np.uint64)
        # facts about input words checked outside the circuit: (list, position, value) / (list, position, variable) / bit lists
        wc = b.word_checks
        self.wc_const = np.array([(c[1][0], c[1][1], c[2]) for c in wc if c[0] == "const"], dtype=np.uint64).reshape(-1, 3)
        self.wc_var = np.array([(c[1][0], c[1][1], c[2]) for c in wc if c[0] == "var"], dtype=np.int64).reshape(-1, 3)
        bits = [c for c in wc if c[0] == "bits"]
        self.wc_bits = np.array([(c[1][0], c[1][1], len(c[2])) for c in bits], dtype=np.int64).reshape(-1, 3)
        self.wc_bit_vars = np.array([v for c in bits for v in c[2]], dtype=np.int64)
        self.stats = {"rows": n, "poseidon_rows": len(b.pos_rows), "arith_gates": sum(len(r) for _, r in arith), "variables": self.n_values,
                      "inputs": self.n_inputs, "sha_rows": len(b.sha_rows), "rows_used": n_rows, "ext_rows": len(ext_rows)}
        self._finish()

    _SAVED = ("consts", "cj", "ci", "cv", "fixed", "pos_row_ids", "sha_row_ids", "sha_kinds", "public_vars", "roots", "prog", "eq_pairs", "wc_const", "wc_var",
              "wc_bits", "wc_bit_vars")
    _STATS = ("rows", "poseidon_rows", "arith_gates", "variables", "inputs", "sha_rows", "rows_used", "ext_rows")

    def _finish(self):
        """what is derived from the recorded arrays: the cell -> variable map of the whole wire matrix (0xFFFFFFFF = zero; an unused gate slot
        that must hold c2 reads it from the tail of the value vector, where device_witness appends those constants)"""
        n = 1 << self.log_n
        cell = np.full((self.W, n), 0xFFFFFFFF, dtype=np.uint32)
        cell[self.cj, self.ci] = self.cv
        if self.fixed.shape[0]:
            cell[self.fixed[:, 0].astype(np.int64), self.fixed[:, 1].astype(np.int64)] = self.n_values + np.arange(self.fixed.shape[0])
        self.cell_index = cell
        self.fixed_values = np.ascontiguousarray(self.fixed[:, 2])
        self._dev = {}                               # prover -> resident cell map

    def save(self, path):
        """the recorded circuit as one .npz of plain arrays (nothing executable): a later process loads it instead of running the builder again —
        the 'build once, prove many times' split of the reference's circuit artifacts"""
        meta = np.array([self.W, self.R, self.log_n, self.n_values, self.n_inputs, -1 if self.input_tags is None else 0, int(self.has_sha),
                         int(self.has_ext)], dtype=np.int64)
        arrays = {k: getattr(self, k) for k in self._SAVED}
        arrays["input_tags"] = self.input_tags if self.input_tags is not None else np.zeros((0, 2), dtype=np.int64)
        arrays["seg_bounds"] = self.seg_bounds if self.seg_bounds is not None else np.zeros(0, dtype=np.uint64)
        arrays["stats"] = np.array([self.stats[k] for k in self._STATS], dtype=np.int64)
        np.savez(path, meta=meta, **arrays)

    @classmethod
    def load(cls, path):
        self = object.__new__(cls)
        with np.load(path, allow_pickle=False) as z:
            self.W, self.R, self.log_n, self.n_values, self.n_inputs, tagged, sha, ext = (int(v) for v in z["meta"])
            self.has_sha, self.has_ext = bool(sha), bool(ext)
            for k in cls._SAVED:
                setattr(self, k, z[k])
            self.input_tags = None if tagged < 0 else z["input_tags"]
            self.seg_bounds = z["seg_bounds"] if z["seg_bounds"].size else None
            self.stats = dict(zip(cls._STATS, (int(v) for v in z["stats"])))
        n = 1 << self.log_n
        if (self.consts.shape != (PLONK_NCONST + (4 if self.has_sha else 0) + (1 if self.has_ext else 0), n) or self.sha_row_ids.size != self.sha_kinds.size or self.roots.size != self.n_values or not (self.cj.size == self.ci.size == self.cv.size)
                or (self.cv.size and (self.cv.max() >= self.n_values or self.cj.max() >= self.R or self.ci.max() >= n))):
            raise ValueError("not a recorded circuit of this format")
        self._finish()
        return self

    def setup(self, prover, cap_height=1):
        """commit the circuit (constants + sigma): one cycle per copy class over its cells, sigma values by the GPU's field multiplication"""
        R, n = self.R, 1 << self.log_n
        cls = self.roots[self.cv]
        order = np.argsort(cls, kind="stable")
        sc = cls[order]
        first = np.ones(order.size, dtype=bool)
        first[1:] = sc[1:] != sc[:-1]
        start = np.maximum.accumulate(np.where(first, np.arange(order.size), 0))      # index of the group's first cell
        last = np.ones(order.size, dtype=bool)
        last[:-1] = first[1:]
        nxt = np.arange(order.size) + 1
        nxt[last] = start[last]                                                        # the last cell of a class points back to its first
        tgt_col = np.tile(np.arange(R, dtype=np.int64)[:, None], (1, n))
        tgt_row = np.tile(np.arange(n, dtype=np.int64)[None, :], (R, 1))
        tgt_col[self.cj[order], self.ci[order]] = self.cj[order[nxt]]
        tgt_row[self.cj[order], self.ci[order]] = self.ci[order[nxt]]
        ks = np.array([pow(7, j, P) for j in range(R)], dtype=np.uint64)
        delta = np.zeros(n, dtype=np.uint64)
        delta[1] = 1
        wp = prover.fft(delta)                                                         # w_n^r, r < n
        sigma = prover.field_op("mul", ks[tgt_col], wp[tgt_row])
        self._sigma = sigma                            # kept for export_raw (a compiled host commits the same circuit from it)
        # the flags follow the rows the circuit has: a selector column that is zero everywhere would still cost its constraints at every LDE point
        return PlonkCircuit(prover, self.consts, sigma, cap_height=cap_height, n_wires=self.W, n_public=len(self.public_vars),
                            poseidon=bool(self.pos_row_ids.size), sha=self.has_sha, ext=self.has_ext)

    def export_raw(self, directory, sample_inputs=None, cap_height=1):
        """the recording as raw little-endian arrays + a text manifest, for a host WITHOUT Python (tests/cpp/host_replay.cpp replays it through the
        C ABI: glp_plonk_setup_ex, glp_witness_eval_mt, glp_gather_u64, the row fillers, glp_plonk_prove_ex).  Call after setup() (sigma is part
        of what a host needs).  sample_inputs: an input vector to ship along (inputs.bin)."""
        import os
        if getattr(self, "_sigma", None) is None:
            raise ValueError("export_raw after setup(): sigma is computed there")
        os.makedirs(directory, exist_ok=True)
        arrays = {"consts": (self.consts, "<u8"), "sigma": (self._sigma, "<u8"), "prog": (self.prog, "<u8"), "eq_pairs": (self.eq_pairs, "<u8"),
                  "seg_bounds": (self.seg_bounds if self.seg_bounds is not None else np.zeros(0, dtype=np.uint64), "<u8"),
                  "cell_index": (self.cell_index, "<u4"), "fixed_values": (self.fixed_values, "<u8"), "pos_rows": (self.pos_row_ids, "<u4"),
                  "sha_rows": (self.sha_row_ids, "<u4"), "sha_kinds": (self.sha_kinds, "<u4"), "public_vars": (self.public_vars, "<u8")}
        if sample_inputs is not None:
            arrays["inputs"] = (np.asarray(sample_inputs, dtype=np.uint64), "<u8")
        if self.input_tags is not None and self.n_inputs:
            # programs whose inputs are words of byte strings (the proofs a verifier circuit consumes): input i = word tags[i][1] of string tags[i][0];
            # plus the facts checked outside the circuit: (string, word, value) constants, (string, word, variable) copies, query-index bit lists
            arrays["input_tags"] = (self.input_tags, "<u8")
            arrays["wc_const"] = (self.wc_const, "<u8")
            arrays["wc_var"] = (self.wc_var, "<u8")
            arrays["wc_bits"] = (self.wc_bits, "<u8")
            arrays["wc_bit_vars"] = (self.wc_bit_vars, "<u8")
        with open(os.path.join(directory, "manifest.txt"), "w") as f:
            f.write(f"log_n {self.log_n}\nn_wires {self.W}\nn_routed {self.R}\nn_public {len(self.public_vars)}\nn_const {self.consts.shape[0]}\n"
                    f"n_values {self.n_values}\nn_inputs {self.n_inputs}\ncap_height {cap_height}\n"
                    f"flags {(1 if self.pos_row_ids.size else 0) | (2 if self.has_sha else 0) | (4 if self.has_ext else 0)}\n")
            for name, (arr, dt) in arrays.items():
                a = np.ascontiguousarray(arr).astype(dt)
                a.tofile(os.path.join(directory, name + ".bin"))
                f.write(f"array {name} {a.size}\n")

    def evaluate(self, poseidon_consts, inputs, threads=None):
        """every variable's value for new inputs (glp_witness_eval_mt: the recorded segments on `threads` host threads, default all cores);
        ValueError when the inputs do not satisfy the circuit's copy constraints"""
        import ctypes
        import os
        from . import load_library
        lib = load_library()
        rc, circ, diag = (np.ascontiguousarray(a, dtype=np.uint64) for a in poseidon_consts)
        inp = np.ascontiguousarray(inputs, dtype=np.uint64)
        if inp.size != self.n_inputs:
            raise ValueError(f"{inp.size} inputs given, the program takes {self.n_inputs}")
        vals = np.zeros(self.n_values, dtype=np.uint64)
        bad = ctypes.c_size_t(0)
        sb = self.seg_bounds
        nt = int(threads) if threads else min(32, os.cpu_count() or 1)
        rcode = lib.glp_witness_eval_mt(rc.ctypes.data, circ.ctypes.data, diag.ctypes.data, self.prog.ctypes.data, self.prog.size,
                                        inp.ctypes.data if inp.size else None, inp.size, vals.ctypes.data, vals.size,
                                        self.eq_pairs.ctypes.data if self.eq_pairs.size else None, self.eq_pairs.size // 2, ctypes.byref(bad),
                                        sb.ctypes.data if sb is not None else None, sb.size - 1 if sb is not None else 0, nt)
        if rcode == -7:
            what = ("a row input is out of range (a SHA word above 32 bits, a swap bit above 1)" if bad.value == ctypes.c_size_t(-1).value
                    else f"copy constraint {bad.value} fails")
            raise ValueError(f"the inputs do not satisfy the circuit ({what})")
        if rcode != 0:
            raise ValueError("witness program or inputs malformed")
        return vals

    def inputs_from_words(self, word_lists):
        """the input vector of a program whose inputs were tagged (list number, word position) — e.g. the proofs a verifier circuit consumes —
        after checking the recorded facts about the non-input words (constants of the statement shape, the circuit's key)"""
        ws = [np.frombuffer(bytes(w), dtype="<u8") if isinstance(w, (bytes, bytearray)) else np.asarray(w, dtype=np.uint64) for w in word_lists]
        sizes = np.array([w.size for w in ws], dtype=np.int64)
        off = np.concatenate(([0], np.cumsum(sizes)))[:-1]
        flat = np.concatenate(ws) if ws else np.zeros(0, dtype=np.uint64)
        def at(tab):
            """the words a table's (list, position) columns name; ValueError when one is missing"""
            k, pos = tab[:, 0].astype(np.int64), tab[:, 1].astype(np.int64)
            if k.size and (k.max() >= len(ws) or np.any(pos >= sizes[k])):
                raise ValueError("an input is shorter than this circuit expects")
            return flat[off[k] + pos]
        if np.any(at(self.wc_const) != self.wc_const[:, 2]):
            raise ValueError("an input is not what this circuit was built for (a word fixed by the statement shape or the circuit's key differs)")
        at(self.wc_var), at(self.wc_bits)
        if self.n_inputs == 0:
            return np.zeros(0, dtype=np.uint64), ws
        if self.input_tags is None:
            raise ValueError("this program has untagged inputs: pass the input vector itself")
        return at(self.input_tags), ws

    def check_words(self, vals, ws):
        """the recorded facts that tie non-input words to computed variables (redundant copies inside a proof, query indices)"""
        sizes = np.array([w.size for w in ws], dtype=np.int64)
        off = np.concatenate(([0], np.cumsum(sizes)))[:-1]
        flat = np.concatenate(ws) if ws else np.zeros(0, dtype=np.uint64)
        vals = np.asarray(vals, dtype=np.uint64)
        if self.wc_var.shape[0]:
            k, pos, var = self.wc_var[:, 0], self.wc_var[:, 1], self.wc_var[:, 2]
            bad = np.nonzero(flat[off[k] + pos] != vals[var])[0]
            if bad.size:
                raise ValueError(f"input {int(k[bad[0]])}: word {int(pos[bad[0]])} differs from the value the circuit derives")
        if self.wc_bits.shape[0]:
            k, pos, nb = self.wc_bits[:, 0], self.wc_bits[:, 1], self.wc_bits[:, 2]
            starts = np.concatenate(([0], np.cumsum(nb)))[:-1]
            shifts = (np.arange(self.wc_bit_vars.size) - np.repeat(starts, nb)).astype(np.uint64)
            packed = np.add.reduceat(vals[self.wc_bit_vars] << shifts, starts)         # bit counts are >= 1 and <= 64, values are 0/1
            bad = np.nonzero(flat[off[k] + pos] != packed)[0]
            if bad.size:
                raise ValueError(f"input {int(k[bad[0]])}: word {int(pos[bad[0]])} differs from the index the transcript derives")

    def _resident(self, prover):
        """what stays on the device per prover: the cell -> variable map, the Poseidon / SHA row lists, an upload buffer for the variables and
        (reuse=True) the wire matrix itself — hipMalloc / hipFree synchronise the WHOLE device, so a witness per proof must not allocate (three
        provers sharing a GPU would serialise on it)"""
        r = self._dev.get(id(prover))
        if r is None or r["cell"].ptr is None or r["cell"].prover is not prover:            # never uploaded, or freed with its prover (Prover.close)
            r = {"cell": prover.to_device(self.cell_index),
                 "pos": prover.to_device(self.pos_row_ids) if self.pos_row_ids.size else None,
                 "sha": prover.to_device(self.sha_row_ids) if self.sha_row_ids.size else None,
                 "kinds": prover.to_device(self.sha_kinds) if self.sha_kinds.size else None,
                 "src": DeviceBuffer(prover, max(8, (self.n_values + self.fixed_values.size) * 8)), "wires": None}
            self._dev[id(prover)] = r
        return r

    def device_witness(self, prover, vals, reuse=False):
        """variable values -> wire matrix on the device: the values are uploaded (8 bytes per VARIABLE, not per cell) and placed by the resident
        cell -> variable map (glp_gather_u64), the Poseidon and SHA rows' derived wires filled by the GPU; returns (buffer, public values).
        reuse=True: the buffer is the program's own per-prover wire matrix (valid until the next call for this prover; do not free it) — no device
        allocation per witness."""
        n = 1 << self.log_n
        r = self._resident(prover)
        vals = np.ascontiguousarray(vals, dtype=np.uint64)
        r["src"].upload(np.concatenate((vals, self.fixed_values)) if self.fixed_values.size else vals)
        if reuse:
            if r["wires"] is None or r["wires"].ptr is None:
                r["wires"] = DeviceBuffer(prover, self.W * n * 8)
            dw = r["wires"]
        else:
            dw = DeviceBuffer(prover, self.W * n * 8)
        prover.gather(dw, r["src"], vals.size + self.fixed_values.size, r["cell"], self.W * n)
        if r["pos"] is not None:
            prover._chk(prover.lib.glp_poseidon_gate_fill_rows(prover.ctx, dw.ptr, self.log_n, self.W, r["pos"].ptr, self.pos_row_ids.size),
                        "glp_poseidon_gate_fill_rows")
        if r["sha"] is not None:
            prover._chk(prover.lib.glp_sha_gate_fill_rows(prover.ctx, dw.ptr, self.log_n, self.W, r["sha"].ptr, r["kinds"].ptr, self.sha_row_ids.size),
                        "glp_sha_gate_fill_rows")
        return dw, [int(v) for v in vals[self.public_vars]]

    def release(self, prover=None):
        """free what is resident (for one prover, or all)"""
        for key in [k for k in self._dev if prover is None or k == id(prover)]:
            for buf in self._dev.pop(key).values():
                if buf is not None and buf.ptr is not None and getattr(buf.prover, "ctx", None):
                    buf.free()


# ---- the Reduce step's aggregation tree -----------------------------------------------------------------------------------------
def merkle_root_host(prover, digests):
    """Poseidon Merkle root (two_to_one) of a power-of-two list of 4-word digests, on the GPU permutation (level by level)"""
    level = [list(map(int, d)) for d in digests]
    assert len(level) & (len(level) - 1) == 0 and level
    while len(level) > 1:
        st = np.array([level[2 * k] + level[2 * k + 1] + [0, 0, 0, 0] for k in range(len(level) // 2)], dtype=np.uint64)
        level = [[int(v) for v in row[:4]] for row in prover.poseidon_permute(st)]
    return level[0]


def build_aggregation_circuit(prover, digests):
    """The circuit of one Reduce: public inputs = the leaf-proof digests (4 words each, leaf order) then the root (4 words);
    constraints = every node of the binary Poseidon tree over them.  Returns (circuit, device wires, public values)."""
    n = len(digests)
    assert n >= 2 and n & (n - 1) == 0, "a power-of-two number of leaves (pad with a fixed digest)"
    b = CircuitBuilder(prover)
    level = []
    for d in digests:
        vs = [b.var(int(x)) for x in d]
        for v in vs:
            b.public_input(v)
        level.append(vs)
    while len(level) > 1:
        level = [b.two_to_one(level[2 * k], level[2 * k + 1]) for k in range(len(level) // 2)]
    for v in level[0]:
        b.public_input(v)
    return b.build()


def aggregate(prover, digests, num_queries=28, pow_bits=16):
    """prove the aggregation tree: returns (root proof bytes, root digest, circuit cap) — the succinct object a Reduce hands on"""
    ck, dw, public = build_aggregation_circuit(prover, digests)
    try:
        proof = ck.prove_(dw, num_queries, pow_bits, public=public)
        return proof, public[-4:], ck.cap()
    finally:
        dw.free()
        ck.free()


def aggregation_key(prover, n_leaves):
    """the verifying key (circuit cap) of the n-leaf aggregation circuit: it depends on n only, not on the digests"""
    ck, dw, _ = build_aggregation_circuit(prover, [[0, 0, 0, 0]] * n_leaves)
    cap = ck.cap()
    dw.free()
    ck.free()
    return cap


# ---- the hashing half of a recursive verifier: every Merkle opening of a child proof, checked in-circuit ---------------------------
PLONK_TAG = 0x32304B4C504C4747
FRI_TAG = 0x32304952464C4747


def parse_proof(proof):
    """the words of a circuit proof by role (the layout written by plonk.hip / fri.hip): statement words, caps,
    and per query the opened leaves and authentication paths of the four batches and of every fold layer.  No verification."""
    w = [int(v) for v in np.frombuffer(bytes(proof), dtype="<u8")]
    if w[0] != PLONK_TAG:
        raise ValueError("not a circuit proof")
    log_n, rb, cap_h, n_pub = w[1], w[4], w[5], w[6]
    log_N = log_n + rb
    capw = 4 << min(cap_h, log_N)
    pos = 8 + n_pub
    stmt_end = pos + 4 * capw                      # header || public inputs || four caps: what glp_plonk_proof_digest hashes
    caps = [w[pos + b * capw: pos + (b + 1) * capw] for b in range(4)]
    pos = stmt_end
    if w[pos] != FRI_TAG:
        raise ValueError("FRI part not found")
    f_log_n, f_rb, cap0, a, fb, nq, _pow, _shift, nb, n_pts = w[pos + 1: pos + 11]
    pos += 11 + n_pts
    n_polys = [w[pos + 2 * b] for b in range(nb)]
    masks = [w[pos + 2 * b + 1] for b in range(nb)]
    pos += 2 * nb + nb * (4 << cap0)
    total = sum(n_polys[b] for p in range(n_pts) for b in range(nb) if (masks[b] >> p) & 1)
    pos += 2 * total
    L = (f_log_n - fb) // a if f_log_n > fb else 0
    final_bits = f_log_n - a * L
    layer_caps, layer_log, layer_caph = [], [], []
    log_len = log_N
    for _ in range(L):
        chh = min(cap0, log_len - a)
        layer_caps.append(w[pos: pos + (4 << chh)])
        pos += 4 << chh
        layer_log.append(log_len)
        layer_caph.append(chh)
        log_len -= a
    pos += (2 << final_bits) + 1
    queries = []
    for _ in range(nq):
        idx = w[pos]
        pos += 1
        trees = []
        for b in range(nb):
            leaf = w[pos: pos + n_polys[b]]
            pos += n_polys[b]
            plen = log_N - cap0
            path = [w[pos + 4 * k: pos + 4 * k + 4] for k in range(plen)]
            pos += 4 * plen
            trees.append({"leaf": leaf, "path": path, "index": idx, "cap": ("batch", b), "cap_log": cap0})
        for l in range(L):
            leaf = w[pos: pos + (2 << a)]
            pos += 2 << a
            plen = (layer_log[l] - a) - layer_caph[l]
            path = [w[pos + 4 * k: pos + 4 * k + 4] for k in range(plen)]
            pos += 4 * plen
            trees.append({"leaf": leaf, "path": path, "index": idx >> (a * (l + 1)), "cap": ("layer", l), "cap_log": layer_caph[l]})
        queries.append(trees)
    if pos != len(w):
        raise ValueError("proof length does not match its header")
    return {"statement": [v % P for v in w[:stmt_end]], "caps": caps, "layer_caps": layer_caps, "queries": queries, "cap_log": cap0}


def opening_check_circuit(prover, proofs):
    """The hashing half of verifying `proofs` (circuit proofs of this library) IN-CIRCUIT.  Public inputs: the 4-word digest of each
    proof (statement + commitments, = glp_plonk_proof_digest).  Constraints, per proof: the digest is the sponge hash of the statement
    words and the four caps (so the caps below are the committed ones); for every FRI query and every tree (4 batches + fold layers) the
    opened leaf hashes, through its authentication path with the index bits choosing left/right, to the cap entry the remaining index bits
    select.  What is NOT constrained here (still the native verifier's job): that the query indices come from the transcript, and the
    arithmetic on the opened values (combination, fold consistency, final polynomial, PLONK identity).
    Returns (circuit, device wires, public values, stats)."""
    b = CircuitBuilder(prover)
    n_pos0 = 0
    stats = {"proofs": len(proofs), "trees": 0}
    for proof in proofs:
        pp = parse_proof(proof)
        stmt = [b.var(v) for v in pp["statement"]]
        digest = b.hash_no_pad(stmt)
        for v in digest:
            b.public_input(v)
        capw = len(pp["caps"][0])
        n_stmt = len(stmt)
        cap_vars = {("batch", k): stmt[n_stmt - (4 - k) * capw: n_stmt - (3 - k) * capw] for k in range(4)}     # bound by the digest
        for l, lc in enumerate(pp["layer_caps"]):
            cap_vars[("layer", l)] = [b.var(v) for v in lc]          # fold-layer caps: transcript data, witnesses here
        for trees in pp["queries"]:
            for t in trees:
                plen, clog = len(t["path"]), t["cap_log"]
                bits = [b.var((t["index"] >> k) & 1) for k in range(plen + clog)]
                top = b.merkle_root_from_path(b.hash_no_pad([b.var(v) for v in t["leaf"]]), bits[:plen],
                                              [[b.var(x) for x in s] for s in t["path"]])
                # the cap entry selected by the remaining index bits: a mux tree over the 2^clog entries
                entries = [cap_vars[t["cap"]][4 * e: 4 * e + 4] for e in range(1 << clog)]
                for k in range(clog):
                    bit = bits[plen + k]
                    b.assert_bool(bit)
                    entries = [[b.select(bit, hi, lo) for lo, hi in zip(entries[2 * e], entries[2 * e + 1])] for e in range(len(entries) // 2)]
                for x, y in zip(top, entries[0]):
                    b.assert_equal(x, y)
                stats["trees"] += 1
    stats["poseidon_rows"] = len(b.pos_rows) - n_pos0
    stats["arith_gates"] = sum(len(r) for rows in b.arith_rows.values() for r in rows)
    ck, dw, public = b.build()
    stats["rows"] = 1 << ck.log_n
    return ck, dw, public, stats
