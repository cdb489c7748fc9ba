"""R1CS for the reference's SHA-256 fixture circuit, rebuilt from its semantics.

`fixtures/sha256/sha256.circom` instantiates circomlib's `Sha256_2`: two 216-bit inputs a, b (private) are
bit-decomposed, laid out big-endian as one padded 512-bit block, compressed once, and the low 216 bits of the
digest are the public output:  out = SHA-256(a as 27 BE bytes || b as 27 BE bytes) mod 2^216.
The compiled `sha256.r1cs` is absent from the reference checkout and circomlib is not vendored,
so the constraint system is regenerated here with the usual gadgets (1 constraint per XOR / Ch, 2 per Maj, bit
decomposition for the modular additions).  It has the same interface (1 public output, 2 private inputs), the
same domain size m = 2^15 and about the same number of wires; it is anchored by the reference's public-output
known answer for a = 1, b = 2 (groth16/examples/sha256.rs:392-393).

Variable layout (ark-relations convention, groth16/src/qap.rs:47-48): index 0 = constant 1, index 1 = `out`
(public), then witnesses.  Rows of A, B, C are lists of (coefficient, variable index).
"""

K = [
    0x428A2F98, 0x71374491, 0xB5C0FBCF, 0xE9B5DBA5, 0x3956C25B, 0x59F111F1, 0x923F82A4, 0xAB1C5ED5, 0xD807AA98,
    0x12835B01, 0x243185BE, 0x550C7DC3, 0x72BE5D74, 0x80DEB1FE, 0x9BDC06A7, 0xC19BF174, 0xE49B69C1, 0xEFBE4786,
    0x0FC19DC6, 0x240CA1CC, 0x2DE92C6F, 0x4A7484AA, 0x5CB0A9DC, 0x76F988DA, 0x983E5152, 0xA831C66D, 0xB00327C8,
    0xBF597FC7, 0xC6E00BF3, 0xD5A79147, 0x06CA6351, 0x14292967, 0x27B70A85, 0x2E1B2138, 0x4D2C6DFC, 0x53380D13,
    0x650A7354, 0x766A0ABB, 0x81C2C92E, 0x92722C85, 0xA2BFE8A1, 0xA81A664B, 0xC24B8B70, 0xC76C51A3, 0xD192E819,
    0xD6990624, 0xF40E3585, 0x106AA070, 0x19A4C116, 0x1E376C08, 0x2748774C, 0x34B0BCB5, 0x391C0CB3, 0x4ED8AA4A,
    0x5B9CCA4F, 0x682E6FF3, 0x748F82EE, 0x78A5636F, 0x84C87814, 0x8CC70208, 0x90BEFFFA, 0xA4506CEB, 0xBEF9A3F7,
    0xC67178F2,
]
IV = [0x6A09E667, 0xBB67AE85, 0x3C6EF372, 0xA54FF53A, 0x510E527F, 0x9B05688C, 0x1F83D9AB, 0x5BE0CD19]

ONE = 0  # index of the constant-1 variable


class R1CS:
    def __init__(self, num_instance, num_witness, a, b, c):
        self.num_instance_variables = num_instance
        self.num_witness_variables = num_witness
        self.a, self.b, self.c = a, b, c
        self.num_constraints = len(a)

    @property
    def num_variables(self):
        return self.num_instance_variables + self.num_witness_variables


class _Builder:
    """Linear combinations are dicts {variable: coefficient}; values are tracked alongside."""

    def __init__(self, p):
        self.p = p
        self.values = [1, 0]          # [1, out]
        self.A, self.B, self.C = [], [], []

    # -- linear combinations ------------------------------------------------------------------------
    @staticmethod
    def const(k):
        return {ONE: k} if k else {}

    def var(self, value):
        self.values.append(value % self.p)
        return {len(self.values) - 1: 1}

    def val(self, lc):
        return sum(c * self.values[i] for i, c in lc.items()) % self.p

    @staticmethod
    def add(*lcs):
        out = {}
        for lc in lcs:
            for i, c in lc.items():
                out[i] = out.get(i, 0) + c
        return {i: c for i, c in out.items() if c}

    @staticmethod
    def scale(lc, k):
        return {i: c * k for i, c in lc.items()} if k else {}

    def sub(self, x, y):
        return self.add(x, self.scale(y, -1))

    def enforce(self, a, b, c):
        p = self.p
        self.A.append([(co % p, i) for i, co in a.items() if co % p])
        self.B.append([(co % p, i) for i, co in b.items() if co % p])
        self.C.append([(co % p, i) for i, co in c.items() if co % p])

    # -- boolean gadgets ----------------------------------------------------------------------------
    def bit(self, value):
        b = self.var(value)
        self.enforce(b, self.sub(self.const(1), b), {})            # b (1 - b) = 0
        return b

    def xor2(self, a, b):
        if not a:
            return b
        if not b:
            return a
        va, vb = self.val(a), self.val(b)
        c = self.var(va ^ vb)
        self.enforce(self.scale(a, 2), b, self.sub(self.add(a, b), c))   # 2ab = a + b - c
        return c

    def xor3(self, a, b, c):
        return self.xor2(self.xor2(a, b), c)

    def ch(self, e, f, g):
        ve, vf, vg = self.val(e), self.val(f), self.val(g)
        out = self.var((ve & vf) ^ ((1 - ve) & vg))
        self.enforce(e, self.sub(f, g), self.sub(out, g))            # e (f - g) = out - g
        return out

    def maj(self, a, b, c):
        va, vb, vc = self.val(a), self.val(b), self.val(c)
        mid = self.var(vb & vc)
        self.enforce(b, c, mid)
        out = self.var((va & vb) ^ (va & vc) ^ (vb & vc))
        self.enforce(a, self.sub(self.add(b, c), self.scale(mid, 2)), self.sub(out, mid))
        return out

    # -- 32-bit words: lists of 32 bit-LCs, index 0 = least significant --------------------------------
    def word_const(self, k):
        return [self.const((k >> i) & 1) for i in range(32)]

    def word_lc(self, w):
        return self.add(*[self.scale(b, 1 << i) for i, b in enumerate(w)])

    @staticmethod
    def rotr(w, k):
        return [w[(i + k) % 32] for i in range(32)]

    @staticmethod
    def shr(w, k):
        return [w[i + k] if i + k < 32 else {} for i in range(32)]

    def xor3_words(self, x, y, z):
        return [self.xor3(x[i], y[i], z[i]) for i in range(32)]

    def reduce32(self, total_lc, nterms):
        """Bits of the integer `total_lc` (< nterms * 2^32); returns the low 32 as a word."""
        nbits = 32 + max(1, (nterms - 1).bit_length())
        v = self.val(total_lc)
        assert v < (1 << nbits)
        bits = [self.bit((v >> i) & 1) for i in range(nbits)]
        self.enforce(total_lc, self.const(1), self.add(*[self.scale(b, 1 << i) for i, b in enumerate(bits)]))
        return bits[:32]


REFERENCE_WIRES = 29823      # witnessSize of the reference's compiled fixture


def build(a_val, b_val, p, pad_wires=None):
    """Returns (R1CS, full_assignment) for inputs a, b < 2^216 over the prime field p.

    pad_wires: total number of variables to pad the system to (REFERENCE_WIRES = 29 823 gives exactly the vector
    lengths of the reference's run: a_share / ax_share 14 911 per party).  circomlib's
    Sha256_2 spends ~3.5k more intermediate signals than the gadgets used here; the padding stands in for them with
    extra bit wires (alternating pattern derived from the digest) under their booleanity constraint
    w * (w - 1) = 0, so the padded system is still satisfied only by a valid witness."""
    assert 0 <= a_val < (1 << 216) and 0 <= b_val < (1 << 216)
    B = _Builder(p)
    a = B.var(a_val)
    b = B.var(b_val)
    a_bits = [B.bit((a_val >> i) & 1) for i in range(216)]
    b_bits = [B.bit((b_val >> i) & 1) for i in range(216)]
    B.enforce(a, B.const(1), B.add(*[B.scale(x, 1 << i) for i, x in enumerate(a_bits)]))
    B.enforce(b, B.const(1), B.add(*[B.scale(x, 1 << i) for i, x in enumerate(b_bits)]))
    # message block, bit 0 = first (most significant) bit of the block
    inp = [a_bits[215 - i] for i in range(216)] + [b_bits[215 - i] for i in range(216)]
    inp.append(B.const(1))
    inp += [{} for _ in range(512 - 433 - 64)]
    inp += [B.const((432 >> (63 - i)) & 1) for i in range(64)]
    assert len(inp) == 512
    w = [[inp[32 * t + 31 - i] for i in range(32)] for t in range(16)]
    for t in range(16, 64):
        x, y = w[t - 15], w[t - 2]
        s0 = B.xor3_words(B.rotr(x, 7), B.rotr(x, 18), B.shr(x, 3))
        s1 = B.xor3_words(B.rotr(y, 17), B.rotr(y, 19), B.shr(y, 10))
        total = B.add(B.word_lc(s1), B.word_lc(w[t - 7]), B.word_lc(s0), B.word_lc(w[t - 16]))
        w.append(B.reduce32(total, 4))
    st = [B.word_const(v) for v in IV]
    for t in range(64):
        av, bv, cv, dv, ev, fv, gv, hv = st
        S1 = B.xor3_words(B.rotr(ev, 6), B.rotr(ev, 11), B.rotr(ev, 25))
        chw = [B.ch(ev[i], fv[i], gv[i]) for i in range(32)]
        S0 = B.xor3_words(B.rotr(av, 2), B.rotr(av, 13), B.rotr(av, 22))
        mjw = [B.maj(av[i], bv[i], cv[i]) for i in range(32)]
        t1 = B.add(B.word_lc(hv), B.word_lc(S1), B.word_lc(chw), B.const(K[t]), B.word_lc(w[t]))
        new_e = B.reduce32(B.add(B.word_lc(dv), t1), 6)
        new_a = B.reduce32(B.add(t1, B.word_lc(S0), B.word_lc(mjw)), 7)
        st = [new_a, av, bv, cv, new_e, ev, fv, gv]
    digest_words = [B.reduce32(B.add(B.const(IV[i]), B.word_lc(st[i])), 2) for i in range(8)]
    # digest bit j (0 = most significant of word 0); out = sum_{i<216} digest_bit[255 - i] 2^i
    dbit = lambda j: digest_words[j // 32][31 - (j % 32)]
    out_lc = B.add(*[B.scale(dbit(255 - i), 1 << i) for i in range(216)])
    B.values[1] = B.val(out_lc)
    B.enforce(out_lc, B.const(1), {1: 1})
    if pad_wires is not None:
        if pad_wires < len(B.values):
            raise ValueError("pad_wires below the size of the circuit (%d variables)" % len(B.values))
        k = 0
        while len(B.values) < pad_wires:
            B.bit((B.values[1] >> (k % 216)) & 1)       # bit() adds the wire and its booleanity constraint
            k += 1
    r1cs = R1CS(2, len(B.values) - 2, B.A, B.B, B.C)
    return r1cs, B.values


def expected_output(a_val, b_val):
    import hashlib
    d = hashlib.sha256(a_val.to_bytes(27, "big") + b_val.to_bytes(27, "big")).digest()
    return int.from_bytes(d, "big") % (1 << 216)
