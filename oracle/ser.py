"""ark-serialize ^0.4.0 ``serialize_compressed`` restated for the types that cross the
reference's wire (mpc-net/src/ser_net.rs:24-25,111-112) and for ``Proof``.

TEST INFRASTRUCTURE ONLY (see oracle/params.py).  ark-serialize is third-party
and not vendored; format per SURVEY.md 8c: Fp -> canonical
little-endian bytes (non-Montgomery); Vec<T> -> u64 LE length + items;
short-Weierstrass affine -> x (Fq2: c0 || c1) with flag bits in the top bits of
the last byte: bit 7 = y is the lexicographically larger of (y, -y), bit 6 =
point at infinity (x = 0).  ark-bls12-381 ^0.4 overrides the point encoding with
the zcash / IETF one (big-endian x, Fq2 as c1 || c0, flags compressed / infinity
/ sort in the three top bits of the FIRST byte): ``*_zcash`` below, pinned on the
published compressed generators in tests/test_circom.py.
"""

BLS12_381_Q = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB


def fp_bytes(v, nbytes):
    return int(v).to_bytes(nbytes, "little")


def fr_vec(vals, nbytes=32):
    return len(vals).to_bytes(8, "little") + b"".join(fp_bytes(v, nbytes) for v in vals)


def _fq_size(q):
    return (q.bit_length() + 7) // 8


def _fq2_gt(a, b):
    # ark-ff QuadExtField Ord: compare c1 first, then c0
    return (a[1], a[0]) > (b[1], b[0])


def g1_zcash(p, q):
    n = _fq_size(q)
    if p is None:
        return bytes([0xC0]) + bytes(n - 1)
    b = bytearray(int(p[0]).to_bytes(n, "big"))
    b[0] |= 0x80 | (0x20 if p[1] > (q - p[1]) % q else 0)
    return bytes(b)


def g2_zcash(p, q):
    n = _fq_size(q)
    if p is None:
        return bytes([0xC0]) + bytes(2 * n - 1)
    x, y = p
    b = bytearray(int(x[1]).to_bytes(n, "big") + int(x[0]).to_bytes(n, "big"))
    neg = ((q - y[0]) % q, (q - y[1]) % q)
    b[0] |= 0x80 | (0x20 if _fq2_gt(y, neg) else 0)
    return bytes(b)


def g1_compressed(p, q):
    if q == BLS12_381_Q:
        return g1_zcash(p, q)
    n = _fq_size(q)
    if p is None:
        b = bytearray(n)
        b[-1] |= 1 << 6
        return bytes(b)
    x, y = p
    b = bytearray(fp_bytes(x, n))
    if y > (q - y) % q:
        b[-1] |= 1 << 7
    return bytes(b)


def g2_compressed(p, q):
    if q == BLS12_381_Q:
        return g2_zcash(p, q)
    n = _fq_size(q)
    if p is None:
        b = bytearray(2 * n)
        b[-1] |= 1 << 6
        return bytes(b)
    x, y = p
    b = bytearray(fp_bytes(x[0], n) + fp_bytes(x[1], n))
    neg = ((q - y[0]) % q, (q - y[1]) % q)
    if _fq2_gt(y, neg):
        b[-1] |= 1 << 7
    return bytes(b)


def proof_compressed(a, b, c, q):
    """ark_groth16::Proof {a: G1, b: G2, c: G1} (128 B for BN254, 192 B for BLS12-381)."""
    return g1_compressed(a, q) + g2_compressed(b, q) + g1_compressed(c, q)
