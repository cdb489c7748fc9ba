"""ark-serialize ^0.4 `serialize_compressed` / `deserialize_compressed` for the values that cross the reference's
wire (mpc-net/src/ser_net.rs:24-25, 111-112) and for `ark_groth16::Proof`, so that a process built on this
library can exchange frames with a real mpc-net peer.

* Fr vectors (the bulk: m/l elements per d_fft message) are converted on the device (`zk_fr_to_bytes` /
  `zk_fr_from_bytes`); `Vec<T>` adds a u64 little-endian length prefix.
* Group elements (one per d_msm message, three per proof) are converted on the host with Python integers; vectors
  of them (CRS shares) on the device (zk_points_decompress).  Two encodings, as in arkworks 0.4:
    - BN254, BLS12-377 (ark-ec's default short-Weierstrass flags): x (Fq2: c0 || c1) little-endian, flags in the two
      top bits of the LAST byte: bit 7 = y is the lexicographically larger of (y, -y) (Fq2 compares c1 first), bit 6
      = point at infinity;
    - BLS12-381 (ark-bls12-381 overrides it with the zcash / IETF encoding): x BIG-endian (Fq2: c1 || c0), flags in
      the three top bits of the FIRST byte: bit 7 = compressed (always set), bit 6 = infinity, bit 5 = y is the
      lexicographically larger.  Pinned on the published compressed generators (tests/test_circom.py).

Marshalling only; nothing here is on the hot path.
"""
import numpy as np

from . import api, fields

# y^2 = x^3 + b (G1) / x^3 + b2 (G2, over Fq2 = Fq[u]/(u^2 + 1))
_CURVE_B = {
    "bn254": (3, None),          # b2 = 3 / (9 + u), filled in below
    "bls12_381": (4, (4, 4)),
    "bls12_377": (1, "none"),    # G2 of BLS12-377 (Fq2 non-residue -5) is not on the reference's hot path
}
ZCASH_CURVES = ("bls12_381",)


def _bn254_b2():
    q = fields.FQ["bn254"]
    inv82 = pow(82, q - 2, q)
    return (27 * inv82 % q, (-3) * inv82 % q)


def _curve_b(curve, g2):
    b1, b2 = _CURVE_B[curve]
    if g2 and b2 == "none":
        raise ValueError("G2 is not available for " + curve)
    if g2 and b2 is None:
        b2 = _bn254_b2()
    return b2 if g2 else b1


# ------------------------------------------------------------------------------------------------ Fq2 helpers
def _f2_mul(a, b, q):
    return ((a[0] * b[0] - a[1] * b[1]) % q, (a[0] * b[1] + a[1] * b[0]) % q)


def _f2_inv(a, q):
    n = pow((a[0] * a[0] + a[1] * a[1]) % q, q - 2, q)
    return (a[0] * n % q, (-a[1]) * n % q)


def _fq_sqrt(a, q):
    """q = 3 mod 4 for BN254 and BLS12-381 (one exponentiation); Tonelli-Shanks otherwise (BLS12-377: q = 1 mod 4)."""
    a %= q
    if a == 0:
        return 0
    if q % 4 == 3:
        r = pow(a, (q + 1) // 4, q)
        return r if r * r % q == a else None
    if pow(a, (q - 1) // 2, q) != 1:
        return None
    s, t = 0, q - 1
    while t % 2 == 0:
        t //= 2
        s += 1
    z = 2
    while pow(z, (q - 1) // 2, q) != q - 1:
        z += 1
    c, x, b, m = pow(z, t, q), pow(a, (t + 1) // 2, q), pow(a, t, q), s
    while b != 1:
        i, b2 = 0, b
        while b2 != 1:
            b2 = b2 * b2 % q
            i += 1
        e = pow(c, 1 << (m - i - 1), q)
        x, c = x * e % q, e * e % q
        b, m = b * c % q, i
    return x


def _f2_sqrt(a, q):
    """Square root in Fq[u]/(u^2+1) via the norm (complex method)."""
    if a == (0, 0):
        return (0, 0)
    if a[1] == 0:
        r = _fq_sqrt(a[0], q)
        if r is not None:
            return (r, 0)
        r = _fq_sqrt((-a[0]) % q, q)
        return None if r is None else (0, r)
    nrm = _fq_sqrt((a[0] * a[0] + a[1] * a[1]) % q, q)
    if nrm is None:
        return None
    inv2 = pow(2, q - 2, q)
    for cand in ((a[0] + nrm) * inv2 % q, (a[0] - nrm) * inv2 % q):
        x0 = _fq_sqrt(cand, q)
        if x0 is not None and x0 != 0:
            x1 = a[1] * pow(2 * x0, q - 2, q) % q
            if _f2_mul((x0, x1), (x0, x1), q) == (a[0] % q, a[1] % q):
                return (x0, x1)
    return None


def _f2_gt(a, b):
    return (a[1], a[0]) > (b[1], b[0])


# ------------------------------------------------------------------------------------------------ points
def jacobian_to_affine(pp, jac, g2):
    """Jacobian Montgomery limbs (as returned by the C ABI) -> affine ints, or None for the identity."""
    q = pp.fq.p
    v = pp.fq.decode(np.asarray(jac, dtype=np.uint64).reshape(-1, pp.fq.nl))
    if g2:
        z = (v[4], v[5])
        if z == (0, 0):
            return None
        zi = _f2_inv(z, q)
        zi2 = _f2_mul(zi, zi, q)
        return _f2_mul((v[0], v[1]), zi2, q), _f2_mul((v[2], v[3]), _f2_mul(zi2, zi, q), q)
    if v[2] == 0:
        return None
    zi = pow(v[2], q - 2, q)
    return v[0] * zi * zi % q, v[1] * zi * zi * zi % q


def affine_to_jacobian(pp, pt, g2):
    """affine ints (or None) -> Jacobian Montgomery limbs for the C ABI."""
    if pt is None:
        vals = [1, 0, 1, 0, 0, 0] if g2 else [1, 1, 0]
    elif g2:
        vals = [pt[0][0], pt[0][1], pt[1][0], pt[1][1], 1, 0]
    else:
        vals = [pt[0], pt[1], 1]
    return pp.fq.encode(vals).reshape(-1)


def point_to_bytes(pp, pt, g2, curve=None):
    curve = curve or pp.curve
    q = pp.fq.p
    n = (q.bit_length() + 7) // 8
    if curve in ZCASH_CURVES:
        if pt is None:
            b = bytearray(n * (2 if g2 else 1))
            b[0] = 0xC0
            return bytes(b)
        x, y = pt
        if g2:
            b = bytearray(x[1].to_bytes(n, "big") + x[0].to_bytes(n, "big"))
            larger = _f2_gt(y, ((-y[0]) % q, (-y[1]) % q))
        else:
            b = bytearray(x.to_bytes(n, "big"))
            larger = y > (-y) % q
        b[0] |= 0x80 | (0x20 if larger else 0)
        return bytes(b)
    if pt is None:
        b = bytearray(n * (2 if g2 else 1))
        b[-1] |= 1 << 6
        return bytes(b)
    x, y = pt
    if g2:
        b = bytearray(x[0].to_bytes(n, "little") + x[1].to_bytes(n, "little"))
        neg = ((-y[0]) % q, (-y[1]) % q)
        larger = _f2_gt(y, neg)
    else:
        b = bytearray(x.to_bytes(n, "little"))
        larger = y > (-y) % q
    if larger:
        b[-1] |= 1 << 7
    return bytes(b)


def _lift_x(q, bb, x, g2, want_larger):
    """(x, y) on y^2 = x^3 + b with the requested root, or ValueError"""
    if g2:
        if x[0] >= q or x[1] >= q:
            raise ValueError("coordinate not below the modulus")
        x3 = _f2_mul(_f2_mul(x, x, q), x, q)
        y = _f2_sqrt(((x3[0] + bb[0]) % q, (x3[1] + bb[1]) % q), q)
        if y is None:
            raise ValueError("x is not on the curve")
        neg = ((-y[0]) % q, (-y[1]) % q)
        if _f2_gt(y, neg) != want_larger:
            y = neg
        return x, y
    if x >= q:
        raise ValueError("coordinate not below the modulus")
    y = _fq_sqrt((x * x * x + bb) % q, q)
    if y is None:
        raise ValueError("x is not on the curve")
    if (y > (-y) % q) != want_larger:
        y = (-y) % q
    return x, y


def point_from_bytes(pp, data, g2, curve):
    """deserialize_compressed with validation: x below the modulus and on the curve (arkworks: InvalidData)."""
    q = pp.fq.p
    n = (q.bit_length() + 7) // 8
    size = n * (2 if g2 else 1)
    if len(data) != size:
        raise ValueError("expected %d bytes" % size)
    b = bytearray(data)
    bb = _curve_b(curve, g2)
    if curve in ZCASH_CURVES:
        flags = b[0] & 0xE0
        b[0] &= 0x1F
        if not flags & 0x80:
            raise ValueError("uncompressed encoding where a compressed one is expected")
        if flags & 0x40:
            if flags & 0x20 or any(b):
                raise ValueError("infinity flag with a non-zero x or a sort flag")
            return None
        if g2:
            x = (int.from_bytes(b[n:], "big"), int.from_bytes(b[:n], "big"))
        else:
            x = int.from_bytes(b, "big")
        return _lift_x(q, bb, x, g2, bool(flags & 0x20))
    flags = b[-1] & 0xC0
    b[-1] &= 0x3F
    if flags == 0xC0:
        raise ValueError("invalid flag combination")
    if flags & 0x40:
        if any(b):
            raise ValueError("infinity flag with a non-zero x")
        return None
    if g2:
        x = (int.from_bytes(b[:n], "little"), int.from_bytes(b[n:], "little"))
    else:
        x = int.from_bytes(b, "little")
    return _lift_x(q, bb, x, g2, bool(flags & 0x80))


def proof_to_bytes(pp, pi_a, pi_b, pi_c):
    """ark_groth16::Proof {a: G1, b: G2, c: G1}.serialize_compressed from one party's Jacobian outputs
    (128 bytes for BN254, 192 for BLS12-381)."""
    return (point_to_bytes(pp, jacobian_to_affine(pp, pi_a, False), False)
            + point_to_bytes(pp, jacobian_to_affine(pp, pi_b, True), True)
            + point_to_bytes(pp, jacobian_to_affine(pp, pi_c, False), False))


def proof_from_bytes(pp, data, curve):
    n = (pp.fq.p.bit_length() + 7) // 8
    if len(data) != 4 * n:
        raise ValueError("expected %d bytes" % (4 * n))
    return (point_from_bytes(pp, data[:n], False, curve), point_from_bytes(pp, data[n:3 * n], True, curve),
            point_from_bytes(pp, data[3 * n:], False, curve))


# ------------------------------------------------------------------------------------------------ Fr vectors
def fr_vec_to_bytes(pp, x_d, count, stream=None):
    """Vec<Fr>::serialize_compressed of a device vector: u64 length + canonical elements."""
    return count.to_bytes(8, "little") + api.fr_to_bytes(pp, x_d, count, stream)


def fr_vec_from_bytes(pp, data, stream=None):
    """Returns (device vector, count)."""
    if len(data) < 8:
        raise ValueError("missing length prefix")
    count = int.from_bytes(data[:8], "little")
    if len(data) != 8 + count * pp.fr.nbytes:
        raise ValueError("length prefix does not match the payload")
    return api.fr_from_bytes(pp, data[8:], stream), count


# ------------------------------------------------------------------------------------------------ point vectors
def points_from_bytes(pp, group, data, stream=None):
    """Vec<G::Affine> payload (without the length prefix) in compressed form -> device vector of affine Montgomery
    points (zk_points_decompress: batched square roots on the GPU).  Raises ZkError (InvalidData) on a bad element."""
    g2 = group == api.ZK_G2
    size = pp.fq.nbytes * (2 if g2 else 1)
    host = np.frombuffer(bytes(data), dtype=np.uint8)
    if host.size % size:
        raise ValueError("byte length is not a multiple of the compressed point size")
    count = host.size // size
    raw = api.DeviceBuffer.from_numpy(pp, host)
    out = api.DeviceBuffer(pp, max(count, 1) * 2 * size)
    pp._check(pp.lib.zk_points_decompress(pp.h, group, raw.ptr, count, out.ptr, stream))
    return out, count


def points_to_bytes(pp, group, pts_d, count, stream=None):
    g2 = group == api.ZK_G2
    size = pp.fq.nbytes * (2 if g2 else 1)
    out = api.DeviceBuffer(pp, max(count, 1) * size)
    pp._check(pp.lib.zk_points_compress(pp.h, group, api._ptr(pts_d), count, out.ptr, stream))
    pp.sync(stream)
    return out.to_numpy(dtype=np.uint8)[: count * size].tobytes()


# ------------------------------------------------------------------------------------------------ whole mpc-net frames
def point_vec_to_bytes(pp, group, pts_d, count, stream=None):
    """Vec<G::Affine>::serialize_compressed of a device vector as it crosses an mpc-net wire
    (mpc-net/src/ser_net.rs:24-25, 111-112): u64 little-endian length + the compressed points."""
    return count.to_bytes(8, "little") + points_to_bytes(pp, group, pts_d, count, stream)


def point_vec_from_bytes(pp, group, data, stream=None):
    """deserialize_compressed of a whole Vec<G::Affine> frame.  Returns (device vector of affine points, count); raises
    ValueError when the length prefix and the payload disagree, ZkError (InvalidData) on an invalid point."""
    if len(data) < 8:
        raise ValueError("missing length prefix")
    count = int.from_bytes(data[:8], "little")
    size = pp.fq.nbytes * (2 if group == api.ZK_G2 else 1)
    if len(data) != 8 + count * size:
        raise ValueError("length prefix does not match the payload")
    out, n = points_from_bytes(pp, group, data[8:], stream)
    return out, count
