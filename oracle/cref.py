"""ctypes front end of oracle/c/libzkref.so (the plain-C restatement; TEST INFRASTRUCTURE / CPU BASELINE ONLY)."""
import ctypes as C
import os
import subprocess

import numpy as np

from .params import CURVES

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "c", "libzkref.so")
_lib = None

NL = 4


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            subprocess.run(["make", "-C", os.path.join(_HERE, "c")], check=True)
        _lib = C.CDLL(_SO)
    return _lib


class FieldT(C.Structure):
    _fields_ = [("mod", C.c_uint64 * NL), ("n0inv", C.c_uint64), ("r1", C.c_uint64 * NL), ("r2", C.c_uint64 * NL)]


class DomainT(C.Structure):
    _fields_ = [("log_size", C.c_int), ("size", C.c_size_t)] + [(k, C.c_uint64 * NL) for k in
                                                                 ("gen", "gen_inv", "size_inv", "offset", "offset_inv")]


class PssT(C.Structure):
    _fields_ = [("F", FieldT), ("l", C.c_int), ("t", C.c_int), ("n", C.c_int), ("share", DomainT),
                ("secret", DomainT), ("secret2", DomainT)]


def _limbs(v):
    return (C.c_uint64 * NL)(*[(v >> (64 * i)) & ((1 << 64) - 1) for i in range(NL)])


class Field:
    """4-limb Montgomery field description + int <-> limb codecs."""

    def __init__(self, p):
        assert p.bit_length() <= 256
        self.p = p
        self.bits = p.bit_length()
        self.R = 1 << 256
        self.Rinv = pow(self.R, -1, p)
        self.ct = FieldT(_limbs(p), (-pow(p, -1, 1 << 64)) % (1 << 64), _limbs(self.R % p), _limbs(self.R * self.R % p))

    def mont(self, v):
        return _limbs(v % self.p * self.R % self.p)

    def enc(self, vals):
        out = np.empty((len(vals), NL), dtype=np.uint64)
        mask = (1 << 64) - 1
        for i, v in enumerate(vals):
            m = v % self.p * self.R % self.p
            for k in range(NL):
                out[i, k] = (m >> (64 * k)) & mask
        return out

    def dec(self, arr):
        arr = np.asarray(arr, dtype=np.uint64).reshape(-1, NL)
        out = []
        for row in arr:
            m = sum(int(row[k]) << (64 * k) for k in range(NL))
            out.append(m * self.Rinv % self.p)
        return out


def _domain(fld, d):
    return DomainT(d.log_size, d.size, fld.mont(d.group_gen), fld.mont(d.group_gen_inv), fld.mont(d.size_inv),
                   fld.mont(d.offset), fld.mont(d.offset_inv))


class CPss:
    """PackedSharingParams for the C side, built from the Python oracle's domain constants."""

    def __init__(self, curve_name, l):
        from .pss import PackedSharingParams
        self.curve = CURVES[curve_name]
        self.opp = PackedSharingParams(self.curve, l)
        self.fr = Field(self.curve.r)
        self.fq = Field(self.curve.q) if self.curve.q.bit_length() <= 256 else None
        o = self.opp
        self.ct = PssT(self.fr.ct, o.l, o.t, o.n, _domain(self.fr, o.share), _domain(self.fr, o.secret),
                       _domain(self.fr, o.secret2))
        self.l, self.t, self.n = o.l, o.t, o.n

    @staticmethod
    def _p(arr):
        return arr.ctypes.data_as(C.c_void_p)

    def fft1(self, vec, gen):
        a = self.fr.enc(vec)
        lib().zkref_fft1(C.byref(self.ct), self._p(a), C.c_size_t(len(vec)), self.fr.mont(gen))
        return self.fr.dec(a)

    def d_fft_arrays(self, shares, mbyl, gen, size_inv, g, rearrange, in_mask, out_mask, seed):
        """shares: uint64 array [n*mbyl][4] modified in place (gen, size_inv, g are ints)."""
        lib().zkref_d_fft(C.byref(self.ct), self._p(shares), C.c_size_t(mbyl), self.fr.mont(gen),
                          None if size_inv is None else self.fr.mont(size_inv), None if g is None else self.fr.mont(g),
                          int(rearrange), None if in_mask is None else self._p(in_mask),
                          None if out_mask is None else self._p(out_mask), C.c_uint64(seed), self.fr.bits)
        return shares

    def d_fft_arrays_mt(self, shares, mbyl, gen, size_inv, g, rearrange, in_mask, out_mask, seed, king_threads=64):
        """d_fft_arrays with the parties' local stages on n threads and the precomputed-matrix king split over
        `king_threads` threads (zkref_d_fft_mt; same shares): the full-size checks at m = 2^24."""
        lib().zkref_set_fast_king(1, max(1, king_threads))
        try:
            lib().zkref_d_fft_mt(C.byref(self.ct), self._p(shares), C.c_size_t(mbyl), self.fr.mont(gen),
                                 None if size_inv is None else self.fr.mont(size_inv), None if g is None else self.fr.mont(g),
                                 int(rearrange), None if in_mask is None else self._p(in_mask),
                                 None if out_mask is None else self._p(out_mask), C.c_uint64(seed), self.fr.bits)
        finally:
            lib().zkref_set_fast_king(0, 1)
        return shares

    def d_fft(self, shares, dom, rearrange, masks=None, seed=0, inverse=False, g=None):
        """list-of-lists front end mirroring oracle.dist.d_fft / d_ifft."""
        n, mbyl = len(shares), len(shares[0])
        a = self.fr.enc([v for s in shares for v in s])
        im = om = None
        if masks is not None:
            im = self.fr.enc([v for mk in masks for v in mk.in_mask])
            om = self.fr.enc([v for mk in masks for v in mk.out_mask])
        gen = dom.group_gen_inv if inverse else dom.group_gen
        self.d_fft_arrays(a, mbyl, gen, dom.size_inv if inverse else None, g, rearrange, im, om, seed)
        flat = self.fr.dec(a)
        return [flat[i * mbyl:(i + 1) * mbyl] for i in range(n)]

    def deg_red_arrays(self, x, ln, in_mask, out_mask, seed):
        lib().zkref_deg_red(C.byref(self.ct), self._p(x), C.c_size_t(ln), None if in_mask is None else self._p(in_mask),
                            None if out_mask is None else self._p(out_mask), C.c_uint64(seed), self.fr.bits)
        return x

    def deg_red(self, x, masks, seed):
        n, ln = len(x), len(x[0])
        a = self.fr.enc([v for s in x for v in s])
        im = om = None
        if masks is not None:
            im = self.fr.enc([v for mk in masks for v in mk.in_mask])
            om = self.fr.enc([v for mk in masks for v in mk.out_mask])
        self.deg_red_arrays(a, ln, im, om, seed)
        flat = self.fr.dec(a)
        return [flat[i * ln:(i + 1) * ln] for i in range(n)]

    def d_pp_arrays(self, num, den, ln, in_mask, out_mask, seed, threads=1):
        """dpp/mod.rs:15-87 (zkref_d_pp): num, den uint64 arrays [n*ln][4]; returns the shares [n*ln][4].  threads > 1
        splits the king's unpack / inversions / pack (same values); ZeroDivisionError on a zero denominator."""
        out = np.empty_like(num)
        L = lib()
        L.zkref_d_pp.restype = C.c_int
        L.zkref_set_fast_king(1 if threads > 1 else 0, max(1, threads))
        try:
            rc = L.zkref_d_pp(C.byref(self.ct), self._p(num), self._p(den), C.c_size_t(ln),
                              None if in_mask is None else self._p(in_mask),
                              None if out_mask is None else self._p(out_mask), C.c_uint64(seed), self.fr.bits, self._p(out))
        finally:
            L.zkref_set_fast_king(0, 1)
        if rc:
            raise ZeroDivisionError("d_pp: zero denominator")
        return out

    def d_pp(self, num, den, masks, seed, threads=1):
        """list-of-lists front end mirroring oracle.dist.d_pp."""
        n, ln = len(num), len(num[0])
        a, b = self.fr.enc([v for s in num for v in s]), self.fr.enc([v for s in den for v in s])
        im = om = None
        if masks is not None:
            im = self.fr.enc([v for mk in masks for v in mk.in_mask])
            om = self.fr.enc([v for mk in masks for v in mk.out_mask])
        flat = self.fr.dec(self.d_pp_arrays(a, b, ln, im, om, seed, threads))
        return [flat[i * ln:(i + 1) * ln] for i in range(n)]

    def mul_sub_arrays(self, a, b, c):
        out = np.empty_like(a)
        lib().zkref_mul_sub(C.byref(self.fr.ct), self._p(a), self._p(b), self._p(c), C.c_size_t(a.shape[0]), self._p(out))
        return out

    # ---- group side (4-limb base fields only: BN254) ----
    def msm_g1_arrays(self, bases, scalars, n, nthreads=1):
        """bases uint64 [n][8] affine Montgomery, scalars uint64 [n][4]; returns uint64[12] Jacobian."""
        out = np.zeros(12, dtype=np.uint64)
        lib().zkref_msm_g1(C.byref(self.fr.ct), C.byref(self.fq.ct), self.fr.bits, self._p(bases), self._p(scalars),
                           C.c_size_t(n), nthreads, self._p(out))
        return out

    def msm_g2_arrays(self, bases, scalars, n, nthreads=1):
        out = np.zeros(24, dtype=np.uint64)
        lib().zkref_msm_g2(C.byref(self.fr.ct), C.byref(self.fq.ct), self.fr.bits, self._p(bases), self._p(scalars),
                           C.c_size_t(n), nthreads, self._p(out))
        return out

    def msm_g1(self, pts, scalars, nthreads=1):
        b = self.fq.enc([c for p in pts for c in ((0, 0) if p is None else p)]).reshape(len(pts), 8)
        s = self.fr.enc(scalars)
        v = self.fq.dec(self.msm_g1_arrays(b, s, len(pts), nthreads))
        return (v[0], v[1], v[2])

    def msm_g2(self, pts, scalars, nthreads=1):
        flat = []
        for p in pts:
            flat += [0, 0, 0, 0] if p is None else [p[0][0], p[0][1], p[1][0], p[1][1]]
        b = self.fq.enc(flat).reshape(len(pts), 16)
        s = self.fr.enc(scalars)
        v = self.fq.dec(self.msm_g2_arrays(b, s, len(pts), nthreads))
        return ((v[0], v[1]), (v[2], v[3]), (v[4], v[5]))

    def doubling_chain_g1(self, p, n):
        """affine uint64 [n][8]: 2^i * p"""
        base = self.fq.enc(list(p)).reshape(-1)
        out = np.zeros((n, 8), dtype=np.uint64)
        lib().zkref_g1_doubling_chain(C.byref(self.fq.ct), self._p(base), C.c_size_t(n), self._p(out))
        return out



# ---------------------------------------------------------------------------------------------- six-limb build
# libzkref6.so = the same C source compiled with six 64-bit limbs per element (Makefile): G1 of BLS12-381 / BLS12-377 at
# sizes the Python oracle cannot reach.  Scalars and coordinates are Montgomery residues to the radix 2^384 there; the
# coordinate layout (12 x u32 little-endian) is the GPU's own.
_SO6 = os.path.join(_HERE, "c", "libzkref6.so")
_lib6 = None
NL6 = 6


def lib6():
    global _lib6
    if _lib6 is None:
        if not os.path.exists(_SO6):
            subprocess.run(["make", "-C", os.path.join(_HERE, "c")], check=True)
        _lib6 = C.CDLL(_SO6)
    return _lib6


class FieldT6(C.Structure):
    _fields_ = [("mod", C.c_uint64 * NL6), ("n0inv", C.c_uint64), ("r1", C.c_uint64 * NL6), ("r2", C.c_uint64 * NL6)]


class Field6:
    """six-limb Montgomery field description (radix 2^384) + codecs"""

    def __init__(self, p):
        assert p.bit_length() <= 384
        self.p, self.bits, self.R = p, p.bit_length(), 1 << 384
        self.Rinv = pow(self.R, -1, p)
        lim = lambda v: (C.c_uint64 * NL6)(*[(v >> (64 * i)) & ((1 << 64) - 1) for i in range(NL6)])
        self.ct = FieldT6(lim(p), (-pow(p, -1, 1 << 64)) % (1 << 64), lim(self.R % p), lim(self.R * self.R % p))

    def enc(self, vals):
        out = np.empty((len(vals), NL6), dtype=np.uint64)
        mask = (1 << 64) - 1
        for i, v in enumerate(vals):
            m = v % self.p * self.R % self.p
            for k in range(NL6):
                out[i, k] = (m >> (64 * k)) & mask
        return out

    def dec(self, arr):
        arr = np.asarray(arr, dtype=np.uint64).reshape(-1, NL6)
        return [sum(int(row[k]) << (64 * k) for k in range(NL6)) * self.Rinv % self.p for row in arr]


class CGroup6:
    """G1 of a curve with a 6-limb base field through libzkref6.so: arkworks' signed-digit Pippenger (zkref.c DEF_MSM)
    and the doubling chain the size tests build their bases with."""

    def __init__(self, curve_name):
        self.curve = CURVES[curve_name]
        self.fr, self.fq = Field6(self.curve.r), Field6(self.curve.q)

    @staticmethod
    def _p(arr):
        return arr.ctypes.data_as(C.c_void_p)

    def scalars_from_gpu_residues(self, arr4):
        """uint64 [n][4] Montgomery residues of the GPU's 4-limb scalar field (radix 2^256) -> the SAME field values as
        uint64 [n][6] residues to the radix 2^384: m6 = m4 * 2^128 mod r."""
        r = self.curve.r
        k = (1 << 128) % r
        a = np.asarray(arr4, dtype=np.uint64)
        out = np.zeros((a.shape[0], NL6), dtype=np.uint64)
        mask = (1 << 64) - 1
        for i in range(a.shape[0]):
            m = (int(a[i, 0]) | (int(a[i, 1]) << 64) | (int(a[i, 2]) << 128) | (int(a[i, 3]) << 192)) * k % r
            for j in range(4):
                out[i, j] = (m >> (64 * j)) & mask
        return out

    def msm_g1_arrays(self, bases, scalars6, n, nthreads=1):
        """bases uint64 [n][12] affine Montgomery (the GPU layout), scalars6 uint64 [n][6]; returns uint64[18] Jacobian."""
        out = np.zeros(18, dtype=np.uint64)
        lib6().zkref_msm_g1(C.byref(self.fr.ct), C.byref(self.fq.ct), self.fr.bits, self._p(bases), self._p(scalars6),
                            C.c_size_t(n), nthreads, self._p(out))
        return out

    def msm_g1(self, pts, scalars, nthreads=1):
        b = self.fq.enc([c for p in pts for c in ((0, 0) if p is None else p)]).reshape(len(pts), 12)
        v = self.fq.dec(self.msm_g1_arrays(b, self.fr.enc(scalars), len(pts), nthreads))
        return (v[0], v[1], v[2])

    def msm_g2_arrays(self, bases, scalars6, n, nthreads=1):
        """bases uint64 [n][24] affine Montgomery over Fq2 (the GPU layout), scalars6 uint64 [n][6]; returns uint64[36]."""
        out = np.zeros(36, dtype=np.uint64)
        lib6().zkref_msm_g2(C.byref(self.fr.ct), C.byref(self.fq.ct), self.fr.bits, self._p(bases), self._p(scalars6),
                            C.c_size_t(n), nthreads, self._p(out))
        return out

    def doubling_chain_g1(self, p, n):
        base = self.fq.enc(list(p)).reshape(-1)
        out = np.zeros((n, 12), dtype=np.uint64)
        lib6().zkref_g1_doubling_chain(C.byref(self.fq.ct), self._p(base), C.c_size_t(n), self._p(out))
        return out
