"""Synthetic Groth16 instances at scale, built entirely on the device (SURVEY.md 8d "C5": BLS12-381, m = 2^24).

The reference's own large benchmark makes its proving key from cheap point chains instead of a real setup
(groth16/examples/local_groth_bench.rs:21-52); the same idea here: every CRS query element is a pseudo-random multiple
d_j * G of the generator, its packed shares are (det_pack of the d_j) * G (pss.rs:69-87 is linear, so this is the share
proving_key.rs:72-86 would produce), the R1CS has two-term / one-term rows over a pseudo-random assignment
(under the circom reduction c = a * b, qap.rs:66-70, every assignment satisfies it).  Such an instance has no
trapdoor, so its proofs cannot be verified by pairing; what it supports is timing at full size and the
size-independent properties the tests use (distributed == local prover, independence of the share randomness).

No Python big-integer loops over the vectors: numpy limbs -> zk_fr_from_bytes -> device kernels.
"""
import types

import numpy as np

from . import api, circom, fields
from . import groth16 as zg
from .api import ZK_G1, ZK_G2, DeviceBuffer


def rand_fr_device(pp, count, seed, head=None, pad=0):
    """`count` pseudo-random field elements (< 2^(bits-1), hence canonical) as a Montgomery device vector, followed by
    `pad` zero elements; `head`: ints overriding the first elements."""
    nl = pp.fr.nl
    rng = np.random.default_rng(seed)
    limbs = rng.integers(0, 1 << 63, size=(count + pad, nl), dtype=np.uint64) << np.uint64(1)
    limbs |= rng.integers(0, 2, size=(count + pad, nl), dtype=np.uint64)
    top_bits = fields.FR[pp.curve].bit_length() - 1 - 64 * (nl - 1)
    limbs[:, nl - 1] &= np.uint64((1 << top_bits) - 1)
    if pad:
        limbs[count:] = 0
    for i, v in enumerate(head or []):
        for k in range(nl):
            limbs[i, k] = (v >> (64 * k)) & ((1 << 64) - 1)
    return api.fr_from_bytes(pp, limbs)


class SyntheticInstance:
    """num_variables = m, one public input besides the constant (ni = 2), nc = m - 2 constraints
    A_i = w[i] + 3 w[7i+1],  B_i = w[i+1]  (indices mod m)."""

    def __init__(self, pp, log_m, seed=1, parties=None):
        """parties = (first, k): build only the CRS rows and witness shares of parties [first, first + k) (one rank of a
        multi-GPU run; the instance itself -- R1CS, assignment, discrete logs -- is the same on every rank)."""
        self.pp, self.log_m = pp, log_m
        self.first, self.k = parties if parties is not None else (0, pp.n)
        m = 1 << log_m
        self.m, self.ni, self.nc, self.nv = m, 2, m - 2, m
        l, n = pp.l, pp.n
        # assignment (w[0] = 1) with l zero elements behind it so that chunked views stay in bounds
        self.w = rand_fr_device(pp, m, seed, head=[1], pad=l)
        idx = np.arange(self.nc, dtype=np.uint64)
        cols_a = np.empty(2 * self.nc, dtype=np.uint32)
        cols_a[0::2] = (idx % m).astype(np.uint32)
        cols_a[1::2] = ((7 * idx + 1) % m).astype(np.uint32)
        one, three = pp.fr.encode_one(1), pp.fr.encode_one(3)
        vals_a = np.empty((2 * self.nc, pp.fr.nl), dtype=np.uint64)
        vals_a[0::2], vals_a[1::2] = one, three
        self.r1cs = circom.DeviceR1cs.from_csr(
            pp, self.nc, self.ni, self.nv,
            (np.arange(0, 2 * self.nc + 1, 2, dtype=np.uint32), cols_a, vals_a),
            (np.arange(self.nc + 1, dtype=np.uint32), ((idx + 1) % m).astype(np.uint32),
             np.tile(one, (self.nc, 1))))
        del vals_a, cols_a
        self.len_a = (self.nv - 1 + l - 1) // l
        self.len_w = (self.nv - self.ni + l - 1) // l
        self.len_u = m // l
        eb = pp.fr.nbytes
        # discrete logs of the query elements (index 0 of a_query / b_query is a single element, proving_key.rs:60-70)
        self.logs = {
            "a": rand_fr_device(pp, l * self.len_a, seed + 10), "b": rand_fr_device(pp, l * self.len_a, seed + 11),
            "l": rand_fr_device(pp, l * self.len_w, seed + 12), "h": rand_fr_device(pp, l * self.len_u, seed + 13),
        }
        singles = [int(x) for x in np.random.default_rng(seed + 20).integers(2, 1 << 62, size=6)]
        self.single_logs = dict(zip(("a0", "b0", "delta", "alpha", "beta"), singles))

        def packed(logs, nch, group):
            sh = pp.det_pack(logs, nch)                                  # [n][nch] discrete logs of the share points
            mine = sh.view(self.first * nch * eb, self.k * nch * eb)
            pts = zg.base_points(pp, group, mine, self.k * nch)
            del mine
            sh.free()
            return pts
        self.s = packed(self.logs["a"], self.len_a, ZK_G1)
        self.h = packed(self.logs["b"], self.len_a, ZK_G1)
        self.v = packed(self.logs["b"], self.len_a, ZK_G2)
        self.wq = packed(self.logs["l"], self.len_w, ZK_G1)
        self.u = packed(self.logs["h"], self.len_u, ZK_G1)
        nl = pp.fq.nl
        sl = self.single_logs
        self.s1 = zg.base_points(pp, ZK_G1, pp.upload_fr([sl["a0"], sl["b0"], sl["delta"], sl["alpha"], sl["beta"]]),
                                 5).to_numpy().reshape(5, 2 * nl)
        self.s2 = zg.base_points(pp, ZK_G2, pp.upload_fr([sl["b0"], sl["delta"], sl["beta"]]), 3).to_numpy().reshape(
            3, 4 * nl)
        ct = zg.CrsShare(self.s.ptr, self.h.ptr, self.v.ptr, self.wq.ptr, self.u.ptr, self.len_a, self.len_w, self.len_u,
                         self.s1[0].ctypes.data, self.s1[1].ctypes.data, self.s1[2].ctypes.data,
                         self.s1[3].ctypes.data, self.s1[4].ctypes.data, self.s2[0].ctypes.data,
                         self.s2[1].ctypes.data, self.s2[2].ctypes.data)
        self.crs = types.SimpleNamespace(ct=ct, len_a=self.len_a, len_w=self.len_w, len_u=self.len_u)
        _ = eb

    def witness(self, seed):
        """QAP::pss (qap.rs:91-135) + pack_from_witness (sha256.rs:131-156) with share randomness `seed`."""
        pp, l, eb = self.pp, self.pp.l, self.pp.fr.nbytes
        qap = []
        sl = lambda buf, ln: buf if self.k == pp.n else buf.view(self.first * ln * eb, self.k * ln * eb)
        for k, d in enumerate(self.r1cs.qap(self.w)):
            pp._check(pp.lib.zk_bitrev(pp.h, d.ptr, self.log_m, None))
            qap.append(sl(pp.pack(d, self.m // l, seed + k, order=1), self.m // l))
            d.free()
        a_share = sl(pp.pack(self.w.view(1 * eb), self.len_a, seed + 3), self.len_a)
        ax_share = sl(pp.pack(self.w.view(self.ni * eb), self.len_w, seed + 4), self.len_w)
        return types.SimpleNamespace(qap=qap, a_share=a_share, ax_share=ax_share, log_m=self.log_m, len_a=self.len_a,
                                     len_w=self.len_w)

    def unpacked_points(self, name, group):
        """The public query elements d_j * G (affine, device) for the local prover / cross-checks."""
        logs = self.logs[name]
        return zg.base_points(self.pp, group, logs, logs.nbytes // self.pp.fr.nbytes)
