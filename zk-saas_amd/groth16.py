"""Host-side harness of the distributed Groth16 prover: mirror of groth16/examples/sha256.rs `main`
(setup -> deal shares and masks -> run the n parties -> collect (A, B, C)) on top of the C ABI.

What runs where:
  * host (Python ints, one-off per circuit): R1CS -> QAP evaluation vectors (groth16/src/qap.rs:42-89) and the
    discrete logs of the CRS for a seeded trapdoor (ark-groth16 generate_parameters with CircomReduction, in
    the exponent);
  * GPU: CRS points and PackedProvingKeyShare (zk_pss_det_pack on the discrete logs + zk_base_mul, because
    det_pack is linear: groth16/src/proving_key.rs:47-123), QAP::pss / pack_from_witness dealing
    (zk_bitrev + zk_pss_pack), and the whole prover (zk_groth16_prove).
"""
import ctypes as C

import numpy as np

from . import fields
from .api import DeviceBuffer, ZK_G1, ZK_G2, _ptr

G1_GEN = {
    "bn254": (1, 2),
    "bls12_381": (
        0x17F1D3A73197D7942695638C4FA9AC0FC3688C4F9774B905A14E3A3F171BAC586C55E83FF97A1AEFFB3AF00ADB22C6BB,
        0x08B3F481E3AAA0F1A09E30ED741D8AE4FCF5E095D5D00AF600DB18CB2C04B3EDD03CC744A2888AE40CAA232946C5E7E1),
}
G2_GEN = {
    "bn254": ((10857046999023057135944570762232829481370756359578518086990519993285655852781,
               11559732032986387107991004021392285783925812861821192530917403151452391805634),
              (8495653923123431417604973247489272438418190587263600148770280649306958101930,
               4082367875863433681332203403145435568316851327593401208105741076214120093531)),
    "bls12_381": (
        (0x024AA2B2F08F0A91260805272DC51051C6E47AD4FA403B02B4510B647AE3D1770BAC0326A805BBEFD48056C8C121BDB8,
         0x13E02B6052719F607DACD3A088274F65596BD0D09920B61AB5DA61BBDC7F5049334CF11213945D57E5AC7D055D042B7E),
        (0x0CE5D527727D6E118CC9CDC6DA2E351AADFD9BAA8CBDD3A76D429A695160D12C923AC9CC3BACA289E193548608B82801,
         0x0606C4A02EA734CC32ACD2B02BC28B99CB3E287E85A763AF267492AB572E99AB3F370D275CEC1DA1AAA9075FF05F79BE)),
}


class Masks(C.Structure):
    """zk_groth16_masks (include/zksaas.h)."""
    _fields_ = [("fft_in", C.c_void_p * 6), ("fft_out", C.c_void_p * 6), ("degred_in", C.c_void_p),
                ("degred_out", C.c_void_p), ("msm_in", C.c_void_p * 5), ("msm_out", C.c_void_p * 5)]


class CrsShare(C.Structure):
    """zk_crs_share (include/zksaas.h)."""
    _fields_ = [("s_d", C.c_void_p), ("h_d", C.c_void_p), ("v_d", C.c_void_p), ("w_d", C.c_void_p),
                ("u_d", C.c_void_p), ("len_a", C.c_size_t), ("len_w", C.c_size_t), ("len_u", C.c_size_t),
                ("a_query0", C.c_void_p), ("b_g1_query0", C.c_void_p), ("delta_g1", C.c_void_p),
                ("alpha_g1", C.c_void_p), ("beta_g1", C.c_void_p), ("b_g2_query0", C.c_void_p),
                ("delta_g2", C.c_void_p), ("beta_g2", C.c_void_p)]


# ------------------------------------------------------------------------------------------------ host math
def _batch_inverse(vals, p):
    pre, acc = [], 1
    for v in vals:
        pre.append(acc)
        acc = acc * v % p
    inv = pow(acc, p - 2, p)
    out = [0] * len(vals)
    for i in reversed(range(len(vals))):
        out[i] = inv * pre[i] % p
        inv = inv * vals[i] % p
    return out


def _root_of_unity(curve, log_size):
    p, g = fields.FR[curve], fields.FR_GENERATOR[curve]
    s, t = 0, p - 1
    while t % 2 == 0:
        t //= 2
        s += 1
    return pow(pow(g, t, p), 1 << (s - log_size), p)


def _ntt(vals, root, p):
    n = len(vals)
    a = list(vals)
    j = 0
    for i in range(1, n):
        bit = n >> 1
        while j & bit:
            j ^= bit
            bit >>= 1
        j |= bit
        if i < j:
            a[i], a[j] = a[j], a[i]
    length = 2
    while length <= n:
        wl = pow(root, n // length, p)
        half = length // 2
        tw = [1] * half
        for k in range(1, half):
            tw[k] = tw[k - 1] * wl % p
        for s in range(0, n, length):
            for k in range(half):
                u, v = a[s + k], a[s + k + half] * tw[k] % p
                a[s + k] = (u + v) % p
                a[s + k + half] = (u - v) % p
        length *= 2
    return a


class SetupScalars:
    """Discrete logs of the CRS for a given trapdoor (what `circuit_specific_setup` computes before the
    fixed-base multiplications): LibsnarkReduction::instance_map_with_evaluation + CircomReduction::h_query_scalars."""

    def __init__(self, curve, r1cs, alpha, beta, gamma, delta, tau):
        p = fields.FR[curve]
        self.curve, self.p = curve, p
        self.alpha, self.beta, self.gamma, self.delta, self.tau = alpha, beta, gamma, delta, tau
        ni, nc = r1cs.num_instance_variables, r1cs.num_constraints
        log_m = max(0, (nc + ni - 1).bit_length())
        m = 1 << log_m
        self.log_m, self.m = log_m, m
        w = _root_of_unity(curve, log_m)
        zt = (pow(tau, m, p) - 1) % p
        pw, den = [], []
        cur = 1
        for _ in range(m):
            pw.append(cur)
            den.append(m * (tau - cur) % p)
            cur = cur * w % p
        u = [zt * x % p * y % p for x, y in zip(pw, _batch_inverse(den, p))]   # Lagrange coefficients at tau
        nv = r1cs.num_variables
        a, b, c = [0] * nv, [0] * nv, [0] * nv
        for i in range(ni):
            a[i] = u[nc + i]
        for i in range(nc):
            ui = u[i]
            for co, j in r1cs.a[i]:
                a[j] = (a[j] + ui * co) % p
            for co, j in r1cs.b[i]:
                b[j] = (b[j] + ui * co) % p
            for co, j in r1cs.c[i]:
                c[j] = (c[j] + ui * co) % p
        ginv, dinv = pow(gamma, p - 2, p), pow(delta, p - 2, p)
        abc = [(beta * x + alpha * y + z) % p for x, y, z in zip(a, b, c)]
        self.a_query, self.b_query = a, b
        self.gamma_abc = [x * ginv % p for x in abc[:ni]]
        self.l_query = [x * dinv % p for x in abc[ni:]]
        # h_query_scalars(m - 1, tau, _, delta^-1): powers 0..2m-2 of tau / delta, ifft on the 2m domain, odd entries
        sc, t = [], dinv
        for _ in range(2 * m - 1):
            sc.append(t)
            t = t * tau % p
        sc.append(0)
        w2 = _root_of_unity(curve, log_m + 1)
        co = _ntt(sc, pow(w2, p - 2, p), p)
        n2inv = pow(2 * m, p - 2, p)
        self.h_query = [co[i] * n2inv % p for i in range(1, 2 * m, 2)]


def _affine_codec(pp, vals, g2):
    """flat coordinate ints -> uint64 rows"""
    per = 4 if g2 else 2
    return pp.fq.encode(vals).reshape(-1, per * pp.fq.nl)


def base_points(pp, group, scalars_d, count):
    """scalars (device Fr) -> affine multiples of the group generator (device)."""
    g2 = group == ZK_G2
    gen = G2_GEN[pp.curve] if g2 else G1_GEN[pp.curve]
    flat = [gen[0][0], gen[0][1], gen[1][0], gen[1][1]] if g2 else list(gen)
    base = _affine_codec(pp, flat, g2)
    out = DeviceBuffer(pp, count * base.shape[1] * 8)
    pp._check(pp.lib.zk_base_mul(pp.h, group, base.ctypes.data, _ptr(scalars_d), count, out.ptr, None))
    return out


def pack_points(pp, group, points_d, nchunks, points_per_chunk):
    """pss.rs:69-122 over curve points: [nchunks][points_per_chunk] affine -> [n][nchunks] affine shares."""
    width = (4 if group == ZK_G2 else 2) * pp.fq.nl * 8
    out = DeviceBuffer(pp, pp.n * nchunks * width)
    pp._check(pp.lib.zk_pss_pack_points(pp.h, group, _ptr(points_d), nchunks, points_per_chunk, out.ptr, None))
    return out


class Crs:
    """Device-resident PackedProvingKeyShare for all parties + the unpacked proving key (for the local prover)."""

    def __init__(self, pp, setup, keep_unpacked=False):
        self.pp, self.setup = pp, setup
        l = pp.l

        def packed(vals, group):
            vals = list(vals)
            if len(vals) % l:
                vals += [0] * (l - len(vals) % l)     # det_pack's resize(t) zero-pads a short chunk (pss.rs:78)
            nch = len(vals) // l
            sh = pp.det_pack(pp.upload_fr(vals), nch)   # [n][nch] discrete logs of the share points
            return base_points(pp, group, sh, pp.n * nch), nch

        self.s, self.len_a = packed(setup.a_query[1:], ZK_G1)
        self.h, _ = packed(setup.b_query[1:], ZK_G1)
        self.v, _ = packed(setup.b_query[1:], ZK_G2)
        self.w, self.len_w = packed(setup.l_query, ZK_G1)
        self.u, self.len_u = packed(setup.h_query, ZK_G1)
        singles1 = [setup.a_query[0], setup.b_query[0], setup.delta, setup.alpha, setup.beta]
        singles2 = [setup.b_query[0], setup.delta, setup.beta]
        nl = pp.fq.nl
        self.s1 = base_points(pp, ZK_G1, pp.upload_fr(singles1), 5).to_numpy().reshape(5, 2 * nl)
        self.s2 = base_points(pp, ZK_G2, pp.upload_fr(singles2), 3).to_numpy().reshape(3, 4 * nl)
        self.unpacked = None
        if keep_unpacked:
            q = lambda vals, grp: base_points(pp, grp, pp.upload_fr(vals), len(vals))
            self.unpacked = {"a_query": q(setup.a_query, ZK_G1), "b_g1_query": q(setup.b_query, ZK_G1),
                             "b_g2_query": q(setup.b_query, ZK_G2), "l_query": q(setup.l_query, ZK_G1),
                             "h_query": q(setup.h_query, ZK_G1)}
        self._make_ct()

    def precompute(self, pp=None):
        """Fixed-base tables for the five query vectors (zk_msm_precompute): for a service that proves many witnesses
        against this CRS.  16 x the size of the packed CRS shares in HBM.  Tables belong to a context: pass `pp` to
        build them for another context that proves against the same CRS buffers."""
        from . import api
        pp = pp or self.pp
        for buf, grp, ln in ((self.s, ZK_G1, self.len_a), (self.h, ZK_G1, self.len_a), (self.v, ZK_G2, self.len_a),
                             (self.w, ZK_G1, self.len_w), (self.u, ZK_G1, self.len_u)):
            api.msm_precompute(pp, grp, buf, pp.n * ln)
        return self

    def _make_ct(self):
        self.ct = CrsShare(self.s.ptr, self.h.ptr, self.v.ptr, self.w.ptr, self.u.ptr, self.len_a, self.len_w,
                           self.len_u, self.s1[0].ctypes.data, self.s1[1].ctypes.data, self.s1[2].ctypes.data,
                           self.s1[3].ctypes.data, self.s1[4].ctypes.data, self.s2[0].ctypes.data,
                           self.s2[1].ctypes.data, self.s2[2].ctypes.data)


class Witness:
    """QAP::pss (qap.rs:91-135) and pack_from_witness (sha256.rs:131-156) on the device."""

    def __init__(self, pp, curve, r1cs, w, seed, dev_r1cs=None):
        from .circom import DeviceR1cs
        dev_r1cs = dev_r1cs or DeviceR1cs(pp, r1cs)
        w_d = w if isinstance(w, DeviceBuffer) else pp.upload_fr(w)
        self.log_m = log_m = dev_r1cs.log_m
        m = 1 << log_m
        self.qap = []
        for k, d in enumerate(dev_r1cs.qap(w_d)):              # qap.rs:42-89 on the device
            pp._check(pp.lib.zk_bitrev(pp.h, d.ptr, log_m, None))
            self.qap.append(pp.pack(d, m // pp.l, seed + k, order=1))
        if isinstance(w, DeviceBuffer):
            w = pp.download_fr(w, r1cs.num_variables)
        ni = r1cs.num_instance_variables

        def deal(vals, sd):
            vals = list(vals)
            if len(vals) % pp.l:
                vals += [0] * (pp.l - len(vals) % pp.l)
            return pp.pack(pp.upload_fr(vals), len(vals) // pp.l, sd), len(vals) // pp.l

        self.a_share, self.len_a = deal(w[1:], seed + 3)
        self.ax_share, self.len_w = deal(w[ni:], seed + 4)


class ProofMasks:
    """All preprocessing material of one proof, dealt as groth16/examples/sha256.rs:226-291 does: six FftMask (three
    for the d_ifft with the coset shift w_2m and rearranged output, three for the d_fft), one DegRedMask and five
    MsmMask (A, B-in-G1, B-in-G2, C.w, C.u), all sampled by the library's dealers (zk_fft_mask_sample,
    zk_degred_mask_sample, zk_msm_mask_sample).  `.ct` is the zk_groth16_masks to pass to prove()."""

    def __init__(self, pp, log_m, seed):
        from . import api
        m = 1 << log_m
        w2m = _root_of_unity(pp.curve, log_m + 1)
        self.fft = [api.FftMask.sample(pp, k < 3, w2m if k < 3 else None, 1 if k < 3 else 0, log_m, seed + k)
                    for k in range(6)]
        self.degred = api.DegRedMask.sample(pp, m // pp.l, seed + 6)
        g1 = _affine_codec(pp, list(G1_GEN[pp.curve]), False)
        gg = G2_GEN[pp.curve]
        g2 = _affine_codec(pp, [gg[0][0], gg[0][1], gg[1][0], gg[1][1]], True)
        self.msm = [api.MsmMask.sample(pp, ZK_G2 if k == 2 else ZK_G1, g2 if k == 2 else g1, seed + 7 + k)
                    for k in range(5)]
        ct = Masks()
        for k in range(6):
            ct.fft_in[k], ct.fft_out[k] = self.fft[k].in_mask.ptr, self.fft[k].out_mask.ptr
        ct.degred_in, ct.degred_out = self.degred.in_mask.ptr, self.degred.out_mask.ptr
        for k in range(5):
            ct.msm_in[k], ct.msm_out[k] = self.msm[k].in_mask.ctypes.data, self.msm[k].out_mask.ctypes.data
        self.ct = ct


def verifying_key(pp, setup):
    """ark_groth16::VerifyingKey of a trapdoor setup as affine coordinate ints (verifier side, not the hot path):
    dict(alpha_g1, beta_g2, gamma_g2, delta_g2, gamma_abc_g1)."""
    nl = pp.fq.nl
    g1 = base_points(pp, ZK_G1, pp.upload_fr([setup.alpha] + list(setup.gamma_abc)), 1 + len(setup.gamma_abc))
    g2 = base_points(pp, ZK_G2, pp.upload_fr([setup.beta, setup.gamma, setup.delta]), 3)
    a1 = pp.fq.decode(g1.to_numpy().reshape(-1, nl))
    a2 = pp.fq.decode(g2.to_numpy().reshape(-1, nl))
    p1 = [(a1[2 * i], a1[2 * i + 1]) for i in range(len(a1) // 2)]
    p2 = [((a2[4 * i], a2[4 * i + 1]), (a2[4 * i + 2], a2[4 * i + 3])) for i in range(3)]
    return {"alpha_g1": p1[0], "gamma_abc_g1": p1[1:], "beta_g2": p2[0], "gamma_g2": p2[1], "delta_g2": p2[2]}


def prove(pp, crs, wit, r, s, masks=None, seed=0, stream=None):
    """dsha256 (sha256.rs:32-129) for all parties. Returns (pi_a [n][3nl], pi_b [n][6nl], pi_c [n][3nl]) Jacobian."""
    nl = pp.fq.nl
    pa = np.zeros((pp.n, 3 * nl), dtype=np.uint64)
    pb = np.zeros((pp.n, 6 * nl), dtype=np.uint64)
    pc = np.zeros((pp.n, 3 * nl), dtype=np.uint64)
    rr, ss = pp.fr.encode_one(r), pp.fr.encode_one(s)
    assert crs.len_a == wit.len_a and crs.len_w == wit.len_w
    if isinstance(masks, ProofMasks):
        masks = masks.ct
    pp._check(pp.lib.zk_groth16_prove(pp.h, C.byref(crs.ct), wit.qap[0].ptr, wit.qap[1].ptr, wit.qap[2].ptr,
                                      wit.a_share.ptr, wit.ax_share.ptr, rr.ctypes.data, ss.ctypes.data, wit.log_m,
                                      None if masks is None else C.byref(masks), seed, pa.ctypes.data,
                                      pb.ctypes.data, pc.ctypes.data, stream))
    return pa, pb, pc


class BatchInFlight:
    """A batch started with prove_batch_async (zk_groth16_prove_batch_async); wait() joins it and returns the list of
    (pi_a, pi_b, pi_c)."""

    def __init__(self, pp, handle, nb, keep):
        self.pp, self.handle, self.nb, self._keep = pp, handle, nb, keep

    def wait(self):
        pp, nb, nl = self.pp, self.nb, self.pp.fq.nl
        pa = np.zeros((nb, pp.n, 3 * nl), dtype=np.uint64)
        pb = np.zeros((nb, pp.n, 6 * nl), dtype=np.uint64)
        pc = np.zeros((nb, pp.n, 3 * nl), dtype=np.uint64)
        pp._check(pp.lib.zk_groth16_batch_wait(pp.h, self.handle, pa.ctypes.data, pb.ctypes.data, pc.ctypes.data))
        self._keep = None
        return [(pa[b], pb[b], pc[b]) for b in range(nb)]


def prove_batch(pp, crs, wits, rs, ss, masks=None, seed=0, stream=None):
    """zk_groth16_prove_batch: len(wits) proofs against one CRS in one pass (each witness with its own r, s and
    masks).  Returns a list of (pi_a, pi_b, pi_c) as prove() gives them."""
    return prove_batch_async(pp, crs, wits, rs, ss, masks=masks, seed=seed, stream=stream).wait()


def prove_batch_async(pp, crs, wits, rs, ss, masks=None, seed=0, stream=None):
    """zk_groth16_prove_batch_async: enqueue a batch and return at once; up to two batches may be in flight."""
    nb = len(wits)
    assert len(rs) == nb and len(ss) == nb and (masks is None or len(masks) == nb)
    rr = np.ascontiguousarray(np.stack([pp.fr.encode_one(v) for v in rs]))
    sv = np.ascontiguousarray(np.stack([pp.fr.encode_one(v) for v in ss]))
    for w in wits:
        assert crs.len_a == w.len_a and crs.len_w == w.len_w and w.log_m == wits[0].log_m
    arr = lambda f: (C.c_void_p * nb)(*[f(w) for w in wits])
    mk = None
    if masks is not None:
        mk = (Masks * nb)()
        for b, m in enumerate(masks):
            src = m.ct if isinstance(m, ProofMasks) else m
            C.memmove(C.byref(mk, b * C.sizeof(Masks)), C.byref(src), C.sizeof(Masks))
    h = C.c_int(-1)
    pp._check(pp.lib.zk_groth16_prove_batch_async(
        pp.h, C.byref(crs.ct), nb, arr(lambda w: w.qap[0].ptr), arr(lambda w: w.qap[1].ptr), arr(lambda w: w.qap[2].ptr),
        arr(lambda w: w.a_share.ptr), arr(lambda w: w.ax_share.ptr), rr.ctypes.data, sv.ctypes.data, wits[0].log_m,
        None if mk is None else C.cast(mk, C.c_void_p), seed, stream, C.byref(h)))
    return BatchInFlight(pp, h.value, nb, (crs, wits, masks, mk))


def reconstruct(pp, proof, parties=None, want_bytes=True, stream=None):
    """zk_groth16_reconstruct (groth16/examples/sha256.rs:375-377): (a, b, c) = pp.unpack2(shares)[0] over the parties'
    shares of one proof.  proof = (pi_a, pi_b, pi_c) rows of the listed parties (all n when parties is None).  Returns
    (affine, bytes): affine = uint64 array [8 * |Fq| limbs] = A (x, y) | B (x.c0, x.c1, y.c0, y.c1) | C (x, y) in
    Montgomery form; bytes = ark_groth16::Proof::serialize_compressed."""
    nl = pp.fq.nl
    pa, pb, pc = (np.ascontiguousarray(x, dtype=np.uint64) for x in proof)
    aff = np.zeros(8 * nl, dtype=np.uint64)
    raw = np.zeros(4 * pp.fq.nbytes, dtype=np.uint8)
    ids, count = None, pp.n
    if parties is not None:
        ids, count = (C.c_uint32 * len(parties))(*parties), len(parties)
    pp._check(pp.lib.zk_groth16_reconstruct(pp.h, pa.ctypes.data, pb.ctypes.data, pc.ctypes.data, ids, count,
                                            aff.ctypes.data, raw.ctypes.data if want_bytes else None, stream))
    return aff, (raw.tobytes() if want_bytes else None)


class ProofInFlight:
    """A proof started with prove_async (zk_groth16_prove_async); wait() joins it and returns the shares."""

    def __init__(self, pp, handle, keep):
        self.pp, self.handle, self._keep = pp, handle, keep

    def wait(self):
        pp = self.pp
        nl = pp.fq.nl
        pa = np.zeros((pp.n, 3 * nl), dtype=np.uint64)
        pb = np.zeros((pp.n, 6 * nl), dtype=np.uint64)
        pc = np.zeros((pp.n, 3 * nl), dtype=np.uint64)
        pp._check(pp.lib.zk_groth16_wait(pp.h, self.handle, pa.ctypes.data, pb.ctypes.data, pc.ctypes.data))
        self._keep = None
        return pa, pb, pc


def prove_async(pp, crs, wit, r, s, masks=None, seed=0, stream=None):
    """zk_groth16_prove_async: enqueue one proof and return at once; up to two may be in flight per context."""
    rr, ss = pp.fr.encode_one(r), pp.fr.encode_one(s)
    assert crs.len_a == wit.len_a and crs.len_w == wit.len_w
    keep = (crs, wit, masks, rr, ss)
    mk = masks.ct if isinstance(masks, ProofMasks) else masks
    h = C.c_int(-1)
    pp._check(pp.lib.zk_groth16_prove_async(pp.h, C.byref(crs.ct), wit.qap[0].ptr, wit.qap[1].ptr, wit.qap[2].ptr,
                                            wit.a_share.ptr, wit.ax_share.ptr, rr.ctypes.data, ss.ctypes.data,
                                            wit.log_m, None if mk is None else C.byref(mk), seed, stream, C.byref(h)))
    return ProofInFlight(pp, h.value, keep)


def libsnark_h(pp, qap, fft_masks, log_m, seed=0):
    """groth16/src/ext_wit.rs:14-102 for all parties (zk_libsnark_h): 3 d_ifft (coset shift g = F::GENERATOR,
    rearranged) -> 3 d_fft (rearranged) -> (a*b - c) / Z(g) -> d_ifft with g^-1.  qap: three device buffers [n][m/l];
    fft_masks: seven api.FftMask (or FftMask.zero()).  Returns the h coefficient shares [n][m/l]."""
    cnt = pp.n * ((1 << log_m) // pp.l)
    h = pp.alloc_fr(cnt)
    mi = (C.c_void_p * 7)(*[_ptr(m.in_mask) for m in fft_masks])
    mo = (C.c_void_p * 7)(*[_ptr(m.out_mask) for m in fft_masks])
    pp._check(pp.lib.zk_libsnark_h(pp.h, _ptr(qap[0]), _ptr(qap[1]), _ptr(qap[2]), log_m, mi, mo, seed, h.ptr, None))
    return h


def crs_from_proving_key(pp, pk, singles_from):
    """PackedProvingKeyShare::pack_from_arkworks_proving_key (proving_key.rs:47-123) WITHOUT a trapdoor: `pk` maps
    "a_query", "b_g1_query", "b_g2_query", "l_query", "h_query" to (device affine buffer, count); every l-chunk of
    a_query[1..], h_query, l_query, b_g1_query[1..], b_g2_query[1..] is det_pack'ed over the group elements on the GPU.
    `singles_from`: a Crs whose single elements (a_query[0], delta, alpha, ...) are reused."""
    l = pp.l
    nl = pp.fq.nl

    def packed(name, group, skip_first):
        buf, count = pk[name]
        width = (4 if group == ZK_G2 else 2) * nl
        arr = buf.to_numpy().reshape(count, width)[1 if skip_first else 0:]
        if arr.shape[0] % l:                                   # det_pack zero-pads a short last chunk (pss.rs:78)
            arr = np.concatenate([arr, np.zeros((l - arr.shape[0] % l, width), dtype=np.uint64)])
        nch = arr.shape[0] // l
        return pack_points(pp, group, DeviceBuffer.from_numpy(pp, arr), nch, l), nch

    crs = Crs.__new__(Crs)
    crs.pp, crs.setup, crs.unpacked = pp, None, None
    crs.s, crs.len_a = packed("a_query", ZK_G1, True)
    crs.h, _ = packed("b_g1_query", ZK_G1, True)
    crs.v, _ = packed("b_g2_query", ZK_G2, True)
    crs.w, crs.len_w = packed("l_query", ZK_G1, False)
    crs.u, crs.len_u = packed("h_query", ZK_G1, False)
    crs.s1, crs.s2 = singles_from.s1, singles_from.s2
    crs._make_ct()
    return crs
