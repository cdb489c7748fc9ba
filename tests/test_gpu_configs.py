"""Every BASELINE.json config under `pytest -m gpu` against the oracle.

  configs[1]  d_fft 2^20 BN254            -> test_gpu_fullsize.py::test_d_fft_2_20_shares_equal_c_oracle
  configs[2]  d_msm 2^20 G1 per party     -> test_c3_d_msm_2_20_points_per_party_equals_c_oracle (here)
  configs[3]  SHA-256 circuit, BN254      -> test_c4_* (here): masks ON, vectors padded to the reference's sizes,
                                             proof VERIFIED BY PAIRING (the reference's own end-to-end assertion,
                                             groth16/examples/sha256.rs:389-415)
  configs[4]  BLS12-381 2^24 composed     -> test_c5_* (here): oracle-checked small composed proof, d_fft at 2^20
                                             against the C oracle (Fr has 4 limbs), and the full 2^24 - 2 constraint
                                             flow with its size-independent properties.
"""
import ctypes as C
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import zksaas_amd as zk
from zksaas_amd import groth16 as zg
from zksaas_amd import sha256_circuit as sc
from zksaas_amd import wire
from zksaas_amd.api import ZK_G1, ZK_G2, msm
from oracle import groth16 as og
from oracle import pairing as op
from oracle import ser
from oracle.cref import CPss
from oracle.curve import g1, g2, GroupOps
from oracle.field import Domain
from oracle.params import BLS12_381, BN254, CURVES
from oracle.prng import rand_fp

from gpu_util import ctx, opp, dec_jacobian, enc_affine
from test_pairing import _small_r1cs_mod

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _rand_fr_array(count, seed, top_bits=60):
    rng = np.random.default_rng(seed)
    a = rng.integers(0, 1 << 62, size=(count, 4), dtype=np.uint64)
    a[:, 3] &= np.uint64((1 << top_bits) - 1)
    return a


def _reconstructed(pp, o, curve, proof):
    """sha256.rs:375-377: unpack2 of the n parties' shares, slot 0; returns affine (A, B, C)."""
    c = CURVES[curve]
    G1, G2 = g1(c), g2(c)
    o1, o2 = GroupOps(G1), GroupOps(G2)
    pa, pb, pc = proof
    A = o.unpack2([dec_jacobian(pp, pa[i]) for i in range(o.n)], o1)[0]
    B = o.unpack2([dec_jacobian(pp, pb[i], True) for i in range(o.n)], o2)[0]
    Cc = o.unpack2([dec_jacobian(pp, pc[i]) for i in range(o.n)], o1)[0]
    return G1.to_affine(A), G2.to_affine(B), G1.to_affine(Cc)


def _oracle_vk(vk):
    return op.VerifyingKey(vk["alpha_g1"], vk["beta_g2"], vk["gamma_g2"], vk["delta_g2"], vk["gamma_abc_g1"])


# ------------------------------------------------------------------------------------------------ configs[3] (C4)
@pytest.mark.parametrize("tables", [False, True])
def test_c4_sha256_proof_with_all_masks_verifies_by_pairing(tables):
    """The reference's run of the SHA-256 fixture (sha256.rs): 29 823 wires (a_share / ax_share 14 911 per party,
    h_share 16 384), all twelve masks sampled, proof reconstructed with unpack2 and checked with
    Groth16::verify_proof against the verifying key and the public output of sha256.rs:392-393."""
    P = BN254.r
    r1, w = sc.build(1, 2, P, pad_wires=sc.REFERENCE_WIRES)
    assert r1.num_variables == 29823 and w[1] == 72587776472194017031617589674261467945970986113287823188107011979
    pp, o = zk.PackedSharingParams("bn254", 2), opp("bn254", 2)
    td = [rand_fp(42, i, P) for i in range(5)]
    setup = zg.SetupScalars("bn254", r1, *td)
    crs = zg.Crs(pp, setup)
    wit = zg.Witness(pp, "bn254", r1, w, seed=7)
    assert (setup.log_m, crs.len_a, crs.len_w, crs.len_u) == (15, 14911, 14911, 16384)
    if tables:
        crs.precompute()
    masks = zg.ProofMasks(pp, setup.log_m, seed=500)
    r, s = rand_fp(43, 0, P), rand_fp(43, 1, P)
    proof = zg.prove(pp, crs, wit, r, s, masks=masks, seed=13)
    # out-masks differ per party: the shares are NOT all equal, only their reconstruction is the proof
    assert not np.array_equal(proof[0][0], proof[0][1])
    A, B, Cc = _reconstructed(pp, o, "bn254", proof)
    vk = _oracle_vk(zg.verifying_key(pp, setup))
    assert op.verify_proof(BN254, vk, (A, B, Cc), [w[1]], g1(BN254))
    assert not op.verify_proof(BN254, vk, (A, B, Cc), [(w[1] + 1) % P], g1(BN254))
    # and the masked proof is the same group elements as the unmasked one and as the trapdoor closed form
    plain = _reconstructed(pp, o, "bn254", zg.prove(pp, crs, wit, r, s, seed=14))
    assert plain == (A, B, Cc)
    R = og.R1CS(2, r1.num_witness_variables, r1.a, r1.b, r1.c)
    okey = og.setup_scalars(BN254, R, og.Trapdoor(*td))
    sa, sb, sc_ = og.prove_scalars(BN254, R, okey, w, r, s)
    G1, G2 = g1(BN254), g2(BN254)
    assert A == G1.to_affine(G1.mul(G1.from_affine(BN254.g1), sa))
    assert B == G2.to_affine(G2.mul(G2.from_affine(BN254.g2), sb))
    assert Cc == G1.to_affine(G1.mul(G1.from_affine(BN254.g1), sc_))
    pp.close()


def test_reference_fixture_points_through_the_gpu_msm():
    """(r - 1) * P == -P for every G1 / G2 point of the reference's verification_key.json, computed by zk_msm."""
    with open(os.path.join(ROOT, "tests", "golden", "verification_key_bn254.json")) as fh:
        d = json.load(fh)
    pp = ctx("bn254", 2)
    G1, G2 = g1(BN254), g2(BN254)
    p1 = [(int(v[0]), int(v[1])) for v in [d["vk_alpha_1"]] + d["IC"]]
    p2 = [((int(v[0][0]), int(v[0][1])), (int(v[1][0]), int(v[1][1]))) for v in
          (d["vk_beta_2"], d["vk_gamma_2"], d["vk_delta_2"])]
    rm1 = zk.DeviceBuffer.from_numpy(pp, pp.fr.encode([BN254.r - 1]))
    for pt in p1:
        got = dec_jacobian(pp, msm(pp, ZK_G1, zk.DeviceBuffer.from_numpy(pp, enc_affine(pp, [pt])), rm1, 1))
        assert G1.to_affine(got) == (pt[0], (-pt[1]) % BN254.q)
    for pt in p2:
        got = dec_jacobian(pp, msm(pp, ZK_G2, zk.DeviceBuffer.from_numpy(pp, enc_affine(pp, [pt], True)), rm1, 1), True)
        assert G2.to_affine(got) == (pt[0], G2.F.neg(pt[1]))
    # all of them at once with scalars that sum the points to a known combination: sum_i i * P_i
    sc_ = list(range(1, len(p1) + 1))
    got = dec_jacobian(pp, msm(pp, ZK_G1, zk.DeviceBuffer.from_numpy(pp, enc_affine(pp, p1)),
                               zk.DeviceBuffer.from_numpy(pp, pp.fr.encode(sc_)), len(p1)))
    assert G1.eq(got, G1.msm_naive(p1, sc_))


# ------------------------------------------------------------------------------------------------ configs[2] (C3)
def test_c3_d_msm_2_20_points_per_party_equals_c_oracle():
    """d_msm with 2^20 G1 points PER PARTY (8 parties fused into one 2^23-point Pippenger on the GPU) against the
    C oracle's per-party G::msm + the king's unpack2 + sum (dmsm/mod.rs:73-92)."""
    pp = ctx("bn254", 2)
    cp = CPss("bn254", 2)
    ln = 1 << 20
    chain = cp.doubling_chain_g1(BN254.g1, pp.n * ln)
    scalars = _rand_fr_array(pp.n * ln, 7)
    out = zk.d_msm(pp, ZK_G1, zk.DeviceBuffer.from_numpy(pp, chain), zk.DeviceBuffer.from_numpy(pp, scalars), ln)
    G = g1(BN254)
    threads = min(32, os.cpu_count() or 1)
    parts = [dec_jacobian(pp, cp.msm_g1_arrays(chain[p * ln:(p + 1) * ln], scalars[p * ln:(p + 1) * ln], ln, threads))
             for p in range(pp.n)]
    want = G.sum(cp.opp.unpack2(parts, GroupOps(G)))
    for p in range(pp.n):
        assert G.eq(dec_jacobian(pp, out[p]), want)
    # one party's slice alone through the multi-GPU building block: coef_p * msm_p
    nl = pp.fq.nl
    o1 = np.zeros(3 * nl, dtype=np.uint64)
    dchain, dscal = zk.DeviceBuffer.from_numpy(pp, chain[3 * ln:4 * ln]), zk.DeviceBuffer.from_numpy(pp, scalars[3 * ln:4 * ln])
    pp._check(pp.lib.zk_d_msm_local(pp.h, ZK_G1, dchain.ptr, dscal.ptr, ln, 3, 1, None, o1.ctypes.data, None))
    o = cp.opp
    unit = [G.identity] * o.n
    unit[3] = parts[3]
    assert G.eq(dec_jacobian(pp, o1), G.sum(o.unpack2(unit, GroupOps(G))))


# ------------------------------------------------------------------------------------------------ configs[4] (C5)
@pytest.mark.parametrize("masked", [False, True])
def test_c5_bls12_381_composed_proof_equals_oracle_and_verifies(masked):
    """BLS12-381, d_fft + d_msm + deg_red composed (zk_groth16_prove) on a small R1CS: the reconstructed proof equals
    the oracle's arkworks-style local prover (og.create_proof_local) point for point and byte for byte, equals the
    trapdoor closed form, and verifies by pairing."""
    c = BLS12_381
    P = c.r
    r1, w = _small_r1cs_mod(P)
    pp, o = ctx("bls12_381", 2), opp("bls12_381", 2)
    td = [rand_fp(52, i, P) for i in range(5)]
    setup = zg.SetupScalars("bls12_381", r1, *td)
    okey = og.setup_scalars(c, r1, og.Trapdoor(*td))
    assert setup.a_query == okey.a_query and setup.h_query == okey.h_query and setup.l_query == okey.l_query
    crs = zg.Crs(pp, setup)
    wit = zg.Witness(pp, "bls12_381", r1, w, seed=5)
    r, s = rand_fp(53, 0, P), rand_fp(53, 1, P)
    masks = zg.ProofMasks(pp, setup.log_m, seed=700) if masked else None
    A, B, Cc = _reconstructed(pp, o, "bls12_381", zg.prove(pp, crs, wit, r, s, masks=masks, seed=9))
    G1, G2 = g1(c), g2(c)
    pk = og.proving_key_points(okey, G1, G2)
    lA, lB, lC = og.create_proof_local(c, r1, pk, G1, G2, w, r, s)
    assert (A, B, Cc) == (G1.to_affine(lA), G2.to_affine(lB), G1.to_affine(lC))
    assert ser.proof_compressed(A, B, Cc, c.q) == ser.proof_compressed(G1.to_affine(lA), G2.to_affine(lB),
                                                                       G1.to_affine(lC), c.q)
    sa, sb, sc_ = og.prove_scalars(c, r1, okey, w, r, s)
    assert A == G1.to_affine(G1.mul(G1.from_affine(c.g1), sa)) and B == G2.to_affine(G2.mul(G2.from_affine(c.g2), sb))
    assert Cc == G1.to_affine(G1.mul(G1.from_affine(c.g1), sc_))
    vk = _oracle_vk(zg.verifying_key(pp, setup))
    assert op.verify_proof(c, vk, (A, B, Cc), [w[1]], G1)
    assert not op.verify_proof(c, vk, (A, B, Cc), [(w[1] + 1) % P], G1)


def test_c5_bls12_381_composed_proof_at_2_12_is_exact():
    """The composed prover on BLS12-381 at a domain of 2^12 (4094 constraints + 2 inputs), all twelve masks on: the proof
    reconstructed from the n parties' shares equals the trapdoor closed form (og.prove_scalars: the discrete logs of A, B,
    C from the QAP evaluated at tau -- exact, not a property) point for point and verifies by pairing; the setup scalars of
    the host mirror equal the oracle's."""
    c = BLS12_381
    P = c.r
    r1, w = _small_r1cs_mod(P, nc=4094)
    pp, o = ctx("bls12_381", 2), opp("bls12_381", 2)
    td = [rand_fp(152, i, P) for i in range(5)]
    setup = zg.SetupScalars("bls12_381", r1, *td)
    assert setup.log_m == 12
    okey = og.setup_scalars(c, r1, og.Trapdoor(*td))
    assert setup.a_query == okey.a_query and setup.h_query == okey.h_query and setup.l_query == okey.l_query
    crs = zg.Crs(pp, setup)
    wit = zg.Witness(pp, "bls12_381", r1, w, seed=15)
    r, s = rand_fp(153, 0, P), rand_fp(153, 1, P)
    masks = zg.ProofMasks(pp, setup.log_m, seed=1700)
    A, B, Cc = _reconstructed(pp, o, "bls12_381", zg.prove(pp, crs, wit, r, s, masks=masks, seed=19))
    G1, G2 = g1(c), g2(c)
    sa, sb, sc_ = og.prove_scalars(c, r1, okey, w, r, s)
    assert A == G1.to_affine(G1.mul(G1.from_affine(c.g1), sa)) and B == G2.to_affine(G2.mul(G2.from_affine(c.g2), sb))
    assert Cc == G1.to_affine(G1.mul(G1.from_affine(c.g1), sc_))
    vk = _oracle_vk(zg.verifying_key(pp, setup))
    assert op.verify_proof(c, vk, (A, B, Cc), [w[1]], G1)


@pytest.mark.parametrize("inverse", [False, True])
def test_c5_bls12_381_d_fft_2_20_shares_equal_c_oracle(inverse):
    """d_fft / d_ifft on the config-5 scalar field at m = 2^20 (16 M-element share matrix): every output share
    equals the C restatement's (BLS12-381 Fr has four 64-bit limbs, so the C oracle covers it)."""
    pp = ctx("bls12_381", 2)
    cp = CPss("bls12_381", 2)
    log_m = 20
    m = 1 << log_m
    dom = Domain(BLS12_381, m)
    shares = _rand_fr_array(pp.n * (m // 2), 15)
    g = Domain(BLS12_381, 2 * m).element(1) if inverse else None
    buf = zk.DeviceBuffer.from_numpy(pp, shares)
    if inverse:
        zk.d_ifft(pp, buf, zk.FftMask.zero(), True, log_m, g=g, seed=78)
    else:
        zk.d_fft(pp, buf, zk.FftMask.zero(), False, log_m, seed=78)
    want = shares.copy()
    cp.d_fft_arrays(want, m // 2, dom.group_gen_inv if inverse else dom.group_gen, dom.size_inv if inverse else None, g,
                    inverse, None, None, 78)
    assert np.array_equal(buf.to_numpy().reshape(-1, 4), want)


def test_c5_bls12_381_full_size_2_24():
    """BASELINE configs[4] at full size: BLS12-381, 2^24 - 2 constraints, n = 8 parties on this one GPU,
    d_fft + d_msm + deg_red composed and d_pp at the same size, through the size-independent properties:
      * proof-byte stability: a second dealing of the same witness (fresh share randomness) gives the same bytes;
      * distributed == local: zk_groth16_assemble over five PLAIN zk_msm's of the public query elements;
      * d_pp telescopes: num_i = x_(i+1), den_i = x_i  =>  prefix product_i * x_0 == x_(i+1);
      * every party holds the same (A, B, C) without out-masks."""
    from zksaas_amd import synthetic
    log_m = int(os.environ.get("ZK_C5_LOG_M", "24"))
    pp = zk.PackedSharingParams("bls12_381", 2)
    try:
        inst = synthetic.SyntheticInstance(pp, log_m, seed=1)
        wit = inst.witness(seed=100)
        r, s = 0x1234567890ABCDEF1234567890ABCDEF, 0xFEDCBA0987654321FEDCBA0987654321
        proof = zg.prove(pp, inst.crs, wit, r, s, seed=7)

        def norm(pf):
            return (wire.jacobian_to_affine(pp, pf[0][0], False), wire.jacobian_to_affine(pp, pf[1][0], True),
                    wire.jacobian_to_affine(pp, pf[2][0], False))
        ref = norm(proof)
        assert all(np.array_equal(proof[k][0], proof[k][p]) for k in range(3) for p in range(pp.n))
        blob = wire.proof_to_bytes(pp, proof[0][0], proof[1][0], proof[2][0])
        wit2 = inst.witness(seed=900)
        p2 = zg.prove(pp, inst.crs, wit2, r, s, seed=8)
        assert wire.proof_to_bytes(pp, p2[0][0], p2[1][0], p2[2][0]) == blob
        del wit2, p2
        m, l, eb = inst.m, pp.l, pp.fr.nbytes
        hsh = pp.alloc_fr(pp.n * (m // l))
        pp._check(pp.lib.zk_circom_h(pp.h, wit.qap[0].ptr, wit.qap[1].ptr, wit.qap[2].ptr, log_m, None, 7, hsh.ptr, None))
        h_pub = pp.unpack(hsh, m // l)
        sums = []
        for name, group, scal, count in (("a", ZK_G1, inst.w.view(eb), l * inst.len_a), ("b", ZK_G1, inst.w.view(eb), l * inst.len_a),
                                         ("b", ZK_G2, inst.w.view(eb), l * inst.len_a),
                                         ("l", ZK_G1, inst.w.view(inst.ni * eb), l * inst.len_w), ("h", ZK_G1, h_pub, m)):
            pts = inst.unpacked_points(name, group)
            sums.append(msm(pp, group, pts, scal, count))
            pts.free()
        nl = pp.fq.nl
        pa, pb, pc = (np.zeros((pp.n, 3 * nl), dtype=np.uint64), np.zeros((pp.n, 6 * nl), dtype=np.uint64),
                      np.zeros((pp.n, 3 * nl), dtype=np.uint64))
        rr, ss = pp.fr.encode_one(r), pp.fr.encode_one(s)
        arr = (C.c_void_p * 5)(*[x.ctypes.data for x in sums])
        pp._check(pp.lib.zk_groth16_assemble(pp.h, C.byref(inst.crs.ct), rr.ctypes.data, ss.ctypes.data, arr, None,
                                             pa.ctypes.data, pb.ctypes.data, pc.ctypes.data))
        assert norm((pa, pb, pc)) == ref
        del wit, hsh, h_pub
        x = synthetic.rand_fr_device(pp, m + 1, 77)
        num_sh, den_sh = pp.pack(x.view(eb), m // l, 78), pp.pack(x, m // l, 79)
        res = zk.d_pp(pp, num_sh, den_sh, zk.DegRedMask.zero(), m // l, seed=80)
        prod = pp.unpack(res, m // l)
        zk.api.vec_scale(pp, prod, pp.download_fr(x, 1)[0], m)
        assert np.array_equal(prod.to_numpy()[: m * pp.fr.nl], x.to_numpy()[pp.fr.nl:(m + 1) * pp.fr.nl])
    finally:
        pp.close()


def test_c5_bls12_381_d_ifft_d_fft_chain_at_2_24_equals_c_oracle():
    """BASELINE configs[4] at its real size, EXACT: one masked d_ifft (coset shift, rearranged output)
    followed by one masked d_fft -- a third of circom_h's transform chain (ext_wit.rs:127-170) -- at m = 2^24 on BLS12-381,
    n = 8, l = 2: every one of the 2 x 67 M output shares equals the C restatement's (oracle/c zkref_d_fft_mt: the parties'
    local stages on 8 threads, the king's pack / unpack2 over 64).  Replay share randomness; masks sampled by the library's
    dealer and handed to the oracle as data."""
    from oracle.cref import CPss
    from oracle.params import CURVES
    cv = CURVES["bls12_381"]
    log_m = int(os.environ.get("ZK_C5_LOG_M", "24"))
    m = 1 << log_m
    pp = zk.PackedSharingParams("bls12_381", 2)
    try:
        cp = CPss("bls12_381", 2)
        dom = Domain(cv, m)
        g = Domain(cv, 2 * m).element(1)
        rng = np.random.default_rng(50)
        shares = rng.integers(0, 1 << 62, size=(pp.n * (m // 2), 4), dtype=np.uint64)
        shares[:, 3] &= np.uint64((1 << 60) - 1)
        buf = zk.DeviceBuffer.from_numpy(pp, shares)
        want = shares
        for step, (inverse, seed) in enumerate(((True, 123), (False, 124))):
            mask = zk.FftMask.sample(pp, inverse, g if inverse else None, 1 if inverse else 0, log_m, 200 + step)
            if inverse:
                zk.d_ifft(pp, buf, mask, True, log_m, g=g, seed=seed)
            else:
                zk.d_fft(pp, buf, mask, False, log_m, seed=seed)
            im = mask.in_mask.to_numpy().reshape(-1, 4)
            om = mask.out_mask.to_numpy().reshape(-1, 4)
            del mask
            cp.d_fft_arrays_mt(want, m // 2, dom.group_gen_inv if inverse else dom.group_gen,
                               dom.size_inv if inverse else None, g if inverse else None, inverse, im, om, seed)
            del im, om
            got = buf.to_numpy().reshape(-1, 4)
            assert np.array_equal(got, want), "step %d" % step
            del got
    finally:
        pp.close()


def test_c5_bls12_381_d_msm_2_23_per_party_equals_six_limb_c_oracle():
    """BASELINE configs[4] at its real size, EXACT: the G1 d_msm of a 2^24-constraint proof -- 2^23 points per party, 8
    parties, ONE 2^26-point Pippenger on the GPU (quad / split kernels, staged sort, wide entry format) -- against arkworks'
    signed-digit Pippenger restated in C with six 64-bit limbs (oracle/c libzkref6.so), every party's G::msm on its own
    threads, then the king's unpack2 + sum (dmsm/mod.rs:73-92) through the Python oracle."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle.cref import CGroup6
    from oracle.curve import GroupOps, g1
    from oracle.params import CURVES
    from oracle.pss import PackedSharingParams as OPP
    cv = CURVES["bls12_381"]
    ln = 1 << int(os.environ.get("ZK_C5_LOG_LEN", "23"))
    pp = zk.PackedSharingParams("bls12_381", 2)
    try:
        cg = CGroup6("bls12_381")
        rng = np.random.default_rng(51)

        def rand(count):
            a = rng.integers(0, 1 << 62, size=(count, 4), dtype=np.uint64)
            a[:, 3] &= np.uint64((1 << 60) - 1)
            return a
        tot = pp.n * ln
        bases = zg.base_points(pp, ZK_G1, zk.DeviceBuffer.from_numpy(pp, rand(tot)), tot)      # distinct multiples of the generator
        scal = zk.DeviceBuffer.from_numpy(pp, rand(tot))
        out = zk.d_msm(pp, ZK_G1, bases, scal, ln)
        G = g1(cv)
        got = [dec_jacobian(pp, out[p]) for p in range(pp.n)]
        # the oracle's inputs: the same points (GPU layout = 12 x u32 little-endian = 6 x u64), the same scalars as
        # residues to the radix 2^384: m6 = m4 * 2^128 mod r, a multiplication by a constant done on the device
        bases_h = bases.to_numpy().reshape(tot, 12)
        zk.api.vec_scale(pp, scal, (1 << 128) % cv.r, tot)
        sc6 = np.zeros((tot, 6), dtype=np.uint64)
        sc6[:, :4] = scal.to_numpy().reshape(tot, 4)
        del scal, bases

        def party(p):
            return cg.msm_g1_arrays(np.ascontiguousarray(bases_h[p * ln:(p + 1) * ln]), np.ascontiguousarray(sc6[p * ln:(p + 1) * ln]),
                                    ln, nthreads=24)
        with ThreadPoolExecutor(max_workers=pp.n) as ex:
            parts = list(ex.map(party, range(pp.n)))
        dec6 = lambda a: tuple(cg.fq.dec(a))
        o = OPP(cv, 2)
        want = G.sum(o.unpack2([dec6(x) for x in parts], GroupOps(G)))
        for p in range(pp.n):
            assert G.eq(got[p], want)
    finally:
        pp.close()


@pytest.mark.parametrize("scalars", ["random", "repeating"])
def test_c5_bls12_381_g2_d_msm_2_20_per_party_equals_six_limb_c_oracle(scalars):
    """The G2 d_msm (V = b_g2_query, prove.rs:134-160) at the size the staged path runs, EXACT: 2^20 points per party, 8
    parties, ONE 2^23-point Pippenger on the GPU -- staged 1024-thread sort, wide entry format, heavy list and the 12-limb
    lane-quad accumulate (msm_accumulate_split_kernel) -- against arkworks' signed-digit Pippenger over Fq2 restated in C with
    six 64-bit limbs (oracle/c libzkref6.so zkref_msm_g2), every party's G::msm on its own threads, then the king's
    unpack2 + sum (dmsm/mod.rs:73-92) through the Python oracle.  `repeating`: witness-like scalars (60 % ones, 25 %
    fives, 5 % r - 1: buckets of hundreds of thousands of entries through the heavy list)."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle.cref import CGroup6
    from oracle.curve import GroupOps, g2
    from oracle.params import CURVES
    from oracle.pss import PackedSharingParams as OPP
    cv = CURVES["bls12_381"]
    ln = 1 << int(os.environ.get("ZK_C5_G2_LOG_LEN", "20"))
    pp = zk.PackedSharingParams("bls12_381", 2)
    try:
        cg = CGroup6("bls12_381")
        rng = np.random.default_rng(52)

        def rand(count):
            a = rng.integers(0, 1 << 62, size=(count, 4), dtype=np.uint64)
            a[:, 3] &= np.uint64((1 << 60) - 1)
            return a
        tot = pp.n * ln
        bases = zg.base_points(pp, ZK_G2, zk.DeviceBuffer.from_numpy(pp, rand(tot)), tot)      # distinct multiples of the generator
        sc = rand(tot)
        if scalars == "repeating":
            sel = rng.random(tot)
            for lo, hi, val in ((0.0, 0.6, 1), (0.6, 0.85, 5), (0.85, 0.9, cv.r - 1)):
                sc[(sel >= lo) & (sel < hi)] = pp.fr.encode([val])[0]
        scal = zk.DeviceBuffer.from_numpy(pp, sc)
        out = zk.d_msm(pp, ZK_G2, bases, scal, ln)
        G = g2(cv)
        got = [dec_jacobian(pp, out[p], True) for p in range(pp.n)]
        bases_h = bases.to_numpy().reshape(tot, 24)
        zk.api.vec_scale(pp, scal, (1 << 128) % cv.r, tot)          # residues to the radix 2^384 (see the G1 test above)
        sc6 = np.zeros((tot, 6), dtype=np.uint64)
        sc6[:, :4] = scal.to_numpy().reshape(tot, 4)
        del scal, bases

        def party(p):
            return cg.msm_g2_arrays(np.ascontiguousarray(bases_h[p * ln:(p + 1) * ln]), np.ascontiguousarray(sc6[p * ln:(p + 1) * ln]),
                                    ln, nthreads=24)
        with ThreadPoolExecutor(max_workers=pp.n) as ex:
            parts = list(ex.map(party, range(pp.n)))

        def dec6(a):
            v = cg.fq.dec(a)
            return ((v[0], v[1]), (v[2], v[3]), (v[4], v[5]))
        o = OPP(cv, 2)
        want = G.sum(o.unpack2([dec6(x) for x in parts], GroupOps(G)))
        for p in range(pp.n):
            assert G.eq(got[p], want)
    finally:
        pp.close()
