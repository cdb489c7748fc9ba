"""Parity at BASELINE.json sizes (configs 2-3) against the plain-C oracle, plus size-independent properties for the
curves the C oracle does not cover."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import zksaas_amd as zk
from zksaas_amd.api import ZK_G1, msm
from oracle.cref import CPss
from oracle.curve import g1
from oracle.field import Domain
from oracle.params import CURVES, BN254

from gpu_util import ctx, dec_jacobian


def _rand_fr_array(count, seed, top_bits=60):
    rng = np.random.default_rng(seed)
    a = rng.integers(0, 1 << 62, size=(count, 4), dtype=np.uint64)
    a[:, 3] &= np.uint64((1 << top_bits) - 1)        # < r for all three scalar fields: valid Montgomery residues
    return a


@pytest.mark.parametrize("masked", [False, True])
@pytest.mark.parametrize("inverse", [False, True])
def test_d_fft_2_20_shares_equal_c_oracle(inverse, masked):
    """Config 2: d_fft / d_ifft, m = 2^20, BN254, l = 2, n = 8: every output SHARE equals the C restatement's -- with
    FftMask::zero() and with a SAMPLED mask pair (dfft/mod.rs:30-85; the masks the headline bench runs with), whose
    in-mask is added after fft1 and whose out-mask after the king's pack."""
    pp = ctx("bn254", 2)
    cp = CPss("bn254", 2)
    log_m = 20
    m = 1 << log_m
    dom = Domain(BN254, m)
    shares = _rand_fr_array(pp.n * (m // 2), 5)
    g = Domain(BN254, 2 * m).element(1) if inverse else None
    buf = zk.DeviceBuffer.from_numpy(pp, shares)
    mask, im, om = zk.FftMask.zero(), None, None
    if masked:
        mask = zk.FftMask.sample(pp, inverse, g, 1 if inverse else 0, log_m, 123)
        im = mask.in_mask.to_numpy().reshape(-1, 4).copy()
        om = mask.out_mask.to_numpy().reshape(-1, 4).copy()
        assert im.any() and om.any() and not np.array_equal(im, om)
    if inverse:
        zk.d_ifft(pp, buf, mask, True, log_m, g=g, seed=77)
    else:
        zk.d_fft(pp, buf, mask, False, log_m, seed=77)
    got = buf.to_numpy().reshape(-1, 4)
    want = shares.copy()
    cp.d_fft_arrays(want, m // 2, dom.group_gen_inv if inverse else dom.group_gen, dom.size_inv if inverse else None, g,
                    inverse, im, om, 77)
    assert np.array_equal(got, want)
    if masked:
        # the masks cancel: unpack2 of the masked output is the transform of the unpacked input (spot check, first
        # chunks) -- i.e. the SAME secrets the unmasked run reconstructs to
        buf0 = zk.DeviceBuffer.from_numpy(pp, shares)
        if inverse:
            zk.d_ifft(pp, buf0, zk.FftMask.zero(), True, log_m, g=g, seed=77)
        else:
            zk.d_fft(pp, buf0, zk.FftMask.zero(), False, log_m, seed=77)
        a = pp.unpack2(buf, m // 2).to_numpy().reshape(-1, 4)
        b = pp.unpack2(buf0, m // 2).to_numpy().reshape(-1, 4)
        assert np.array_equal(a, b)


def test_msm_2_20_equals_c_oracle():
    """Config 3 size: 2^20-point G1 MSM (doubling-chain bases as in local_groth_bench.rs:25-29, random scalars)."""
    pp = ctx("bn254", 2)
    cp = CPss("bn254", 2)
    n = 1 << 20
    bases = cp.doubling_chain_g1(BN254.g1, n)
    scalars = _rand_fr_array(n, 6)
    got = msm(pp, ZK_G1, zk.DeviceBuffer.from_numpy(pp, bases), zk.DeviceBuffer.from_numpy(pp, scalars), n)
    want = cp.msm_g1_arrays(bases, scalars, n, nthreads=8)
    G = g1(BN254)
    assert G.eq(dec_jacobian(pp, got), dec_jacobian(pp, want))


@pytest.mark.parametrize("table,log_n", [(False, 20), (True, 20), (False, 22)])
def test_msm_2_20_with_repeating_scalars_equals_c_oracle(table, log_n):
    """Real witnesses repeat values (boolean wires): 60 % of the scalars are 1, 25 % are 5, 5 % are r - 1, the rest random.
    Buckets of hundreds of thousands of entries, spread over thousands of accumulate lanes, go through the heavy list with
    SEVERAL virtual workgroups each (msm_heavy_kernel's chunk sums + the finalize kernel's extra workgroups: round 4,
    a lone 2^24-point BLS12-381 MSM with such scalars took 121 ms instead of 39); with a table all windows share the
    bucket set."""
    from zksaas_amd import api
    pp = ctx("bn254", 2)
    cp = CPss("bn254", 2)
    n = 1 << log_n                  # 2^22: the staged (1024-thread) sort kernels, one bin holding 60 % of the entries
    bases = cp.doubling_chain_g1(BN254.g1, n)
    scalars = _rand_fr_array(n, 16)
    sel = np.random.default_rng(17).random(n)
    for lo, hi, val in ((0.0, 0.6, 1), (0.6, 0.85, 5), (0.85, 0.9, BN254.r - 1)):
        scalars[(sel >= lo) & (sel < hi)] = pp.fr.encode([val])[0]
    bd = zk.DeviceBuffer.from_numpy(pp, bases)
    if table:
        api.msm_precompute(pp, ZK_G1, bd, n)
    try:
        got = msm(pp, ZK_G1, bd, zk.DeviceBuffer.from_numpy(pp, scalars), n)
    finally:
        if table:
            api.msm_forget(pp, bd)
    want = cp.msm_g1_arrays(bases, scalars, n, nthreads=8)
    G = g1(BN254)
    assert G.eq(dec_jacobian(pp, got), dec_jacobian(pp, want))


def test_d_msm_2_17_per_party_equals_c_oracle():
    """d_msm with 2^17 points per party (8 parties, fused into one 2^20-point Pippenger on the GPU)."""
    pp = ctx("bn254", 2)
    cp = CPss("bn254", 2)
    ln = 1 << 17
    chain = cp.doubling_chain_g1(BN254.g1, pp.n * ln)
    scalars = _rand_fr_array(pp.n * ln, 7)
    out = zk.d_msm(pp, ZK_G1, zk.DeviceBuffer.from_numpy(pp, chain), zk.DeviceBuffer.from_numpy(pp, scalars), ln)
    G = g1(BN254)
    o = cp.opp
    # reference: per-party G::msm, king unpack2 + sum (dmsm/mod.rs:73-92)
    from oracle.curve import GroupOps
    parts = [dec_jacobian(pp, cp.msm_g1_arrays(chain[p * ln:(p + 1) * ln], scalars[p * ln:(p + 1) * ln], ln, 8))
             for p in range(pp.n)]
    want = G.sum(o.unpack2(parts, GroupOps(G)))
    for p in range(pp.n):
        assert G.eq(dec_jacobian(pp, out[p]), want)


@pytest.mark.parametrize("curve", ["bls12_381", "bls12_377"])
def test_msm_properties_on_12_limb_curves(curve):
    """BLS12 base fields (12 limbs, no C oracle): msm(b, s) + msm(b, -s) = 0 and msm(b, s1) + msm(b, s2) = msm(b, s1+s2)
    at 2^14 points, and a small exact comparison with the Python oracle."""
    c = CURVES[curve]
    pp = ctx(curve, 2)
    G = g1(c)
    n = 1 << 14
    gen = G.from_affine(c.g1)
    # bases: 64 distinct multiples of the generator, tiled
    pts = G.batch_to_affine([G.mul(gen, 3 + 7 * i) for i in range(64)])
    rows = np.stack([pp.fq.encode([p[0], p[1]]).reshape(-1) for p in pts])
    bases = zk.DeviceBuffer.from_numpy(pp, np.tile(rows, (n // 64, 1)))
    rng = np.random.default_rng(8)
    s1 = [int.from_bytes(rng.bytes(32), "little") % c.r for _ in range(n)]
    s2 = [int.from_bytes(rng.bytes(32), "little") % c.r for _ in range(n)]
    up = lambda v: zk.DeviceBuffer.from_numpy(pp, pp.fr.encode(v))
    m1 = dec_jacobian(pp, msm(pp, ZK_G1, bases, up(s1), n))
    m1n = dec_jacobian(pp, msm(pp, ZK_G1, bases, up([(-x) % c.r for x in s1]), n))
    m2 = dec_jacobian(pp, msm(pp, ZK_G1, bases, up(s2), n))
    m12 = dec_jacobian(pp, msm(pp, ZK_G1, bases, up([(x + y) % c.r for x, y in zip(s1, s2)]), n))
    assert G.is_identity(G.add(m1, m1n))
    assert G.eq(G.add(m1, m2), m12)
    # exact: sum_i s_i * P_(i mod 64) = sum_j (sum_{i = j mod 64} s_i) * P_j
    agg = [sum(s1[j::64]) % c.r for j in range(64)]
    assert G.eq(m1, G.msm(pts, agg))


def test_d_fft_roundtrip_bls12_381_2_18():
    """Config 5 curve at a moderate size: d_ifft(rearrange) o d_fft is the identity on the reconstructed secrets."""
    pp = ctx("bls12_381", 2)
    log_m = 18
    m = 1 << log_m
    sec = _rand_fr_array(m, 9)
    sec_d = zk.DeviceBuffer.from_numpy(pp, sec)
    pp._check(pp.lib.zk_bitrev(pp.h, sec_d.ptr, log_m, None))
    shares = pp.pack(sec_d, m // 2, seed=9, order=1)
    zk.d_ifft(pp, shares, zk.FftMask.zero(), True, log_m, seed=10)
    zk.d_fft(pp, shares, zk.FftMask.zero(), False, log_m, seed=11)
    assert np.array_equal(pp.unpack(shares, m // 2).to_numpy().reshape(m, 4), sec)


@pytest.mark.parametrize("curve,log_m", [("bn254", 20), ("bls12_381", 18)])
def test_synthetic_prover_distributed_equals_local_and_ignores_share_randomness(curve, log_m):
    """Size-independent properties of the whole prover on a synthetic instance built on the device
    (zksaas_amd/synthetic.py; SURVEY.md 8d C5 at a size that runs in seconds; tools/c5_bls381.py runs 2^24):
    * dealing the same witness again (new share randomness) and proving IMMEDIATELY after the dealing kernels were
      queued gives the same proof -- also the regression test for the stream-ordering bug where the MSM streams read
      the shares before the caller's stream had written them;
    * distributed == local: zk_groth16_assemble over five plain zk_msm's of the public query elements."""
    import ctypes as C
    from zksaas_amd import synthetic, wire
    from zksaas_amd import groth16 as zg
    from zksaas_amd.api import ZK_G2
    pp = ctx(curve, 2)
    inst = synthetic.SyntheticInstance(pp, log_m, seed=3)
    r, s = 0x1234567890ABCDEF1234567890ABCDEF, 0xFEDCBA0987654321FEDCBA0987654321

    def norm(pf):
        return (wire.jacobian_to_affine(pp, pf[0][0], False), wire.jacobian_to_affine(pp, pf[1][0], True),
                wire.jacobian_to_affine(pp, pf[2][0], False))
    wit = inst.witness(seed=100)
    pp.sync()
    ref = norm(zg.prove(pp, inst.crs, wit, r, s, seed=7))
    for ws in (900, 901, 902):
        w2 = inst.witness(seed=ws)                      # kernels still in flight on the default stream
        assert norm(zg.prove(pp, inst.crs, w2, r, s, seed=ws)) == ref
        del w2
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
    # the proof survives the ark-compressed wire format
    blob = wire.proof_to_bytes(pp, pa[0], pb[0], pc[0])
    assert wire.proof_from_bytes(pp, blob, curve) == ref


@pytest.mark.parametrize("curve,masked", [("bn254", True), ("bls12_381", False)])
def test_d_pp_2_20_shares_equal_c_oracle(curve, masked):
    """d_pp (dpp/mod.rs:15-87) at m = 2^20: every output SHARE equals the C restatement's, which follows the reference
    step for step (unpack, one inverse() per element, serial prefix product, pack_vec, deg_red) -- the GPU path reaches
    the same field elements through a prefix scan of the numerators, a suffix scan of the denominators and one inversion
    (csrc/dpp.hpp), with the deg_red round fused into its last kernel."""
    import os
    pp = ctx(curve, 2)
    cp = CPss(curve, 2)
    m = 1 << 20
    ln = m // 2
    num, den = _rand_fr_array(pp.n * ln, 31), _rand_fr_array(pp.n * ln, 32)
    mask, im, om = zk.DegRedMask.zero(), None, None
    if masked:
        mask = zk.DegRedMask.sample(pp, ln, 33)
        im = mask.in_mask.to_numpy().reshape(-1, 4).copy()
        om = mask.out_mask.to_numpy().reshape(-1, 4).copy()
        assert im.any() and om.any()
    res = zk.d_pp(pp, zk.DeviceBuffer.from_numpy(pp, num), zk.DeviceBuffer.from_numpy(pp, den), mask, ln, seed=34)
    want = cp.d_pp_arrays(num, den, ln, im, om, 34, threads=min(32, os.cpu_count() or 1))
    assert np.array_equal(res.to_numpy().reshape(-1, 4), want)


def test_d_pp_telescopes_bls12_381_2_18():
    """d_pp (dpp/mod.rs:15-87) at 2^18 on the config-5 curve: num_i = x_(i+1), den_i = x_i  =>  prefix product_i
    times x_0 equals x_(i+1) (size-independent property; the small exact comparison is in test_gpu_dfft.py)."""
    from zksaas_amd import synthetic
    pp = ctx("bls12_381", 2)
    m, l, eb, nl = 1 << 18, 2, pp.fr.nbytes, pp.fr.nl
    x = synthetic.rand_fr_device(pp, m + 1, 77)
    num_sh, den_sh = pp.pack(x.view(eb), m // l, 78), pp.pack(x, m // l, 79)
    res = zk.d_pp(pp, num_sh, den_sh, zk.DegRedMask.zero(), m // l, seed=80)
    prod = pp.unpack(res, m // l)
    zk.api.vec_scale(pp, prod, pp.download_fr(x, 1)[0], m)
    assert np.array_equal(prod.to_numpy()[: m * nl], x.to_numpy()[nl:(m + 1) * nl])



@pytest.mark.parametrize("curve", ["bls12_381", "bls12_377"])
def test_msm_2_20_on_12_limb_curves_equals_six_limb_c_oracle(curve):
    """Config-3 size on the 12-limb base fields (round 2 checked these through properties only): a 2^20-point G1 MSM
    (doubling-chain bases, random scalars) equals arkworks' Pippenger restated in C with six 64-bit limbs
    (oracle/c libzkref6.so, pinned on the Python oracle in tests/test_oracle_c.py)."""
    from oracle.cref import CGroup6
    cv = CURVES[curve]
    pp = ctx(curve, 2)
    cg = CGroup6(curve)
    n = 1 << 20
    bases = cg.doubling_chain_g1(cv.g1, n)
    scalars = _rand_fr_array(n, 16)
    got = msm(pp, ZK_G1, zk.DeviceBuffer.from_numpy(pp, bases), zk.DeviceBuffer.from_numpy(pp, scalars), n)
    want = cg.msm_g1_arrays(bases, cg.scalars_from_gpu_residues(scalars), n, nthreads=8)
    G = g1(cv)
    assert G.eq(dec_jacobian(pp, got), dec_jacobian(pp, want))


@pytest.mark.parametrize("curve,g2", [("bn254", False), ("bls12_381", False), ("bn254", True), ("bls12_381", True)])
def test_msm_at_every_lane_shape_equals_the_sum_of_small_msms(curve, g2):
    """Sizes across the accumulate kernel's lane shapes (fewer than one round of waves, between one and two -- the
    case that once left entries uncovered: 2^18 .. 2^19 points at 15-bit windows --, several rounds), both kernels
    (one lane / lane pair per range).  Reference: the sum of 8192-point MSMs of the same vectors (a size the oracle
    comparisons of test_gpu_msm.py cover), added with the oracle's group law.  G::msm is linear (dmsm/mod.rs:73)."""
    from zksaas_amd import groth16 as zg, wire
    from zksaas_amd.api import ZK_G1, ZK_G2
    from oracle.curve import g1, g2 as g2f
    from oracle.params import CURVES
    pp = ctx(curve, 2)
    G = g2f(CURVES[curve]) if g2 else g1(CURVES[curve])
    grp = ZK_G2 if g2 else ZK_G1
    nl = pp.fr.nl
    rng = np.random.default_rng(5)

    def rand(count):
        a = rng.integers(0, 1 << 62, size=(count, nl), dtype=np.uint64)
        a[:, nl - 1] &= np.uint64((1 << 58) - 1)
        return a
    sizes = [60000, 100000, 150000, 200000, 300000] if g2 else [150000, 262144, 300000, 400000, 524288, 700000]
    mx = max(sizes)
    eb, w = pp.fr.nbytes, (4 if g2 else 2) * pp.fq.nbytes
    pts = zg.base_points(pp, grp, zk.DeviceBuffer.from_numpy(pp, rand(mx)), mx)
    sc = zk.DeviceBuffer.from_numpy(pp, rand(mx))
    pt = lambda j: G.from_affine(wire.jacobian_to_affine(pp, j, g2))
    ch = 8192
    prefix = {0: G.identity}
    acc = G.identity
    for c0 in range(0, mx, ch):
        n_ = min(ch, mx - c0)
        acc = G.add(acc, pt(msm(pp, grp, pts.view(c0 * w, n_ * w), sc.view(c0 * eb, n_ * eb), n_)))
        prefix[c0 + n_] = acc
    for n_ in sizes:
        full = (n_ // ch) * ch
        want = prefix[full]
        if n_ > full:
            want = G.add(want, pt(msm(pp, grp, pts.view(full * w, (n_ - full) * w), sc.view(full * eb, (n_ - full) * eb),
                                      n_ - full)))
        assert G.eq(pt(msm(pp, grp, pts.view(0, n_ * w), sc.view(0, n_ * eb), n_)), want), n_


@pytest.mark.parametrize("table", [False, True])
def test_msm_batch_at_every_lane_shape_equals_single_msms(table):
    """zk_msm_batch of 1..6 scalar vectors over the SHA-256 circuit's 119 296-point base vector (and G2 at half of it):
    the batch multiplies the sorted entries, so these sizes walk through every lane shape of the accumulate kernels --
    B = 3 (5.7 M entries) sat in the gap between one and two rounds of waves that once left entries uncovered.
    Every result equals the same vector's single zk_msm as a group element."""
    from zksaas_amd import groth16 as zg, wire, api
    from zksaas_amd.api import ZK_G1, ZK_G2
    pp = ctx("bn254", 2)
    rng = np.random.default_rng(15)

    def rand(count):
        a = rng.integers(0, 1 << 62, size=(count, 4), dtype=np.uint64)
        a[:, 3] &= np.uint64((1 << 58) - 1)
        return a
    for group, npts, g2 in ((ZK_G1, 119296, False), (ZK_G2, 59648, True)):
        bases = zg.base_points(pp, group, zk.DeviceBuffer.from_numpy(pp, rand(npts)), npts)
        if table:
            api.msm_precompute(pp, group, bases, npts)
        vecs = [zk.DeviceBuffer.from_numpy(pp, rand(npts)) for _ in range(6)]
        single = [wire.jacobian_to_affine(pp, msm(pp, group, bases, v, npts), g2) for v in vecs]
        for nb in (2, 3, 5, 6):
            got = api.msm_batch(pp, group, bases, vecs[:nb], npts)
            assert [wire.jacobian_to_affine(pp, got[b], g2) for b in range(nb)] == single[:nb], (group, nb)


@pytest.mark.parametrize("log_n", [21, 22])
def test_table_msm_with_wide_sort_entries_equals_the_table_free_msm(log_n):
    """2^21 points with a fixed-base table: the sorted entry's index (window * stride + point: 25 bits) + sign + low bucket
    bits no longer fit one 32-bit word, so the sort runs on its WIDE entry format (word + 16-bit low part) through the
    small-launch kernels -- a combination no other test reaches.  2^22 points: the same through the STAGED (1024-thread)
    sort kernels, whose tile shrinks to one point per thread so that a tile's 16 windows fit the stage (round 4 found
    that launch shape faulting: the scatter kernel ignored the shrunken tile).  Same group element as the table-free MSM
    of the same vectors (which the other tests pin on the oracle)."""
    from zksaas_amd import api, groth16 as zg, wire
    from zksaas_amd.api import ZK_G1
    pp = ctx("bn254", 2)
    n = 1 << log_n
    rng = np.random.default_rng(8)

    def rand(count):
        a = rng.integers(0, 1 << 62, size=(count, 4), dtype=np.uint64)
        a[:, 3] &= np.uint64((1 << 58) - 1)
        return a
    pts = zg.base_points(pp, ZK_G1, zk.DeviceBuffer.from_numpy(pp, rand(n)), n)
    sc = zk.DeviceBuffer.from_numpy(pp, rand(n))
    aff = lambda j: wire.jacobian_to_affine(pp, j, False)
    free = aff(msm(pp, ZK_G1, pts, sc, n))
    api.msm_precompute(pp, ZK_G1, pts, n)
    try:
        # window bits by the vector's length (csrc/msm.hpp table_c_auto): 17 bits = 15 windows below 2^22 points, 20 = 13 above
        assert api.msm_table_info(pp, ZK_G1, pts) == ({"window_bits": 17, "windows": 15} if log_n < 22 else
                                                      {"window_bits": 20, "windows": 13})
        assert aff(msm(pp, ZK_G1, pts, sc, n)) == free
    finally:
        api.msm_forget(pp, pts)
