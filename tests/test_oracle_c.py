"""The plain-C restatement (oracle/c/zkref.c) pinned against the Python big-int oracle."""
import ctypes as C

import numpy as np
import pytest

from oracle import dist as od
from oracle.cref import CPss, Field, lib
from oracle.curve import g1, g2
from oracle.dist import transpose
from oracle.field import Domain, bitrev_permute
from oracle.params import CURVES, BN254
from oracle.prng import rand_fp, rand_vec


@pytest.mark.parametrize("name", ["bn254", "bls12_377", "bls12_381"])
def test_field_ops(name):
    p = CURVES[name].r
    F = Field(p)
    a, b = rand_vec(1, 50, p), rand_vec(2, 50, p)
    a[0], b[0], a[1] = 0, p - 1, p - 1
    ea, eb = F.enc(a), F.enc(b)
    out = np.zeros((1, 4), dtype=np.uint64)
    vp = lambda x: x.ctypes.data_as(C.c_void_p)
    for i in range(50):
        for fn, want in (("zkref_mul", a[i] * b[i] % p), ("zkref_add", (a[i] + b[i]) % p),
                         ("zkref_sub", (a[i] - b[i]) % p)):
            getattr(lib(), fn)(C.byref(F.ct), vp(ea[i]), vp(eb[i]), vp(out))
            assert F.dec(out)[0] == want
    lib().zkref_inv(C.byref(F.ct), vp(eb[3]), vp(out))
    assert F.dec(out)[0] == pow(b[3], p - 2, p)


@pytest.mark.parametrize("name,l,m,rearrange,inverse", [
    ("bls12_377", 2, 8, False, False), ("bls12_377", 2, 1024, True, False), ("bn254", 4, 64, True, True),
    ("bn254", 2, 256, False, True), ("bls12_381", 8, 64, False, False)])
def test_d_fft_shares_equal_python_oracle(name, l, m, rearrange, inverse):
    cp = CPss(name, l)
    o = cp.opp
    dom = Domain(CURVES[name], m)
    x = rand_vec(3, m, o.p)
    y = list(x)
    bitrev_permute(y)
    shares = transpose(od.stride_pack(y, o, 4))
    g = Domain(CURVES[name], 2 * m).element(1) if inverse else 1
    gen = dom.group_gen_inv if inverse else dom.group_gen
    masks = od.FftMask.sample(rearrange, g, gen, m, o, 5)
    if inverse:
        want = od.d_ifft(shares, masks, rearrange, dom, g, o, seed=6)
    else:
        want = od.d_fft(shares, masks, rearrange, dom, o, seed=6)
    got = cp.d_fft(shares, dom, rearrange, masks, seed=6, inverse=inverse, g=g if inverse else None)
    assert got == want


def test_fft1_and_deg_red():
    cp = CPss("bn254", 2)
    o = cp.opp
    dom = Domain(BN254, 128)
    v = rand_vec(7, 64, o.p)
    assert cp.fft1(v, dom.group_gen) == od.fft1_in_place(list(v), o, dom.group_gen)
    shares = transpose(od.pack_vec(rand_vec(8, 32, o.p), o, 9))
    mul = [[a * a % o.p for a in s] for s in shares]
    masks = od.DegRedMask.sample(o, 1, 16, 10)
    assert cp.deg_red(mul, masks, 11) == od.deg_red(mul, masks, o, seed=11)


@pytest.mark.parametrize("name,threads", [("bn254", 1), ("bn254", 3), ("bls12_381", 1)])
def test_d_pp_c_matches_python_oracle(name, threads):
    """zkref_d_pp (dpp/mod.rs:15-87 in C, one inverse() per element) == oracle/dist.py d_pp, and the reference's own
    check: num = den reconstructs to all ones (dpp_test.rs:51,62-65); a zero denominator is an error (:55)."""
    cp = CPss(name, 2)
    o = cp.opp
    m = 2200 if threads > 1 else 64        # > 1024 chunks: the threaded spans really split
    num, den = rand_vec(20, m, o.p), rand_vec(21, m, o.p)
    ns, ds = transpose(od.pack_vec(num, o, 22)), transpose(od.pack_vec(den, o, 23))
    masks = od.DegRedMask.sample(o, 1, m // 2, 24)
    got = cp.d_pp(ns, ds, masks, 25, threads)
    assert got == od.d_pp(ns, ds, masks, o, seed=25)
    x = list(range(1, m + 1))
    xs = transpose(od.pack_vec(x, o, 26))
    ones = cp.d_pp(xs, xs, None, 27, threads)
    assert [v for ch in transpose(ones) for v in o.unpack(ch)] == [1] * m
    den[5] = 0
    with pytest.raises(ZeroDivisionError):
        cp.d_pp(ns, transpose(od.pack_vec(den, o, 28)), None, 29, threads)


@pytest.mark.parametrize("count,threads", [(1, 1), (31, 1), (200, 1), (200, 4)])
def test_msm_g1(count, threads):
    cp = CPss("bn254", 2)
    G = g1(BN254)
    gen = G.from_affine(G.gen)
    pts = G.batch_to_affine([G.mul(gen, rand_fp(12, i, BN254.r)) for i in range(count)])
    if count > 3:
        pts[2] = None
    sc = rand_vec(13, count, BN254.r)
    if count > 5:
        sc[4], sc[5] = 0, BN254.r - 1
    assert G.eq(cp.msm_g1(pts, sc, threads), G.msm(pts, sc))


def test_msm_g2():
    cp = CPss("bn254", 2)
    G = g2(BN254)
    gen = G.from_affine(G.gen)
    pts = G.batch_to_affine([G.mul(gen, rand_fp(14, i, BN254.r)) for i in range(40)])
    sc = rand_vec(15, 40, BN254.r)
    assert G.eq(cp.msm_g2(pts, sc), G.msm(pts, sc))


def test_doubling_chain():
    cp = CPss("bn254", 2)
    G = g1(BN254)
    chain = cp.doubling_chain_g1(BN254.g1, 6)
    P = G.from_affine(BN254.g1)
    for i in range(6):
        aff = G.to_affine(P)
        assert cp.fq.dec(chain[i]) == [aff[0], aff[1]]
        P = G.double(P)



@pytest.mark.parametrize("curve", ["bls12_381", "bls12_377"])
def test_six_limb_build_msm_and_doubling_chain_equal_the_python_oracle(curve):
    """libzkref6.so (zkref.c with NL = 6): arkworks' signed-digit Pippenger over the 381 / 377-bit base fields and the
    doubling chain, pinned on the Python big-integer oracle (edge scalars, an identity base, threads)."""
    from oracle.cref import CGroup6
    from oracle.curve import g1
    from oracle.params import CURVES
    from oracle.prng import rand_vec
    cv = CURVES[curve]
    G, cg = g1(cv), CGroup6(curve)
    gen = G.from_affine(cv.g1)
    pts = G.batch_to_affine([G.mul(gen, 3 + i * i) for i in range(70)])
    pts[7] = None
    sc = rand_vec(5, 70, cv.r)
    sc[3], sc[4], sc[5] = 0, cv.r - 1, 1
    want = G.msm(pts, sc)
    for th in (1, 3):
        assert G.eq(cg.msm_g1(pts, sc, nthreads=th), want)
    chain = cg.doubling_chain_g1(cv.g1, 6)
    for i in range(6):
        v = cg.fq.dec(chain[i])
        assert (v[0], v[1]) == G.to_affine(G.mul(gen, 1 << i))
    # residues of the GPU's 4-limb scalar field carry over: m6 = m4 * 2^128
    import numpy as np
    vals = [0, 1, cv.r - 1, 123456789 ** 4 % cv.r]
    m4 = np.array([[(v * (1 << 256) % cv.r >> (64 * k)) & ((1 << 64) - 1) for k in range(4)] for v in vals], dtype=np.uint64)
    assert cg.fr.dec(cg.scalars_from_gpu_residues(m4)) == vals



def test_tuned_king_gives_the_reference_kings_shares():
    """cpu_baseline's `all_cores_tuned_king` leg (zkref_set_fast_king: pack / unpack2 as precomputed matrices over threads)
    must produce exactly the shares of the serial FFT-form king it stands in for: d_fft, d_ifft (rearranged, coset), masked,
    and deg_red."""
    import numpy as np
    from oracle.cref import CPss, lib
    from oracle.field import Domain
    from oracle.params import BN254
    cp = CPss("bn254", 2)
    m = 1 << 12
    dom = Domain(BN254, m)
    rng = np.random.default_rng(1)

    def rand(count):
        a = rng.integers(0, 1 << 62, size=(count, 4), dtype=np.uint64)
        a[:, 3] &= np.uint64((1 << 60) - 1)
        return a
    a, im, om = rand(8 * (m // 2)), rand(8 * (m // 2)), rand(8 * (m // 2))
    res = []
    try:
        for fast in (0, 1):
            lib().zkref_set_fast_king(fast, 4)
            for rearr, inv in ((False, False), (True, True)):
                w = a.copy()
                g = Domain(BN254, 2 * m).element(1) if inv else None
                cp.d_fft_arrays(w, m // 2, dom.group_gen_inv if inv else dom.group_gen, dom.size_inv if inv else None, g,
                                rearr, im, om, 7)
                res.append(w)
            x = a.copy()
            cp.deg_red_arrays(x, m // 2, im, om, 9)
            res.append(x)
    finally:
        lib().zkref_set_fast_king(0, 1)
    assert all(np.array_equal(res[i], res[i + 3]) for i in range(3))


def test_threaded_d_fft_equals_the_serial_restatement():
    """zkref_d_fft_mt (parties' local stages on n threads, matrix king over several) is what the 2^24 checks of
    tests/test_gpu_configs.py run: same shares as zkref_d_fft, masks and coset shift included, on both scalar fields."""
    import numpy as np
    from oracle.cref import CPss
    from oracle.field import Domain
    from oracle.params import CURVES
    for cv in ("bn254", "bls12_381"):
        cp, c = CPss(cv, 2), CURVES[cv]
        m = 1 << 10
        dom = Domain(c, m)
        rng = np.random.default_rng(1)

        def rand():
            a = rng.integers(0, 1 << 62, size=(cp.n * (m // 2), 4), dtype=np.uint64)
            a[:, 3] &= np.uint64((1 << 60) - 1)
            return a
        a, im, om = rand(), rand(), rand()
        g = Domain(c, 2 * m).element(1)
        for inverse in (True, False):
            x, y = a.copy(), a.copy()
            args = (m // 2, dom.group_gen_inv if inverse else dom.group_gen, dom.size_inv if inverse else None,
                    g if inverse else None, inverse, im, om, 7)
            cp.d_fft_arrays(x, *args)
            cp.d_fft_arrays_mt(y, *args, king_threads=4)
            assert np.array_equal(x, y)


@pytest.mark.parametrize("threads,r_zero", [(1, False), (8, False), (1, True)])
def test_local_prover_in_c_equals_the_python_oracle(threads, r_zero):
    """BASELINE configs[0] (groth16/examples/sha256.rs:191-199): CpuProver.prove_local -- zkref_circom_ref + five G::msm over
    the unpacked proving key + prove.rs' assembly in C -- gives the proof oracle/groth16.py create_proof_local gives, and
    its h is circom_ref's (ext_wit.rs:239-285)."""
    from oracle import groth16 as g
    from oracle.cpu_prover import CpuProver
    from test_oracle_groth16 import small_r1cs
    Cv = BN254
    P = Cv.r
    r1, w = small_r1cs(nc=40)
    key = g.setup_scalars(Cv, r1, g.Trapdoor.from_seed(42, P))
    r, s = (0 if r_zero else rand_fp(43, 0, P)), rand_fp(43, 1, P)
    G1, G2 = g1(Cv), g2(Cv)
    pk = g.proving_key_points(key, G1, G2)
    want = g.create_proof_local(Cv, r1, pk, G1, G2, w, r, s)
    cpu = CpuProver("bn254", 2)
    cp = cpu.cp
    q = g.qap(Cv, r1, w)
    log_m = q.domain.log_size
    h = cpu.circom_ref(cp.fr.enc(q.a), cp.fr.enc(q.b), cp.fr.enc(q.c), log_m, threads)
    assert cp.fr.dec(h) == g.circom_ref(q.a, q.b, q.c, q.domain)

    def enc1(pts):
        return np.stack([cp.fq.enc([0, 0] if p is None else [p[0], p[1]]).reshape(-1) for p in pts])

    def enc2(pts):
        return np.stack([cp.fq.enc([0, 0, 0, 0] if p is None else [p[0][0], p[0][1], p[1][0], p[1][1]]).reshape(-1) for p in pts])
    inp = {"qap_a": cp.fr.enc(q.a), "qap_b": cp.fr.enc(q.b), "qap_c": cp.fr.enc(q.c), "log_m": log_m, "w": cp.fr.enc(w),
           "ni": r1.num_instance_variables, "a_query": enc1(pk.a_query), "b_g1_query": enc1(pk.b_g1_query),
           "b_g2_query": enc2(pk.b_g2_query), "l_query": enc1(pk.l_query), "h_query": enc1(pk.h_query),
           "delta_g1": enc1([pk.delta_g1])[0], "delta_g2": enc2([pk.delta_g2])[0], "alpha_g1": enc1([pk.alpha_g1])[0],
           "beta_g1": enc1([pk.beta_g1])[0], "beta_g2": enc2([pk.beta_g2])[0], "r": r, "s_": s}
    (A, B, Cc), tm = cpu.prove_local(inp, threads)
    assert cpu.affine(A) == G1.to_affine(want[0]) and cpu.affine(B, True) == G2.to_affine(want[1])
    assert cpu.affine(Cc) == G1.to_affine(want[2])
    assert set(tm) == {"circom_h_s", "msm_s", "assemble_s", "total_s"}
