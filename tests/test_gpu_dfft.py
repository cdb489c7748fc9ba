"""GPU parity for d_fft / d_ifft (dist-primitives/src/dfft): fft1, the fused king kernel, masks, dropout."""
import ctypes as C

import pytest

pytestmark = pytest.mark.gpu

import zksaas_amd as zk
from oracle import dist as od
from oracle.field import Domain, bitrev_permute
from oracle.params import CURVES
from oracle.prng import rand_vec

from gpu_util import ctx, opp, up, up_parties, down_parties


def _deal(x, o, seed):
    y = list(x)
    bitrev_permute(y)
    return od.transpose(od.stride_pack(y, o, seed))


@pytest.mark.parametrize("curve,l,log_m,inverse", [
    ("bls12_377", 2, 3, 0), ("bls12_377", 2, 7, 1), ("bn254", 4, 8, 0), ("bn254", 2, 12, 0),   # tiled, 1 pass
    ("bn254", 2, 13, 1), ("bn254", 2, 14, 0), ("bls12_381", 2, 13, 0), ("bn254", 4, 15, 1),      # tiled, 2 passes
])
def test_fft1_matches_oracle(curve, l, log_m, inverse):
    pp, o = ctx(curve, l), opp(curve, l)
    m = 1 << log_m
    dom = Domain(CURVES[curve], m)
    batch = 3
    vecs = [rand_vec(20 + b, m // l, o.p) for b in range(batch)]
    buf = up_parties(pp, vecs)
    pp._check(pp.lib.zk_fft1(pp.h, buf.ptr, log_m, inverse, batch, None, None))
    gen = dom.group_gen_inv if inverse else dom.group_gen
    want = [od.fft1_in_place(list(v), o, gen) for v in vecs]
    assert down_parties(pp, buf, batch, m // l) == want


@pytest.mark.parametrize("log_m", [16, 22])
def test_fft1_large_closed_form(log_m):
    """Large sizes (3 passes at 2^22) against the closed form of fft1 on a sparse input."""
    pp, o = ctx("bn254", 2), opp("bn254", 2)
    m = 1 << log_m
    dom = Domain(CURVES["bn254"], m)
    n = m // 2
    # sparse input so the oracle can evaluate single outputs: x = e_i + 3 e_j
    import numpy as np
    arr = np.zeros((n, pp.fr.nl), dtype=np.uint64)
    i, j = 5, n - 7
    arr[i] = pp.fr.encode_one(1)
    arr[j] = pp.fr.encode_one(3)
    buf = zk.DeviceBuffer.from_numpy(pp, arr)
    pp._check(pp.lib.zk_fft1(pp.h, buf.ptr, log_m, 0, 1, None, None))
    out = buf.to_numpy().reshape(n, pp.fr.nl)
    # fft1 is the DIT over w_n with shifted twiddles: out[k] = sum_i x[i] * w_n^(rev(i) * (k+1))  
    wn = dom.group_gen * dom.group_gen % o.p if False else pow(dom.group_gen, 2, o.p)   # w_n = w_m^l, l = 2
    logn = log_m - 1
    rev = lambda v: int(format(v, "0%db" % logn)[::-1], 2)
    for k in (0, 1, 12345 % n, n - 1):
        want = (pow(wn, rev(i) * (k + 1), o.p) + 3 * pow(wn, rev(j) * (k + 1), o.p)) % o.p
        assert pp.fr.decode(out[k])[0] == want


@pytest.mark.parametrize("curve,l,m", [("bls12_377", 2, 8), ("bls12_377", 2, 1024), ("bn254", 4, 64),
                                        ("bn254", 2, 4096), ("bls12_381", 2, 256), ("bn254", 8, 64), ("bn254", 1, 16)])
@pytest.mark.parametrize("rearrange", [False, True])
def test_d_fft_shares_match_oracle(curve, l, m, rearrange):
    """dfft_test.rs / dfft/tests.rs d_fft_works: same inputs, same seeds -> identical output SHARES."""
    pp, o = ctx(curve, l), opp(curve, l)
    dom = Domain(CURVES[curve], m)
    x = rand_vec(30, m, o.p)
    shares = _deal(x, o, 31)
    masks = od.FftMask.sample(rearrange, 1, dom.group_gen, m, o, 32)
    want = od.d_fft(shares, masks, rearrange, dom, o, seed=33)
    buf = up_parties(pp, shares)
    fm = zk.FftMask(up_parties(pp, [mk.in_mask for mk in masks]), up_parties(pp, [mk.out_mask for mk in masks]))
    zk.d_fft(pp, buf, fm, rearrange, m.bit_length() - 1, seed=33)
    got = down_parties(pp, buf, pp.n, m // l)
    assert got == want
    if not rearrange:   # reconstructs to the DFT with arkworks' root of unity (dfft_test.rs:78)
        assert pp.download_fr(pp.unpack(buf, m // l)) == dom.fft(x)


@pytest.mark.parametrize("l,m", [(2, 8), (2, 512), (4, 128)])
@pytest.mark.parametrize("rearrange", [False, True])
def test_d_ifft_with_coset_shift_matches_oracle(l, m, rearrange):
    curve = "bn254"
    pp, o = ctx(curve, l), opp(curve, l)
    dom = Domain(CURVES[curve], m)
    g = Domain(CURVES[curve], 2 * m).element(1)          # ext_wit.rs:120-125
    x = rand_vec(34, m, o.p)
    shares = _deal(x, o, 35)
    masks = od.FftMask.sample(rearrange, g, dom.group_gen_inv, m, o, 36)
    want = od.d_ifft(shares, masks, rearrange, dom, g, o, seed=37)
    buf = up_parties(pp, shares)
    fm = zk.FftMask(up_parties(pp, [mk.in_mask for mk in masks]), up_parties(pp, [mk.out_mask for mk in masks]))
    zk.d_ifft(pp, buf, fm, rearrange, m.bit_length() - 1, g=g, seed=37)
    assert down_parties(pp, buf, pp.n, m // l) == want


def test_d_ifft_then_d_fft_roundtrip_zero_masks():  # dfft/tests.rs:142-220
    pp, o = ctx("bls12_377", 2), opp("bls12_377", 2)
    m = 2048
    x = rand_vec(38, m, o.p)
    buf = up_parties(pp, _deal(x, o, 39))
    zk.d_ifft(pp, buf, zk.FftMask.zero(), True, 11, seed=40)
    zk.d_fft(pp, buf, zk.FftMask.zero(), False, 11, seed=41)
    assert pp.download_fr(pp.unpack(buf, m // 2)) == x


@pytest.mark.parametrize("rearrange", [False, True])
def test_fft_mask_sample_matches_oracle(rearrange):
    curve, l, m = "bn254", 2, 64
    pp, o = ctx(curve, l), opp(curve, l)
    dom = Domain(CURVES[curve], m)
    g = Domain(CURVES[curve], 2 * m).element(1)
    want = od.FftMask.sample(rearrange, g, dom.group_gen_inv, m, o, 50)
    got = zk.FftMask.sample(pp, rearrange, g, 1, 6, 50)
    assert down_parties(pp, got.in_mask, pp.n, m // l) == [w.in_mask for w in want]
    assert down_parties(pp, got.out_mask, pp.n, m // l) == [w.out_mask for w in want]


def test_king_with_dropout_matches_oracle():  # ser_net.rs:57-94 -> lagrange_unpack
    curve, l, m = "bls12_377", 2, 32
    pp, o = ctx(curve, l), opp(curve, l)
    dom = Domain(CURVES[curve], m)
    shares = _deal(rand_vec(42, m, o.p), o, 43)
    pxs = [od.fft1_in_place(list(s), o, dom.group_gen) for s in shares]
    parties = [0, 1, 2, 4, 5, 6, 7]
    want = od.king_fft2([pxs[i] for i in parties], parties, False, 1, o, dom.group_gen, 44)
    inb = up_parties(pp, [pxs[i] for i in parties])
    out = pp.alloc_fr(pp.n * (m // l))
    arr = (C.c_uint32 * len(parties))(*parties)
    pp._check(pp.lib.zk_fft2_king(pp.h, inb.ptr, arr, len(parties), 5, 0, None, 0, 0, 44, out.ptr, None, None))
    assert down_parties(pp, out, pp.n, m // l) == want


def test_size_mismatch_is_bad_input():
    pp = ctx("bn254", 4)
    with pytest.raises(zk.ZkError) as e:
        zk.d_fft(pp, pp.alloc_fr(16), zk.FftMask.zero(), False, 1)     # m = 2 < l = 4
    assert e.value.code == 4


def test_large_roundtrip_2_20():
    """BASELINE config 2 size: d_ifft(rearrange) o d_fft is the identity on the reconstructed secrets."""
    import numpy as np
    pp = ctx("bn254", 2)
    log_m = 20
    m = 1 << log_m
    rng = np.random.default_rng(7)
    sec = rng.integers(0, 1 << 62, size=(m, 4), dtype=np.uint64)
    sec[:, 3] &= np.uint64((1 << 60) - 1)            # < r: valid (Montgomery-form) field elements
    sec_d = zk.DeviceBuffer.from_numpy(pp, sec)
    pp._check(pp.lib.zk_bitrev(pp.h, sec_d.ptr, log_m, None))
    shares = pp.pack(sec_d, m // 2, seed=9, order=1)
    zk.d_ifft(pp, shares, zk.FftMask.zero(), True, log_m, seed=10)
    zk.d_fft(pp, shares, zk.FftMask.zero(), False, log_m, seed=11)
    back = pp.unpack(shares, m // 2).to_numpy().reshape(m, 4)
    assert np.array_equal(back, sec)
