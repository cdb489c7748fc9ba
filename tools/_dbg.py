import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.environ.get('DBG_ROOT') or os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["ZK_RNG_REPLAY"] = "1"
import zksaas_amd as zk
from zksaas_amd import synthetic, wire, api
from zksaas_amd import groth16 as zg
from zksaas_amd.api import ZK_G1, ZK_G2
curve, log_m = sys.argv[1], int(sys.argv[2])
pp = zk.PackedSharingParams(curve, 2)
inst = synthetic.SyntheticInstance(pp, log_m, seed=3)
r, s = 0x1234567890ABCDEF1234567890ABCDEF, 0xFEDCBA0987654321FEDCBA0987654321
print('inst ok', flush=True)
wit = inst.witness(seed=100)
pp.sync()
print('wit ok', flush=True)
l, eb, m = pp.l, pp.fr.nbytes, inst.m
A = lambda j, g2=False: wire.jacobian_to_affine(pp, j, g2)
# d_msm route vs plain msm route, per product
hsh = pp.alloc_fr(pp.n * (m // l))
pp._check(pp.lib.zk_circom_h(pp.h, wit.qap[0].ptr, wit.qap[1].ptr, wit.qap[2].ptr, log_m, None, 7, hsh.ptr, None))
pp.sync()
print('circom_h ok', flush=True)
h_pub = pp.unpack(hsh, m // l)
pp.sync()
print('unpack ok', flush=True)
items = (("a", ZK_G1, inst.w.view(eb), l * inst.len_a, inst.s, wit.a_share, inst.len_a),
         ("b", ZK_G1, inst.w.view(eb), l * inst.len_a, inst.h, wit.a_share, inst.len_a),
         ("b", ZK_G2, inst.w.view(eb), l * inst.len_a, inst.v, wit.a_share, inst.len_a),
         ("l", ZK_G1, inst.w.view(inst.ni * eb), l * inst.len_w, inst.wq, wit.ax_share, inst.len_w),
         ("h", ZK_G1, h_pub, m, inst.u, hsh, inst.len_u))
for name, group, scal, count, crsv, shares, ln in items[:int(os.environ.get('DBG_ITEMS', '5'))]:
    pts = inst.unpacked_points(name, group)
    pp.sync()
    print('pts ok', flush=True)
    loc = api.msm(pp, group, pts, scal, count)
    print('msm ok', flush=True)
    half = count // 2
    w = pp.fq.nbytes * (4 if group == ZK_G2 else 2)
    lo = api.msm(pp, group, pts.view(0, half * w), scal.view(0, half * eb), half)
    hi = api.msm(pp, group, pts.view(half * w), scal.view(half * eb), count - half)
    print('halves ok', flush=True)
    dm = zk.d_msm(pp, group, crsv, shares, ln)
    print('d_msm ok', flush=True)
    g2 = group == ZK_G2
    print(name, group, "count", count, "local==d_msm", A(loc, g2) == A(dm[0], g2), flush=True)
    # linearity of the local msm: lo + hi == whole, checked through d_msm-free path: msm over [pts_lo..], use the oracle add
    from oracle.curve import g1 as G1f, g2 as G2f
    from oracle import params
    cv = {"bn254": params.BN254, "bls12_381": params.BLS12_381}[curve]
    G = G2f(cv) if g2 else G1f(cv)
    sm = G.add(G.from_affine(A(lo, g2)) if hasattr(G, "from_affine") else A(lo, g2), G.from_affine(A(hi, g2)) if hasattr(G, "from_affine") else A(hi, g2))
    print("   halves sum == whole:", G.eq(sm, G.from_affine(A(loc, g2)) if hasattr(G, "from_affine") else A(loc, g2)), flush=True)
    pts.free()
