import os, sys
import numpy as np
sys.path.insert(0, os.environ.get('DBG_ROOT') or os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["ZK_RNG_REPLAY"] = "1"
import zksaas_amd as zk
from zksaas_amd import wire, api
from zksaas_amd import groth16 as zg
from zksaas_amd.api import ZK_G1, ZK_G2, DeviceBuffer
from oracle.curve import g1 as G1f
from oracle import params
curve = sys.argv[1]
sizes = [int(x) for x in sys.argv[2].split(",")]
pp = zk.PackedSharingParams(curve, 2)
if os.environ.get("DBG_BIGSORT") is not None:
    pp.set_option("msm_bigsort_min", int(os.environ["DBG_BIGSORT"]))
cv = {"bn254": params.BN254, "bls12_381": params.BLS12_381}[curve]
G = G1f(cv)
nl = pp.fr.nl
rng = np.random.default_rng(5)
def rand(count):
    a = rng.integers(0, 1 << 62, size=(count, nl), dtype=np.uint64)
    a[:, nl - 1] &= np.uint64((1 << 58) - 1)
    return a
mx = max(sizes)
eb, w = pp.fr.nbytes, 2 * pp.fq.nbytes
pts = zg.base_points(pp, ZK_G1, DeviceBuffer.from_numpy(pp, rand(mx)), mx)
sc = DeviceBuffer.from_numpy(pp, rand(mx))
A = lambda j: G.from_affine(wire.jacobian_to_affine(pp, j, False))
CH = 8192
parts = []
acc = G.identity
pref = {0: acc}
for c0 in range(0, mx, CH):
    n_ = min(CH, mx - c0)
    acc = G.add(acc, A(api.msm(pp, ZK_G1, pts.view(c0 * w, n_ * w), sc.view(c0 * eb, n_ * eb), n_)))
    pref[c0 + n_] = acc
for n_ in sizes:
    whole = A(api.msm(pp, ZK_G1, pts.view(0, n_ * w), sc.view(0, n_ * eb), n_))
    # reference: prefix over whole chunks + remainder chunk
    full = (n_ // CH) * CH
    ref = pref[full]
    if n_ > full:
        ref = G.add(ref, A(api.msm(pp, ZK_G1, pts.view(full * w, (n_ - full) * w), sc.view(full * eb, (n_ - full) * eb), n_ - full)))
    print(curve, n_, api.msm_plan(pp, ZK_G1, n_), "OK" if G.eq(whole, ref) else "WRONG", flush=True)
