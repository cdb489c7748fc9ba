"""BASELINE configs[4] / SURVEY.md 8d "C5": a full distributed Groth16 proof on BLS12-381 at m = 2^LOG_M (default 24:
16.7 M constraints, n = 8 parties, l = 2, all parties on ONE GPU), synthetic instance built on the device
(zksaas_amd/synthetic.py).  Prints one JSON line with timings and two size-independent checks:
  * the proof does not depend on the share randomness (two dealings of the same witness);
  * d_pp at the same size telescopes (num_i = x_(i+1), den_i = x_i  =>  product_i * x_0 = x_(i+1));
  * distributed == local: zk_groth16_assemble over five PLAIN zk_msm's of the public query elements and the public
    witness (h taken from the unpacked zk_circom_h output) is the same proof.
usage: python tools/c5_bls381.py [LOG_M] > gpurun_out/c5.json"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zksaas_amd as zk
from zksaas_amd import groth16 as zg
from zksaas_amd import synthetic, wire
from zksaas_amd.api import ZK_G1, ZK_G2

log_m = int(sys.argv[1]) if len(sys.argv) > 1 else 24
curve = sys.argv[2] if len(sys.argv) > 2 else "bls12_381"
pp = zk.PackedSharingParams(curve, 2)
out = {"curve": curve, "log_m": log_m, "constraints": (1 << log_m) - 2, "parties": pp.n, "l": pp.l}
t0 = time.perf_counter()
inst = synthetic.SyntheticInstance(pp, log_m, seed=1)
pp.sync()
out["build_crs_s"] = round(time.perf_counter() - t0, 2)
t0 = time.perf_counter()
wit = inst.witness(seed=100)
pp.sync()
out["deal_witness_s"] = round(time.perf_counter() - t0, 3)
r, s = 0x1234567890ABCDEF1234567890ABCDEF, 0xFEDCBA0987654321FEDCBA0987654321
ts = []
for i in range(3):
    t0 = time.perf_counter()
    proof = zg.prove(pp, inst.crs, wit, r, s, seed=7)
    ts.append(time.perf_counter() - t0)
out["prove_s"] = [round(t, 3) for t in ts]
out["constraints_per_s"] = round(out["constraints"] / min(ts), 1)


def norm(pf):
    return (wire.jacobian_to_affine(pp, pf[0][0], False), wire.jacobian_to_affine(pp, pf[1][0], True),
            wire.jacobian_to_affine(pp, pf[2][0], False))


ref = norm(proof)
out["all_parties_equal"] = all(np.array_equal(proof[k][0], proof[k][p]) for k in range(3) for p in range(pp.n))
# (1) another dealing of the same witness (different share randomness everywhere)
wit2 = inst.witness(seed=900)
out["independent_of_share_randomness"] = norm(zg.prove(pp, inst.crs, wit2, r, s, seed=8)) == ref
del wit2
# (2) local prover from plain MSMs
m, l, eb = inst.m, pp.l, pp.fr.nbytes
cnt = pp.n * (m // l)
hsh = pp.alloc_fr(cnt)
pp._check(pp.lib.zk_circom_h(pp.h, wit.qap[0].ptr, wit.qap[1].ptr, wit.qap[2].ptr, log_m, None, 7, hsh.ptr, None))
h_pub = pp.unpack(hsh, m // l)
t0 = time.perf_counter()
sums = []
for name, group, scal, count in (("a", ZK_G1, inst.w.view(eb), l * inst.len_a), ("b", ZK_G1, inst.w.view(eb), l * inst.len_a),
                                 ("b", ZK_G2, inst.w.view(eb), l * inst.len_a),
                                 ("l", ZK_G1, inst.w.view(inst.ni * eb), l * inst.len_w), ("h", ZK_G1, h_pub, m)):
    pts = inst.unpacked_points(name, group)
    sums.append(zk.api.msm(pp, group, pts, scal, count))
    pts.free()
pp.sync()
out["local_msms_s"] = round(time.perf_counter() - t0, 3)
nl = pp.fq.nl
pa, pb, pc = (np.zeros((pp.n, 3 * nl), dtype=np.uint64), np.zeros((pp.n, 6 * nl), dtype=np.uint64),
              np.zeros((pp.n, 3 * nl), dtype=np.uint64))
rr, ss = pp.fr.encode_one(r), pp.fr.encode_one(s)
arr = (C.c_void_p * 5)(*[x.ctypes.data for x in sums])
pp._check(pp.lib.zk_groth16_assemble(pp.h, C.byref(inst.crs.ct), rr.ctypes.data, ss.ctypes.data, arr, None,
                                     pa.ctypes.data, pb.ctypes.data, pc.ctypes.data))
out["distributed_equals_local"] = norm((pa, pb, pc)) == ref
# (3) d_pp at the same size (dpp/mod.rs:15-87): num_i = x_(i+1), den_i = x_i  =>  prefix product_i = x_(i+1) / x_0
del wit, hsh, h_pub
x = synthetic.rand_fr_device(pp, m + 1, 77)
num_sh, den_sh = pp.pack(x.view(eb), m // l, 78), pp.pack(x, m // l, 79)
zk.d_pp(pp, num_sh, den_sh, zk.DegRedMask.zero(), m // l, seed=80).free()      # warm-up (tables, workspaces)
pp.sync()
t0 = time.perf_counter()
res = zk.d_pp(pp, num_sh, den_sh, zk.DegRedMask.zero(), m // l, seed=80)
pp.sync()
out["d_pp_s"] = round(time.perf_counter() - t0, 4)
prod = pp.unpack(res, m // l)
x0 = pp.download_fr(x, 1)[0]
zk.api.vec_scale(pp, prod, x0, m)
out["d_pp_telescopes"] = bool(np.array_equal(prod.to_numpy()[: m * pp.fr.nl], x.to_numpy()[pp.fr.nl:(m + 1) * pp.fr.nl]))
out["proof_compressed_hex"] = wire.proof_to_bytes(pp, proof[0][0], proof[1][0], proof[2][0]).hex()
print(json.dumps(out))
sys.exit(0 if (out["independent_of_share_randomness"] and out["distributed_equals_local"] and out["d_pp_telescopes"]) else 1)
