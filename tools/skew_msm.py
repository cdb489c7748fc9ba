"""One G1 MSM whose scalars repeat like a real witness's (60 % ones, 25 % fives, 5 % r - 1, the rest random) against one with
random scalars, BN254 and BLS12-381:   python tools/skew_msm.py [log2 points]      (python tools/ab_run.py <other build> tools/skew_msm.py .. for an A/B)"""
import json
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import zksaas_amd as zk  # noqa: E402
from zksaas_amd import groth16 as zg  # noqa: E402
from zksaas_amd.api import ZK_G1, msm  # noqa: E402

log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 22
n = 1 << log_n
out = {"points": n}
for curve in ("bn254", "bls12_381"):
    pp = zk.Context(curve, 2)
    rng = np.random.default_rng(5)

    def rand(count):
        a = rng.integers(0, 1 << 62, size=(count, 4), dtype=np.uint64)
        a[:, 3] &= np.uint64((1 << 58) - 1)
        return a
    pts = zg.base_points(pp, ZK_G1, zk.DeviceBuffer.from_numpy(pp, rand(n)), n)
    sc = rand(n)
    skew = sc.copy()
    sel = rng.random(n)
    for lo, hi, val in ((0.0, 0.6, 1), (0.6, 0.85, 5), (0.85, 0.9, pp.fr.p - 1)):
        skew[(sel >= lo) & (sel < hi)] = pp.fr.encode([val])[0]
    res = {}
    for name, arr in (("random", sc), ("repeating", skew)):
        d = zk.DeviceBuffer.from_numpy(pp, arr)
        msm(pp, ZK_G1, pts, d, n)
        t0 = time.perf_counter()
        for _ in range(5):
            msm(pp, ZK_G1, pts, d, n)
        res[name + "_ms"] = round((time.perf_counter() - t0) / 5 * 1e3, 3)
    out[curve] = res
    del pts
print(json.dumps(out))
