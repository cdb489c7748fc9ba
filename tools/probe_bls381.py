"""BASELINE config 5 components at scale on BLS12-381 (12-limb base field): d_fft at m = 2^24, G1 / G2 MSMs.
Timing probe only (parity of these kernels is covered by tests/test_gpu_fullsize.py at smaller sizes)."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zksaas_amd as zk
from zksaas_amd import groth16 as zg
from zksaas_amd.api import ZK_G1, ZK_G2


def rand_fr(pp, count, seed):
    rng = np.random.default_rng(seed)
    a = rng.integers(0, 1 << 62, size=(count, 4), dtype=np.uint64)
    a[:, 3] &= np.uint64((1 << 60) - 1)
    return zk.DeviceBuffer.from_numpy(pp, a)


def med(pp, fn, reps=3):
    fn()
    pp.sync()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        pp.sync()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts))


pp = zk.PackedSharingParams("bls12_381", 2)
out = {}
for log_m in (20, 24):
    m = 1 << log_m
    sh, dst = rand_fr(pp, pp.n * m // 2, 1), pp.alloc_fr(pp.n * m // 2)
    t = med(pp, lambda: zk.d_fft(pp, sh, zk.FftMask.zero(), False, log_m, seed=3, out=dst))
    alg = 32 * m * 32
    out["d_fft_bls12_381_m2^%d" % log_m] = {"ms": round(t * 1e3, 3), "achieved_GBps": round(alg / t / 1e9, 1),
                                            "frac_hbm": round(alg / t / 8e12, 4)}
    del sh, dst
for grp, name, ln in ((ZK_G1, "g1", 1 << 18), (ZK_G2, "g2", 1 << 16)):
    sc = rand_fr(pp, pp.n * ln, 2)
    pts = zg.base_points(pp, grp, rand_fr(pp, 4096, 4), 4096)          # 4096 distinct multiples of the generator
    width = (4 if grp == ZK_G2 else 2) * pp.fq.nl
    rows = pts.to_numpy().reshape(4096, width)
    bases = zk.DeviceBuffer.from_numpy(pp, np.tile(rows, (pp.n * ln // 4096, 1)))
    t = med(pp, lambda: zk.d_msm(pp, grp, bases, sc, ln))
    out["d_msm_bls12_381_%s_8x2^%d" % (name, ln.bit_length() - 1)] = {"ms": round(t * 1e3, 2),
                                                                     "points_per_s": round(pp.n * ln / t / 1e6, 1)}
print(json.dumps(out))
