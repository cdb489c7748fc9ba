"""One MSM at a time (nothing else on the chip): kernel-level view of a single Pippenger pipeline.
usage: G=1|2 [TABLES=1] [NPTS=119288] python tools/msm_solo.py      (run it under rocprofv3 for per-kernel times / counters)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import zksaas_amd as zk                          # noqa: E402
from zksaas_amd import api, groth16 as zg        # noqa: E402
from zksaas_amd.api import ZK_G1, ZK_G2, DeviceBuffer   # noqa: E402

pp = zk.PackedSharingParams(os.environ.get("CURVE", "bn254"), 2)
for kv in filter(None, os.environ.get("ZK_BENCH_OPTIONS", "").split(",")):     # A/B runs: name=value context options
    pp.set_option(kv.split("=")[0], int(kv.split("=")[1]))
n = int(os.environ.get("NPTS", 8 * 14911))
rng = np.random.default_rng(3)


def rand_fr(count):
    a = rng.integers(0, 1 << 62, size=(count, 4), dtype=np.uint64)
    a[:, 3] &= np.uint64((1 << 60) - 1)
    return DeviceBuffer.from_numpy(pp, a)


grp = ZK_G2 if os.environ.get("G", "2") == "2" else ZK_G1
sc = rand_fr(n)
bases = zg.base_points(pp, grp, rand_fr(n), n)
if os.environ.get("TABLES"):
    api.msm_precompute(pp, grp, bases, n)
for _ in range(3):
    api.msm(pp, grp, bases, sc, n)
pp.sync()
t0 = time.perf_counter()
reps = int(os.environ.get("REPS", "20"))
for _ in range(reps):
    api.msm(pp, grp, bases, sc, n)
pp.sync()
print("ms per msm %.3f (%d points, group %s, tables %s)" % ((time.perf_counter() - t0) / reps * 1e3, n,
                                                              "G2" if grp == ZK_G2 else "G1", bool(os.environ.get("TABLES"))))
