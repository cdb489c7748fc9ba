"""One G2 d_msm alone on the chip (counter / timing probe of the extension-field accumulate kernel):
   python tools/g2_solo.py <curve> <log2 points per party> [reps]      (ZK_SOLO_G1=1: the G1 kernel on the same sizes)"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zksaas_amd as zk
from zksaas_amd import groth16 as zg
from zksaas_amd.api import ZK_G1, ZK_G2

curve = sys.argv[1] if len(sys.argv) > 1 else "bls12_381"
ln = 1 << (int(sys.argv[2]) if len(sys.argv) > 2 else 19)
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
group = ZK_G1 if os.environ.get("ZK_SOLO_G1") else ZK_G2
pp = zk.PackedSharingParams(curve, 2)
rng = np.random.default_rng(4)


def rand(count):
    a = rng.integers(0, 1 << 62, size=(count, 4), dtype=np.uint64)
    a[:, 3] &= np.uint64((1 << 60) - 1)
    return zk.DeviceBuffer.from_numpy(pp, a)


tot = pp.n * ln
bases = zg.base_points(pp, group, rand(tot), tot)
sc = rand(tot)
pp._check(pp.lib.zk_profile_enable(pp.h, 1))
zk.d_msm(pp, group, bases, sc, ln)
pp.sync()
ts = []
for _ in range(reps):
    t0 = time.perf_counter()
    zk.d_msm(pp, group, bases, sc, ln)
    pp.sync()
    ts.append(time.perf_counter() - t0)
from bench import read_profile
prof = {e["kernel"]: round(e["total_ms"] / max(1, e["launches"]), 3) for e in read_profile(pp) if e["launches"]}
plan = zk.api.msm_plan(pp, group, tot)
print(json.dumps({"curve": curve, "group": "g1" if group == ZK_G1 else "g2", "points": tot, "ms": round(min(ts) * 1e3, 3),
                  "slots_ms_per_launch": prof, "plan": plan}))
