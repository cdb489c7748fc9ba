"""circom_h (groth16/src/ext_wit.rs:104-181) of the SHA-256 proof ALONE on the chip: wall per call and, under
`rocprofv3 --kernel-trace`, the lone duration of each of its seven launches (what they take without the four MSMs of a
proof competing for issue slots).  usage: python tools/circom_h_solo.py [reps]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zksaas_amd as zk
from zksaas_amd import groth16 as zg
from bench import build_inputs

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
pp = zk.PackedSharingParams("bn254", 2)
r1, w, setup, crs, wit, r, s = build_inputs(pp, zk)
masks = zg.ProofMasks(pp, wit.log_m, seed=77)
m, l = 1 << wit.log_m, pp.l
h = pp.alloc_fr(pp.n * (m // l))
out = {}
for label, mk in (("masks", masks.ct), ("zero_masks", None)):
    import ctypes as C
    arg = None if mk is None else C.byref(mk)
    for _ in range(5):
        pp._check(pp.lib.zk_circom_h(pp.h, wit.qap[0].ptr, wit.qap[1].ptr, wit.qap[2].ptr, wit.log_m, arg, 7, h.ptr, None))
    pp.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        pp._check(pp.lib.zk_circom_h(pp.h, wit.qap[0].ptr, wit.qap[1].ptr, wit.qap[2].ptr, wit.log_m, arg, 7, h.ptr, None))
    pp.sync()
    out[label + "_us_per_call_back_to_back"] = round((time.perf_counter() - t0) / reps * 1e6, 1)
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        pp._check(pp.lib.zk_circom_h(pp.h, wit.qap[0].ptr, wit.qap[1].ptr, wit.qap[2].ptr, wit.log_m, arg, 7, h.ptr, None))
        pp.sync()
        ts.append(time.perf_counter() - t0)
    out[label + "_us_per_call_synced_median"] = round(sorted(ts)[len(ts) // 2] * 1e6, 1)
print(json.dumps(out))
