"""d_pp (dist-primitives/src/dpp/mod.rs:15-87) timed on one GPU, all n parties on the device.
usage: python tools/dpp_bench.py [CURVE LOG_M [REPS]] ...  -> one JSON line per (curve, log_m)
Per case: wall per call (sync on both sides, zero masks and sampled masks), per-kernel average launch durations from
the library's HIP-event slots, algorithmic bytes (3 n (m/l) B + 2 m B) and the telescoping check
(num_i = x_(i+1), den_i = x_i  =>  prefix_i * x_0 = x_(i+1))."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zksaas_amd as zk
from zksaas_amd import synthetic


def slots(pp):
    out = {}
    for slot in range(pp.lib.zk_profile_slots()):
        ms, units, calls = C.c_double(), C.c_double(), C.c_long()
        pp._check(pp.lib.zk_profile_read(pp.h, slot, C.byref(ms), C.byref(units), C.byref(calls)))
        if calls.value:
            out[pp.lib.zk_profile_name(slot).decode()] = round(ms.value / calls.value * 1e3, 1)   # us per launch
    return out


def run(curve, log_m, reps):
    pp = zk.PackedSharingParams(curve, 2)
    for kv in filter(None, os.environ.get("ZK_BENCH_OPTIONS", "").split(",")):       # A/B runs: name=value context options
        pp.set_option(kv.split("=")[0], int(kv.split("=")[1]))
    m, l, eb = 1 << log_m, pp.l, pp.fr.nbytes
    x = synthetic.rand_fr_device(pp, m + 1, 77)
    num_sh, den_sh = pp.pack(x.view(eb), m // l, 78), pp.pack(x, m // l, 79)
    out = {"curve": curve, "log_m": log_m, "reps": reps}
    alg = 3 * pp.n * (m // l) * eb + 2 * m * eb
    out["algorithmic_bytes"] = alg
    for label, mask in (("zero_masks", zk.DegRedMask.zero()), ("masks", zk.DegRedMask.sample(pp, m // l, 81))):
        res = zk.d_pp(pp, num_sh, den_sh, mask, m // l, seed=80)      # warm-up (tables, workspaces)
        pp.sync()
        pp._check(pp.lib.zk_profile_enable(pp.h, 1))
        t0 = time.perf_counter()
        for _ in range(reps):
            zk.d_pp(pp, num_sh, den_sh, mask, m // l, seed=80, out=res)
        pp.sync()
        dt = (time.perf_counter() - t0) / reps
        out[label] = {"ms": round(dt * 1e3, 4), "algorithmic_TBps": round(alg / dt / 1e12, 3),
                      "frac_hbm": round(alg / dt / 8e12, 4), "kernels_us": slots(pp)}
        pp._check(pp.lib.zk_profile_enable(pp.h, 0))
        if label == "masks":
            mask.in_mask.free(), mask.out_mask.free()
        else:
            prod = pp.unpack(res, m // l)
            x0 = pp.download_fr(x, 1)[0]
            zk.api.vec_scale(pp, prod, x0, m)
            out["telescopes"] = bool(np.array_equal(prod.to_numpy()[: m * pp.fr.nl],
                                                    x.to_numpy()[pp.fr.nl:(m + 1) * pp.fr.nl]))
            prod.free()
        res.free()
    print(json.dumps(out), flush=True)
    return out.get("telescopes", False)


if __name__ == "__main__":
    args = sys.argv[1:] or ["bn254", "20", "20", "bls12_381", "24", "5"]
    ok = True
    for i in range(0, len(args), 3):
        ok &= run(args[i], int(args[i + 1]), int(args[i + 2]) if i + 2 < len(args) else 10)
    sys.exit(0 if ok else 1)
