"""What ONE rank of an N-GPU SHA-256 proof spends on its five MSMs, measured on one GPU without any network: the MSM half
of zk_dist_groth16_prove (zk_groth16_msms_begin / _finish) over k = 8 / N parties' rows, fixed-base tables on.  The king
rounds of circom_h come on top (DESIGN.md section 7 prices them from the link bandwidth); this is the part of the strong-
scaling curve that can be measured here.   usage: python tools/rank_latency.py [reps]"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import zksaas_amd as zk                      # noqa: E402
from zksaas_amd import groth16 as zg        # noqa: E402
from zksaas_amd import multigpu as mg       # noqa: E402
from bench import build_inputs              # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
pp = zk.PackedSharingParams("bn254", 2)
if os.environ.get("ZK_TABLE_C"):
    pp.set_option("msm_table_c", int(os.environ["ZK_TABLE_C"]))
r1, w, setup, crs, wit, r, s = build_inputs(pp, zk)
Lc = (1 << wit.log_m) // pp.l
h_full = pp.alloc_fr(pp.n * Lc)
pp._check(pp.lib.zk_circom_h(pp.h, wit.qap[0].ptr, wit.qap[1].ptr, wit.qap[2].ptr, wit.log_m, None, 7, h_full.ptr, None))
pp.sync()
nl = pp.fq.nl
out = {}
for k in [int(x) for x in os.environ.get("RANK_K", "8,4,2,1").split(",")]:
    lcrs = mg.LocalCrs(pp, crs, 0, k)
    lcrs.precompute(pp)
    qap, a_sh, ax_sh = mg.local_witness(pp, wit, 0, k)
    h = mg.rows(h_full, 0, k, Lc * pp.fr.nbytes)
    bufs = [np.zeros(3 * nl, dtype=np.uint64), np.zeros(3 * nl, dtype=np.uint64), np.zeros(6 * nl, dtype=np.uint64),
            np.zeros(3 * nl, dtype=np.uint64), np.zeros(3 * nl, dtype=np.uint64)]
    arr = (C.c_void_p * 5)(*[b.ctypes.data for b in bufs])

    def once():
        pp._check(pp.lib.zk_groth16_msms_begin(pp.h, C.byref(lcrs.ct), a_sh.ptr, ax_sh.ptr, 0, k, 0, None, None))
        pp._check(pp.lib.zk_groth16_msms_finish(pp.h, h.ptr, arr, None))
    for _ in range(5):
        once()
        if os.environ.get("RANK_GAP"):           # profiling: idle gaps so that tools/timeline.py can tell the proofs apart
            pp.sync()
            time.sleep(0.003)
    pp.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        once()
    pp.sync()
    out["%d_parties_per_rank" % k] = {"gpus": 8 // k, "ms_per_proof_msm_part": round((time.perf_counter() - t0) / reps * 1e3, 3)}
print(json.dumps({"workload": "c4 (SHA-256, BN254, 29 823 wires), five MSMs of one rank, tables on", "results": out}))
