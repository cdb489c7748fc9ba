"""Experiment: proofs/s with K proofs in flight (one context + host thread each, shared read-only CRS/witness)."""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zksaas_amd as zk
from zksaas_amd import groth16 as zg
from bench import build_inputs

pp = zk.PackedSharingParams("bn254", 2)
r1, w, setup, crs, wit, r, s = build_inputs(pp, zk)
for inflight in (1, 2, 3, 4):
    ctxs = [pp] + [zk.PackedSharingParams("bn254", 2) for _ in range(inflight - 1)]
    for c in ctxs:
        zg.prove(c, crs, wit, r, s, seed=1)
    total = 48
    per = total // inflight

    def work(c):
        for _ in range(per):
            zg.prove(c, crs, wit, r, s, seed=1)

    t0 = time.perf_counter()
    ths = [threading.Thread(target=work, args=(c,)) for c in ctxs]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    dt = time.perf_counter() - t0
    print("inflight", inflight, "proofs/s %.1f" % (per * inflight / dt), "ms/proof amortised %.2f" % (dt / (per * inflight) * 1e3))
