"""BASELINE configs 2 and 3 as stand-alone primitives: d_fft (m = 2^20, BN254, l = 2, n = 8) and d_msm (2^20 G1 points
per party), GPU vs the plain-C port on this host, with the algorithmic-byte accounting of SURVEY.md 8d.
Writes one JSON object to stdout (committed under profiles/)."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zksaas_amd as zk
from zksaas_amd.api import ZK_G1
from oracle.cref import CPss
from oracle.field import Domain
from oracle.params import BN254


def rand_fr(count, seed):
    rng = np.random.default_rng(seed)
    a = rng.integers(0, 1 << 62, size=(count, 4), dtype=np.uint64)
    a[:, 3] &= np.uint64((1 << 60) - 1)
    return a


def gpu_time(fn, pp, reps=10):
    fn()
    pp.sync()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        pp.sync()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts))


def main():
    pp = zk.PackedSharingParams("bn254", 2)
    cp = CPss("bn254", 2)
    out = {"host_cpus": os.cpu_count()}
    # ---- config 2: d_fft 2^20
    log_m = 20
    m = 1 << log_m
    B = 32
    shares = rand_fr(pp.n * m // 2, 1)
    buf = zk.DeviceBuffer.from_numpy(pp, shares)
    dst = pp.alloc_fr(pp.n * m // 2)
    t_gpu = gpu_time(lambda: zk.d_fft(pp, buf, zk.FftMask.zero(), False, log_m, seed=3, out=dst), pp)
    dom = Domain(BN254, m)
    work = shares.copy()
    t0 = time.perf_counter()
    cp.d_fft_arrays(work, m // 2, dom.group_gen, None, None, False, None, None, 3)
    t_cpu = time.perf_counter() - t0
    alg = 32 * m * B        # d_fft end-to-end, all parties = 32 m B
    out["d_fft_2^20"] = {"gpu_ms": round(t_gpu * 1e3, 3), "cpu_port_1thread_ms": round(t_cpu * 1e3, 1),
                         "algorithmic_bytes": alg, "achieved_GBps": round(alg / t_gpu / 1e9, 1),
                         "frac_of_8TBps": round(alg / t_gpu / 8e12, 4),
                         "modmul_estimate_per_s": round((pp.n * (m // 2) * 10.5 + (m // 2) * 32) / t_gpu / 1e9, 1)}
    # ---- config 3: d_msm, 2^20 G1 points per party (8 parties -> one fused 2^23-point Pippenger)
    ln = 1 << 20
    chain = cp.doubling_chain_g1(BN254.g1, ln)
    bases = zk.DeviceBuffer.from_numpy(pp, np.tile(chain, (pp.n, 1)))
    sc = rand_fr(pp.n * ln, 2)
    scal = zk.DeviceBuffer.from_numpy(pp, sc)
    t_gpu = gpu_time(lambda: zk.d_msm(pp, ZK_G1, bases, scal, ln), pp, reps=5)
    t0 = time.perf_counter()
    cp.msm_g1_arrays(chain, sc[:ln], ln, nthreads=1)
    t_cpu1 = time.perf_counter() - t0
    alg = pp.n * ln * 96
    out["d_msm_8x2^20"] = {"gpu_ms": round(t_gpu * 1e3, 2), "cpu_port_one_party_1thread_s": round(t_cpu1, 2),
                           "cpu_port_all_parties_8threads_s_estimate": round(t_cpu1, 2),
                           "algorithmic_bytes": alg, "achieved_GBps": round(alg / t_gpu / 1e9, 1),
                           "frac_of_8TBps": round(alg / t_gpu / 8e12, 5)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
