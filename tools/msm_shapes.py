"""Sweep of MSM launch shapes against the plain table-free single-vector MSM (which the parity tests pin on the oracle):
sizes around the sort classes' thresholds x batch sizes x fixed-base tables of several window sizes x G1 / G2.
    python tools/msm_shapes.py [quick]
Prints one line per shape and a final count; exits non-zero on any mismatch."""
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import zksaas_amd as zk  # noqa: E402
from zksaas_amd import api, groth16 as zg, wire  # noqa: E402
from zksaas_amd.api import ZK_G1, ZK_G2  # noqa: E402

quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
pp = zk.Context("bn254", 2)
rng = np.random.default_rng(11)


def rand(count):
    a = rng.integers(0, 1 << 62, size=(count, 4), dtype=np.uint64)
    a[:, 3] &= np.uint64((1 << 58) - 1)
    return a


def sweep(group, sizes, batches, tables):
    g2 = group == ZK_G2
    bad = 0
    nmax = max(sizes)
    pts_all = zg.base_points(pp, group, zk.DeviceBuffer.from_numpy(pp, rand(nmax)), nmax)
    scs = [zk.DeviceBuffer.from_numpy(pp, rand(nmax)) for _ in range(max(batches))]
    aff = lambda j: wire.jacobian_to_affine(pp, j, g2)
    for n in sizes:
        want = [aff(api.msm(pp, group, pts_all, s, n)) for s in scs]
        for tc in tables:
            if tc:
                pp.set_option("msm_table_c", tc)
                api.msm_precompute(pp, group, pts_all, n)
            try:
                for nb in batches:
                    t0 = time.perf_counter()
                    if nb == 1:
                        got = [aff(api.msm(pp, group, pts_all, scs[0], n))]
                    else:
                        got = [aff(j) for j in api.msm_batch(pp, group, pts_all, scs[:nb], n)]
                    ok = got == want[:nb]
                    bad += 0 if ok else 1
                    print("%s n=%d table=%s batch=%d %s %.1f ms" % ("G2" if g2 else "G1", n, tc or "-", nb,
                                                                    "ok" if ok else "MISMATCH", (time.perf_counter() - t0) * 1e3), flush=True)
            finally:
                if tc:
                    api.msm_forget(pp, pts_all)
    return bad


bad = 0
if quick:
    bad += sweep(ZK_G1, [(1 << 19) + 123, 1 << 22], [1, 3], [0, 16])
else:
    bad += sweep(ZK_G1, [(1 << 19) + 123, (1 << 20) + 1, (1 << 21) + 7, 1 << 22, (1 << 22) + (1 << 20) + 5, 1 << 23],
                 [1, 2, 3, 8], [0, 16, 20])
    bad += sweep(ZK_G2, [(1 << 19) + 123, (1 << 21) + 7, 1 << 22], [1, 3, 8], [0, 15, 19])
print("mismatches:", bad)
sys.exit(1 if bad else 0)
