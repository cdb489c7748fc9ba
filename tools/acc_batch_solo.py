"""Each MSM of the proof ALONE as a batch (zk_msm_batch over the CRS query and B copies of the witness shares): duration
and multiplication rate of the accumulate kernel with the chip to itself, sort and finalize+reduce slot times.
usage: python tools/acc_batch_solo.py [B]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import zksaas_amd as zk
from zksaas_amd import api
from zksaas_amd.api import ZK_G1, ZK_G2
from bench import build_inputs, read_profile

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
pp = zk.PackedSharingParams("bn254", 2)
r1, w, setup, crs, wit, r, s = build_inputs(pp, zk)
crs.precompute()
idf = {"S": 0.0, "H": 0.511, "V": 0.511, "W": 0.0}
for name, grp, bases, ln, scal in (("S", ZK_G1, crs.s, crs.len_a, wit.a_share), ("H", ZK_G1, crs.h, crs.len_a, wit.a_share),
                                   ("W", ZK_G1, crs.w, crs.len_w, wit.ax_share), ("V", ZK_G2, crs.v, crs.len_a, wit.a_share)):
    npts = pp.n * ln
    for _ in range(2):
        api.msm_batch(pp, grp, bases, [scal] * B, npts)
    pp._check(pp.lib.zk_profile_enable(pp.h, 1))
    reps = 5
    for _ in range(reps):
        api.msm_batch(pp, grp, bases, [scal] * B, npts)
    prof = {e["kernel"]: e for e in read_profile(pp)}
    pp._check(pp.lib.zk_profile_enable(pp.h, 0))
    acc = prof["msm_accumulate_kernel<G2>" if grp == ZK_G2 else "msm_accumulate_kernel<G1>"]
    red = prof["msm_finalize+reduce<G2>" if grp == ZK_G2 else "msm_finalize+reduce<G1>"]
    srt = prof["msm_digits+scan+expand"]
    us = acc["total_ms"] / acc["launches"] * 1e3
    nwin = api.msm_table_info(pp, grp, bases)["windows"]
    adds = npts * B * nwin * (1.0 - idf[name])
    muls = adds * (28 if grp == ZK_G2 else 10)
    print("%s batch %d: accumulate %.0f us = %.1f G modmul/s (%.0f M additions), sort %.0f us, finalize+reduce %.0f us" % (
        name, B, us, muls / us / 1e3, adds / 1e6, srt["total_ms"] / srt["launches"] * 1e3, red["total_ms"] / red["launches"] * 1e3),
        flush=True)
