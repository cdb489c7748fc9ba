"""Solo duration of the accumulate kernels of the proof's MSMs (no other stream active): multiplications per second of
the G1 and G2 mixed-addition kernels when they have the chip to themselves.  usage: python tools/acc_solo.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zksaas_amd as zk
from zksaas_amd.api import ZK_G1, ZK_G2
from bench import build_inputs, read_profile

pp = zk.PackedSharingParams("bn254", 2)
r1, w, setup, crs, wit, r, s = build_inputs(pp, zk)
for tables in (False, True):
    if tables:
        crs.precompute()
    for name, grp, bases, ln in (("G1 (W)", ZK_G1, crs.w, crs.len_w), ("G2 (V)", ZK_G2, crs.v, crs.len_a)):
        scal = wit.ax_share if grp == ZK_G1 else wit.a_share
        for _ in range(3):
            zk.d_msm(pp, grp, bases, scal, ln)
        pp._check(pp.lib.zk_profile_enable(pp.h, 1))
        for _ in range(10):
            zk.d_msm(pp, grp, bases, scal, ln)
        prof = {e["kernel"]: e for e in read_profile(pp)}
        pp._check(pp.lib.zk_profile_enable(pp.h, 0))
        acc = prof["msm_accumulate_kernel<G2>" if grp == ZK_G2 else "msm_accumulate_kernel<G1>"]
        red = prof["msm_finalize+reduce<G2>" if grp == ZK_G2 else "msm_finalize+reduce<G1>"]
        srt = prof["msm_digits+scan+expand"]
        us = acc["total_ms"] / acc["launches"] * 1e3
        pts = acc["units"] / acc["launches"]
        plan = zk.api.msm_plan(pp, grp, int(pts))
        nwin = 16 if tables else plan["windows"]
        muls = pts * nwin * plan["muls_per_add"]
        print("%s tables=%d: accumulate %.0f us (%.1f G mul/s), sort %.0f us, finalize+reduce %.0f us" % (
            name, tables, us, muls / us / 1e3, srt["total_ms"] / srt["launches"] * 1e3, red["total_ms"] / red["launches"] * 1e3))
