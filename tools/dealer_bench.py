"""The dealer's kernels, timed on one GPU: what the reference does ONCE per circuit / per proof
before the parties start, and calls its slowest step (`groth16/src/proving_key.rs:47-123`, `.github/workflows/ci.yml:54-67`).

usage: python tools/dealer_bench.py [--quick] [--reps R]      -> ONE JSON line (`bench.py --workload dealer` prints the same)

Keys (per entry: wall per call with a sync on both sides, units / s, the base-field products one unit costs as the
kernel executes them, and `frac_issue_bound` = products / s over the multiplier's issue bound -- 153.6 G/s for 8 limbs,
68.3 G/s for 12 limbs; these kernels are multiplier-bound, HBM traffic is a few hundred bytes per ~10^3 products):
  crs_pack_points   `pack_from_arkworks_proving_key` at the SHA-256 circuit's sizes: det_pack over group elements of
                    a_query[1..], b_g1_query[1..], l_query, h_query (G1) and b_g2_query[1..] (G2), chunks of l = 2 -> n = 8
  fixed_base_mul    scalars -> multiples of a generator (the trapdoor dealer; `proving_key.rs:125-176` `rand()` dummy CRS)
  msm_table_build   zk_msm_precompute over the five packed query vectors
  fft_mask_sample   `FftMask::sample` (`dfft/mod.rs:30-85`) at 2^15 and 2^20
  degred_mask_sample / msm_mask_sample   `deg_red.rs:40-66`, `dmsm/mod.rs:21-47`
"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zksaas_amd as zk
from zksaas_amd import api, synthetic
from zksaas_amd import groth16 as zg

ISSUE_BOUND = {8: 153.6e9, 12: 153.6e9 * (2 * 64 + 8) / (2 * 144 + 12)}     # products / s: 2 N^2 + N multiply instructions each
# base-field products per group operation as executed (api.MULS_PER_ADD; doubling dbl-2008-s-1: 8.5 / 3 x for Fq2)
MADD = {api.ZK_G1: 9.47, api.ZK_G2: 23.76}
DBL = {api.ZK_G1: 8.5, api.ZK_G2: 21.0}


def timed(pp, fn, reps):
    fn()
    pp.sync()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        pp.sync()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    return ts[len(ts) // 2]


def entry(dt, units, unit_name, products_per_unit, limbs, **extra):
    e = {"ms": round(dt * 1e3, 4), unit_name + "_per_s": round(units / dt, 1),
         "products_per_" + unit_name: round(products_per_unit, 1),
         "frac_issue_bound": round(units * products_per_unit / dt / ISSUE_BOUND[limbs], 4)}
    e.update(extra)
    return e


def random_points(pp, group, count, seed):
    return zg.base_points(pp, group, synthetic.rand_fr_device(pp, count, seed), count)


def pack_points_cost(group, nv, inv):
    """products per OUTPUT share of the form the kernel runs (see csrc/groth16.hpp); inv = products of one inversion"""
    ext = 3 if group == api.ZK_G2 else 1
    if nv == 2:      # joint sparse form: 257 doublings, ~128 mixed additions, the two-sum table (one inversion), normalisation
        return 257 * DBL[group] + 128.5 * MADD[group] + (2 * inv + 10) * ext
    return 256 * DBL[group] + 128 * nv * MADD[group] + inv * ext


def run(curve, quick, reps):
    pp = zk.PackedSharingParams(curve, 2)
    for kv in filter(None, os.environ.get("ZK_BENCH_OPTIONS", "").split(",")):     # A/B runs: name=value context options
        pp.set_option(kv.split("=")[0], int(kv.split("=")[1]))
    limbs = pp.fq.nl * 2                    # 32-bit limbs of the base field
    out = {"curve": curve, "base_field_limbs": limbs}
    # what the kernels run (csrc/groth16.hpp, csrc/ec.hpp, round 6): divstep inversion (~35 / 25 product equivalents of issue on
    # 8 / 12 limbs), 16-bit windows from 2^19 scalars up, joint-sparse-form packing
    info = {"inversion": "safegcd", "base_mul_wide_from": 1 << 19, "pack_points_form": "jsf"}
    out["build"] = info
    inv = 35.0 if limbs == 8 else 25.0
    # ---- CRS share packing at the SHA-256 circuit's sizes (proving_key.rs:47-123)
    sizes = [("a_query", api.ZK_G1, 14911), ("b_g1_query", api.ZK_G1, 14911), ("l_query", api.ZK_G1, 14910),
             ("h_query", api.ZK_G1, 16384), ("b_g2_query", api.ZK_G2, 14911)]
    if curve != "bn254" or quick:
        sizes = [("h_query", api.ZK_G1, 16384), ("b_g2_query", api.ZK_G2, 14911)]
    crs, total = {}, 0.0
    for name, grp, nch in sizes:
        pts = random_points(pp, grp, nch * pp.l, 100 + nch)
        dt = timed(pp, lambda: zg.pack_points(pp, grp, pts, nch, pp.l).free(), reps)
        crs[name] = entry(dt, nch * pp.n, "share", pack_points_cost(grp, pp.l, inv), limbs, chunks=nch,
                          group="G2" if grp == api.ZK_G2 else "G1")
        total += dt
        pts.free()
    out["crs_pack_points"] = {"vectors": crs, "total_ms": round(total * 1e3, 3)}

    # ---- fixed-base multiplication
    fb = {}
    big = 20 if quick else (22 if curve == "bn254" else 24)
    for grp, gname in ((api.ZK_G1, "G1"), (api.ZK_G2, "G2")):
        for log_n in (17, big):
            cnt = 1 << log_n
            sc = synthetic.rand_fr_device(pp, cnt, 7 + log_n)
            dt = timed(pp, lambda: zg.base_points(pp, grp, sc, cnt).free(), max(2, reps // 2))
            nwin = 32
            if info.get("base_mul_wide_from") and cnt >= info["base_mul_wide_from"]:
                nwin = 16
            fb["%s_2^%d" % (gname, log_n)] = entry(dt, cnt, "point", nwin * MADD[grp] + inv * (3 if grp == api.ZK_G2 else 1),
                                                   limbs, windows=nwin)
            sc.free()
    out["fixed_base_mul"] = fb

    # ---- fixed-base MSM tables over packed query vectors (zk_msm_precompute)
    tb = {}
    for grp, gname, nch in ((api.ZK_G1, "G1", 16384), (api.ZK_G2, "G2", 14911)):
        pts = random_points(pp, grp, nch * pp.n, 300 + nch)

        def build():
            api.msm_precompute(pp, grp, pts, nch * pp.n)
            api.msm_forget(pp, pts)
        dt = timed(pp, build, max(2, reps // 2))
        api.msm_precompute(pp, grp, pts, nch * pp.n)
        ti = api.msm_table_info(pp, grp, pts)
        api.msm_forget(pp, pts)
        tb[gname] = entry(dt, nch * pp.n, "point", 254 * DBL[grp] + ti["windows"] * inv * (3 if grp == api.ZK_G2 else 1), limbs, **ti)
        pts.free()
    out["msm_table_build"] = tb

    # ---- masks
    fm = {}
    for log_m in ((15,) if quick else (15, 20)):
        for rearr, inverse in ((True, True), (False, False)):
            g = 5 if inverse else None

            def sample():
                m_ = zk.FftMask.sample(pp, rearr, g, inverse, log_m, 9)
                m_.in_mask.free(), m_.out_mask.free()
            dt = timed(pp, sample, reps)
            m = 1 << log_m
            fm["2^%d_%s" % (log_m, "ifft_rearranged" if inverse else "fft")] = {
                "ms": round(dt * 1e3, 4), "elements_per_s": round(m / dt, 1),
                "algorithmic_bytes": 2 * pp.n * (m // pp.l) * 32, "frac_hbm": round(2 * pp.n * (m // pp.l) * 32 / dt / 8e12, 4)}
    out["fft_mask_sample"] = fm

    def dm():
        k = zk.DegRedMask.sample(pp, 1 << 14, 10)
        k.in_mask.free(), k.out_mask.free()
    dt = timed(pp, dm, reps)
    out["degred_mask_sample_2^14"] = {"ms": round(dt * 1e3, 4), "chunks_per_s": round((1 << 14) / dt, 1)}
    gen = zg._affine_codec(pp, list(zg.G1_GEN[pp.curve]), False)
    dt = timed(pp, lambda: zk.MsmMask.sample(pp, api.ZK_G1, gen, 11), reps)
    out["msm_mask_sample_g1"] = {"ms": round(dt * 1e3, 4)}
    return out


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    quick = "--quick" in argv
    reps = int(argv[argv.index("--reps") + 1]) if "--reps" in argv else 5
    res = {"workload": "dealer (SURVEY.md 8 f1 / f2): CRS share packing, fixed-base multiplication, table build, mask sampling",
           "curves": [run("bn254", quick, reps), run("bls12_381", quick, reps)]}
    print(json.dumps(res), flush=True)
    return res


if __name__ == "__main__":
    main()
