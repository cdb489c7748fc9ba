"""Per-kernel HBM traffic from two rocprofv3 counter passes (FETCH_SIZE and WRITE_SIZE collected separately, as the
MI355X guide prescribes: they do not fit one pass) -> profiles/<name>.json, the file bench.py's `roofline.traffic` reads.

Corrections (MI355X_MICROARCH.md, HBM section): the counters are reported in KB; on gfx950 FETCH_SIZE reports half of
the bytes of wide coalesced streaming reads (16 bytes per lane) and "other access widths are uncalibrated".  Round 5
calibrates: tools/pmc_calib.hip moves a known byte count in each access shape the library's kernels use (32-byte elements
as two 16-byte accesses, the NTT tile loads of a first / later pass, the king kernels' eight party rows, 64-byte point
gathers) and tools/pmc_calibrate.py records known / reported per shape (profiles/r05_pmc_calibration.json); every kernel
is corrected by the factor of ITS shape (`access_shape`, `fetch_factor`, `write_factor` in the output).

usage: python tools/pmc_summary.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass> <out.json> [note]"""
import csv
import glob
import json
import re
import sys
from collections import defaultdict

# kernel-name prefix -> access shape of its dominant loads (tools/pmc_calib.hip measures the counter factor of each shape;
# profiles/r05_pmc_calibration.json).  Without a calibration file the guide's rule is used: x2 for 16-byte-per-lane
# streaming loads ("lane16"-like shapes), x1 otherwise.
SHAPE = [("ntt_pass_kernel", "ntt"), ("king_fft2_kernel", "rows8"), ("king_degred_kernel", "rows8"),
         ("dpp_tile_kernel", "rows8"), ("dpp_finish_kernel", "elem32"), ("dpp_carry_kernel", "elem32"), ("vec_", "elem32"),
         ("pss_", "rows8"), ("bitrev_kernel", "elem32"), ("r1cs_qap_kernel", "elem32"), ("msm_scatter_kernel<", "elem32"),
         ("msm_hist_kernel", "elem32"), ("msm_accumulate_kernel<Fp<", "gather64")]
CALIB = None
for cand in ("profiles/r05_pmc_calibration.json", "profiles/r06_pmc_calibration.json"):   # the last one found wins
    try:
        import os
        CALIB = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), cand)))["shapes"]
    except (OSError, ValueError, KeyError):
        pass


def factors(kernel):
    """(shape, fetch factor, write factor) of a kernel"""
    shape = next((sh for pre, sh in SHAPE if kernel.startswith(pre)), None)
    if shape is None:
        return None, 1.0, 1.0
    if CALIB is None:
        return shape, (1.0 if shape == "gather64" else 2.0), 1.0
    if shape == "ntt":          # the passes of one transform share a kernel name: the mean of the first / later pass shapes
        f = (CALIB["ntt_pass0"]["fetch_factor"] + CALIB["ntt_pass1"]["fetch_factor"]) / 2
        w = ((CALIB["ntt_pass0"]["write_factor"] or 1.0) + (CALIB["ntt_pass1"]["write_factor"] or 1.0)) / 2
        return shape, f, w
    return shape, CALIB[shape]["fetch_factor"], CALIB[shape]["write_factor"] or 1.0


def norm(name):
    name = re.sub(r"^void ", "", name)
    name = name.replace("zk::", "")
    return re.sub(r"\(.*$", "", name).strip()


def collect(d, counter):
    agg = defaultdict(lambda: [0, 0.0])
    for path in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(path)):
            if row["Counter_Name"] != counter:
                continue
            k = norm(row["Kernel_Name"])
            agg[k][0] += 1
            agg[k][1] += float(row["Counter_Value"])
    return agg


fetch, write = collect(sys.argv[1], "FETCH_SIZE"), collect(sys.argv[2], "WRITE_SIZE")
kernels = []
for k in sorted(fetch, key=lambda k: -fetch[k][1]):
    n = fetch[k][0]
    wn, wv = write.get(k, [0, 0.0])
    f_kb = fetch[k][1] / n
    w_kb = wv / wn if wn else 0.0
    shape, ff, wf = factors(k)
    kernels.append({"kernel": k, "launches": n, "FETCH_SIZE_KB_per_launch": round(f_kb, 1),
                    "WRITE_SIZE_KB_per_launch": round(w_kb, 1), "access_shape": shape, "fetch_factor": round(ff, 3),
                    "write_factor": round(wf, 3), "fetch_corrected": abs(ff - 1.0) > 0.05,
                    "hbm_bytes_per_launch": int((ff * f_kb + wf * w_kb) * 1024)})
out = {"note": (sys.argv[4] if len(sys.argv) > 4 else "") + " -- rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate "
               "passes (program directly after `--`); KB as reported; hbm_bytes_per_launch = fetch_factor x FETCH + write_factor "
               "x WRITE with the factors of the kernel's access shape measured by tools/pmc_calib.hip (%s)" % (
                   "profiles/r06_pmc_calibration.json (r05 on a tree without it)" if CALIB else "no calibration file: the guide's x2 / x1 rule"),
       "kernels": kernels}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(kernels[:8], indent=1))
