"""Per-kernel HBM traffic from two rocprofv3 counter passes (FETCH_SIZE and WRITE_SIZE collected separately, as the
MI355X guide prescribes: they do not fit one pass) -> profiles/<name>.json, the file bench.py's `roofline.traffic` reads.

Corrections (MI355X_MICROARCH.md, HBM section): the counters are reported in KB; on gfx950 FETCH_SIZE reports half of
the bytes of wide coalesced streaming reads (16 bytes per lane) -- doubled here for the kernels whose loads are of that
kind (NTT passes, king kernels, vector helpers: every lane reads whole 32-byte elements as two 16-byte accesses of
consecutive addresses); the MSM kernels gather 64/128-byte points at random addresses, a pattern the guide calls
uncalibrated: reported as measured, flagged `fetch_corrected: false`.  WRITE_SIZE is exact for 16-byte stores.

usage: python tools/pmc_summary.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass> <out.json> [note]"""
import csv
import glob
import json
import re
import sys
from collections import defaultdict

STREAMING = ("ntt_pass_kernel", "king_fft2_kernel", "king_degred_kernel", "vec_", "pss_", "bitrev_kernel", "r1cs_qap_kernel",
             "msm_scatter_kernel<", "msm_hist_kernel")     # 16-byte-per-lane loads of 32-byte scalars (the bin sort reads dwords: uncalibrated)


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
    corr = k.startswith(STREAMING)
    kernels.append({"kernel": k, "launches": n, "FETCH_SIZE_KB_per_launch": round(f_kb, 1),
                    "WRITE_SIZE_KB_per_launch": round(w_kb, 1), "fetch_corrected": corr,
                    "hbm_bytes_per_launch": int(((2 if corr else 1) * f_kb + w_kb) * 1024)})
out = {"note": (sys.argv[4] if len(sys.argv) > 4 else "") + " -- rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate "
               "passes (program directly after `--`); KB as reported; hbm_bytes_per_launch = (2 x FETCH for the 16-byte-per-"
               "lane streaming kernels, 1 x otherwise) + WRITE, see the docstring of tools/pmc_summary.py",
       "kernels": kernels}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(kernels[:8], indent=1))
