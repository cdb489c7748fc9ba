"""Per-kernel HBM traffic from two rocprofv3 counter passes (FETCH_SIZE and WRITE_SIZE collected separately, as the
MI355X guide prescribes) -> profiles/<name>.json, the file bench.py's `roofline.traffic` reads.

usage: python tools/pmc_summary.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass> <out.json>"""
import csv
import glob
import json
import re
import sys
from collections import defaultdict


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
    kernels.append({"kernel": k, "launches": n, "FETCH_SIZE_KB_per_launch": round(fetch[k][1] / n, 1),
                    "WRITE_SIZE_KB_per_launch": round(wv / wn, 1) if wn else None})
out = {"note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of `bench.py --steps 5 --warmup 1 "
               "--no-cpu-baseline --no-primitives` (setup launches included for the dealer kernels only); values are KB "
               "as reported. gfx950: FETCH_SIZE under-reports wide coalesced streaming reads by 2x; msm_accumulate "
               "gathers 64-byte points at random (uncalibrated, no correction applied).",
       "kernels": kernels}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(kernels[:6], indent=1))
