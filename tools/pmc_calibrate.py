"""FETCH_SIZE / WRITE_SIZE factors per access shape from two rocprofv3 passes over tools/pmc_calib (see its header):
factor = known bytes / reported bytes (the counters are in KB).  Writes profiles/<out>.json, which tools/pmc_summary.py
reads to correct per kernel by ACCESS SHAPE instead of a per-kernel boolean.

usage: python tools/pmc_calibrate.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass> <out.json>"""
import csv
import glob
import json
import re
import sys
from collections import defaultdict

BYTES = 2 << 30          # every calibration kernel reads 2 GiB and writes 2 GiB (tools/pmc_calib.hip main)


def collect(d, counter):
    agg = defaultdict(list)
    for path in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(path)):
            if row["Counter_Name"] == counter:
                name = re.sub(r"\(.*$", "", row["Kernel_Name"].replace("void ", "")).strip()
                agg[name].append(float(row["Counter_Value"]))
    return agg


if __name__ == "__main__":
    fetch, write = collect(sys.argv[1], "FETCH_SIZE"), collect(sys.argv[2], "WRITE_SIZE")
    shapes = {}
    for k in sorted(fetch):
        if not k.startswith("calib_"):
            continue
        f_kb = sorted(fetch[k])[len(fetch[k]) // 2]
        w = write.get(k, [])
        w_kb = sorted(w)[len(w) // 2] if w else None
        shapes[k[len("calib_"):]] = {
            "FETCH_SIZE_KB": round(f_kb, 1), "WRITE_SIZE_KB": None if w_kb is None else round(w_kb, 1),
            "fetch_factor": round(BYTES / (f_kb * 1024), 4),
            "write_factor": None if not w_kb else round(BYTES / (w_kb * 1024), 4), "launches": len(fetch[k])}
    out = {"note": "tools/pmc_calib.hip under rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes); every kernel "
                   "reads 2 GiB and writes 2 GiB of buffers far beyond the 256 MiB Infinity Cache; factor = known bytes / "
                   "reported bytes (median of the launches)", "bytes_per_launch": BYTES, "shapes": shapes}
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    print(json.dumps(shapes, indent=1))
