"""Lane utilisation (rocprofv3's VALUUtilization) and VALU instruction counts per kernel from one counter pass
(--pmc SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_WAVES) over the headline bench -> profiles/<name>.json.
usage: python tools/lane_util.py <dir of the pass> <out.json>"""
import collections
import csv
import glob
import json
import sys

agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("void ", "").replace("zk::", "").split("(")[0]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_INSTS_VALU":
            cnt[k] += 1
out = []
for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_INSTS_VALU", 0))[:16]:
    n = cnt[k] or 1
    out.append({"kernel": k, "launches": n, "valu_instructions_per_launch": round(v["SQ_INSTS_VALU"] / n),
                "lane_utilisation": round(v["SQ_THREAD_CYCLES_VALU"] / (v["SQ_INSTS_VALU"] * 64), 3)})
json.dump({"command": "rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_WAVES -- python3 bench.py --no-cpu-baseline "
                      "--no-primitives --steps 10 --warmup 3",
           "note": "lane utilisation = SQ_THREAD_CYCLES_VALU / (64 x SQ_INSTS_VALU)", "kernels": out},
          open(sys.argv[2], "w"), indent=1)
