"""Share of VALU work by kernel from one rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU pass (counter collection
serialises the kernels: these are each kernel's own instruction counts, not concurrent behaviour).
usage: python tools/valu_share.py <dir of the pass> [skip first N launches per kernel]"""
import collections
import csv
import glob
import re
import sys

agg = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.match(r"(?:void )?(?:zk::)?(\w+)", r["Kernel_Name"])
        k = (m.group(1) if m else r["Kernel_Name"]) + ("<G2>" if "Fp2" in r["Kernel_Name"] else "")
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
tot = sum(v.get("SQ_INSTS_VALU", 0) for v in agg.values())
tot_a = sum(v.get("SQ_ACTIVE_INST_VALU", 0) for v in agg.values())
for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_ACTIVE_INST_VALU", 0))[:18]:
    print(f"{v.get('SQ_INSTS_VALU', 0) / tot:7.3f} of VALU instructions  {v.get('SQ_ACTIVE_INST_VALU', 0) / max(tot_a, 1):7.3f} of VALU-active cycles  {k}")
