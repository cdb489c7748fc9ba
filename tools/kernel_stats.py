"""Per-kernel statistics (calls, total, average, share) of a rocprofv3 --kernel-trace run as CSV on stdout, from the
rocpd database or the kernel-trace csv.   usage: python tools/kernel_stats.py <p_results.db | *_kernel_trace.csv>"""
import csv
import re
import sqlite3
import sys
from collections import defaultdict

path = sys.argv[1]
agg = defaultdict(lambda: [0, 0, 0])
if path.endswith(".db"):
    rows = sqlite3.connect(path).cursor().execute("select name, end - start from kernels")
else:
    rows = ((r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in csv.DictReader(open(path)))
for name, d in rows:
    n = re.sub(r"\(.*$", "", re.sub(r"^void ", "", name).replace("zk::", "")).strip()
    a = agg[n]
    a[0] += 1
    a[1] += d
    a[2] = max(a[2], d)
tot = sum(a[1] for a in agg.values())
w = csv.writer(sys.stdout)
w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MaxNs", "Percentage"])
for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    w.writerow([n, a[0], a[1], round(a[1] / a[0], 1), a[2], round(100.0 * a[1] / tot, 2)])
