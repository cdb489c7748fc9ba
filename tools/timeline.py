"""Per-stream kernel timeline of one proof from a rocprofv3 --kernel-trace result (rocpd sqlite database or csv).

usage: python tools/timeline.py gpurun_out/prof/p_results.db [proof_index_from_end [gap_us [min_kernel_us]]]
Proof boundaries: gaps of more than gap_us (default 70) with no kernel running.
"""
import csv
import re
import sqlite3
import sys
from collections import defaultdict

path = sys.argv[1]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
gap_ns = int(float(sys.argv[3]) * 1000) if len(sys.argv) > 3 else 70_000
min_ms = float(sys.argv[4]) / 1000 if len(sys.argv) > 4 else 0.02          # kernels shorter than this are left out
if path.endswith(".db"):
    cur = sqlite3.connect(path).cursor()
    rows = [{"Kernel_Name": r[0], "Start_Timestamp": r[1], "End_Timestamp": r[2], "Stream_Id": r[3], "Grid_Size_X": r[4]}
            for r in cur.execute("select name, start, end, stream_id, grid_x from kernels")]
else:
    rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))


def short(n):
    n = re.sub(r"^void zk::", "", n)
    n = re.sub(r"^zk::", "", n)
    m = re.match(r"(\w+)", n)
    s = m.group(1) if m else n
    if "Fp2" in n:
        s += "<G2>"
    return s


bounds, end = [0], 0
for i, r in enumerate(rows):
    st = int(r["Start_Timestamp"])
    if i and st - end > gap_ns:
        bounds.append(i)
    end = max(end, int(r["End_Timestamp"]))
bounds.append(len(rows))
# windows that are whole proofs (the table build, a lone copy between two proofs etc. are not)
wins = [(bounds[i], bounds[i + 1]) for i in range(len(bounds) - 1) if bounds[i + 1] - bounds[i] >= 40]
lo, hi = wins[-back] if len(wins) >= back else wins[0]
sel = rows[lo:hi]
t0 = int(sel[0]["Start_Timestamp"])
print(f"proof window: {len(sel)} launches, {(max(int(r['End_Timestamp']) for r in sel) - t0) / 1e6:.3f} ms busy, "
      f"period {(int(rows[hi]['Start_Timestamp']) - t0) / 1e6 if hi < len(rows) else float('nan'):.3f} ms")
by = defaultdict(list)
for r in sel:
    by[r.get("Stream_Id", r.get("Queue_Id"))].append(r)
for q, rs in sorted(by.items(), key=lambda kv: int(kv[1][0]["Start_Timestamp"])):
    print(f"-- stream/queue {q}")
    for r in rs:
        s, e = (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6
        if e - s >= min_ms:
            print(f"   {s:7.3f} -> {e:7.3f}  ({e - s:6.3f})  {short(r['Kernel_Name'])}  grid={r.get('Grid_Size_X', r.get('Grid_Size'))}")
