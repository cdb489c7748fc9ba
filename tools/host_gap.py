"""What the host does between two proofs: HIP API calls around the idle gap of the GPU, from a rocprofv3
--kernel-trace --hip-trace result (rocpd sqlite database).

usage: python tools/host_gap.py gpurun_out/prof/p_results.db [proof_index_from_end [gap_us]]
Prints the last kernels of one proof, the first kernels of the next one and every HIP API call (per host thread) that
overlaps the span between them, all relative to the end of the proof's last kernel.
"""
import re
import sqlite3
import sys

path = sys.argv[1]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
gap_ns = int(float(sys.argv[3]) * 1000) if len(sys.argv) > 3 else 70_000
con = sqlite3.connect(path)
cur = con.cursor()
names = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
if "--schema" in sys.argv:
    for n in names:
        cols = [c[1] for c in cur.execute(f"pragma table_info('{n}')")]
        print(n, cols)
    sys.exit(0)
ks = sorted(cur.execute("select name, start, end, stream_id from kernels").fetchall(), key=lambda r: r[1])
bounds, end = [0], 0
for i, r in enumerate(ks):
    if i and r[1] - end > gap_ns:
        bounds.append(i)
    end = max(end, r[2])
bounds.append(len(ks))
wins = [(bounds[i], bounds[i + 1]) for i in range(len(bounds) - 1) if bounds[i + 1] - bounds[i] >= 40]
lo, hi = wins[-back]
t_end = max(r[2] for r in ks[lo:hi])
t_next = ks[hi][1]


def short(n):
    m = re.match(r"(?:void )?(?:zk::)?(\w+)", n)
    return m.group(1) if m else n


print(f"gap: {(t_next - t_end) / 1e3:.1f} us between the last kernel of a proof and the first of the next one")
for r in ks[hi - 3:hi + 6]:
    print(f"  kernel {(r[1] - t_end) / 1e3:9.1f} -> {(r[2] - t_end) / 1e3:9.1f}  stream {r[3]}  {short(r[0])}")
src = "regions" if "regions" in names else None
if not src:
    print("no `regions` view in this database; views:", names)
    sys.exit(0)
cols = [c[1] for c in cur.execute(f"pragma table_info('{src}')")]
tid = "tid" if "tid" in cols else ("thread_id" if "thread_id" in cols else None)
q = f"select name, start, end{', ' + tid if tid else ''} from {src} where end >= ? and start <= ? order by start"
for r in cur.execute(q, (t_end - 60_000, t_next + 60_000)):
    print(f"  api    {(r[1] - t_end) / 1e3:9.1f} -> {(r[2] - t_end) / 1e3:9.1f}  ({(r[2] - r[1]) / 1e3:7.1f})  "
          f"tid {r[3] if tid else '-'}  {r[0]}")
