"""GPU occupancy of the batched prover from a rocprofv3 --kernel-trace result (rocpd sqlite database): over the steady
part of the run, the fraction of wall time with at least one kernel running, the sum of kernel time per kernel name and
the union time per stream.
usage: python tools/batch_timeline.py gpurun_out/prof/p_results.db [skip_fraction]"""
import re
import sqlite3
import sys
from collections import defaultdict

cur = sqlite3.connect(sys.argv[1]).cursor()
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
ks = sorted(cur.execute("select name, start, end, stream_id from kernels").fetchall(), key=lambda r: r[1])
t0, t1 = ks[0][1], max(k[2] for k in ks)
lo = t0 + (t1 - t0) * skip                       # steady part: the tail of the run (the batches come last)
sel = [k for k in ks if k[1] >= lo]
wall = max(k[2] for k in sel) - sel[0][1]
busy, end = 0, sel[0][1]
for k in sel:
    if k[2] > end:
        busy += k[2] - max(k[1], end)
        end = k[2]


def short(n):
    m = re.match(r"(?:void )?(?:zk::)?(\w+)", n)
    s = m.group(1) if m else n
    return s + ("<G2>" if "Fp2" in n else "")


by = defaultdict(float)
for k in sel:
    by[short(k[0])] += k[2] - k[1]
print(f"window {wall / 1e6:.2f} ms, some kernel running {busy / wall:.3f} of it, summed kernel time {sum(by.values()) / wall:.2f} x wall")
for n, t in sorted(by.items(), key=lambda kv: -kv[1])[:16]:
    print(f"  {t / wall:6.3f} x wall  {n}")
