"""Per-kernel SQ / GRBM counters of one rocprofv3 pass -> profiles/<name>.json (any command; the kernel-specific views are
tools/sq_summary.py for d_msm and tools/sq_c2_summary.py for d_fft).
usage: python tools/sq_generic.py <dir of the pass> <out.json> "<command that was profiled>" [top N = 8]
Derived per kernel: issue-cycle split of the wave cycles (active / issue stall / parked on waitcnt), VALU instructions per
SIMD cycle (1024 SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs), effective shader clock when a kernel-trace duration is
not at hand is left out.  SQ_WAVE_CYCLES counts in units of FOUR cycles on this ROCm (a kernel compiled for 4 waves per
SIMD that fills the chip reads 0.9-1.0 wave-cycles per SIMD cycle), hence the factor in waves_in_flight_per_simd."""
import collections
import csv
import glob
import json
import sys

src, out_path, command = sys.argv[1], sys.argv[2], sys.argv[3]
top = int(sys.argv[4]) if len(sys.argv) > 4 else 8
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for f in glob.glob(src + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("zk::", "")
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            cnt[k] += 1
out = []
for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0))[:top]:
    n = cnt[k] or 1
    d = {c: round(x / n) for c, x in v.items()}
    e = {"kernel": k, "launches": n, "per_launch": d}
    wc = d.get("SQ_WAVE_CYCLES")
    if wc:
        simd_cycles = 1024 * d["GRBM_GUI_ACTIVE"] / 8
        e["derived"] = {"valu_instructions_per_simd_cycle": round(d.get("SQ_INSTS_VALU", 0) / simd_cycles, 3),
                        "wave_cycle_split": {"active": round(d.get("SQ_ACTIVE_INST_ANY", 0) / wc, 3),
                                             "issue_stall": round(d.get("SQ_WAIT_INST_ANY", 0) / wc, 3),
                                             "parked_waitcnt": round(d.get("SQ_WAIT_ANY", 0) / wc, 3)},
                        "waves_in_flight_per_simd": round(4 * wc / simd_cycles, 2)}
    out.append(e)
json.dump({"command": command, "note": "GRBM_GUI_ACTIVE is summed over the 8 XCDs; per_launch = counter sum / launches",
           "kernels": out}, open(out_path, "w"), indent=1)
print(json.dumps([{"kernel": e["kernel"], **e.get("derived", {})} for e in out[:4]], indent=1))
