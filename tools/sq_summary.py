"""Issue-cycle view of the d_msm accumulate kernel from one rocprofv3 SQ / GRBM counter pass over `bench.py --workload c3`
-> profiles/<name>.json.   usage: python tools/sq_summary.py <dir of the pass> <kernel_stats.csv of the same command> <out>
Issue costs (tools/mulbench.hip): v_mad_u64_u32 8 cycles per wave64 instruction, full-rate VALU 2."""
import collections
import csv
import glob
import json
import sys

src, kstats, out_path = sys.argv[1], sys.argv[2], sys.argv[3]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for f in glob.glob(src + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            cnt[k] += 1
ks = list(csv.reader(open(kstats)))
acc_ms = [float(r[3]) / 1e6 for r in ks[1:] if r[0].startswith("msm_accumulate_kernel")][0]
out = []
for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0))[:6]:
    n = cnt[k] or 1
    d = {c: round(x / n) for c, x in v.items()}
    e = {"kernel": k.split("(")[0].replace("void ", "").replace("zk::", ""), "launches": n, "per_launch": d}
    if "accumulate" in k:
        cyc = d["GRBM_GUI_ACTIVE"] / 8
        simd_cycles = 1024 * cyc
        mads = 8.4e6 * 15 * 10 * 128 / 64      # wave-level v_mad_u64_u32 of one launch: 8 x 2^20 points, 15 windows, 10 products per addition
        e["derived"] = {
            "shader_cycles_per_xcd": round(cyc), "kernel_ms_kernel_trace": round(acc_ms, 2),
            "effective_clock_GHz": round(cyc / (acc_ms * 1e-3) / 1e9, 2),
            "valu_instructions_per_simd_cycle": round(d["SQ_INSTS_VALU"] / simd_cycles, 3),
            "mad_u64_u32_issue_cycles_over_simd_cycles": round(mads * 8 / simd_cycles, 3),
            "other_valu_issue_cycles_over_simd_cycles": round((d["SQ_INSTS_VALU"] - mads) * 2 / simd_cycles, 3),
            "non_mad_valu_instructions_per_mad": round((d["SQ_INSTS_VALU"] - mads) / mads, 2),
            "wave_cycle_split": {"active": round(d["SQ_ACTIVE_INST_ANY"] / d["SQ_WAVE_CYCLES"], 3),
                                 "issue_stall": round(d["SQ_WAIT_INST_ANY"] / d["SQ_WAVE_CYCLES"], 3),
                                 "parked_waitcnt": round(d["SQ_WAIT_ANY"] / d["SQ_WAVE_CYCLES"], 3)}}
    out.append(e)
json.dump({"command": "rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY "
                      "SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_BUSY_CYCLES -- python3 bench.py --workload c3 --no-cpu-baseline "
                      "--steps 5 --warmup 2",
           "note": "GRBM_GUI_ACTIVE is summed over the 8 XCDs; on this ROCm SQ_ACTIVE_INST_VALU reports the same value as "
                   "SQ_INSTS_VALU (instructions)", "kernels": out}, open(out_path, "w"), indent=1)
print(json.dumps(out[0].get("derived", {}), indent=1))
