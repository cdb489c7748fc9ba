"""Issue-cycle view of the two d_fft kernels (ntt_pass_kernel, king_fft2_kernel) from two rocprofv3 SQ counter passes over
`bench.py --workload c2` -> profiles/<name>.json.   usage: python tools/sq_c2_summary.py <dir of pass A> <dir of pass B> <out>
Issue costs (tools/mulbench.hip): v_mad_u64_u32 8 cycles per wave64 instruction, other VALU 2."""
import collections
import csv
import glob
import json
import sys

dir_a, dir_b, out_path = sys.argv[1], sys.argv[2], sys.argv[3]
out = {"command": ["rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY "
                   "SQ_BUSY_CYCLES -- python3 bench.py --workload c2 --no-cpu-baseline --steps 5 --warmup 2",
                   "rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR "
                   "SQ_INSTS_SALU SQ_INSTS_SMEM -- (same)"],
       "note": "per launch, averaged over the launches of the run; GRBM_GUI_ACTIVE is summed over the 8 XCDs. Issue costs "
               "(tools/mulbench.hip): v_mad_u64_u32 8 cycles per wave64 instruction, other VALU 2.",
       "kernels": []}
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for d in (dir_a, dir_b):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("zk::", "")
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[(k, r["Counter_Name"])] += 1
spec = {"king_fft2_kernel<Bn254Fr, 2, false>": dict(
            waves=8192, mads_per_wave=2 * 584 + 14 * 136,
            what="2^19 chunks, one per lane; unpack2 as two 8-term dot products with one reduction each (Fp::dot_k, 584 multiply "
                 "instructions each instead of 8 x 136), 4 products of butterfly / g^pos, 10 of the structured pack, one "
                 "ChaCha12 block for the two draws of a chunk"),
        "ntt_pass_kernel<Fp<Bn254Fr>, 11>": dict(
            waves=16384, mads_per_wave=24 * 136,
            what="8 x 2^19 elements, 4 per lane; 22-26 Montgomery products per lane (stages / 2 x 4 + the pre-twiddle of the "
                 "second pass)")}
for k, v in agg.items():
    if k not in spec:
        continue
    d = {c: round(x / cnt[(k, c)]) for c, x in v.items()}
    sp = spec[k]
    cyc = d["GRBM_GUI_ACTIVE"] / 8
    valu_per_wave = d["SQ_INSTS_VALU"] / sp["waves"]
    other = valu_per_wave - sp["mads_per_wave"]
    issue = sp["mads_per_wave"] * 8 + other * 2
    wps = sp["waves"] / 1024
    out["kernels"].append({"kernel": k, "work": sp["what"], "per_launch": d, "derived": {
        "shader_cycles_per_xcd": round(cyc), "valu_instructions_per_wave": round(valu_per_wave),
        "of_which_v_mad_u64_u32": sp["mads_per_wave"], "other_valu": round(other),
        "valu_issue_cycles_per_wave": round(issue), "waves_per_simd": wps,
        "valu_issue_cycles_per_simd_over_kernel_cycles": round(issue * wps / cyc, 3),
        "share_of_issue_cycles_that_is_not_multiply": round(other * 2 / issue, 3),
        "wave_cycle_split": {"active": round(d["SQ_ACTIVE_INST_ANY"] / d["SQ_WAVE_CYCLES"], 3),
                             "issue_stall": round(d["SQ_WAIT_INST_ANY"] / d["SQ_WAVE_CYCLES"], 3),
                             "parked_waitcnt": round(d["SQ_WAIT_ANY"] / d["SQ_WAVE_CYCLES"], 3)},
        "lds_bank_conflict_cycles_per_lds_instruction": round(d["SQ_LDS_BANK_CONFLICT"] / max(1, d["SQ_INSTS_LDS"]), 2)}})
json.dump(out, open(out_path, "w"), indent=1)
for e in out["kernels"]:
    print(e["kernel"], json.dumps(e["derived"]))
