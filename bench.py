#!/usr/bin/env python3
"""bench.py -- Groth16 proofs/s on the SHA-256 fixture circuit (BASELINE.json metric), MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

One step = one distributed Groth16 proof (n = 8 parties, l = 2, BN254, m = 2^15) of the SHA-256 circuit with
every share (QAP, witness, packed CRS) already resident in HBM: circom_h (3 d_ifft + 3 d_fft + deg_red) and the
five d_msm, ending with the parties' (A, B, C) shares on the host.  With N ranks the 8 parties are split over
the ranks (king = rank 0) and the gather / scatter / broadcast of the star network run over RCCL.

Rank 0 prints ONE JSON line.  `roofline` is the dominant kernel's algorithmic bytes per launch
over its average launch duration measured here with HIP events on the launching stream; `cpu_baseline` is the
plain-C restatement of the reference's CPU path (oracle/c, kind "port") timed on this host on the same inputs,
and its proof is compared with the GPU's.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)

# algorithmic bytes per unit of each timed slot (DESIGN.md "Measurement")
#   ntt_pass      : one of P passes of fft1 over an element: (2 * 32 B) / P is charged per launch (see below)
#   king_fft2     : per chunk, l = 2: n shares in + n shares out = 16 * 32 B
#   msm accumulate: per point: affine base (2 |Fq|) + scalar (32 B)  -> 96 B (G1), 160 B (G2)
MUL_PEAK_G = 93.0   # measured on MI355X with tools/mulbench.hip (profiles/r01_mulbench.txt)
SLOT_BYTES = {"king_fft2_kernel": 512.0, "msm_accumulate_kernel<G1>": 96.0, "msm_accumulate_kernel<G2>": 160.0,
              "msm_digits+scan+expand": 32.0, "msm_finalize+reduce": 0.0, "king_degred_kernel": 512.0}


def build_inputs(pp, zk, seed=42):
    from zksaas_amd import groth16 as zg
    from zksaas_amd import sha256_circuit as sc
    from zksaas_amd.fields import FR
    p = FR["bn254"]
    r1, w = sc.build(1, 2, p)
    assert w[1] == sc.expected_output(1, 2)
    rng = np.random.default_rng(seed)
    td = [int.from_bytes(rng.bytes(32), "little") % p for _ in range(5)]
    setup = zg.SetupScalars("bn254", r1, *td)
    crs = zg.Crs(pp, setup)
    wit = zg.Witness(pp, "bn254", r1, w, seed=seed + 1)
    r = int.from_bytes(rng.bytes(32), "little") % p
    s = int.from_bytes(rng.bytes(32), "little") % p
    return r1, w, setup, crs, wit, r, s


def read_profile(pp):
    lib = pp.lib
    out = []
    for slot in range(lib.zk_profile_slots()):
        ms, units, calls = C.c_double(), C.c_double(), C.c_long()
        pp._check(lib.zk_profile_read(pp.h, slot, C.byref(ms), C.byref(units), C.byref(calls)))
        out.append({"kernel": lib.zk_profile_name(slot).decode(), "total_ms": ms.value, "units": units.value,
                    "launches": calls.value})
    return out


PMC_KERNEL = {"msm_accumulate_kernel<G1>": "msm_accumulate_kernel<Fp<", "msm_accumulate_kernel<G2>": "msm_accumulate_kernel<Fp2",
              "ntt_pass_kernel": "ntt_pass_kernel", "king_fft2_kernel": "king_fft2_kernel",
              "king_degred_kernel": "king_degred_kernel"}


def pmc_traffic(slot_name):
    """HBM bytes per launch of the roofline kernel from the committed rocprofv3 PMC summary of this same command
    (profiles/r01_final_pmc_hbm.json: FETCH_SIZE and WRITE_SIZE collected in separate --pmc passes, KB units;
    no 2x streaming-read correction is applied because the accesses are 64-byte gathers)."""
    path = os.path.join(ROOT, "profiles", "r01_final_pmc_hbm.json")
    prefix = PMC_KERNEL.get(slot_name)
    if not prefix or not os.path.exists(path):
        return None
    try:
        for k in json.load(open(path))["kernels"]:
            if k["kernel"].startswith(prefix):
                return int((k["FETCH_SIZE_KB_per_launch"] + k["WRITE_SIZE_KB_per_launch"]) * 1024)
    except (ValueError, KeyError):
        return None
    return None


def roofline_of(prof, ntt_passes, pp=None, table_windows=None):
    # the dominant STREAMING kernel: the sort and the bucket finalize/reduce helpers are latency-bound tree
    # kernels without a per-unit byte figure in SURVEY.md 8d; they are listed under "kernels"
    cands = [e for e in prof if e["launches"] and (SLOT_BYTES.get(e["kernel"]) or e["kernel"] == "ntt_pass_kernel")
             and not e["kernel"].startswith("msm_digits")]
    if not cands:
        return None
    best = max(cands, key=lambda e: e["total_ms"])
    name = best["kernel"]
    per_unit = SLOT_BYTES.get(name)
    if name == "ntt_pass_kernel":
        per_unit = 64.0 / max(1, ntt_passes)
    if not best["launches"] or not per_unit:
        return None
    avg_ms = best["total_ms"] / best["launches"]
    bytes_per_launch = per_unit * best["units"] / best["launches"]
    achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9
    # ALU view of the same launch: a mixed addition is 10 (G1) / 28 (G2) base-field Montgomery
    # multiplications per point and window -- the figure that actually bounds this kernel.  The peak is the measured
    # chip-wide rate of independent 256-bit Montgomery multiplications (tools/mulbench.hip, v_mad_u64_u32 bound).
    alu = None
    if pp is not None and name.startswith("msm_accumulate"):
        from zksaas_amd.api import ZK_G1, ZK_G2, msm_plan
        pts = int(best["units"] / best["launches"])
        plan = msm_plan(pp, ZK_G2 if "G2" in name else ZK_G1, pts)
        if table_windows:           # fixed-base tables: every MSM runs with the table's window layout
            plan["windows"], plan["window_bits"], plan["fixed_base_table"] = table_windows, -(-256 // table_windows), True
        muls = pts * plan["windows"] * plan["muls_per_add"]
        rate = muls / (avg_ms * 1e-3) / 1e9
        alu = {"achieved": round(rate, 2), "peak": MUL_PEAK_G, "unit": "G modmul/s (256-bit Montgomery)",
               "frac": round(rate / MUL_PEAK_G, 3), "plan": plan}
    return {"bound": "hbm", "alu": alu, "kernel": name, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": pmc_traffic(name),
            "avg_launch_us": round(avg_ms * 1e3, 2), "algorithmic_bytes_per_launch": int(bytes_per_launch),
            "launches": best["launches"]}


def primitives(pp, zk):
    """GPU-side timings of BASELINE configs 2 (d_fft, m = 2^20) and 3-like (d_msm, 8 x 2^17 points) with the
    achieved fraction of HBM bandwidth on SURVEY.md 8d's algorithmic bytes (the CPU side of these two is in
    profiles/r01_primitives_c2_c3.json; here only a few milliseconds of GPU time are spent)."""
    from zksaas_amd.api import ZK_G1
    rng = np.random.default_rng(3)

    def rand_fr(count):
        a = rng.integers(0, 1 << 62, size=(count, 4), dtype=np.uint64)
        a[:, 3] &= np.uint64((1 << 60) - 1)
        return zk.DeviceBuffer.from_numpy(pp, a)

    def med(fn, reps):
        fn()
        pp.sync()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            pp.sync()
            ts.append(time.perf_counter() - t0)
        return float(np.median(ts))

    out = {}
    log_m = 20
    m = 1 << log_m
    sh, dst = rand_fr(pp.n * m // 2), pp.alloc_fr(pp.n * m // 2)
    t = med(lambda: zk.d_fft(pp, sh, zk.FftMask.zero(), False, log_m, seed=3, out=dst), 10)
    alg = 32 * m * 32
    out["d_fft_m2^20_bn254_l2_n8"] = {"ms": round(t * 1e3, 3), "algorithmic_bytes": alg,
                                      "achieved_GBps": round(alg / t / 1e9, 1), "frac_hbm": round(alg / t / 8e12, 4)}
    ln = 1 << 17
    g1 = pp.fq.encode([1, 2]).reshape(-1)
    bases = zk.DeviceBuffer.from_numpy(pp, np.tile(g1, (pp.n * ln, 1)))
    sc = rand_fr(pp.n * ln)
    t = med(lambda: zk.d_msm(pp, ZK_G1, bases, sc, ln), 5)
    alg = pp.n * ln * 96
    out["d_msm_g1_8x2^17_bn254"] = {"ms": round(t * 1e3, 3), "algorithmic_bytes": alg,
                                    "achieved_GBps": round(alg / t / 1e9, 1), "frac_hbm": round(alg / t / 8e12, 5)}
    return out


def pipelined(zk, zg, pp, crs, wit, r, s, device, total, tables):
    """Informational, outside the timed K steps: the same proofs with TWO in flight (a second context = second set of
    workspaces and streams, its own host thread; CRS and witness shares are shared read-only).  A prover service
    would run like this; `value` above stays the one-proof-at-a-time rate."""
    import threading
    ctxs = [pp, zk.PackedSharingParams("bn254", 2, device=device)]
    if tables:
        crs.precompute(ctxs[1])
    for c in ctxs:
        zg.prove(c, crs, wit, r, s, seed=1)
    per = total // len(ctxs)

    def work(c):
        for _ in range(per):
            zg.prove(c, crs, wit, r, s, seed=1)
    t0 = time.perf_counter()
    ths = [threading.Thread(target=work, args=(c,)) for c in ctxs]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    pp.sync()
    dt = time.perf_counter() - t0
    return {"proofs_in_flight": len(ctxs), "proofs": per * len(ctxs), "proofs_per_s": round(per * len(ctxs) / dt, 2)}


def cpu_baseline(pp, crs, wit, r, s, seed, gpu_proof):
    """Plain-C port of the CPU path on the same inputs (one proof), 8 threads = one per party."""
    from oracle.cpu_prover import CpuProver
    n, nl = pp.n, pp.fr.nl
    Lc = (1 << wit.log_m) // pp.l
    dl = lambda buf, *shape: buf.to_numpy().reshape(*shape)
    inp = {
        "qap": [dl(q, n * Lc, nl) for q in wit.qap], "log_m": wit.log_m, "seed": seed,
        "a_share": dl(wit.a_share, n, wit.len_a, nl), "ax_share": dl(wit.ax_share, n, wit.len_w, nl),
        "s": dl(crs.s, n, crs.len_a, 8), "h": dl(crs.h, n, crs.len_a, 8), "v": dl(crs.v, n, crs.len_a, 16),
        "w": dl(crs.w, n, crs.len_w, 8), "u": dl(crs.u, n, crs.len_u, 8),
        "a_query0": crs.s1[0], "b_g1_query0": crs.s1[1], "delta_g1": crs.s1[2], "alpha_g1": crs.s1[3],
        "beta_g1": crs.s1[4], "b_g2_query0": crs.s2[0], "delta_g2": crs.s2[1], "beta_g2": crs.s2[2],
        "r": r, "s_": s,
    }
    threads = min(8, os.cpu_count() or 1)
    cpu = CpuProver("bn254", pp.l)
    (A, B, Cc), tm = cpu.prove(inp, threads=threads)
    ok = (cpu.affine(A) == cpu.affine(gpu_proof[0][0]) and cpu.affine(B, True) == cpu.affine(gpu_proof[1][0], True)
          and cpu.affine(Cc) == cpu.affine(gpu_proof[2][0]))
    return {"value": round(1.0 / tm["total_s"], 4), "unit": "proofs/s", "cores": threads, "kind": "port",
            "sample": "1 proof of the same SHA-256 circuit shares (circom_h %.2fs + 8x5 MSM %.2fs + king/assembly %.2fs)"
                      % (tm["circom_h_s"], tm["msm_s"], tm["king_assemble_s"]),
            "host_cpus": os.cpu_count(), "proof_matches_gpu": bool(ok)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-tables", action="store_true", help="prove without the fixed-base tables of the CRS "
                    "(zk_msm_precompute); default: tables built once at setup, as a prover service would")
    ap.add_argument("--no-primitives", action="store_true", help="skip the d_fft / d_msm side measurements "
                    "(used for the rocprofv3 runs so that every profiled launch belongs to the proof loop)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus must equal WORLD_SIZE")

    import torch
    import zksaas_amd as zk
    from zksaas_amd import groth16 as zg

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback)")
    if os.environ.get("ZK_DIST_VIA_CPU"):
        local_rank = 0           # debugging mode: every rank drives GPU 0 (see multigpu.StarNet)
    torch.cuda.set_device(local_rank)

    if world > 1 or os.environ.get("ZK_BENCH_FORCE_SHARDED"):
        from zksaas_amd import multigpu
        res = multigpu.bench(args, rank, local_rank, world)
        if rank == 0:
            print(json.dumps(res))
        return

    pp = zk.PackedSharingParams("bn254", 2, device=local_rank)
    r1, w, setup, crs, wit, r, s = build_inputs(pp, zk)
    table_windows = None
    if not args.no_tables:
        from zksaas_amd import api
        if os.environ.get("ZK_TABLE_C"):
            pp.set_option("msm_table_c", int(os.environ["ZK_TABLE_C"]))
        crs.precompute()
        table_windows = api.msm_table_info(pp, api.ZK_G1, crs.s)["windows"]
    seed = 1000
    for _ in range(args.warmup):
        proof = zg.prove(pp, crs, wit, r, s, seed=seed)
    torch.cuda.synchronize()
    pp._check(pp.lib.zk_profile_enable(pp.h, 1))
    t0 = time.perf_counter()
    for _ in range(args.steps):
        proof = zg.prove(pp, crs, wit, r, s, seed=seed)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    prof = read_profile(pp)
    pp._check(pp.lib.zk_profile_enable(pp.h, 0))

    proofs_per_s = args.steps / dt
    res = {
        "metric": "Groth16 proofs/sec (SHA-256 circuit)", "value": round(proofs_per_s, 3), "unit": "proofs/s",
        "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u32 limbs (256-bit Montgomery)",
        "data": "synthetic: SHA-256(a=1,b=2) circuit rebuilt from its semantics, seeded trapdoor CRS, seeded shares",
        "config": {"workload": "BASELINE configs[3]: full distributed Groth16 on the SHA-256 circuit, BN254, l=2, "
                               "n=8 parties on one GPU, zero masks", "constraints": r1.num_constraints,
                   "wires": r1.num_variables, "domain": 1 << wit.log_m, "parties": pp.n, "packing_factor": pp.l,
                   "fixed_base_tables": not args.no_tables},
        "constraints_per_sec": round(proofs_per_s * r1.num_constraints, 1),
        "roofline": roofline_of(prof, ntt_passes=2, pp=pp, table_windows=table_windows),
        "kernels": [{**e, "total_ms": round(e["total_ms"], 3)} for e in prof if e["launches"]],
    }
    if not args.no_primitives:
        res["primitives"] = primitives(pp, zk)
        res["pipelined"] = pipelined(zk, zg, pp, crs, wit, r, s, local_rank, max(8, args.steps // 2 * 2),
                                     not args.no_tables)
    if not args.no_cpu_baseline:
        res["cpu_baseline"] = cpu_baseline(pp, crs, wit, r, s, seed, proof)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
