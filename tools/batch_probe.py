"""Throughput of zk_groth16_prove_batch on the SHA-256 circuit (bench.py's inputs) for a list of batch sizes.
    python tools/batch_probe.py [B ...] [--reps R] [--inflight N] [--options host_threads=4] [--no-masks] [--no-tables] [--profile]
Prints one JSON line per batch size: proofs/s, ms per batch, and the profile slots (HIP events around the kernels)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    args = sys.argv[1:]
    reps = 5
    if "--reps" in args:
        i = args.index("--reps")
        reps = int(args[i + 1])
        del args[i:i + 2]
    inflight = 1
    if "--inflight" in args:
        i = args.index("--inflight")
        inflight = int(args[i + 1])
        del args[i:i + 2]
    options = []
    if "--options" in args:                      # name=value[,name=value]: context options set before the first proof
        i = args.index("--options")
        options = [kv.split("=") for kv in args[i + 1].split(",") if kv]
        del args[i:i + 2]
    no_masks = "--no-masks" in args
    no_tables = "--no-tables" in args
    prof = "--profile" in args
    sizes = [int(a) for a in args if not a.startswith("--")] or [1, 2, 4, 8]
    import torch
    import zksaas_amd as zk
    from zksaas_amd import groth16 as zg
    import bench
    pp = zk.PackedSharingParams("bn254", 2)
    for name, value in options:
        pp.set_option(name, int(value))
    r1, w, setup, crs, wit, r, s = bench.build_inputs(pp, zk)
    masks = None if no_masks else zg.ProofMasks(pp, wit.log_m, seed=77)
    if not no_tables:
        crs.precompute()
    ref = zg.prove(pp, crs, wit, r, s, masks=masks, seed=1)
    for nb in sizes:
        mk = None if masks is None else [masks] * nb
        out = zg.prove_batch(pp, crs, [wit] * nb, [r] * nb, [s] * nb, masks=mk, seed=3)
        same = all(bench.same_shares(pp, o, ref) for o in out)
        torch.cuda.synchronize()
        if prof:
            pp._check(pp.lib.zk_profile_enable(pp.h, 1))
        ts = []
        if inflight > 1:
            # two batches in flight: start batch i + 1 before waiting for batch i; time = whole run / batches
            for rep in range(3):
                nbat = max(reps, 6)
                t0 = time.perf_counter()
                from collections import deque
                q = deque()
                for i in range(nbat):
                    q.append(zg.prove_batch_async(pp, crs, [wit] * nb, [r] * nb, [s] * nb, masks=mk, seed=100 + i))
                    if len(q) >= inflight:
                        last = q.popleft().wait()
                while q:
                    last = q.popleft().wait()
                ts.append((time.perf_counter() - t0) / nbat)
            same = same and all(bench.same_shares(pp, o, ref) for o in last)
        else:
            for i in range(reps):
                t0 = time.perf_counter()
                zg.prove_batch(pp, crs, [wit] * nb, [r] * nb, [s] * nb, masks=mk, seed=100 + i)
                ts.append(time.perf_counter() - t0)
        ts.sort()
        med = ts[len(ts) // 2]
        line = {"batch": nb, "proofs_per_s": round(nb / med, 1), "ms_per_batch": round(med * 1e3, 3),
                "ms_per_proof": round(med * 1e3 / nb, 4), "min_ms": round(ts[0] * 1e3, 3), "max_ms": round(ts[-1] * 1e3, 3),
                "same_proof": same, "inflight": inflight}
        if prof:
            line["kernels"] = [{**e, "total_ms": round(e["total_ms"] / reps, 3), "launches": e["launches"] // reps}
                               for e in bench.read_profile(pp) if e["launches"]]
            pp._check(pp.lib.zk_profile_enable(pp.h, 0))
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
