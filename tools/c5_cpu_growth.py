"""The C5 CPU leg (bench.cpu_baseline_c5: the plain-C port proving the synthetic BLS12-381 instance) measured at 2^20 AND at
2^22 constraints on this host, so that the 2^24 figure of the c5 line rests on a measured growth instead of a linear guess
(VERDICT r5 next #7).  Needs a GPU (the instance is built and checked on the device).  Writes one JSON object:
    python tools/c5_cpu_growth.py > profiles/r06_c5_cpu_growth.json      (about 4 minutes on 16 cores)"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zksaas_amd as zk
import bench


def main():
    out = {}
    for lg in (20, 22):
        os.environ["ZK_C5_CPU_LOG_M"] = str(lg)
        pp = zk.PackedSharingParams("bls12_381", 2)
        r = bench.cpu_baseline_c5(pp, zk, 24)
        out["seconds_2^%d" % lg] = r["measured_s_at_sample"]
        out["matches_gpu_2^%d" % lg] = r["matches_gpu_at_sample"]
        out["cores"] = r["cores"]
        out["cpu_model"] = r["cpu_model"]
        del pp
    out["ratio"] = round(out["seconds_2^22"] / out["seconds_2^20"], 3)
    out["note"] = "one proof each, all usable cores; a linear law would give 4.0"
    print(json.dumps(out))


if __name__ == "__main__":
    main()
