"""The per-rank collective entry points (zk_dist_* over zk_net_*, csrc/net.hpp): the reference's call shape -- every
party calls d_fft / d_ifft / deg_red / d_pp / d_msm / circom_h / dsha256 with a net and a stream id -- on the GPU.
  * world = 1 (local and RCCL transports): results identical to the all-parties-in-one-call entry points;
  * 2 and 4 ranks sharing this box's one GPU through the shared-memory transport (RCCL needs a GPU per rank): every
    rank's rows identical to the single-context results, masks on;
  * 8 ranks, one enters late: it is left out (ZK_ERR_PROTOCOL for it) and the others finish through lagrange_unpack
    (ser_net.rs:57-94);
  * bench.py --gpus 2 for the c2 / c3 / c4 workloads as the driver launches it."""
import json
import multiprocessing as mp
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _spawn(world, scenario, transport="shm", nresults=1):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import dist_worker
    from zksaas_amd.net import StarNet
    net_id = StarNet.unique_id()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=dist_worker.run, args=(r, world, net_id, scenario, q, transport)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=600) for _ in range(len(procs) * nresults)]
    for p in procs:
        p.join(timeout=120)
    bad = [r for r in results if not r[1]]
    assert not bad, bad


@pytest.mark.parametrize("transport", ["local", "rccl"])
def test_world_of_one_equals_the_all_parties_calls(transport):
    _spawn(1, "flow", transport)


@pytest.mark.parametrize("world", [2, 4])
def test_ranks_sharing_the_gpu_over_shared_memory(world):
    _spawn(world, "flow")


@pytest.mark.parametrize("world", [2, 4])
def test_arbitrary_party_to_rank_map(world):
    """MpcNet ids are arbitrary (mpc-net/src/lib.rs:43-53): the same flow with the parties dealt to the ranks by a
    non-contiguous map; every rank's rows (in its ascending party order) equal the single-context results."""
    _spawn(world, "map")


@pytest.mark.parametrize("world", [2, 4])
def test_zero_denominator_fails_the_round_on_every_rank_at_once(world):
    """dist-primitives/src/dpp/mod.rs:55 (the king's `inverse().unwrap()` panics): every rank returns Generic within
    seconds, not after the net's timeout, and the channel serves the next round."""
    _spawn(world, "dpp_zero", nresults=2)


def test_late_rank_is_left_out_on_the_gpu():
    _spawn(8, "late")


@pytest.mark.parametrize("world", [2, 4, 8])
def test_alltoall_king_equals_the_star_king(world):
    """Option king_alltoall: every rank reconstructs and re-packs a contiguous chunk range (two all-to-all exchanges per
    round instead of gather + scatter through rank 0); shares bit-identical to the single-context results."""
    _spawn(world, "flow_a2a")


def test_alltoall_king_with_a_late_rank():
    _spawn(8, "late_a2a")


def _torchrun(ranks, bench_args, env, timeout):
    """bench.py under torch.distributed.run on 127.0.0.1; a probed-free port can be taken again before the store binds it
    (EADDRINUSE on a busy box): retry on another one."""
    for attempt in range(4):
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks), "--master-addr",
               "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(ranks)] + bench_args
        out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
        if out.returncode == 0 or "EADDRINUSE" not in out.stderr:
            return out
    return out


@pytest.mark.parametrize("workload,ranks", [("c4", 2), ("c4", 4), ("c2", 2), ("c3", 2)])
def test_bench_ranks_as_the_driver_launches_it(workload, ranks):
    """bench.py --gpus N through torch.distributed.run, all ranks on the one GPU of this box (ZK_NET=shm), WITHOUT the
    replay stream the other tests run on (the driver does not set it: every rank's context then has its own production
    stream, and the bench has to deal its inputs from one dealer all the same); for c4 bench.py itself compares the
    sharded proof with the single-context proof."""
    env = dict(os.environ, ZK_NET="shm", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = _torchrun(ranks, ["--steps", "2", "--warmup", "1", "--workload", workload], env, 900)
    assert out.returncode == 0, out.stderr[-3000:]
    res = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert res["n_gpus"] == ranks and res["parties_per_gpu"] == 8 // ranks and res["value"] > 0
    # the headline runs north_star's topology (star: king on GPU 0); the all-to-all king is timed in the same run beside it
    assert res["king"] == "star" and "transport_note" not in res and "degraded" not in res      # (shm was ASKED for here)
    # what crossed the net per step and the committed xGMI prediction for it (the first hardware curve is checked against it)
    nps, xg = res["net_per_step"], res["xgmi_prediction"]
    assert len(nps["bytes_sent_by_rank"]) == ranks and nps["bytes_total"] == sum(nps["bytes_sent_by_rank"])
    assert xg["links_per_direction"] == ranks - 1
    if workload == "c3":            # d_msm's king step is one small HOST message per rank and one answer (dmsm/mod.rs:76-92): no device bytes
        assert nps["bytes_total"] == 0 and xg["seconds_per_step"] == 0
    else:
        assert nps["gathers"] >= 1 and nps["scatters"] >= 1 and nps["bytes_total"] > 0
        assert 0 < xg["seconds_per_step"] < res["ms_per_step"] / 1e3
    if workload == "c2":            # a 2^20 d_fft round: (n - k) parties' 2^19-element rows into GPU 0 and back, 32 B each
        assert nps["bytes_total"] == 2 * (8 - 8 // ranks) * (1 << 19) * 32
    if workload != "c3":
        assert res["alltoall"]["king"] == "alltoall" and res["alltoall"]["value"] > 0, res["alltoall"]
        assert "king on GPU 0" in res["config"]["workload"]
    if workload == "c4":
        assert res["proof_matches_single_gpu"] is True and res["config"]["masks"] is True
        # throughput modes of the sharded prover: a batch per collective call, and two proofs in flight per rank
        assert res["batched"]["same_proof"] is True and res["batched"]["proofs_per_s"] > 0
        assert res["in_flight"]["same_proof"] is True and res["in_flight"]["proofs_per_s"] > 0


def test_bench_falls_back_to_shared_memory_when_rccl_cannot_start():
    """Two ranks on ONE GPU is a configuration RCCL refuses; the bench must notice before its timed region (probe round),
    agree across ranks, and finish over the shared-memory transport with the fact recorded in its line."""
    env = dict(os.environ, ZK_DIST_VIA_CPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("ZK_NET", None)
    out = _torchrun(2, ["--steps", "2", "--warmup", "1", "--workload", "c2"], env, 600)
    assert out.returncode == 0, out.stderr[-3000:]
    res = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert res["n_gpus"] == 2 and res["value"] > 0
    assert res["transport"] == "shm" and res["king"] == "star" and "rccl transport" in res["transport_note"]
    assert res["degraded"] is True and res["rccl_ranks"] == 0        # a reader of `value` alone is told: not a scaling point
