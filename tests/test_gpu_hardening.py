"""Round 6: no library call may block for ever, and calls on different streams do not share
working memory.
  * batches in flight on a FOUR-thread host pool, 500 pairs, every proof equal to the one-at-a-time proof; the same pattern
    in a process pinned to TWO CPUs (tools/batch_probe.py 8 --inflight 2 --options host_threads=4);
  * `wait_deadline_ms`: a wait that cannot be met returns ZK_ERR_GENERIC naming what it waited for, dumps the job's
    gates / events on stderr, wedges the context (later proofs fail at once), and a fresh context proves normally;
  * d_fft / d_ifft / d_pp on TWO streams at once equal the one-at-a-time results bit for bit (the reference runs three
    d_ifft concurrently on three stream ids, groth16/src/ext_wit.rs:127-159; dist-primitives/src/dpp/mod.rs:15-22 takes a
    sid for the same reason)."""
import json
import os
import subprocess
import sys
import time
from collections import deque

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import zksaas_amd as zk
from zksaas_amd import groth16 as zg
from zksaas_amd import synthetic

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _sha256_inputs(options=()):
    from bench import build_inputs
    pp = zk.PackedSharingParams("bn254", 2)
    for k, v in options:
        pp.set_option(k, v)
    r1, w, setup, crs, wit, r, s = build_inputs(pp, zk)
    crs.precompute()
    masks = zg.ProofMasks(pp, wit.log_m, seed=77)
    return pp, crs, wit, r, s, masks


def test_five_hundred_pairs_of_batches_in_flight_on_a_four_thread_pool():
    from bench import same_shares
    pp, crs, wit, r, s, masks = _sha256_inputs([("host_threads", 4)])
    ref = zg.prove(pp, crs, wit, r, s, masks=masks, seed=1)
    nb = 8
    q, done, t0 = deque(), 0, time.perf_counter()
    for i in range(1000):                                   # 500 pairs: batch i + 1 starts before batch i is waited for
        q.append(zg.prove_batch_async(pp, crs, [wit] * nb, [r] * nb, [s] * nb, masks=[masks] * nb, seed=100 + i))
        if len(q) >= 2:
            out = q.popleft().wait()
            done += 1
            if done % 100 == 1:
                assert all(same_shares(pp, o, ref) for o in out)
    while q:
        out = q.popleft().wait()
        done += 1
    assert done == 1000 and all(same_shares(pp, o, ref) for o in out)
    print("1000 batches of %d, two in flight, 4 host threads: %.1f s" % (nb, time.perf_counter() - t0))


def test_batches_in_flight_in_a_process_pinned_to_two_cpus():
    """what round 5 could not explain hung under exactly this call pattern on one box: here with the least host the library
    accepts (4 pool threads on 2 CPUs); the probe's own deadline is the library's (a hang would now be an error, not a
    timeout of this test)"""
    cpus = sorted(os.sched_getaffinity(0))[:2]
    cmd = [sys.executable, os.path.join(ROOT, "tools", "batch_probe.py"), "8", "--inflight", "2", "--reps", "170",
           "--options", "host_threads=4"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT,
                       preexec_fn=lambda: os.sched_setaffinity(0, set(cpus)))
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["same_proof"] and line["inflight"] == 2 and line["batch"] == 8


def test_a_wait_that_cannot_be_met_is_an_error_not_a_hang(capfd):
    pp, crs, wit, r, s, masks = _sha256_inputs()
    good = zg.prove(pp, crs, wit, r, s, masks=masks, seed=1)
    with pytest.raises(zk.ZkError):
        pp.set_option("wait_deadline_ms", -1)
    pp.set_option("wait_deadline_ms", 1)                    # a batch of 16 takes ~15 ms: its events cannot signal in 1 ms
    nb = 16
    t0 = time.perf_counter()
    with pytest.raises(zk.ZkError) as e:
        zg.prove_batch(pp, crs, [wit] * nb, [r] * nb, [s] * nb, masks=[masks] * nb, seed=3)
    assert time.perf_counter() - t0 < 5.0
    assert e.value.code == 1 and "deadline" in str(e.value)                 # ZK_ERR_GENERIC = MpcNetError::Generic
    err = capfd.readouterr().err
    assert "[zksaas]" in err and "batch slot" in err and "chain event" in err
    # the context refuses further proofs at once (its slot still holds work), a fresh one is unaffected
    pp.set_option("wait_deadline_ms", 120000)
    with pytest.raises(zk.ZkError) as e2:
        zg.prove(pp, crs, wit, r, s, masks=masks, seed=1)
    assert "wedged" in str(e2.value)
    pp.sync()
    pp2 = zk.PackedSharingParams("bn254", 2)
    crs.precompute(pp2)
    from bench import same_shares
    again = zg.prove(pp2, crs, wit, r, s, masks=masks, seed=1)
    assert same_shares(pp2, again, good)


@pytest.mark.parametrize("curve", ["bn254", "bls12_381"])
def test_transforms_and_d_pp_on_two_streams_at_once_equal_the_serial_results(curve):
    import torch
    pp = zk.PackedSharingParams(curve, 2)
    log_m = 16
    m, L = 1 << log_m, (1 << log_m) // pp.l
    xs = [synthetic.rand_fr_device(pp, m + 1, 900 + i) for i in range(2)]
    shares = [pp.pack(x, L, 910 + i) for i, x in enumerate(xs)]
    fmasks = [zk.FftMask.sample(pp, True, 5, True, log_m, 920 + i) for i in range(2)]
    dmasks = [zk.DegRedMask.sample(pp, L, 930 + i) for i in range(2)]
    den = [pp.pack(x.view(pp.fr.nbytes), L, 940 + i) for i, x in enumerate(xs)]
    pp.sync()

    def run(i, stream):
        a = zk.d_ifft(pp, shares[i], fmasks[i], True, log_m, g=5, seed=950 + i, out=pp.alloc_fr(pp.n * L), stream=stream)
        b = zk.d_fft(pp, a, zk.FftMask.zero(), False, log_m, seed=960 + i, out=pp.alloc_fr(pp.n * L), stream=stream)
        c = zk.d_pp(pp, shares[i], den[i], dmasks[i], L, seed=970 + i, stream=stream)
        return a, b, c

    serial = []
    for i in range(2):
        serial.append([x.to_numpy().copy() for x in run(i, None)])
        pp.sync()
    streams = [torch.cuda.Stream() for _ in range(2)]
    for rep in range(20):                                    # interleaved: both streams hold work of all three kinds at once
        outs = [run(i, streams[i].cuda_stream) for i in range(2)]
        for st in streams:
            st.synchronize()
        for i in range(2):
            for got, want in zip(outs[i], serial[i]):
                assert np.array_equal(got.to_numpy(), want), (rep, i)
            for buf in outs[i]:
                buf.free()
