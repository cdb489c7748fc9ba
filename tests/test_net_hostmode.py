"""The star network of the library (zk-saas_amd/csrc/net.hpp behind zk_net_*) exercised WITHOUT a GPU: N processes,
shared-memory transport, host buffers (ctx = NULL), with the oracle standing in for the device compute (test
infrastructure only).  What is covered: the control plane (rounds, verdicts), gather = client_send_or_king_receive and
scatter = client_receive_or_king_send (mpc-net/src/lib.rs:89-176) incl. multi-chunk transfers, three channels
interleaved (ext_wit.rs:158-170), the small host messages of d_msm, and the timeout / `Partial` emulation
(ser_net.rs:57-94): a rank that never shows up is left out, the king continues through lagrange_unpack, a late rank
gets the protocol error.  The GPU forms of the same rounds (zk_dist_*) are tested in tests/test_gpu_dist.py."""
import multiprocessing as mp
import os
import sys
import time

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _setup():
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)


def _enc(vals):
    return np.array([[(v >> (64 * k)) & ((1 << 64) - 1) for k in range(4)] for v in vals], dtype=np.uint64)


def _dec(arr):
    return [sum(int(row[k]) << (64 * k) for k in range(4)) for row in np.asarray(arr).reshape(-1, 4)]


def _d_fft_round(net, o, dom, sid, mask, mine_rows, rearrange, seed, present=None):
    """One king round of d_fft as zk_dist_d_fft performs it, oracle compute on host arrays."""
    from oracle import dist as od
    k, Lc = net.k, len(mine_rows[0])
    local = _enc([v for row in mine_rows for v in od.fft1_in_place(list(row), o, dom.group_gen)])
    full = np.zeros((o.n * Lc, 4), dtype=np.uint64) if net.rank == 0 else None
    net.gather(sid, mask, local, k * Lc * 32, full)
    out = np.zeros((o.n * Lc, 4), dtype=np.uint64) if net.rank == 0 else None
    if net.rank == 0:
        parties = present or [r * k + p for r in range(net.world) if mask >> r & 1 for p in range(k)]
        flat = _dec(full[: len(parties) * Lc])
        shares = [flat[i * Lc:(i + 1) * Lc] for i in range(len(parties))]
        res = od.king_fft2(shares, parties, rearrange, 1, o, dom.group_gen, seed)
        out[:] = _enc([v for s in res for v in s])
    net.scatter(sid, mask, out, k * Lc * 32, local)
    flat = _dec(local)
    return [flat[i * Lc:(i + 1) * Lc] for i in range(k)]


def _worker(rank, world, net_id, scenario, q):
    _setup()
    try:
        from zksaas_amd.net import StarNet
        from zksaas_amd._lib import ZkError
        from oracle import dist as od
        from oracle.field import Domain, bitrev_permute
        from oracle.params import BLS12_377
        from oracle.prng import rand_vec
        from oracle.pss import PackedSharingParams
        o = PackedSharingParams(BLS12_377, 2)
        m = 64
        dom = Domain(BLS12_377, m)
        Lc = m // 2
        x = rand_vec(5, m, o.p)
        y = list(x)
        bitrev_permute(y)
        shares = od.transpose(od.stride_pack(y, o, 2))          # all parties' inputs (every rank deals the same)
        pmap = {2: [1, 0, 0, 1, 1, 0, 1, 0], 4: [3, 1, 0, 2, 2, 0, 1, 3]}[world] if scenario == "map" else None
        net = StarNet(None, rank, world, net_id, "shm", n_parties=o.n,
                      shm_bytes=world * (768 if pmap else 1024),       # 1 KiB (768 B: not a whole party row) slots: chunked
                      timeout_ms=400 if scenario not in ("flow", "map") else 20000, party_to_rank=pmap)
        k, first = net.k, net.first
        ok = True
        if scenario == "map":
            # MpcNet ids are arbitrary (lib.rs:43-53): the king sees the rows in ascending party order, every rank gets
            # the rows of ITS parties back, whatever the map
            sel = net.parties
            ok = sel == [p for p in range(o.n) if pmap[p] == rank] and first == sel[0]
            mask = net.enter(0)
            want = od.d_fft(shares, [od.FftMask.zero(Lc)] * o.n, False, dom, o, seed=3)
            for rnd in range(2):
                got = _d_fft_round(net, o, dom, 0, mask, [shares[p] for p in sel], False, 3, present=list(range(o.n)))
                ok = ok and got == [want[p] for p in sel]
            q.put((rank, bool(ok), ""))
        elif scenario == "flow":
            # three channels entered together, their rounds interleaved as zk_dist_circom_h does
            masks = [net.enter(sid) for sid in range(3)]
            assert masks == [(1 << world) - 1] * 3
            want = od.d_fft(shares, [od.FftMask.zero(Lc)] * o.n, False, dom, o, seed=3)
            for sid in (2, 0, 1):
                got = _d_fft_round(net, o, dom, sid, masks[sid], shares[first:first + k], False, 3)
                ok = ok and got == want[first:first + k]
            # a second round on channel 0 without re-entering, and an empty transfer
            got = _d_fft_round(net, o, dom, 0, masks[0], shares[first:first + k], True, 4)
            ok = ok and got == od.d_fft(shares, [od.FftMask.zero(Lc)] * o.n, True, dom, o, seed=4)[first:first + k]
            net.gather(1, masks[1], np.zeros(0, dtype=np.uint64), 0, np.zeros(0, dtype=np.uint64) if rank == 0 else None)
            # d_msm's small messages: every rank sends one value, the king sums, everybody gets the sum
            mk = net.enter(3)
            mine = np.array([rank + 1, 7], dtype=np.uint64)
            allv = np.zeros((world, 2), dtype=np.uint64) if rank == 0 else None
            net.gather_host(3, mk, mine, allv)
            tot = np.array([int(allv[:, 0].sum()), 0], dtype=np.uint64) if rank == 0 else np.zeros(2, dtype=np.uint64)
            net.bcast_host(3, mk, tot)
            ok = ok and int(tot[0]) == world * (world + 1) // 2
            q.put((rank, bool(ok), ""))
        elif scenario == "a2a":
            # the all-to-all of the second-stage king: multi-chunk blocks, slot reuse across rounds, a star verb in between
            mask = net.enter(1)
            nb = 1000                                            # bytes per peer: several sub-slot chunks

            def block(src, dst, rnd):
                return ((np.arange(nb, dtype=np.uint32) * 7 + src * 131 + dst * 17 + rnd) % 251).astype(np.uint8)
            for rnd in range(3):
                send = np.concatenate([block(rank, r, rnd) for r in range(world)])
                recv = np.zeros(world * nb, dtype=np.uint8)
                net.alltoall(1, mask, send, nb, recv)
                for i in range(world):
                    ok = ok and np.array_equal(recv[i * nb:(i + 1) * nb], block(i, rank, rnd))
                if rnd == 1:
                    g = np.full(8, rank, dtype=np.uint64)
                    full = np.zeros(8 * world, dtype=np.uint64) if rank == 0 else None
                    net.gather(1, mask, g, g.nbytes, full)
                    if rank == 0:
                        ok = ok and all(int(full[8 * r]) == r for r in range(world))
            st = net.stats()
            ok = ok and st["alltoalls"] == 3 and st["gathers"] == 1 and st["scatters"] == 0
            ok = ok and st["bytes_sent"] == 3 * nb * (world - 1) + (64 if rank else 0)
            q.put((rank, bool(ok), ""))
        elif scenario == "late":
            # the last rank shows up after the timeout: everyone else continues without its parties (lagrange path)
            if rank == world - 1:
                time.sleep(1.5)                                 # enters after the king's 400 ms timeout
                try:
                    net.enter(0)
                    q.put((rank, False, "late rank was admitted"))
                except ZkError as e:
                    ok = e.code == 2 and e.party == first
                    try:                                        # and it stays out: no later round admits it
                        net.enter(1)
                        ok = False
                    except ZkError as e2:
                        ok = ok and e2.code == 2
                    q.put((rank, ok, "code %d party %d" % (e.code, e.party)))
            else:
                mask = net.enter(0)
                assert mask == (1 << (world - 1)) - 1, mask
                t0 = time.time()
                assert net.enter(1) == mask                     # the king no longer waits for the rank it left out
                assert time.time() - t0 < 0.35, time.time() - t0
                got = _d_fft_round(net, o, dom, 0, mask, shares[first:first + k], False, 3)
                parties = list(range((world - 1) * k))
                want = od.d_fft(shares, [od.FftMask.zero(Lc)] * o.n, False, dom, o, seed=3, parties=parties)
                ok = got == want[first:first + k]
                # the all-to-all among the present ranks: blocks addressed by rank, received compacted
                send = np.array([rank * 100 + r for r in range(world)], dtype=np.uint64)
                recv = np.zeros(world - 1, dtype=np.uint64)
                net.alltoall(0, mask, send, 8, recv)
                ok = ok and [int(v) for v in recv] == [r * 100 + rank for r in range(world - 1)]
                # reconstructs: the outputs of the present parties alone determine the transform (dropout tolerance)
                q.put((rank, bool(ok), ""))
        net.close()
    except Exception as e:      # noqa: BLE001
        import traceback
        q.put((rank, False, traceback.format_exc()[-1500:] + repr(e)))


def _run(world, scenario):
    _setup()
    from zksaas_amd.net import StarNet
    net_id = StarNet.unique_id()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, net_id, scenario, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    bad = [r for r in results if not r[1]]
    assert not bad, bad


@pytest.mark.parametrize("world", [2, 4])
def test_star_rounds_over_shared_memory(world):
    _run(world, "flow")


@pytest.mark.parametrize("world", [2, 4])
def test_arbitrary_party_to_rank_map(world):
    _run(world, "map")


@pytest.mark.parametrize("world", [2, 4, 8])
def test_alltoall_over_shared_memory(world):
    _run(world, "a2a")


def test_a_late_rank_is_left_out_and_the_king_continues():
    # n = 8, l = t = 2: reconstruction needs more than 2 (t + l - 1) = 6 shares (pss.rs:183-186), so exactly one party may
    # be missing: one party per rank, the last rank is late
    _run(8, "late")


def test_net_create_rejects_bad_layouts():
    _setup()
    from zksaas_amd.net import StarNet
    from zksaas_amd._lib import ZkError
    with pytest.raises(ZkError):
        StarNet(None, 0, 3, StarNet.unique_id(), "shm", n_parties=8)      # 3 does not divide 8
    with pytest.raises(ZkError):
        StarNet(None, 0, 2, StarNet.unique_id(), "rccl", n_parties=8)     # RCCL needs a GPU context
    with pytest.raises(ZkError):
        StarNet(None, 0, 2, StarNet.unique_id(), "shm", n_parties=8, party_to_rank=[0, 0, 0, 0, 0, 1, 1, 1])   # 5 + 3
    net = StarNet(None, 0, 1, None, "local", n_parties=8)                 # world 1: local transport, no id needed
    assert (net.rank, net.world, net.first, net.k) == (0, 1, 0, 8)
    assert net.enter(0) == 1
    a = np.arange(16, dtype=np.uint64)
    b = np.zeros(16, dtype=np.uint64)
    net.gather(0, 1, a, a.nbytes, b)
    assert np.array_equal(a, b)
    net.close()
