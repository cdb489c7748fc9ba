"""One process per GPU: the launcher side of the star network (the network itself is in the library: csrc/net.hpp
behind zk_net_* / zk_dist_*, mirrored by zksaas_amd/net.py).

What is here: slicing a full dealing into the rows of this rank's parties, and the `bench.py --gpus N --workload ...`
drivers for the BASELINE configs:
    c2  d_fft, m = 2^20, BN254          (dist-primitives/examples/dfft_test.rs via scripts/dfft_test.zsh)
    c3  d_msm, 2^20 G1 points per party (dist-primitives/examples/dmsm_bench.rs via scripts/dmsm_bench.zsh)
    c4  Groth16 on the SHA-256 circuit  (groth16/examples/sha256.rs)  -- the headline metric
    c5  BLS12-381, 2^24 - 2 constraints, d_fft + d_msm + deg_red composed
torch.distributed (gloo, CPU) is used for the bootstrap only -- handing rank 0's net id to the other ranks, the
barriers that bracket the timed region and the max over ranks; nothing on the data path goes through it.
"""
import os
import time

import numpy as np

from . import groth16 as zg
from . import net as znet
from . import api
from .api import ZK_G1, ZK_G2, DeviceBuffer, FftMask


def party_range(rank, world, n):
    k = n // world
    return rank * k, k


def rows(buf, first, k, row_bytes):
    """the contiguous rows [first, first + k) of a party-major device buffer [n][row]"""
    return buf.view(first * row_bytes, k * row_bytes)


def rows_of(pp, buf, parties, row_bytes):
    """the rows of the listed parties of a party-major device buffer [n][row]: a view when they are a contiguous range,
    otherwise a new buffer (setup-time copy through the host; a rank's parties under a general party_to_rank map)"""
    parties = [int(p) for p in parties]
    if parties == list(range(parties[0], parties[0] + len(parties))):
        return rows(buf, parties[0], len(parties), row_bytes)
    full = buf.to_numpy(np.uint8).reshape(-1, row_bytes)
    return DeviceBuffer.from_numpy(pp, np.ascontiguousarray(full[parties]))


class LocalCrs:
    """zk_crs_share over this rank's rows of a full PackedProvingKeyShare dealing (views, no copies)."""

    def __init__(self, pp, crs, first, k, parties=None):
        e1, e2 = 2 * pp.fq.nbytes, 4 * pp.fq.nbytes
        ps = list(range(first, first + k)) if parties is None else parties
        self.s = rows_of(pp, crs.s, ps, crs.len_a * e1)
        self.h = rows_of(pp, crs.h, ps, crs.len_a * e1)
        self.v = rows_of(pp, crs.v, ps, crs.len_a * e2)
        self.w = rows_of(pp, crs.w, ps, crs.len_w * e1)
        self.u = rows_of(pp, crs.u, ps, crs.len_u * e1)
        self.len_a, self.len_w, self.len_u, self.k = crs.len_a, crs.len_w, crs.len_u, k
        s1, s2 = crs.s1, crs.s2
        self._keep = (crs, s1, s2)
        self.ct = zg.CrsShare(self.s.ptr, self.h.ptr, self.v.ptr, self.w.ptr, self.u.ptr, crs.len_a, crs.len_w, crs.len_u,
                              s1[0].ctypes.data, s1[1].ctypes.data, s1[2].ctypes.data, s1[3].ctypes.data,
                              s1[4].ctypes.data, s2[0].ctypes.data, s2[1].ctypes.data, s2[2].ctypes.data)

    def precompute(self, pp):
        from . import api
        for buf, grp, ln in ((self.s, ZK_G1, self.len_a), (self.h, ZK_G1, self.len_a), (self.v, ZK_G2, self.len_a),
                             (self.w, ZK_G1, self.len_w), (self.u, ZK_G1, self.len_u)):
            api.msm_precompute(pp, grp, buf, self.k * ln)


def local_witness(pp, wit, first, k, parties=None):
    eb = pp.fr.nbytes
    Lc = (1 << wit.log_m) // pp.l
    ps = list(range(first, first + k)) if parties is None else parties
    qap = [rows_of(pp, q, ps, Lc * eb) for q in wit.qap]
    return qap, rows_of(pp, wit.a_share, ps, wit.len_a * eb), rows_of(pp, wit.ax_share, ps, wit.len_w * eb)


def local_masks(pp, masks, log_m, first, k, parties=None):
    """zk_groth16_masks over this rank's rows of a ProofMasks dealing; returns (struct, keep-alive list)"""
    if masks is None:
        return None, None
    eb = pp.fr.nbytes
    Lc = (1 << log_m) // pp.l
    ps = list(range(first, first + k)) if parties is None else [int(p) for p in parties]
    keep = []
    ct = zg.Masks()
    for i in range(6):
        a, b = rows_of(pp, masks.fft[i].in_mask, ps, Lc * eb), rows_of(pp, masks.fft[i].out_mask, ps, Lc * eb)
        keep += [a, b]
        ct.fft_in[i], ct.fft_out[i] = a.ptr, b.ptr
    a, b = rows_of(pp, masks.degred.in_mask, ps, Lc * eb), rows_of(pp, masks.degred.out_mask, ps, Lc * eb)
    keep += [a, b]
    ct.degred_in, ct.degred_out = a.ptr, b.ptr
    for i in range(5):
        a = np.ascontiguousarray(masks.msm[i].in_mask[ps])
        b = np.ascontiguousarray(masks.msm[i].out_mask[ps])
        keep += [a, b]
        ct.msm_in[i], ct.msm_out[i] = a.ctypes.data, b.ctypes.data
    return ct, keep


def _bootstrap(rank, world):
    """rank 0's net id to everyone (setup, untimed); returns (dist module or None, id)"""
    if world == 1:
        return None, znet.StarNet.unique_id()
    import torch.distributed as dist
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    box = [znet.StarNet.unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    return dist, box[0]


def _open_net(pp, rank, world, dist, net_id, transport, king):
    """The star network of the run and the king mode it runs in.  RCCL between several ranks has never run on the 1-GPU
    boxes this was developed on, so the multi-rank bench probes its configuration with one small d_fft round before the
    timed region; if any rank reports a failure every rank moves to the next configuration -- (rccl, all-to-all king) ->
    (rccl, star king) -> (shared memory, star king; the same verbs staged through host memory: slower, and said so in
    the result line) -- instead of losing the measurement.  Returns (net, transport, king, note or None)."""
    from .api import ZkError
    if world == 1:
        return znet.StarNet(pp, rank, world, net_id, transport), transport, "star", None
    import torch
    chain = [(transport, king)]
    if king == "alltoall":
        chain.append((transport, "star"))
    if transport == "rccl":
        chain.append(("shm", "star"))
    notes = []
    net = None
    for attempt, (tr, kg) in enumerate(chain):
        err = None
        try:
            if net is None or tr != net_tr:
                if net is not None:
                    net.close()
                    net = None
                if attempt:
                    box = [znet.StarNet.unique_id() if rank == 0 else None]
                    dist.broadcast_object_list(box, src=0)
                    net_id = box[0]
                net = znet.StarNet(pp, rank, world, net_id, tr)
                net_tr = tr
            pp.set_option("king_alltoall", 1 if kg == "alltoall" else 0)
            net.lib.zk_net_set_timeout_ms(net.h, 20000)
            log_m = 14                                   # 32 king workgroup columns: enough for 8 chunk ranges
            x = _rand_fr(pp, net.k * ((1 << log_m) // pp.l), 5 + rank)
            znet.dist_d_fft(pp, net, 0, x, None, False, log_m, seed=1)
            # the channel first: zk_net_sync waits with the timeout as a DEADLINE (a hung RCCL collective is aborted
            # there and reported); a plain stream synchronise on the caller's stream -- which zk_dist_d_fft has made
            # wait for the channel stream -- would block forever behind the same hung collective
            net.sync(0)
            pp.sync()
            net.lib.zk_net_set_timeout_ms(net.h, 30000)
        except ZkError as e:
            err = "%s" % (e,)
        flag = torch.tensor([1 if err else 0], dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        if not int(flag.item()):
            return net, tr, kg, ("; ".join(notes) or None)
        msgs = [None] * world
        dist.all_gather_object(msgs, err)
        notes.append("%s transport with the %s king failed (%s)" % (tr, kg, next((m for m in msgs if m), "unknown")[:160]))
        if net is not None and err and "net:" in err:     # a failed collective leaves the channel state undefined
            net.close()
            net = None
    raise RuntimeError("no working transport: " + "; ".join(notes))


def _timed(dist, torch, step, steps, warmup):
    """W untimed steps, barrier + device sync, K timed steps, device sync + barrier, max over ranks"""
    for i in range(warmup):
        step(i)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    for i in range(steps):
        step(warmup + i)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    return dt


def _alltoall_leg(pp, dist, torch, step, steps, warmup, unit_scale=1.0):
    """N > 1, informational, after the headline (north_star's star: king on GPU 0, gather / scatter): the same K steps with
    every rank king of a chunk range (option king_alltoall, two all-to-all exchanges per round).  Every rank switches
    alike; a failure on any rank is reported in the leg's place and every rank returns to the star."""
    from .api import ZkError
    err, dt = None, None
    try:
        pp.set_option("king_alltoall", 1)
        dt = _timed(dist, torch, step, steps, max(1, warmup // 2))
    except ZkError as e:
        err = "%s" % (e,)
    finally:
        pp.set_option("king_alltoall", 0)
    flag = torch.tensor([1 if err else 0], dtype=torch.int32)
    dist.all_reduce(flag, op=dist.ReduceOp.MAX)
    if int(flag.item()):
        return {"king": "alltoall", "error": err or "another rank failed"}
    return {"king": "alltoall", "value": round(unit_scale * steps / dt, 4), "ms_per_step": round(dt / steps * 1e3, 4)}


def _step_traffic(pmc_file, parts):
    """HBM bytes of one step of a primitive from the committed counter summary of the same command (profiles/<pmc_file>,
    tools/refresh_profiles.sh; FETCH_SIZE / WRITE_SIZE in separate --pmc passes, corrected per access shape):
    sum over `parts` = [(kernel-name prefix, launches per step)] (None: once per step every msm_* kernel of the file).
    One-GPU figure (world = 1); None when the file is missing."""
    import json
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", pmc_file)
    try:
        ks = json.load(open(path))["kernels"]
    except (OSError, ValueError, KeyError):
        return None
    if parts is None:
        return int(sum(k["hbm_bytes_per_launch"] for k in ks if k["kernel"].startswith("msm_")))
    tot = 0
    for prefix, per_step in parts:
        hit = [k for k in ks if k["kernel"].startswith(prefix)]
        if not hit:
            return None
        tot += per_step * hit[0]["hbm_bytes_per_launch"]
    return int(tot)


XGMI_LINK_GBS = 153.0          # per direction and link; 7 links per GPU (MI355X_MICROARCH.md): a star's traffic crosses GPU 0's links


def _net_model(net, dist, torch, world, king, before, steps, s_per_step):
    """What crossed the net per step (zk_net_stats deltas of every rank over warm-up + timed steps) and what that costs on
    xGMI by a committed model, so that the first multi-GPU hardware curve can be checked against a prediction made before it
    (RCCL between several ranks has never run: the pool's boxes have one GPU).
    Star (gather / scatter through rank 0): every byte crosses exactly one of GPU 0's links, the world - 1 peers' links carry
    their shares in parallel, gather and scatter of a round follow each other: t = bytes_total / (world - 1) / 153 GB/s.
    All-to-all king: every rank sends world - 1 equal blocks over world - 1 links at once: t = max_rank(bytes) / (world - 1) / 153 GB/s."""
    after = net.stats()
    d = {k: (after[k] - before[k]) / steps for k in after}
    per_rank = [d["bytes_sent"]]
    if dist is not None:
        t = torch.tensor([d["bytes_sent"]], dtype=torch.float64)
        allr = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(allr, t)
        per_rank = [float(x.item()) for x in allr]
    out = {"net_per_step": {"gathers": round(d["gathers"], 2), "scatters": round(d["scatters"], 2),
                            "alltoalls": round(d["alltoalls"], 2), "bytes_sent_by_rank": [int(b) for b in per_rank],
                            "bytes_total": int(sum(per_rank))}}
    if world > 1:
        links = world - 1
        if king == "star":
            t_x = sum(per_rank) / links / (XGMI_LINK_GBS * 1e9)
        else:
            t_x = max(per_rank) / links / (XGMI_LINK_GBS * 1e9)
        out["xgmi_prediction"] = {"links_per_direction": links, "link_GBps": XGMI_LINK_GBS, "king": king,
                                  "seconds_per_step": round(t_x, 9), "share_of_ms_per_step": round(t_x / s_per_step, 4),
                                  "model": "star: bytes_total / (world - 1) links / 153 GB/s (every byte crosses one link of GPU 0; "
                                           "gather and scatter are sequential); all-to-all king: max over ranks instead of the total. "
                                           "Latency per exchange (~10-20 us per ncclGroupEnd) is NOT in the model: %d exchanges per "
                                           "step" % int(round(d["gathers"] + d["scatters"] + 2 * d["alltoalls"]))}
    return out


def _king_words(king, world):
    if world == 1:
        return "all parties on one GPU"
    return ("king on GPU 0, RCCL gather / scatter (the reference's star)" if king == "star"
            else "all-to-all king: every GPU reconstructs and re-packs a chunk range")


def _rand_fr(pp, count, seed):
    rng = np.random.default_rng(seed)
    a = rng.integers(0, 1 << 62, size=(count, 4), dtype=np.uint64)
    a[:, 3] &= np.uint64((1 << 60) - 1)
    return DeviceBuffer.from_numpy(pp, a)


def bench(args, rank, local_rank, world):
    """bench.py's sharded leg (N > 1, or any workload other than the single-GPU c4 line)."""
    import torch
    import zksaas_amd as zk
    from bench import HBM_PEAK_GBS, MAD_ISSUE_BOUND_G, build_inputs, read_profile, roofline_of, same_shares

    transport = os.environ.get("ZK_NET", "rccl" if world > 1 else "local")
    dist, net_id = _bootstrap(rank, world)
    wl = args.workload
    curve = "bls12_381" if wl == "c5" else "bn254"
    pp = zk.PackedSharingParams(curve, 2, device=local_rank)
    for kv in filter(None, os.environ.get("ZK_BENCH_OPTIONS", "").split(",")):     # A/B runs: name=value context options
        pp.set_option(kv.split("=")[0], int(kv.split("=")[1]))
    if pp.n % world:
        raise SystemExit("the number of GPUs must divide n = %d parties" % pp.n)
    king = os.environ.get("ZK_KING", getattr(args, "king", "star"))
    net, transport, king, note = _open_net(pp, rank, world, dist, net_id, transport, king)
    first, k = net.first, net.k
    base = {"n_gpus": world, "steps": args.steps, "warmup": args.warmup, "higher_is_better": True, "vs_baseline": None,
            "dtype": "u32 limbs (%s Montgomery)" % ("255-bit Fr / 381-bit Fq" if wl == "c5" else "256-bit"),
            "transport": transport, "king": king, "parties_per_gpu": k,
            # size of the RCCL communicators this run actually initialised and used (0: the data plane did NOT run over
            # RCCL -- one rank, or the shared-memory fallback; see transport_note)
            "rccl_ranks": world if transport == "rccl" else 0}
    if note:
        base["transport_note"] = note
    if world > 1 and transport != "rccl" and os.environ.get("ZK_NET", "rccl") == "rccl":
        # asked for RCCL over xGMI, measured something else (host-staged shared memory): the number is NOT a scaling point
        base["degraded"] = True
    eb = pp.fr.nbytes
    per = lambda dt: dt / args.steps
    # ONE dealer: every rank derives the same dealing (witness / QAP shares, masks) from the seeds below and keeps its
    # parties' rows, so the dealing runs on the replayable stream; a context's production stream is keyed from
    # /dev/urandom and two ranks would otherwise hold rows of two different sharings (found by running 4 ranks without
    # the replay option: every rank's proof shares were right for ITS dealing and the assembled proof was not a proof).
    # The timed proofs run on the production stream again (the kings' share randomness need not agree).
    pp.set_option("rng_replay", 1)

    def timed(step):
        if not api.DEFAULT_OPTIONS.get("rng_replay"):
            pp.set_option("rng_replay", 0)
        before = net.stats()
        dt = _timed(dist, torch, step, args.steps, args.warmup)
        if "net_per_step" not in base:               # the headline's region (later calls are side legs)
            base.update(_net_model(net, dist, torch, world, king, before, args.steps + args.warmup, dt / args.steps))
        return dt
    if wl == "c2":
        log_m = 20
        m = 1 << log_m
        Lc = m // pp.l
        sh = _rand_fr(pp, k * Lc, 100 + rank)
        full_mask = None if args.no_masks else FftMask.sample(pp, False, None, 0, log_m, 11)
        mk = FftMask.zero() if full_mask is None else FftMask(rows(full_mask.in_mask, first, k, Lc * eb),
                                                             rows(full_mask.out_mask, first, k, Lc * eb))
        step = lambda i: znet.dist_d_fft(pp, net, 0, sh, mk, False, log_m, seed=i)
        dt = timed(step)
        alg = (32 if full_mask is not None else 16) * m * 32          # all parties, with / without masks
        gbs = alg / per(dt) / 1e9
        res = dict(base, metric="d_fft per second (m = 2^20, BN254 Fr, l = 2, n = 8)", value=round(args.steps / dt, 3),
                   unit="d_fft/s", ms_per_step=round(per(dt) * 1e3, 4), scaling="strong",
                   data="synthetic: seeded random share vectors",
                   config={"workload": "BASELINE configs[1]: d_fft 2^20-point BN254 Fr NTT, %s" % _king_words(king, world),
                           "m": m, "masks": full_mask is not None},
                   roofline={"bound": "hbm", "kernel": "d_fft end to end (all ranks)", "achieved": round(gbs, 1),
                             "peak": HBM_PEAK_GBS * world, "unit": "GB/s", "frac": round(gbs / (HBM_PEAK_GBS * world), 5),
                             "traffic": _step_traffic("r06_c2_pmc_hbm.json", [("ntt_pass_kernel", 2), ("king_fft2_kernel", 1)])
                             if world == 1 and full_mask is not None else None,
                             "traffic_source": "profiles/r06_c2_pmc_hbm.json: two ntt_pass_kernel launches + one "
                                               "king_fft2_kernel launch per d_fft",
                             "algorithmic_bytes_per_launch": alg})
    elif wl == "c3":
        ln = 1 << 20
        # distinct bases: seeded random multiples of the generator (a tiled single point makes doublings and
        # cancellations the common case of the mixed addition: 64 % lane utilisation in the accumulate kernel)
        bases = zg.base_points(pp, ZK_G1, _rand_fr(pp, k * ln, 300 + rank), k * ln)
        sc = _rand_fr(pp, k * ln, 200 + rank)
        dt = timed(lambda i: znet.dist_d_msm(pp, net, 0, ZK_G1, bases, sc, ln))
        # the same d_msm over a REGISTERED base vector (zk_msm_precompute: a prover's CRS is fixed; window bits by the
        # vector's length): a side figure, `value` stays the reference's call (bases handed over per call, no table)
        fixed = None
        if not args.no_tables:
            from .api import msm_forget, msm_precompute, msm_table_info
            msm_precompute(pp, ZK_G1, bases, k * ln)
            try:
                dtf = timed(lambda i: znet.dist_d_msm(pp, net, 0, ZK_G1, bases, sc, ln))
                info = msm_table_info(pp, ZK_G1, bases)
            finally:
                msm_forget(pp, bases)
            fixed = {"value": round(args.steps / dtf, 3), "ms_per_step": round(per(dtf) * 1e3, 4), "table": info,
                     "table_GB_per_rank": round(info["windows"] * k * ln * 2 * pp.fq.nbytes / 2**30, 2)}
        alg = pp.n * ln * 96
        from .api import msm_plan
        plan = msm_plan(pp, ZK_G1, k * ln)
        muls = pp.n * ln * plan["windows"] * plan["muls_per_add"]
        gbs, gm = alg / per(dt) / 1e9, muls / per(dt) / 1e9
        res = dict(base, metric="d_msm per second (2^20 G1 points per party, BN254)", value=round(args.steps / dt, 3),
                   unit="d_msm/s", ms_per_step=round(per(dt) * 1e3, 4), scaling="strong",
                   points_per_sec=round(pp.n * ln * args.steps / dt, 1), fixed_base=fixed,
                   data="synthetic: seeded random multiples of the generator as bases, seeded scalars",
                   config={"workload": "BASELINE configs[2]: d_msm 2^20 G1 Pippenger per party (BN254), 8 parties",
                           "plan": plan},
                   roofline={"bound": "hbm", "kernel": "d_msm end to end (all ranks)", "achieved": round(gbs, 1),
                             "peak": HBM_PEAK_GBS * world, "unit": "GB/s", "frac": round(gbs / (HBM_PEAK_GBS * world), 5),
                             "traffic": _step_traffic("r06_c3_pmc_hbm.json", None) if world == 1 and not fixed else None,
                             "traffic_source": "profiles/r06_c3_pmc_hbm.json: every msm_* kernel of one table-free d_msm "
                                               "(the accumulate kernel's gathers are 13.1 of the 16 GB)",
                             "algorithmic_bytes_per_launch": alg,
                             "alu": {"achieved": round(gm, 1), "peak": round(MAD_ISSUE_BOUND_G * world, 1),
                                     "unit": "G modmul/s", "frac": round(gm / (MAD_ISSUE_BOUND_G * world), 3)}})
    elif wl == "c4":
        r1, w, setup, crs, wit, r, s = build_inputs(pp, zk)
        masks = None if args.no_masks else zg.ProofMasks(pp, wit.log_m, seed=77)
        lcrs = LocalCrs(pp, crs, first, k)
        qap, a_sh, ax_sh = local_witness(pp, wit, first, k)
        mct, keep = local_masks(pp, masks, wit.log_m, first, k)
        if not args.no_tables:
            lcrs.precompute(pp)
        out = {}

        def step(i):
            out["proof"] = znet.dist_prove(pp, net, lcrs.ct, qap, a_sh, ax_sh, r, s, wit.log_m, masks=mct, seed=2000 + i)
        step(0)
        pp._check(pp.lib.zk_profile_enable(pp.h, 1))
        dt = timed(step)
        prof = read_profile(pp)
        pp._check(pp.lib.zk_profile_enable(pp.h, 0))
        # untimed cross-check: every party's share of the sharded proof equals the single-context one (as group elements)
        ok = None
        mine = np.concatenate([x.reshape(-1) for x in out["proof"]])
        if dist is not None:
            parts = [None] * world
            dist.all_gather_object(parts, mine)
        else:
            parts = [mine]
        if rank == 0:
            nl = pp.fq.nl
            sizes = [k * 3 * nl, k * 6 * nl, k * 3 * nl]
            cols = [[], [], []]
            for p in parts:
                o = 0
                for j, sz in enumerate(sizes):
                    cols[j].append(p[o:o + sz].reshape(k, -1))
                    o += sz
            full = tuple(np.concatenate(c) for c in cols)
            ref = zg.prove(pp, crs, wit, r, s, masks=masks, seed=1)
            ok = bool(same_shares(pp, full, ref))
        # ---- throughput modes, outside the timed K steps (`value` stays one proof at a time).  A leg must never cost the
        # headline: it runs without launcher collectives inside (a rank that fails would leave the others in a mismatched
        # one), waits for the channels with the net's deadline before synchronising the device (a hung collective then
        # raises instead of hanging), and the ranks agree on the outcome afterwards; a failure is reported in the leg's place.
        from . import wire

        def same_as_headline(p_):
            return all(wire.jacobian_to_affine(pp, p_[kk][q_], g2) == wire.jacobian_to_affine(pp, out["proof"][kk][q_], g2)
                       for kk, g2 in ((0, False), (1, True), (2, False)) for q_ in range(k))

        def drain():
            for sid in range(4):
                net.sync(sid)
            torch.cuda.synchronize()

        def guarded(fn):
            err, got = None, None
            try:
                got = fn()
            except Exception as e:      # noqa: BLE001
                err = repr(e)[:300]
            vals = torch.tensor([0.0 if err else 1.0, 0.0 if err else got[0], 0.0 if err else float(got[1])],
                                dtype=torch.float64)
            if dist is not None:
                okv = vals.clone()
                dist.all_reduce(okv, op=dist.ReduceOp.MIN)
                dist.all_reduce(vals, op=dist.ReduceOp.MAX)
                vals[0], vals[2] = okv[0], okv[2]
            if float(vals[0]) == 0.0:
                return None, None, {"error": err or "another rank failed"}
            return float(vals[1]), bool(float(vals[2])), None

        # batches of 8 proofs per collective call (zk_dist_groth16_prove_batch): every rank runs each of its MSMs once per batch
        batched = in_flight = None
        if not os.environ.get("ZK_BENCH_NO_BATCH"):
            nbp, nbat = 8, max(2, min(8, args.steps // 8))
            mk_b = None if mct is None else [mct] * nbp

            def leg_batched():
                proofs = None
                znet.dist_prove_batch(pp, net, lcrs.ct, [qap] * nbp, [a_sh] * nbp, [ax_sh] * nbp, [r] * nbp, [s] * nbp,
                                      wit.log_m, masks=mk_b, seed=5000)
                drain()
                t0 = time.perf_counter()
                for i in range(nbat):
                    proofs = znet.dist_prove_batch(pp, net, lcrs.ct, [qap] * nbp, [a_sh] * nbp, [ax_sh] * nbp, [r] * nbp,
                                                   [s] * nbp, wit.log_m, masks=mk_b, seed=5001 + i)
                drain()
                # every proof of the batch is the same statement: its local shares equal the one-at-a-time proof's as
                # group elements (compared on this rank's rows)
                return time.perf_counter() - t0, all(same_as_headline(p_) for p_ in proofs)
            bdt, eq, err = guarded(leg_batched)
            batched = err or {"batch": nbp, "batches": nbat, "proofs_per_s": round(nbp * nbat / bdt, 2),
                              "ms_per_proof": round(bdt / (nbp * nbat) * 1e3, 4), "same_proof": eq,
                              "api": "zk_dist_groth16_prove_batch"}
            # two proofs in flight per rank (zk_dist_groth16_prove_async / _wait): a proof's king rounds overlap the
            # previous proof's MSMs.  Rolling window over 2 * steps proofs, every rank in the same order.
            nfl = max(4, 2 * args.steps)

            def leg_in_flight():
                def fly(i):
                    return znet.dist_prove_async(pp, net, lcrs.ct, qap, a_sh, ax_sh, r, s, wit.log_m, masks=mct, seed=7000 + i)
                fly(0).wait()
                drain()
                t0 = time.perf_counter()
                prev, last = fly(0), None
                for i in range(1, nfl):
                    cur = fly(i)
                    last = prev.wait()
                    prev = cur
                last = prev.wait()
                drain()
                return time.perf_counter() - t0, same_as_headline(last)
            fdt, eq, err = guarded(leg_in_flight)
            in_flight = err or {"proofs_in_flight_per_rank": 2, "proofs": nfl, "proofs_per_s": round(nfl / fdt, 2),
                                "ms_per_proof": round(fdt / nfl * 1e3, 4), "same_proof": eq,
                                "api": "zk_dist_groth16_prove_async / zk_dist_groth16_wait"}
        proofs_per_s = args.steps / dt
        res = dict(base, metric="Groth16 proofs/sec (SHA-256 circuit)", value=round(proofs_per_s, 3), unit="proofs/s",
                   batched=batched, in_flight=in_flight,
                   ms_per_step=round(per(dt) * 1e3, 4), scaling="strong",
                   data="synthetic: SHA-256(a=1,b=2) circuit rebuilt from its semantics and padded to the reference "
                        "fixture's 29 823 wires, seeded trapdoor CRS, seeded shares and masks",
                   config={"workload": "BASELINE configs[3]: full distributed Groth16 on the SHA-256 circuit, BN254, l=2, "
                                       "n=8 parties sharded over %d GPUs (%s), %s" % (
                                           world, _king_words(king, world),
                                           "zero masks" if masks is None else "all 12 masks sampled and applied"),
                           "masks": masks is not None, "constraints": r1.num_constraints, "wires": r1.num_variables,
                           "domain": 1 << wit.log_m, "len_a": crs.len_a, "len_w": crs.len_w, "len_u": crs.len_u,
                           "parties": pp.n, "packing_factor": pp.l, "fixed_base_tables": not args.no_tables},
                   constraints_per_sec=round(proofs_per_s * r1.num_constraints, 1), proof_matches_single_gpu=ok,
                   roofline=roofline_of(prof, ntt_passes=2, masks_on=masks is not None, pp=pp),
                   cpu_baseline={"value": None, "unit": "proofs/s", "kind": "port",
                                 "note": "timed on rank 0 at N = 1 only (bench contract): see the N = 1 line"},
                   kernels=[{**e, "total_ms": round(e["total_ms"], 3)} for e in prof if e["launches"]])
    else:   # c5
        from . import synthetic
        log_m = int(os.environ.get("ZK_C5_LOG_M", "24"))
        inst = synthetic.SyntheticInstance(pp, log_m, seed=1, parties=(first, k))
        wit = inst.witness(seed=100)
        r, s = 0x1234567890ABCDEF1234567890ABCDEF, 0xFEDCBA0987654321FEDCBA0987654321
        out = {}
        # all twelve masks of a proof (six FftMask, the DegRedMask, five MsmMask), dealt as sha256.rs:226-291 does and
        # applied, as the c4 line does: 28 GiB of mask vectors at 2^24 (--no-masks: the zero masks the reference's bench
        # scripts run with)
        masks = None if args.no_masks else zg.ProofMasks(pp, log_m, seed=77)
        mct, keep = local_masks(pp, masks, log_m, first, k)

        def step(i):
            out["proof"] = znet.dist_prove(pp, net, inst.crs.ct, wit.qap, wit.a_share, wit.ax_share, r, s, log_m,
                                           masks=mct, seed=7 + i)
        step(0)
        pp._check(pp.lib.zk_profile_enable(pp.h, 1))
        from bench import msm_stats
        st0 = msm_stats(pp)
        dt = timed(step)
        st1 = msm_stats(pp)
        prof = read_profile(pp)
        pp._check(pp.lib.zk_profile_enable(pp.h, 0))
        # whole-proof multiplier utilisation of THIS rank (the per-launch `roofline.alu` divides one
        # launch's products by a duration during which four other MSMs share the chip): products of the mixed additions the
        # accumulate kernels performed (zk_msm_stats) over the wall time.  12 limbs: a product is 2 N^2 + N = 300 multiply
        # instructions; G1 addition 9.47 products as executed, G2 28 (8 Fq2 products + 2 squarings, schoolbook per lane pair).
        npf = args.steps + args.warmup
        a1, a2 = (st1[0] - st0[0]) / npf, (st1[1] - st0[1]) / npf
        muls12 = a1 * 9.47 + a2 * 28.0
        bound12 = MAD_ISSUE_BOUND_G * 136.0 / 300.0
        palu = {"modmuls_per_proof_this_rank": int(muls12), "additions_per_proof": {"g1": int(a1), "g2": int(a2)},
                "achieved": round(muls12 / per(dt) / 1e9, 2), "unit": "G modmul/s (381-bit) over the proof's wall time",
                "frac_issue_bound": round(muls12 / per(dt) / 1e9 / bound12, 3),
                "frac_of_measured_multiplier": round(muls12 / per(dt) / 1e9 / 60.1, 3),
                "issue_bound_G": round(bound12, 1), "measured_multiplier_G": 60.1}
        # dominant slot of the timed region with BLS12-381's algorithmic bytes (affine base + scalar per point:
        # 2 x 48 + 32 = 128 B in G1, 4 x 48 + 32 = 224 B in G2) and the 12-limb multiplier's issue bound
        # (the sort slot is left out of the choice: with five multi-hundred-millisecond MSMs in flight its HIP-event span
        # is mostly the time its one-workgroup-per-CU kernels WAIT for CUs the accumulate kernels of the other MSMs hold --
        # 0.43 s per sort against 25 ms of execution, profiles/r04_c5_kernel_stats.csv; `kernels` below still lists it)
        roof = roofline_of(prof, ntt_passes=3, masks_on=masks is not None, pp=pp, limbs=12,
                           slot_bytes={"msm_accumulate_kernel<G1>": 128.0, "msm_accumulate_kernel<G2>": 224.0},
                           pmc_file="r06_c5_pmc_hbm.json")
        cpu = None
        if rank == 0 and world == 1 and not args.no_cpu_baseline:
            del wit
            from bench import cpu_baseline_c5
            cpu = cpu_baseline_c5(pp, zk, log_m)
        res = dict(base, metric="Groth16 proofs/sec (BLS12-381, 2^%d - 2 constraints)" % log_m,
                   value=round(args.steps / dt, 4), unit="proofs/s", ms_per_step=round(per(dt) * 1e3, 2), scaling="strong",
                   constraints_per_sec=round(inst.nc * args.steps / dt, 1),
                   data="synthetic R1CS and pseudo-random CRS built on the device (zksaas_amd/synthetic.py)",
                   config={"workload": "BASELINE configs[4]: BLS12-381 2^%d-constraint synthetic R1CS, d_fft + d_msm + "
                                       "deg_red composed, %s" % (log_m, "zero masks" if masks is None else
                                                                 "all 12 masks sampled and applied"),
                           "masks": masks is not None, "constraints": inst.nc, "parties": pp.n},
                   roofline=roof, proof_alu=palu, cpu_baseline=cpu,
                   kernels=[{**e, "total_ms": round(e["total_ms"], 3)} for e in prof if e["launches"]])
    # N > 1: the headline ran in the mode `king` names (default: north_star's star); the other stage of SURVEY.md 8e in the
    # same run, beside it
    if world > 1 and king == "star" and wl in ("c2", "c4", "c5"):
        res["alltoall"] = _alltoall_leg(pp, dist, torch, step, args.steps, args.warmup)
    net.close()
    return res
