"""The MPC star topology on one multi-GPU node: one process per GPU (torch.distributed, backend "nccl" = RCCL
over xGMI), the n = 4l parties split over the ranks, king = party 0 on rank 0.

mpc-net/src/lib.rs:89-176 defines two collectives -- every party sends to the king
(`client_send_or_king_receive`) and the king sends each party its own answer
(`client_receive_or_king_send`) -- i.e. gather and scatter.  Here they are
`torch.distributed.gather` / `scatter` on device tensors holding raw Montgomery limbs (no serialization);
d_msm's "king sums and broadcasts one point" is an all-gather of one partial point per rank.

Data-path compute goes through a small backend object so that the protocol flow can be exercised on CPU
(gloo, world_size 2) with a stand-in backend in the tests; the product backend is `GpuBackend` (C ABI).
"""
import ctypes as C
import os
import time

import numpy as np

from . import groth16 as zg
from .api import ZK_G1, ZK_G2


def party_range(rank, world, n):
    k = n // world
    return rank * k, k


class GpuBackend:
    """All compute through libzksaas_hip.so on torch int64 tensors [..., limbs] resident on this rank's GPU."""

    def __init__(self, pp):
        import torch
        self.torch = torch
        self.pp = pp
        self.device = torch.device("cuda", torch.cuda.current_device())

    def stream(self):
        return C.c_void_p(self.torch.cuda.current_stream().cuda_stream)

    def empty(self, *shape):
        return self.torch.empty(*shape, dtype=self.torch.int64, device=self.device)

    def from_numpy(self, arr):
        return self.torch.from_numpy(np.ascontiguousarray(arr).view(np.int64)).to(self.device)

    def fft1(self, t, log_m, inverse):
        pp = self.pp
        batch = t.numel() // (4 * ((1 << log_m) // pp.l))
        pp._check(pp.lib.zk_fft1(pp.h, t.data_ptr(), log_m, int(inverse), batch, None, self.stream()))

    def king_fft2(self, tin, tout, log_m, inverse, g, scale, rearrange, seed):
        pp = self.pp
        garr = None if g is None else pp.fr.encode_one(g)
        pp._check(pp.lib.zk_fft2_king(pp.h, tin.data_ptr(), None, pp.n, log_m, int(inverse),
                                      None if garr is None else garr.ctypes.data, int(scale), int(rearrange), seed,
                                      tout.data_ptr(), None, self.stream()))

    def mul_sub(self, out, a, b, c):
        pp = self.pp
        pp._check(pp.lib.zk_vec_mul_sub(pp.h, out.data_ptr(), a.data_ptr(), b.data_ptr(), c.data_ptr(),
                                        out.numel() // 4, self.stream()))

    def deg_red(self, x, length, seed):
        pp = self.pp
        pp._check(pp.lib.zk_deg_red(pp.h, x.data_ptr(), None, None, length, seed, self.stream()))

    def msm_local(self, group, bases, scalars, length, first, count):
        pp = self.pp
        nl = pp.fq.nl * (2 if group == ZK_G2 else 1)
        out = np.zeros(3 * nl, dtype=np.uint64)
        pp._check(pp.lib.zk_d_msm_local(pp.h, group, bases.data_ptr(), scalars.data_ptr(), length, first, count, None,
                                        out.ctypes.data, self.stream()))
        return out

    def msms_begin(self, inp, first, count, skip_h):
        """start S, H, V, W for this rank's parties (overlaps the circom_h rounds)"""
        pp = self.pp
        self._local_crs = zg.CrsShare(inp["s"].data_ptr(), inp["h"].data_ptr(), inp["v"].data_ptr(),
                                      inp["w"].data_ptr(), inp["u"].data_ptr(), inp["len_a"], inp["len_w"],
                                      inp["u"].shape[1], *[getattr(inp["crs_ct"], f) for f in (
                                          "a_query0", "b_g1_query0", "delta_g1", "alpha_g1", "beta_g1", "b_g2_query0",
                                          "delta_g2", "beta_g2")])
        pp._check(pp.lib.zk_groth16_msms_begin(pp.h, C.byref(self._local_crs), inp["a_share"].data_ptr(),
                                               inp["ax_share"].data_ptr(), first, count, int(skip_h), None,
                                               self.stream()))

    def msms_finish(self, h, first, count):
        pp = self.pp
        nl = pp.fq.nl
        outs = [np.zeros(3 * nl * (2 if i == 2 else 1), dtype=np.uint64) for i in range(5)]
        arr = (C.c_void_p * 5)(*[x.ctypes.data for x in outs])
        pp._check(pp.lib.zk_groth16_msms_finish(pp.h, h.data_ptr(), arr, self.stream()))
        return outs

    def group_add(self, group, a, b):
        pp = self.pp
        out = np.zeros_like(a)
        pp._check(pp.lib.zk_group_add(pp.h, group, a.ctypes.data, b.ctypes.data, out.ctypes.data))
        return out

    def assemble(self, crs_ct, r, s, sums):
        pp = self.pp
        nl = pp.fq.nl
        pa = np.zeros((pp.n, 3 * nl), dtype=np.uint64)
        pb = np.zeros((pp.n, 6 * nl), dtype=np.uint64)
        pc = np.zeros((pp.n, 3 * nl), dtype=np.uint64)
        rr, ss = pp.fr.encode_one(r), pp.fr.encode_one(s)
        arr = (C.c_void_p * 5)(*[x.ctypes.data for x in sums])
        pp._check(pp.lib.zk_groth16_assemble(pp.h, C.byref(crs_ct), rr.ctypes.data, ss.ctypes.data, arr, None,
                                             pa.ctypes.data, pb.ctypes.data, pc.ctypes.data))
        return pa, pb, pc

    def point_to_tensor(self, arr):
        return self.from_numpy(arr)

    def tensor_to_points(self, t):
        return t.cpu().numpy().view(np.uint64)

    def sync(self):
        self.torch.cuda.synchronize()


class StarNet:
    """gather-to-king / scatter-from-king over torch.distributed (rank 0 hosts the king).

    `via_cpu` stages every collective through host tensors (gloo): a debugging mode that lets two ranks share ONE
    GPU (RCCL refuses duplicate devices), used to exercise the complete multi-rank GPU code path on a 1-GPU box."""

    def __init__(self, dist, rank, world, via_cpu=False):
        self.dist, self.rank, self.world, self.via_cpu = dist, rank, world, via_cpu

    def gather(self, local, full):
        """local [k, ...] from every rank -> full [n, ...] on rank 0 (party-major = rank-major)."""
        if self.world == 1:
            full.copy_(local)
            return
        if self.via_cpu:
            loc = local.cpu()
            if self.rank == 0:
                parts = [loc.new_empty(loc.shape) for _ in range(self.world)]
                self.dist.gather(loc, gather_list=parts, dst=0)
                for dst, src in zip(full.chunk(self.world, dim=0), parts):
                    dst.copy_(src)
            else:
                self.dist.gather(loc, dst=0)
            return
        if self.rank == 0:
            parts = list(full.chunk(self.world, dim=0))
            self.dist.gather(local, gather_list=parts, dst=0)
        else:
            self.dist.gather(local, dst=0)

    def scatter(self, full, local):
        """full [n, ...] on rank 0 -> local [k, ...] on every rank."""
        if self.world == 1:
            local.copy_(full)
            return
        if self.via_cpu:
            loc = local.cpu()
            if self.rank == 0:
                self.dist.scatter(loc, scatter_list=[p.cpu().contiguous() for p in full.chunk(self.world, dim=0)], src=0)
            else:
                self.dist.scatter(loc, src=0)
            local.copy_(loc)
            return
        if self.rank == 0:
            parts = [p.contiguous() for p in full.chunk(self.world, dim=0)]
            self.dist.scatter(local, scatter_list=parts, src=0)
        else:
            self.dist.scatter(local, src=0)

    def all_gather(self, local):
        if self.world == 1:
            return [local]
        if self.via_cpu:
            loc = local.cpu()
            outs = [loc.new_empty(loc.shape) for _ in range(self.world)]
            self.dist.all_gather(outs, loc)
            return outs
        outs = [local.new_empty(local.shape) for _ in range(self.world)]
        self.dist.all_gather(outs, local)
        return outs


class DistProver:
    """dsha256 (groth16/examples/sha256.rs:32-129) with the parties sharded over ranks.

    Per rank: qap [3][k][Lc], a_share [k][len_a], ax_share [k][len_w], CRS share vectors [k][len] (all tensors of
    Montgomery limbs), k = n / world."""

    def __init__(self, backend, net, n, l, log_m, w2m):
        self.be, self.net, self.n, self.l, self.log_m, self.w2m = backend, net, n, l, log_m, w2m
        self.first, self.k = party_range(net.rank, net.world, n)
        self.Lc = (1 << log_m) // l
        be = backend
        self.W = be.empty(3, self.k, self.Lc, 4)
        self.h = be.empty(self.k, self.Lc, 4)
        if net.rank == 0:
            self.full_in = be.empty(n, self.Lc, 4)
            self.full_out = be.empty(n, self.Lc, 4)
        else:
            self.full_in = self.full_out = None

    def _round(self, x, inverse, g, scale, rearrange, seed):
        """fft1 on the local parties was done by the caller; king step for one vector: gather -> king -> scatter."""
        self.net.gather(x, self.full_in)
        if self.net.rank == 0:
            self.be.king_fft2(self.full_in, self.full_out, self.log_m, inverse, g, scale, rearrange, seed)
        self.net.scatter(self.full_out, x)

    def circom_h(self, qap, seed):
        """ext_wit.rs:104-181 (zero masks)."""
        be = self.be
        self.W.copy_(qap)
        be.fft1(self.W, self.log_m, True)                       # 3 x d_ifft: local stages for a, b, c at once
        for j in range(3):
            self._round(self.W[j], True, self.w2m, True, True, seed + j)
        be.fft1(self.W, self.log_m, False)                      # 3 x d_fft
        for j in range(3):
            self._round(self.W[j], False, None, False, False, seed + 3 + j)
        be.mul_sub(self.h, self.W[0], self.W[1], self.W[2])     # ext_wit.rs:173-177
        self.net.gather(self.h, self.full_in)                   # deg_red (ext_wit.rs:179)
        if self.net.rank == 0:
            be.deg_red(self.full_in, self.Lc, seed + 6)
        self.net.scatter(self.full_in, self.h)
        return self.h

    def prove(self, inp, r, s, seed):
        be = self.be
        first, k = self.first, self.k
        if hasattr(be, "msms_begin"):
            # the four MSMs over the witness shares overlap the king rounds of circom_h (prove.rs try_join!)
            be.msms_begin(inp, first, k, skip_h=(r == 0))
            h = self.circom_h(inp["qap"], seed)
            parts = be.msms_finish(h, first, k)
        else:
            h = self.circom_h(inp["qap"], seed)
            parts = [
                be.msm_local(ZK_G1, inp["s"], inp["a_share"], inp["len_a"], first, k),
                be.msm_local(ZK_G1, inp["h"], inp["a_share"], inp["len_a"], first, k) if r else None,
                be.msm_local(ZK_G2, inp["v"], inp["a_share"], inp["len_a"], first, k),
                be.msm_local(ZK_G1, inp["w"], inp["ax_share"], inp["len_w"], first, k),
                be.msm_local(ZK_G1, inp["u"], h, self.Lc, first, k),
            ]
            if parts[1] is None:
                parts[1] = np.zeros_like(parts[0])
        # d_msm's king: sum of the ranks' partial points, known to every rank afterwards (dmsm/mod.rs:85-92)
        flat = np.concatenate(parts)
        gathered = [be.tensor_to_points(t).reshape(-1) for t in self.net.all_gather(be.point_to_tensor(flat))]
        sizes = [len(p) for p in parts]
        sums = []
        off = 0
        for idx, sz in enumerate(sizes):
            grp = ZK_G2 if idx == 2 else ZK_G1
            acc = np.ascontiguousarray(gathered[0][off:off + sz])
            for g in gathered[1:]:
                acc = be.group_add(grp, acc, np.ascontiguousarray(g[off:off + sz]))
            sums.append(acc)
            off += sz
        return be.assemble(inp["crs_ct"], r, s, sums)


def _local_inputs(be, pp, crs, wit, rank, world):
    """Slice this rank's parties out of the full dealing (setup time, not timed)."""
    n, nl = pp.n, pp.fr.nl
    first, k = party_range(rank, world, n)
    Lc = (1 << wit.log_m) // pp.l
    sl = lambda buf, length, width: buf.to_numpy().reshape(n, length, width)[first:first + k]
    qap = np.stack([sl(q, Lc, nl) for q in wit.qap])
    return {
        "qap": be.from_numpy(qap), "a_share": be.from_numpy(sl(wit.a_share, wit.len_a, nl)),
        "ax_share": be.from_numpy(sl(wit.ax_share, wit.len_w, nl)), "s": be.from_numpy(sl(crs.s, crs.len_a, 2 * nl)),
        "h": be.from_numpy(sl(crs.h, crs.len_a, 2 * nl)), "v": be.from_numpy(sl(crs.v, crs.len_a, 4 * nl)),
        "w": be.from_numpy(sl(crs.w, crs.len_w, 2 * nl)), "u": be.from_numpy(sl(crs.u, crs.len_u, 2 * nl)),
        "len_a": crs.len_a, "len_w": crs.len_w, "crs_ct": crs.ct,
    }


def bench(args, rank, local_rank, world):
    """bench.py's N > 1 leg (also runnable with world == 1 for testing the sharded code path)."""
    import torch
    import torch.distributed as dist
    import zksaas_amd as zk
    from bench import build_inputs, read_profile, roofline_of

    via_cpu = bool(os.environ.get("ZK_DIST_VIA_CPU"))      # debugging: several ranks on one GPU, gloo collectives
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if via_cpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    pp = zk.PackedSharingParams("bn254", 2, device=local_rank)
    if pp.n % world:
        raise SystemExit("the number of GPUs must divide n = %d parties" % pp.n)
    r1, w, setup, crs, wit, r, s = build_inputs(pp, zk)
    be = GpuBackend(pp)
    net = StarNet(dist, rank, world, via_cpu=via_cpu)
    w2m = zg._root_of_unity("bn254", wit.log_m + 1)
    prover = DistProver(be, net, pp.n, pp.l, wit.log_m, w2m)
    inp = _local_inputs(be, pp, crs, wit, rank, world)
    table_windows = None
    if not getattr(args, "no_tables", False):
        # fixed-base tables for THIS rank's slices of the five query vectors (setup time, as a prover service would)
        from . import api
        k = pp.n // world
        for key, grp, ln in (("s", api.ZK_G1, crs.len_a), ("h", api.ZK_G1, crs.len_a), ("v", api.ZK_G2, crs.len_a),
                             ("w", api.ZK_G1, crs.len_w), ("u", api.ZK_G1, crs.len_u)):
            api.msm_precompute(pp, grp, inp[key], k * ln)
        table_windows = api.msm_table_info(pp, api.ZK_G1, inp["s"])["windows"]
    seed = 1000
    for _ in range(args.warmup):
        proof = prover.prove(inp, r, s, seed)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    pp._check(pp.lib.zk_profile_enable(pp.h, 1))        # HIP-event kernel slots of THIS rank (rank 0 reports its own)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        proof = prover.prove(inp, r, s, seed)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    tt = torch.tensor([dt], dtype=torch.float64, device="cpu" if via_cpu else be.device)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt = float(tt.item())
    prof = read_profile(pp)
    pp._check(pp.lib.zk_profile_enable(pp.h, 0))
    # untimed cross-check on rank 0 against the single-GPU prover on the full dealing
    ok = None
    if rank == 0:
        ref = zg.prove(pp, crs, wit, r, s, seed=seed)
        ok = bool(all(np.array_equal(_norm(pp, a[0], g2), _norm(pp, b[0], g2))
                      for a, b, g2 in ((proof[0], ref[0], False), (proof[1], ref[1], True), (proof[2], ref[2], False))))
    if world > 1:
        dist.barrier()
    proofs_per_s = args.steps / dt
    return {
        "metric": "Groth16 proofs/sec (SHA-256 circuit)", "value": round(proofs_per_s, 3), "unit": "proofs/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u32 limbs (256-bit Montgomery)",
        "data": "synthetic: SHA-256(a=1,b=2) circuit rebuilt from its semantics, seeded trapdoor CRS, seeded shares",
        "config": {"workload": "BASELINE configs[3]: full distributed Groth16 on the SHA-256 circuit, BN254, l=2, "
                               "n=8 parties sharded over %d GPUs (king on GPU 0), zero masks" % world,
                   "constraints": r1.num_constraints, "wires": r1.num_variables, "domain": 1 << wit.log_m,
                   "parties": pp.n, "parties_per_gpu": pp.n // world, "packing_factor": pp.l,
                   "fixed_base_tables": table_windows is not None},
        "constraints_per_sec": round(proofs_per_s * r1.num_constraints, 1),
        "proof_matches_single_gpu": ok,
        "roofline": roofline_of(prof, ntt_passes=2, masks_on=False, pp=pp, table_windows=table_windows),   # rank 0's dominant kernel
        "kernels": [{**e, "total_ms": round(e["total_ms"], 3)} for e in prof if e["launches"]],
    }


def _norm(pp, jac, g2):
    """canonical affine ints of a Jacobian point (host-side, for the untimed cross-check only)."""
    q = pp.fq.p
    v = pp.fq.decode(np.asarray(jac).reshape(-1, pp.fq.nl))
    if g2:
        z = (v[4], v[5])
        if z == (0, 0):
            return np.zeros(1)
        def mul(a, b):
            return ((a[0] * b[0] - a[1] * b[1]) % q, (a[0] * b[1] + a[1] * b[0]) % q)
        nrm = pow((z[0] * z[0] + z[1] * z[1]) % q, q - 2, q)
        zi = (z[0] * nrm % q, (-z[1]) * nrm % q)
        zi2 = mul(zi, zi)
        zi3 = mul(zi2, zi)
        x, y = mul((v[0], v[1]), zi2), mul((v[2], v[3]), zi3)
        return np.array([x[0], x[1], y[0], y[1]], dtype=object)
    if v[2] == 0:
        return np.zeros(1)
    zi = pow(v[2], q - 2, q)
    return np.array([v[0] * zi * zi % q, v[1] * zi * zi * zi % q], dtype=object)
