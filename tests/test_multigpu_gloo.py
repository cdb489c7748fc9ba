"""The N > 1 protocol flow (zk-saas_amd/multigpu.py: party sharding, gather-to-king / scatter / all-gather) on CPU:
two processes, gloo backend, with the oracle standing in for the device compute (test infrastructure only --
the product backend is GpuBackend).  Checks that the sharded prover reproduces the oracle's proof."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleBackend:
    """Same interface as multigpu.GpuBackend, computed with the Python oracle on CPU tensors."""

    def __init__(self, n, l):
        import torch
        from oracle.curve import g1, g2, GroupOps
        from oracle.params import BN254
        from oracle.pss import PackedSharingParams
        import zksaas_amd as zk
        self.torch = torch
        self.o = PackedSharingParams(BN254, l)
        self.fr = zk.fields.MontCodec(BN254.r)
        self.fq = zk.fields.MontCodec(BN254.q)
        self.G1, self.G2 = g1(BN254), g2(BN254)
        self.o1, self.o2 = GroupOps(self.G1), GroupOps(self.G2)
        self.curve = BN254
        self.coef = []
        for p in range(self.o.n):
            e = [0] * self.o.n
            e[p] = 1
            self.coef.append(sum(self.o.unpack2(e)) % BN254.r)

    # tensors <-> ints
    def _ints(self, t):
        return self.fr.decode(t.numpy().view(np.uint64).reshape(-1, 4))

    def _put(self, t, vals):
        t.copy_(self.torch.from_numpy(self.fr.encode(vals).view(np.int64)).reshape(t.shape))

    def empty(self, *shape):
        return self.torch.zeros(*shape, dtype=self.torch.int64)

    def from_numpy(self, arr):
        return self.torch.from_numpy(np.ascontiguousarray(arr).view(np.int64).copy())

    def fft1(self, t, log_m, inverse):
        from oracle.dist import fft1_in_place
        from oracle.field import Domain
        dom = Domain(self.curve, 1 << log_m)
        Lc = (1 << log_m) // self.o.l
        flat = self._ints(t)
        out = []
        for i in range(0, len(flat), Lc):
            out += fft1_in_place(flat[i:i + Lc], self.o, dom.group_gen_inv if inverse else dom.group_gen)
        self._put(t, out)

    def king_fft2(self, tin, tout, log_m, inverse, g, scale, rearrange, seed):
        from oracle.dist import king_fft2
        from oracle.field import Domain
        dom = Domain(self.curve, 1 << log_m)
        Lc = (1 << log_m) // self.o.l
        flat = self._ints(tin)
        if scale:
            flat = [x * dom.size_inv % self.curve.r for x in flat]       # no masks: scaling commutes with fft1
        shares = [flat[i * Lc:(i + 1) * Lc] for i in range(self.o.n)]
        out = king_fft2(shares, list(range(self.o.n)), rearrange, 1 if g is None else g, self.o,
                        dom.group_gen_inv if inverse else dom.group_gen, seed)
        self._put(tout, [v for s in out for v in s])

    def mul_sub(self, out, a, b, c):
        p = self.curve.r
        self._put(out, [(x * y - z) % p for x, y, z in zip(self._ints(a), self._ints(b), self._ints(c))])

    def deg_red(self, x, length, seed):
        from oracle.dist import king_deg_red
        flat = self._ints(x)
        shares = [flat[i * length:(i + 1) * length] for i in range(self.o.n)]
        out = king_deg_red(shares, list(range(self.o.n)), self.o, seed)
        self._put(x, [v for s in out for v in s])

    def _dec_aff(self, row, g2):
        v = self.fq.decode(np.asarray(row).view(np.uint64).reshape(-1, 4))
        if g2:
            return None if not any(v) else ((v[0], v[1]), (v[2], v[3]))
        return None if not any(v) else (v[0], v[1])

    def _enc_jac(self, P, g2):
        coords = [P[0][0], P[0][1], P[1][0], P[1][1], P[2][0], P[2][1]] if g2 else list(P)
        return self.fq.encode(coords).reshape(-1)

    def _dec_jac(self, arr, g2):
        v = self.fq.decode(np.asarray(arr).reshape(-1, 4))
        return ((v[0], v[1]), (v[2], v[3]), (v[4], v[5])) if g2 else (v[0], v[1], v[2])

    def msm_local(self, group, bases, scalars, length, first, count):
        g2 = group == 2
        G = self.G2 if g2 else self.G1
        b = bases.numpy().view(np.uint64).reshape(count, length, -1)
        sc = self._ints(scalars)
        acc = G.identity
        for p in range(count):
            pts = [self._dec_aff(b[p, i], g2) for i in range(length)]
            acc = G.add(acc, G.mul(G.msm(pts, sc[p * length:(p + 1) * length]), self.coef[first + p]))
        return self._enc_jac(acc, g2)

    def group_add(self, group, a, b):
        g2 = group == 2
        G = self.G2 if g2 else self.G1
        return self._enc_jac(G.add(self._dec_jac(a, g2), self._dec_jac(b, g2)), g2)

    def assemble(self, crs, r, s, sums):
        G1, G2, q = self.G1, self.G2, self.curve.r
        S, H, V, W, U = [self._dec_jac(x, i == 2) for i, x in enumerate(sums)]
        fa, fb = G1.from_affine, G2.from_affine
        A = G1.sum([fa(crs["a_query0"]), G1.mul(fa(crs["delta_g1"]), r), S, fa(crs["alpha_g1"])])
        B1 = G1.identity if r % q == 0 else G1.sum([fa(crs["b_g1_query0"]), G1.mul(fa(crs["delta_g1"]), s), H,
                                                     fa(crs["beta_g1"])])
        B2 = G2.sum([fb(crs["b_g2_query0"]), G2.mul(fb(crs["delta_g2"]), s), V, fb(crs["beta_g2"])])
        Cc = G1.sum([G1.mul(A, s), G1.mul(B1, r), G1.neg(G1.mul(fa(crs["delta_g1"]), r * s % q)), W, U])
        n = self.o.n
        return ([self._enc_jac(A, False)] * n, [self._enc_jac(B2, True)] * n, [self._enc_jac(Cc, False)] * n)

    def point_to_tensor(self, arr):
        return self.from_numpy(arr)

    def tensor_to_points(self, t):
        return t.numpy().view(np.uint64)

    def sync(self):
        pass


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import zksaas_amd as zk
        from zksaas_amd.multigpu import DistProver, StarNet, party_range
        from oracle import dist as od
        from oracle import groth16 as og
        from oracle.curve import GroupOps
        from oracle.field import Domain
        from oracle.params import BN254
        from oracle.prng import rand_fp
        from test_oracle_groth16 import small_r1cs

        P = BN254.r
        be = OracleBackend(8, 2)
        o = be.o
        r1, w = small_r1cs()
        key = og.setup_scalars(BN254, r1, og.Trapdoor.from_seed(42, P))
        pk = og.proving_key_points(key, be.G1, be.G2)
        crs = og.pack_proving_key(pk, o, be.G1, be.G2, be.o1, be.o2)
        qp = og.qap(BN254, r1, w)
        dom, m = qp.domain, qp.domain.size
        qs = qp.pss(o, 5)
        a_sh = og.pack_from_witness(o, w[1:], 12)
        ax = og.pack_from_witness(o, w[r1.num_instance_variables:], 11)
        r, s = rand_fp(43, 0, P), rand_fp(43, 1, P)
        first, k = party_range(rank, world, o.n)
        enc_fr = lambda rows: be.from_numpy(np.stack([be.fr.encode(v) for v in rows]))

        def enc_pts(rows, g2):
            out = []
            for v in rows:
                flat = []
                for p in v:
                    if g2:
                        flat += [0, 0, 0, 0] if p is None else [p[0][0], p[0][1], p[1][0], p[1][1]]
                    else:
                        flat += [0, 0] if p is None else list(p)
                out.append(be.fq.encode(flat).reshape(len(v), -1))
            return be.from_numpy(np.stack(out))

        mine = range(first, first + k)
        inp = {
            "qap": be.from_numpy(np.stack([np.stack([be.fr.encode(qs[i][j]) for i in mine]) for j in range(3)])),
            "a_share": enc_fr([a_sh[i] for i in mine]), "ax_share": enc_fr([ax[i] for i in mine]),
            "s": enc_pts([crs[i].s for i in mine], False), "h": enc_pts([crs[i].h for i in mine], False),
            "v": enc_pts([crs[i].v for i in mine], True), "w": enc_pts([crs[i].w for i in mine], False),
            "u": enc_pts([crs[i].u for i in mine], False), "len_a": len(a_sh[0]), "len_w": len(ax[0]),
            "crs_ct": {kk: getattr(crs[0], kk) for kk in ("a_query0", "b_g1_query0", "delta_g1", "alpha_g1", "beta_g1",
                                                          "b_g2_query0", "delta_g2", "beta_g2")},
        }
        net = StarNet(dist, rank, world)
        prover = DistProver(be, net, o.n, o.l, dom.log_size, Domain(BN254, 2 * m).element(1))
        pa, pb, pc = prover.prove(inp, r, s, seed=9)
        A, B, Cc = og.create_proof_local(BN254, r1, pk, be.G1, be.G2, w, r, s)
        ok = (be.G1.eq(be._dec_jac(pa[0], False), A) and be.G2.eq(be._dec_jac(pb[0], True), B)
              and be.G1.eq(be._dec_jac(pc[0], False), Cc))
        # the sharded circom_h also matches the all-in-one oracle run share for share
        hs = og.circom_h(qs, [[od.FftMask.zero(m // 2)] * o.n] * 6, [od.DegRedMask.zero(m // 2)] * o.n, o, dom, seed=9)
        got_h = be._ints(prover.h)
        ok = ok and got_h == [v for i in mine for v in hs[i]]
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2])
def test_sharded_prover_two_ranks_gloo(world):
    import torch.multiprocessing as mp
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(results) == [(r, True) for r in range(world)]


def test_party_ranges_cover_all_parties():
    sys.path.insert(0, ROOT)
    from zksaas_amd.multigpu import party_range
    for world in (1, 2, 4, 8):
        seen = []
        for rank in range(world):
            first, k = party_range(rank, world, 8)
            seen += list(range(first, first + k))
        assert seen == list(range(8))
        assert party_range(0, world, 8)[0] == 0          # the king (party 0) lives on rank 0
