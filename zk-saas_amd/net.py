"""Host-side mirror of the reference's `MpcNet` star (mpc-net/src/lib.rs:43-53, 89-176; ser_net.rs) over the C ABI
(include/zksaas.h "the star network"): one process per GPU, rank rho drives k = n / world parties -- the block
[rho*k, (rho+1)*k) or, with `party_to_rank`, the parties mapped to it (`net.parties`, ascending); king = rank 0.

    net = StarNet(pp, rank, world, net_id, transport="rccl")     # collective: every rank constructs it
    dist_d_fft(pp, net, sid, shares_local, FftMask.zero(), rearrange, log_m)   # == d_fft(.., &net, sid)

`net_id` is made once by `StarNet.unique_id()` on rank 0 and handed to the other ranks by the launcher (an
environment variable, a file, torch.distributed's store -- setup, not data path).
"""
import ctypes as C

import numpy as np

from ._lib import ZkError, load
from .api import ZK_G2, DeviceBuffer, _ptr

TRANSPORTS = {"local": 0, "rccl": 1, "shm": 2}
ID_BYTES = 512


class StarNet:
    def __init__(self, pp, rank, world, net_id=None, transport="rccl", n_parties=None, shm_bytes=0, timeout_ms=None,
                 party_to_rank=None):
        """pp = None: host-memory mode (the raw verbs move host buffers; used to test the protocol flow without a GPU).
        party_to_rank: rank of every party (any map giving each rank n / world parties), None = contiguous blocks."""
        self.lib = load()
        self.pp = pp
        self.h = C.c_void_p()
        idbuf = None if net_id is None else (C.c_ubyte * ID_BYTES).from_buffer_copy(bytes(net_id))
        pmap = None if party_to_rank is None else (C.c_int * len(party_to_rank))(*[int(v) for v in party_to_rank])
        rc = self.lib.zk_net_create(None if pp is None else pp.h, TRANSPORTS[transport], rank, world,
                                    0 if n_parties is None else n_parties, pmap, idbuf, shm_bytes, C.byref(self.h))
        if rc != 0:
            msg = self.lib.zk_net_last_error(self.h, None).decode() if self.h else "zk_net_create failed"
            if self.h:
                self.lib.zk_net_destroy(self.h)
                self.h = None
            raise ZkError(rc, msg)
        info = (C.c_int * 4)()
        self.lib.zk_net_info(self.h, info)
        self.rank, self.world, self.first, self.k = info[0], info[1], info[2], info[3]
        ids = (C.c_int * self.k)()
        self.lib.zk_net_parties(self.h, self.rank, ids)
        self.parties = list(ids)              # this rank's rows, in this order
        if timeout_ms is not None:
            self.lib.zk_net_set_timeout_ms(self.h, int(timeout_ms))

    @staticmethod
    def unique_id():
        buf = (C.c_ubyte * ID_BYTES)()
        rc = load().zk_net_unique_id(buf)
        if rc != 0:
            raise ZkError(rc, "zk_net_unique_id failed")
        return bytes(buf)

    def stats(self):
        """dict(gathers, scatters, alltoalls, bytes_sent) since creation"""
        st = (C.c_uint64 * 4)()
        self.lib.zk_net_stats(self.h, st)
        return {"gathers": st[0], "scatters": st[1], "alltoalls": st[2], "bytes_sent": st[3]}

    def close(self):
        if self.h:
            self.lib.zk_net_destroy(self.h)
            self.h = None

    def _check(self, rc):
        if rc != 0:
            party = C.c_int(-1)
            msg = self.lib.zk_net_last_error(self.h, C.byref(party)).decode()
            raise ZkError(rc, msg, party.value)

    # ---- raw verbs (host-mode buffers are numpy arrays, device-mode buffers DeviceBuffer / tensors)
    def enter(self, sid):
        m = C.c_uint32(0)
        self._check(self.lib.zk_net_enter(self.h, sid, C.byref(m)))
        return m.value

    @staticmethod
    def _p(x):
        if x is None:
            return None
        if isinstance(x, np.ndarray):
            return x.ctypes.data
        return _ptr(x)

    def gather(self, sid, mask, local, bytes_per_rank, full):
        self._check(self.lib.zk_net_gather(self.h, sid, mask, self._p(local), bytes_per_rank, self._p(full)))

    def scatter(self, sid, mask, full, bytes_per_rank, local):
        self._check(self.lib.zk_net_scatter(self.h, sid, mask, self._p(full), bytes_per_rank, self._p(local)))

    def alltoall(self, sid, mask, send, bytes_per_peer, recv):
        """block for rank r read at send + r * bytes_per_peer; block from the i-th present rank written at recv + i * ..."""
        self._check(self.lib.zk_net_alltoall(self.h, sid, mask, self._p(send), bytes_per_peer, self._p(recv)))

    def gather_host(self, sid, mask, mine, all_out):
        self._check(self.lib.zk_net_gather_host(self.h, sid, mask, mine.ctypes.data, mine.nbytes,
                                                None if all_out is None else all_out.ctypes.data))

    def bcast_host(self, sid, mask, buf):
        self._check(self.lib.zk_net_bcast_host(self.h, sid, mask, buf.ctypes.data, buf.nbytes))

    def sync(self, sid):
        self._check(self.lib.zk_net_sync(self.h, sid))


def _mask_ptrs(m):
    return (None, None) if m is None else (_ptr(m.in_mask), _ptr(m.out_mask))


def dist_d_fft(pp, net, sid, shares_local, fft_mask, rearrange, log2_m, seed=0, stream=None):
    """dfft/mod.rs:99-134 for this rank's k parties: shares_local [k][m/l] in place."""
    im, om = _mask_ptrs(fft_mask)
    pp._check(pp.lib.zk_dist_d_fft(pp.h, net.h, sid, _ptr(shares_local), im, om, int(rearrange), log2_m, seed, stream))
    return shares_local


def dist_d_ifft(pp, net, sid, shares_local, fft_mask, rearrange, log2_m, g=None, seed=0, stream=None):
    im, om = _mask_ptrs(fft_mask)
    garr = None if g is None else pp.fr.encode_one(g)
    pp._check(pp.lib.zk_dist_d_ifft(pp.h, net.h, sid, _ptr(shares_local), im, om, int(rearrange), log2_m,
                                    None if garr is None else garr.ctypes.data, seed, stream))
    return shares_local


def dist_deg_red(pp, net, sid, x_local, mask, length, seed=0, stream=None):
    im, om = _mask_ptrs(mask)
    pp._check(pp.lib.zk_dist_deg_red(pp.h, net.h, sid, _ptr(x_local), im, om, length, seed, stream))
    return x_local


def dist_d_pp(pp, net, sid, num_local, den_local, mask, length, seed=0, out=None, stream=None):
    out = out or pp.alloc_fr(net.k * length)
    im, om = _mask_ptrs(mask)
    pp._check(pp.lib.zk_dist_d_pp(pp.h, net.h, sid, _ptr(num_local), _ptr(den_local), im, om, length, seed, _ptr(out),
                                  stream))
    return out


def dist_d_msm(pp, net, sid, group, bases_local, scalars_local, length, msm_mask=None, stream=None):
    """dmsm/mod.rs:59-102 for this rank's k parties; returns their k output shares [k][3 * coord limbs]."""
    nl = pp.fq.nl * (2 if group == ZK_G2 else 1)
    out = np.zeros((net.k, 3 * nl), dtype=np.uint64)
    im = om = None
    if msm_mask is not None and msm_mask.in_mask is not None:
        im = np.ascontiguousarray(msm_mask.in_mask, dtype=np.uint64)
    if msm_mask is not None and msm_mask.out_mask is not None:
        om = np.ascontiguousarray(msm_mask.out_mask, dtype=np.uint64)
    pp._check(pp.lib.zk_dist_d_msm(pp.h, net.h, sid, group, _ptr(bases_local), _ptr(scalars_local), length,
                                   None if im is None else im.ctypes.data, None if om is None else om.ctypes.data,
                                   out.ctypes.data, stream))
    return out


def dist_circom_h(pp, net, qap_local, log2_m, masks=None, seed=0, out=None, stream=None):
    out = out or pp.alloc_fr(net.k * ((1 << log2_m) // pp.l))
    pp._check(pp.lib.zk_dist_circom_h(pp.h, net.h, _ptr(qap_local[0]), _ptr(qap_local[1]), _ptr(qap_local[2]), log2_m,
                                      None if masks is None else C.byref(masks), seed, out.ptr, stream))
    return out


def dist_deg_red_points(pp, net, sid, group, x_local, in_mask_local, out_mask_local, length, gen_affine, seed=0, out=None,
                        stream=None):
    """zk_dist_deg_red_points (deg_red.rs:80-126 with T = G): this rank's rows [k][length] of affine points."""
    width = (4 if group == 2 else 2) * pp.fq.nbytes
    out = out or DeviceBuffer(pp, net.k * length * width)
    pp._check(pp.lib.zk_dist_deg_red_points(pp.h, net.h, sid, group, _ptr(x_local), _ptr(in_mask_local),
                                            _ptr(out_mask_local), length, gen_affine.ctypes.data, seed, out.ptr, stream))
    return out


def dist_libsnark_h(pp, net, qap_local, log2_m, fft_in=None, fft_out=None, seed=0, out=None, stream=None):
    """zk_dist_libsnark_h (ext_wit.rs:14-102): fft_in / fft_out = lists of seven local mask buffers (or None)."""
    out = out or pp.alloc_fr(net.k * ((1 << log2_m) // pp.l))
    mi = None if fft_in is None else (C.c_void_p * 7)(*[_ptr(m) for m in fft_in])
    mo = None if fft_out is None else (C.c_void_p * 7)(*[_ptr(m) for m in fft_out])
    pp._check(pp.lib.zk_dist_libsnark_h(pp.h, net.h, _ptr(qap_local[0]), _ptr(qap_local[1]), _ptr(qap_local[2]), log2_m,
                                        mi, mo, seed, out.ptr, stream))
    return out


def dist_prove(pp, net, crs_ct, qap_local, a_share_local, ax_share_local, r, s, log2_m, masks=None, seed=0, stream=None):
    """dsha256 (sha256.rs:32-129) for this rank's k parties; crs_ct: groth16.CrsShare over the LOCAL [k][len] vectors.
    Returns (pi_a [k][3nl], pi_b [k][6nl], pi_c [k][3nl])."""
    nl = pp.fq.nl
    pa = np.zeros((net.k, 3 * nl), dtype=np.uint64)
    pb = np.zeros((net.k, 6 * nl), dtype=np.uint64)
    pc = np.zeros((net.k, 3 * nl), dtype=np.uint64)
    rr, ss = pp.fr.encode_one(r), pp.fr.encode_one(s)
    pp._check(pp.lib.zk_dist_groth16_prove(pp.h, net.h, C.byref(crs_ct), _ptr(qap_local[0]), _ptr(qap_local[1]),
                                           _ptr(qap_local[2]), _ptr(a_share_local), _ptr(ax_share_local),
                                           rr.ctypes.data, ss.ctypes.data, log2_m,
                                           None if masks is None else C.byref(masks), seed, pa.ctypes.data,
                                           pb.ctypes.data, pc.ctypes.data, stream))
    return pa, pb, pc


class DistProofInFlight:
    """Handle of zk_dist_groth16_prove_async; wait() is collective like the call that made it."""

    def __init__(self, pp, net, handle, keep):
        self.pp, self.net, self.handle, self._keep = pp, net, handle, keep

    def wait(self):
        pp, nl, k = self.pp, self.pp.fq.nl, self.net.k
        pa = np.zeros((k, 3 * nl), dtype=np.uint64)
        pb = np.zeros((k, 6 * nl), dtype=np.uint64)
        pc = np.zeros((k, 3 * nl), dtype=np.uint64)
        pp._check(pp.lib.zk_dist_groth16_wait(pp.h, self.net.h, self.handle, pa.ctypes.data, pb.ctypes.data, pc.ctypes.data))
        self._keep = None
        return pa, pb, pc


def dist_prove_async(pp, net, crs_ct, qap_local, a_share_local, ax_share_local, r, s, log2_m, masks=None, seed=0,
                     stream=None):
    """First half of dist_prove: the proof is admitted and its device work enqueued; up to two may be in flight per rank,
    issued and waited for in the same order on every rank."""
    rr, ss = pp.fr.encode_one(r), pp.fr.encode_one(s)
    h = C.c_int(-1)
    pp._check(pp.lib.zk_dist_groth16_prove_async(pp.h, net.h, C.byref(crs_ct), _ptr(qap_local[0]), _ptr(qap_local[1]),
                                                 _ptr(qap_local[2]), _ptr(a_share_local), _ptr(ax_share_local),
                                                 rr.ctypes.data, ss.ctypes.data, log2_m,
                                                 None if masks is None else C.byref(masks), seed, stream, C.byref(h)))
    return DistProofInFlight(pp, net, h.value, (crs_ct, qap_local, a_share_local, ax_share_local, masks))


def dist_prove_batch(pp, net, crs_ct, qaps_local, a_shares_local, ax_shares_local, rs, ss, log2_m, masks=None, seed=0,
                     stream=None):
    """zk_dist_groth16_prove_batch: len(rs) proofs per collective call for this rank's k parties.  qaps_local: per proof
    the three local QAP buffers; a_shares_local / ax_shares_local: per proof the local witness shares; masks: per proof a
    groth16.Masks over the LOCAL rows (or None).  Returns a list of (pi_a [k][3nl], pi_b [k][6nl], pi_c [k][3nl])."""
    from .groth16 import Masks
    nb, nl = len(rs), pp.fq.nl
    pa = np.zeros((nb, net.k, 3 * nl), dtype=np.uint64)
    pb = np.zeros((nb, net.k, 6 * nl), dtype=np.uint64)
    pc = np.zeros((nb, net.k, 3 * nl), dtype=np.uint64)
    rr = np.ascontiguousarray(np.stack([pp.fr.encode_one(v) for v in rs]))
    sv = np.ascontiguousarray(np.stack([pp.fr.encode_one(v) for v in ss]))
    arr = lambda vals: (C.c_void_p * nb)(*[_ptr(v) for v in vals])
    mk = None
    if masks is not None:
        mk = (Masks * nb)()
        for b, m in enumerate(masks):
            C.memmove(C.byref(mk, b * C.sizeof(Masks)), C.byref(m), C.sizeof(Masks))
    pp._check(pp.lib.zk_dist_groth16_prove_batch(
        pp.h, net.h, C.byref(crs_ct), nb, arr([q[0] for q in qaps_local]), arr([q[1] for q in qaps_local]),
        arr([q[2] for q in qaps_local]), arr(a_shares_local), arr(ax_shares_local), rr.ctypes.data, sv.ctypes.data, log2_m,
        None if mk is None else C.cast(mk, C.c_void_p), seed, pa.ctypes.data, pb.ctypes.data, pc.ctypes.data, stream))
    return [(pa[b], pb[b], pc[b]) for b in range(nb)]
