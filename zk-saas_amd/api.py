"""Host-side mirror of the reference interface (same names and argument meaning) over the C ABI.

reference                                           here
---------                                           ----
secret_sharing::pss::PackedSharingParams::new(l)     PackedSharingParams(curve, l)  (owns a zk_ctx)
  .pack / .det_pack / .unpack / .unpack2              same names, batched over chunks
  .lagrange_unpack(shares, parties)                   same
dist_primitives::dfft::{d_fft, d_ifft, FftMask}       d_fft, d_ifft, FftMask.sample / FftMask.zero
dist_primitives::dmsm::{d_msm, MsmMask}               d_msm, MsmMask.zero
dist_primitives::utils::deg_red::{deg_red, DegRedMask} deg_red, DegRedMask.sample / .zero
dist_primitives::dpp::d_pp                            d_pp

The reference runs n parties as tasks that meet at a king; here the n parties' share vectors live in ONE
device buffer [n][m/l] and each call performs the whole round (clients + king) on the GPU.
Errors raise ZkError (MpcNetError mirror).
"""
import ctypes as C

import numpy as np

from . import fields
from ._lib import ZkError, load

ZK_G1, ZK_G2 = 1, 2


class DeviceBuffer:
    """A hipMalloc'ed buffer owned through the C ABI (zk_malloc / zk_free)."""

    def __init__(self, ctx, nbytes):
        self.ctx = ctx
        self.nbytes = int(nbytes)
        p = C.c_void_p()
        ctx._check(ctx.lib.zk_malloc(ctx.h, self.nbytes, C.byref(p)))
        self.ptr = p.value

    @classmethod
    def from_numpy(cls, ctx, arr):
        arr = np.ascontiguousarray(arr)
        b = cls(ctx, arr.nbytes)
        if arr.nbytes:
            ctx._check(ctx.lib.zk_memcpy_h2d(ctx.h, b.ptr, arr.ctypes.data, arr.nbytes, None))
        return b

    def to_numpy(self, dtype=np.uint64, shape=None):
        out = np.empty(self.nbytes // np.dtype(dtype).itemsize, dtype=dtype)
        if self.nbytes:
            self.ctx._check(self.ctx.lib.zk_memcpy_d2h(self.ctx.h, out.ctypes.data, self.ptr, self.nbytes, None))
        return out.reshape(shape) if shape is not None else out

    def view(self, offset_bytes, nbytes=None):
        """A non-owning window into this buffer (keeps the parent alive)."""
        v = DeviceBuffer.__new__(DeviceBuffer)
        v.ctx, v.parent, v.owner = self.ctx, self, False
        v.nbytes = self.nbytes - offset_bytes if nbytes is None else int(nbytes)
        if offset_bytes < 0 or v.nbytes < 0 or offset_bytes + v.nbytes > self.nbytes:
            raise ValueError("view outside the buffer")
        v.ptr = self.ptr + offset_bytes
        return v

    def free(self):
        if not getattr(self, "owner", True):
            self.ptr = None
            return
        if self.ptr:
            self.ctx.lib.zk_free(self.ctx.h, self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def _ptr(x):
    if x is None:
        return None
    if isinstance(x, DeviceBuffer):
        return x.ptr
    if hasattr(x, "data_ptr"):       # torch tensor on the GPU
        return x.data_ptr()
    return int(x)


# Options every context created by THIS process gets right after zk_ctx_create (zk_ctx_set_option): the test suite puts
# {"rng_replay": 1} here (tests/conftest.py; shares are compared bit for bit with the oracle).  Nothing reads the
# environment: a deployment cannot end up on the replayable share randomness by inheriting a variable.
DEFAULT_OPTIONS = {}


class Context:
    """zk_ctx: PackedSharingParams + cached tables on one device."""

    def __init__(self, curve="bn254", l=2, device=0):
        self.lib = load()
        self.curve = curve
        self.h = C.c_void_p()
        rc = self.lib.zk_ctx_create(fields.CURVE_IDS[curve], l, device, C.byref(self.h))
        if rc != 0:
            msg = self.lib.zk_last_error(self.h, None).decode() if self.h else "zk_ctx_create failed (no GPU?)"
            raise ZkError(rc, msg)
        self.l = self.lib.zk_ctx_l(self.h)
        self.n = self.lib.zk_ctx_n(self.h)
        self.t = self.l
        self.fr = fields.MontCodec(fields.FR[curve])
        self.fq = fields.MontCodec(fields.FQ[curve])
        for name, value in DEFAULT_OPTIONS.items():
            self.set_option(name, value)

    def _check(self, rc):
        if rc != 0:
            party = C.c_int(-1)
            msg = self.lib.zk_last_error(self.h, C.byref(party)).decode()
            raise ZkError(rc, msg, party.value)

    def close(self):
        if self.h:
            self.lib.zk_ctx_destroy(self.h)
            self.h = None

    def set_option(self, name, value):
        """zk_ctx_set_option: context tunables ("msm_bigsort_min")."""
        self._check(self.lib.zk_ctx_set_option(self.h, name.encode(), int(value)))

    # --- marshalling helpers ---------------------------------------------------------------------
    def upload_fr(self, vals):
        return DeviceBuffer.from_numpy(self, self.fr.encode(vals))

    def upload_u32(self, arr):
        arr = np.ascontiguousarray(arr, dtype=np.uint32)
        if arr.size == 0:
            arr = np.zeros(1, dtype=np.uint32)
        return DeviceBuffer.from_numpy(self, arr)

    def download_fr(self, buf, count=None):
        arr = buf.to_numpy()
        if count is not None:
            arr = arr[: count * self.fr.nl]
        return self.fr.decode(arr)

    def alloc_fr(self, count):
        return DeviceBuffer(self, count * self.fr.nbytes)

    def sync(self, stream=None):
        self._check(self.lib.zk_stream_sync(self.h, stream))


class PackedSharingParams(Context):
    """secret-sharing/src/pss.rs:19-66."""

    def pack(self, secrets_d, nchunks, seed, order=0, out=None, stream=None):
        out = out or self.alloc_fr(self.n * nchunks)
        self._check(self.lib.zk_pss_pack(self.h, _ptr(secrets_d), nchunks, order, seed, _ptr(out), stream))
        return out

    def det_pack(self, secrets_d, nchunks, order=0, out=None, stream=None):
        out = out or self.alloc_fr(self.n * nchunks)
        self._check(self.lib.zk_pss_det_pack(self.h, _ptr(secrets_d), nchunks, order, _ptr(out), stream))
        return out

    def unpack(self, shares_d, nchunks, out=None, stream=None):
        out = out or self.alloc_fr(self.l * nchunks)
        self._check(self.lib.zk_pss_unpack(self.h, _ptr(shares_d), nchunks, _ptr(out), stream))
        return out

    def unpack2(self, shares_d, nchunks, parties=None, out=None, stream=None):
        out = out or self.alloc_fr(self.l * nchunks)
        if parties is None:
            arr, np_ = None, self.n
        else:
            arr = (C.c_uint32 * len(parties))(*parties)
            np_ = len(parties)
        self._check(self.lib.zk_pss_unpack2(self.h, _ptr(shares_d), arr, np_, nchunks, _ptr(out), stream))
        return out

    def lagrange_unpack(self, shares_d, nchunks, parties, out=None, stream=None):
        return self.unpack2(shares_d, nchunks, parties, out, stream)

    unpack_missing_shares = unpack2


class FftMask:
    """dist-primitives/src/dfft/mod.rs:16-95; buffers are [n][m/l] for all parties."""

    def __init__(self, in_mask, out_mask):
        self.in_mask, self.out_mask = in_mask, out_mask

    @staticmethod
    def sample(pp, rearrange, g, inverse, log2_m, seed, stream=None):
        cnt = pp.n * ((1 << log2_m) // pp.l)
        im, om = pp.alloc_fr(cnt), pp.alloc_fr(cnt)
        gp = None if g is None else pp.fr.encode_one(g).ctypes.data
        garr = None if g is None else pp.fr.encode_one(g)
        pp._check(pp.lib.zk_fft_mask_sample(pp.h, int(rearrange), None if garr is None else garr.ctypes.data,
                                            int(inverse), log2_m, seed, im.ptr, om.ptr, stream))
        del gp
        return FftMask(im, om)

    @staticmethod
    def zero():
        return FftMask(None, None)


class DegRedMask:
    """dist-primitives/src/utils/deg_red.rs:14-77 over Fr."""

    def __init__(self, in_mask, out_mask):
        self.in_mask, self.out_mask = in_mask, out_mask

    @staticmethod
    def sample(pp, num, seed, stream=None):
        im, om = pp.alloc_fr(pp.n * num), pp.alloc_fr(pp.n * num)
        pp._check(pp.lib.zk_degred_mask_sample(pp.h, num, seed, im.ptr, om.ptr, stream))
        return DegRedMask(im, om)

    @staticmethod
    def zero():
        return DegRedMask(None, None)


class MsmMask:
    """dist-primitives/src/dmsm/mod.rs:10-57: n Jacobian points each (host numpy arrays) or None."""

    def __init__(self, in_mask=None, out_mask=None):
        self.in_mask, self.out_mask = in_mask, out_mask

    @staticmethod
    def zero():
        return MsmMask(None, None)

    @staticmethod
    def sample(pp, group, gen_affine, seed):
        """MsmMask::sample (dmsm/mod.rs:21-47); gen_affine: the group generator as Montgomery limbs (uint64 array)."""
        nl = pp.fq.nl * (2 if group == ZK_G2 else 1)
        gen = np.ascontiguousarray(gen_affine, dtype=np.uint64).reshape(-1)
        if gen.size != 2 * nl:
            raise ValueError("generator must be an affine point (%d limbs)" % (2 * nl))
        im = np.zeros((pp.n, 3 * nl), dtype=np.uint64)
        om = np.zeros((pp.n, 3 * nl), dtype=np.uint64)
        pp._check(pp.lib.zk_msm_mask_sample(pp.h, group, gen.ctypes.data, seed, im.ctypes.data, om.ctypes.data))
        return MsmMask(im, om)


def d_fft(pp, shares_d, fft_mask, rearrange, log2_m, seed=0, out=None, stream=None):
    """dfft/mod.rs:99-134 for all parties; shares_d [n][m/l].  Result in `out` (or back in shares_d if None)."""
    pp._check(pp.lib.zk_d_fft(pp.h, _ptr(shares_d), _ptr(fft_mask.in_mask), _ptr(fft_mask.out_mask), int(rearrange),
                              log2_m, seed, _ptr(out), stream))
    return shares_d if out is None else out


def d_ifft(pp, shares_d, fft_mask, rearrange, log2_m, g=None, seed=0, out=None, stream=None):
    """dfft/mod.rs:137-175; g is an int (coset shift, 1 if None)."""
    garr = None if g is None else pp.fr.encode_one(g)
    pp._check(pp.lib.zk_d_ifft(pp.h, _ptr(shares_d), _ptr(fft_mask.in_mask), _ptr(fft_mask.out_mask), int(rearrange),
                               log2_m, None if garr is None else garr.ctypes.data, seed, _ptr(out), stream))
    return shares_d if out is None else out


def deg_red(pp, x_d, mask, length, seed=0, stream=None):
    """utils/deg_red.rs:80-126 over Fr; x_d [n][length] in place."""
    pp._check(pp.lib.zk_deg_red(pp.h, _ptr(x_d), _ptr(mask.in_mask), _ptr(mask.out_mask), length, seed, stream))
    return x_d


def d_pp(pp, num_d, den_d, mask, length, seed=0, out=None, stream=None):
    """dpp/mod.rs:15-87; num_d, den_d [n][length]."""
    out = out or pp.alloc_fr(pp.n * length)
    pp._check(pp.lib.zk_d_pp(pp.h, _ptr(num_d), _ptr(den_d), _ptr(mask.in_mask), _ptr(mask.out_mask), length, seed,
                             _ptr(out), stream))
    return out


def msm(pp, group, bases_d, scalars_d, length, len_scalars=None, stream=None):
    """G::msm (dmsm/mod.rs:73): returns one Jacobian point as a uint64 array [3 * coord limbs]."""
    nl = pp.fq.nl * (2 if group == ZK_G2 else 1)
    out = np.zeros(3 * nl, dtype=np.uint64)
    ls = length if len_scalars is None else len_scalars
    pp._check(pp.lib.zk_msm(pp.h, group, _ptr(bases_d), length, _ptr(scalars_d), ls, out.ctypes.data, stream))
    return out


def msm_batch(pp, group, bases_d, scalar_vectors, length, stream=None):
    """zk_msm_batch: ONE base vector against several scalar vectors (G::msm once per witness, one Pippenger pass):
    returns [nvec][3 * coord limbs] Jacobian points."""
    nl = pp.fq.nl * (2 if group == ZK_G2 else 1)
    nv = len(scalar_vectors)
    out = np.zeros((nv, 3 * nl), dtype=np.uint64)
    arr = (C.c_void_p * nv)(*[_ptr(v) for v in scalar_vectors])
    pp._check(pp.lib.zk_msm_batch(pp.h, group, _ptr(bases_d), length, arr, nv, out.ctypes.data, stream))
    return out


def unpack_points(pp, group, shares_d, nchunks, parties=None, two=True, out=None, stream=None):
    """pss.rs:125-166 over group elements: shares_d [nparties][nchunks] affine -> [nchunks][l] affine (device).
    two=False: unpack (all n parties); parties: ascending ids of the shares present (lagrange_unpack)."""
    width = pp.fq.nl * (4 if group == ZK_G2 else 2) * 8
    out = out or DeviceBuffer(pp, nchunks * pp.l * width)
    if not two:
        pp._check(pp.lib.zk_pss_unpack_points(pp.h, group, _ptr(shares_d), nchunks, _ptr(out), stream))
        return out
    if parties is None:
        pp._check(pp.lib.zk_pss_unpack2_points(pp.h, group, _ptr(shares_d), None, pp.n, nchunks, _ptr(out), stream))
    else:
        ids = (C.c_uint32 * len(parties))(*parties)
        pp._check(pp.lib.zk_pss_unpack2_points(pp.h, group, _ptr(shares_d), ids, len(parties), nchunks, _ptr(out), stream))
    return out


def fr_to_bytes(pp, x_d, count, stream=None):
    """ark-serialize CanonicalSerialize of `count` Fr elements of a device vector: bytes (32 or 48 per element,
    little-endian canonical integers) -- the payload of an mpc-net frame (ser_net.rs:24-25)."""
    out = pp.alloc_fr(count)
    pp._check(pp.lib.zk_fr_to_bytes(pp.h, _ptr(x_d), count, out.ptr, stream))
    pp.sync(stream)
    return out.to_numpy(dtype=np.uint8)[: count * pp.fr.nbytes].tobytes()


def fr_from_bytes(pp, data, stream=None):
    """CanonicalDeserialize of a byte string of Fr elements into a device vector (Montgomery form); raises if an
    element is not below the modulus, as arkworks does."""
    nb = pp.fr.nbytes
    host = (np.ascontiguousarray(data).view(np.uint8).reshape(-1) if isinstance(data, np.ndarray)
            else np.frombuffer(data, dtype=np.uint8))
    if host.size % nb:
        raise ValueError("byte length is not a multiple of the element size")
    count = host.size // nb
    raw = DeviceBuffer.from_numpy(pp, host)
    out = pp.alloc_fr(max(count, 1))
    pp._check(pp.lib.zk_fr_from_bytes(pp.h, raw.ptr, count, out.ptr, stream))
    return out


def msm_precompute(pp, group, bases_d, length, stream=None):
    """zk_msm_precompute: fixed-base table for the affine vector bases_d [length]; later MSMs over it use the table."""
    pp._check(pp.lib.zk_msm_precompute(pp.h, group, _ptr(bases_d), length, stream))


def msm_forget(pp, bases_d):
    pp._check(pp.lib.zk_msm_forget(pp.h, _ptr(bases_d)))


def msm_table_info(pp, group, bases_d):
    import ctypes as C
    info = (C.c_int * 2)()
    pp._check(pp.lib.zk_msm_table_info(pp.h, group, _ptr(bases_d), info))
    return {"window_bits": info[0], "windows": info[1]}


# Base-field Montgomery products (one product = 2 N^2 + N multiply instructions: 136 v_mad_u64_u32 for 8 limbs) that ONE mixed
# XYZZ addition of the accumulate kernels costs, counted in multiply instructions (csrc/field.hpp):
#   G1: 8 products + Y3 = R (Q - X3) - Y1 PPP as one pass with one reduction (3 N^2 + N)      -> 1288 / 136 = 9.47
#   G2: 8 Fq2 products with lazy reduction (3 N^2 + 2 (N^2 + N) = 336) + 2 Fq2 squarings (272)   -> 3232 / 136 = 23.76
# (the textbook counts are 10 and 28; the accounting of bench.py uses what the kernels execute)
MULS_PER_ADD = {ZK_G1: 1288 / 136, ZK_G2: 3232 / 136}


def msm_plan(pp, group, length):
    """The Pippenger plan zk_msm uses for `length` points: dict(window_bits, windows, lane_points, muls_per_add,
    muls_per_add_textbook); muls_per_add = base-field products per mixed addition as executed (MULS_PER_ADD)."""
    import ctypes as C
    plan = (C.c_int * 4)()
    pp._check(pp.lib.zk_msm_plan(pp.h, group, length, plan))
    return {"window_bits": plan[0], "windows": plan[1], "lane_points": plan[2],
            "muls_per_add": round(MULS_PER_ADD[group], 2), "muls_per_add_textbook": plan[3]}


def d_msm(pp, group, bases_d, scalars_d, length, msm_mask=None, stream=None):
    """dmsm/mod.rs:59-102 for all parties: bases_d [n][length] affine, scalars_d [n][length].
    Returns the n parties' output shares as a uint64 array [n][3 * coord limbs] (Jacobian)."""
    msm_mask = msm_mask or MsmMask.zero()
    nl = pp.fq.nl * (2 if group == ZK_G2 else 1)
    out = np.zeros((pp.n, 3 * nl), dtype=np.uint64)
    im = None if msm_mask.in_mask is None else np.ascontiguousarray(msm_mask.in_mask, dtype=np.uint64)
    om = None if msm_mask.out_mask is None else np.ascontiguousarray(msm_mask.out_mask, dtype=np.uint64)
    pp._check(pp.lib.zk_d_msm(pp.h, group, _ptr(bases_d), _ptr(scalars_d), length,
                              None if im is None else im.ctypes.data, None if om is None else om.ctypes.data,
                              out.ctypes.data, stream))
    return out


def deg_red_parties(pp, x_d, parties, mask, length, seed=0, out=None, stream=None):
    """deg_red when only `parties` reached the king (ser_net.rs:57-94 -> pss.rs:170-221); x_d [len(parties)][length]."""
    out = out or pp.alloc_fr(pp.n * length)
    arr = (C.c_uint32 * len(parties))(*parties)
    pp._check(pp.lib.zk_deg_red_parties(pp.h, _ptr(x_d), arr, len(parties), _ptr(mask.in_mask), _ptr(mask.out_mask),
                                        length, seed, _ptr(out), stream))
    return out


def d_msm_parties(pp, group, bases_d, scalars_d, length, parties, msm_mask=None, stream=None):
    """d_msm when only `parties` reached the king; bases_d / scalars_d [len(parties)][length]."""
    msm_mask = msm_mask or MsmMask.zero()
    nl = pp.fq.nl * (2 if group == ZK_G2 else 1)
    out = np.zeros((pp.n, 3 * nl), dtype=np.uint64)
    arr = (C.c_uint32 * len(parties))(*parties)
    im = None if msm_mask.in_mask is None else np.ascontiguousarray(msm_mask.in_mask, dtype=np.uint64)
    om = None if msm_mask.out_mask is None else np.ascontiguousarray(msm_mask.out_mask, dtype=np.uint64)
    pp._check(pp.lib.zk_d_msm_parties(pp.h, group, _ptr(bases_d), _ptr(scalars_d), length, arr, len(parties),
                                      None if im is None else im.ctypes.data, None if om is None else om.ctypes.data,
                                      out.ctypes.data, stream))
    return out


def vec_scale(pp, x_d, k, length, stream=None):
    karr = pp.fr.encode_one(k)
    pp._check(pp.lib.zk_vec_scale(pp.h, _ptr(x_d), karr.ctypes.data, length, stream))
    return x_d


def vec_mul_sub(pp, out_d, a_d, b_d, c_d, length, stream=None):
    pp._check(pp.lib.zk_vec_mul_sub(pp.h, _ptr(out_d), _ptr(a_d), _ptr(b_d), _ptr(c_d), length, stream))
    return out_d
