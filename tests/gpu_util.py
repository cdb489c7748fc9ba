"""Shared helpers for the -m gpu parity tests: every compute call goes through the C ABI (ctypes)."""
import functools

import numpy as np

import zksaas_amd as zk
from oracle.params import CURVES
from oracle.pss import PackedSharingParams as OraclePP


@functools.lru_cache(maxsize=None)
def ctx(curve="bn254", l=2):
    return zk.PackedSharingParams(curve, l)


@functools.lru_cache(maxsize=None)
def opp(curve="bn254", l=2):
    return OraclePP(CURVES[curve], l)


def up(pp, vals):
    return pp.upload_fr(vals)


def up_parties(pp, per_party):
    """list over parties of equal-length int vectors -> device [n][len]"""
    flat = [v for vec in per_party for v in vec]
    return pp.upload_fr(flat)


def down_parties(pp, buf, nparties, length):
    flat = pp.download_fr(buf, nparties * length)
    return [flat[i * length:(i + 1) * length] for i in range(nparties)]


def enc_affine(pp, pts, g2=False):
    """affine oracle points -> uint64 array [len][2*coord limbs]; None -> (0,0)"""
    rows = []
    for p in pts:
        if p is None:
            coords = [0, 0, 0, 0] if g2 else [0, 0]
        elif g2:
            coords = [p[0][0], p[0][1], p[1][0], p[1][1]]
        else:
            coords = [p[0], p[1]]
        rows.append(pp.fq.encode(coords).reshape(-1))
    return np.stack(rows) if rows else np.zeros((0, (4 if g2 else 2) * pp.fq.nl), dtype=np.uint64)


def dec_jacobian(pp, arr, g2=False):
    """uint64 [3*coord limbs] -> oracle Jacobian tuple"""
    vals = pp.fq.decode(np.asarray(arr).reshape(-1, pp.fq.nl))
    if g2:
        return ((vals[0], vals[1]), (vals[2], vals[3]), (vals[4], vals[5]))
    return (vals[0], vals[1], vals[2])


def enc_jacobian(pp, P, g2=False):
    coords = [P[0][0], P[0][1], P[1][0], P[1][1], P[2][0], P[2][1]] if g2 else list(P)
    return pp.fq.encode(coords).reshape(-1)
