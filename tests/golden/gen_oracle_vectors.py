#!/usr/bin/env python3
"""Regenerates oracle_vectors.json: small input/expected-output vectors of the hot path computed by the Python
big-int oracle on the inputs the reference's own tests use (dfft_test.rs: x[i] = i over BLS12-377, l = 2;
dpp_test.rs: x = 1..32; pss.rs tests).  The GPU tests replay them through the C ABI."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import dist as od
from oracle.field import Domain, bitrev_permute
from oracle.params import BLS12_377, BN254
from oracle.pss import PackedSharingParams


def main():
    out = {}
    c = BLS12_377
    o = PackedSharingParams(c, 2)
    for m in (8, 64):
        dom = Domain(c, m)
        x = list(range(m))
        y = list(x)
        bitrev_permute(y)
        shares = od.transpose(od.stride_pack(y, o, 1))
        res = od.d_fft(shares, [od.FftMask.zero(m // 2)] * o.n, False, dom, o, seed=2)
        out["d_fft_bls12_377_l2_m%d" % m] = {
            "curve": "bls12_377", "l": 2, "m": m, "input_x": [str(v) for v in x], "deal_seed": 1, "king_seed": 2,
            "input_shares": [[str(v) for v in s] for s in shares], "output_shares": [[str(v) for v in s] for s in res],
            "reconstructed": [str(v) for v in dom.fft(x)]}
    o = PackedSharingParams(BN254, 2)
    secrets = [3, 11, 2 ** 200 + 7, BN254.r - 1]
    out["pack_bn254_l2"] = {"curve": "bn254", "l": 2, "secrets": [str(v) for v in secrets], "seed": 77,
                            "shares": [[str(v) for v in s] for s in od.transpose(od.pack_vec(secrets, o, 77))]}
    with open(os.path.join(HERE, "oracle_vectors.json"), "w") as fh:
        json.dump(out, fh)


if __name__ == "__main__":
    main()
