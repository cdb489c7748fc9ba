"""Host-side field constants and arkworks in-memory (Montgomery, little-endian u64 limbs) conversion.

This is data marshalling for the C ABI (SURVEY.md 8b "data layout at the boundary"), not arithmetic on the
hot path.  Moduli are the published curve parameters.
"""
import numpy as np

CURVE_IDS = {"bn254": 0, "bls12_381": 1, "bls12_377": 2}

FR = {
    "bn254": 21888242871839275222246405745257275088548364400416034343698204186575808495617,
    "bls12_381": 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001,
    "bls12_377": 8444461749428370424248824938781546531375899335154063827935233455917409239041,
}
FQ = {
    "bn254": 21888242871839275222246405745257275088696311157297823662689037894645226208583,
    "bls12_381": 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB,
    "bls12_377": 0x01AE3A4617C510EAC63B05C06CA1493B1A22D9F300F5138F1EF3622FBA094800170B5D44300000008508C00000000001,
}
FR_GENERATOR = {"bn254": 5, "bls12_381": 7, "bls12_377": 22}


def nlimbs64(p):
    return (p.bit_length() + 63) // 64


class MontCodec:
    """ints <-> numpy uint64 limb arrays in Montgomery form for modulus p."""

    def __init__(self, p):
        self.p = p
        self.nl = nlimbs64(p)
        self.R = 1 << (64 * self.nl)
        self.Rinv = pow(self.R, -1, p)
        self.nbytes = 8 * self.nl

    def encode(self, vals):
        """list of ints -> array [len, nl] uint64"""
        out = np.empty((len(vals), self.nl), dtype=np.uint64)
        mask = (1 << 64) - 1
        for i, v in enumerate(vals):
            m = (v % self.p) * self.R % self.p
            for k in range(self.nl):
                out[i, k] = (m >> (64 * k)) & mask
        return out

    def decode(self, arr):
        arr = np.asarray(arr, dtype=np.uint64).reshape(-1, self.nl)
        out = []
        for row in arr:
            m = 0
            for k in range(self.nl):
                m |= int(row[k]) << (64 * k)
            out.append(m * self.Rinv % self.p)
        return out

    def encode_one(self, v):
        return self.encode([v])[0]
