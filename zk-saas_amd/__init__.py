"""zksaas_amd: MI355X-native hot path of zkSaaS (d_fft / d_msm / d_pp over packed secret shares and the
Groth16 prover that composes them).  Host-side mirror of the reference's dist-primitives / secret-sharing
interface over the C ABI in include/zksaas.h; all arithmetic runs in libzksaas_hip.so on the GPU."""
from . import fields  # noqa: F401
from ._lib import LIB_PATH, SYMBOLS, ZkError, load  # noqa: F401
from .api import (Context, DeviceBuffer, FftMask, DegRedMask, MsmMask, PackedSharingParams, d_fft, d_ifft, d_msm,  # noqa: F401
                  d_pp, deg_red)
from . import groth16, sha256_circuit  # noqa: F401,E402
