"""Curve / field constants for the CPU oracle.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` may be imported by the
product (`zk-saas_amd/`); only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` use it, and only as the checker.

PARITY STATUS: pinned by every check the reference's own tests hold for this path (restated in
tests/test_oracle_*.py: SURVEY.md 8c items 2-9), by the reference's SHA-256 public-output known answer and by the
output of the reference's own wasm witness calculator run under node (tests/golden/).  Byte-level parity against an
arkworks-PRODUCED proof is UNPINNED: the reference holds no expected proof / NTT / MSM bytes and cannot be built
here (no Rust toolchain, arkworks not vendored) -- see DESIGN.md section 5.

The arithmetic of the reference lives in third-party arkworks crates that are
not vendored under /root/reference: ark-ff / ark-ec / ark-poly
``^0.4``, ark-bn254 / ark-bls12-377 ``^0.4.0``.  BLS12-381 is not a dependency
of the reference at all; it is instantiated here from its
published parameters.  Constants below are the published curve parameters; the
2-adic roots of unity are *derived* (GENERATOR^((r-1)/2^s), which is how
ark-ff's ``MontConfig`` derives ``TWO_ADIC_ROOT_OF_UNITY``) and asserted
against the values recorded in SURVEY.md section 8c.
"""


class Curve:
    def __init__(self, name, r, r_gen, q, b1, g1, nonres, b2, g2, fq_limbs64):
        self.name = name
        self.r = r            # scalar field modulus (Fr)
        self.r_gen = r_gen    # F::GENERATOR of Fr (multiplicative generator)
        self.q = q            # base field modulus (Fq)
        self.b1 = b1          # G1: y^2 = x^3 + b1
        self.g1 = g1          # G1 generator (affine)
        self.nonres = nonres  # Fq2 = Fq[u]/(u^2 - nonres)
        self.b2 = b2          # G2: y^2 = x^3 + b2, b2 in Fq2 as (c0, c1)
        self.g2 = g2          # G2 generator ((x0,x1),(y0,y1)) or None
        self.fq_limbs64 = fq_limbs64
        # two-adicity of Fr and the 2^s-th root of unity
        s, t = 0, r - 1
        while t % 2 == 0:
            t //= 2
            s += 1
        self.two_adicity = s
        self.two_adic_root = pow(r_gen, t, r)


_BN254_R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
_BN254_Q = 21888242871839275222246405745257275088696311157297823662689037894645226208583

# b2 = 3 / (9 + u) in Fq2 = Fq[u]/(u^2+1)
def _bn254_b2():
    q = _BN254_Q
    # 1/(9+u) = (9-u)/(81+1)
    inv82 = pow(82, q - 2, q)
    return (3 * 9 * inv82 % q, (-3) * inv82 % q)


BN254 = Curve(
    "bn254",
    r=_BN254_R,
    r_gen=5,
    q=_BN254_Q,
    b1=3,
    g1=(1, 2),
    nonres=_BN254_Q - 1,
    b2=_bn254_b2(),
    g2=(
        (10857046999023057135944570762232829481370756359578518086990519993285655852781,
         11559732032986387107991004021392285783925812861821192530917403151452391805634),
        (8495653923123431417604973247489272438418190587263600148770280649306958101930,
         4082367875863433681332203403145435568316851327593401208105741076214120093531),
    ),
    fq_limbs64=4,
)

_BLS381_R = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
_BLS381_Q = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB

BLS12_381 = Curve(
    "bls12_381",
    r=_BLS381_R,
    r_gen=7,
    q=_BLS381_Q,
    b1=4,
    g1=(
        0x17F1D3A73197D7942695638C4FA9AC0FC3688C4F9774B905A14E3A3F171BAC586C55E83FF97A1AEFFB3AF00ADB22C6BB,
        0x08B3F481E3AAA0F1A09E30ED741D8AE4FCF5E095D5D00AF600DB18CB2C04B3EDD03CC744A2888AE40CAA232946C5E7E1,
    ),
    nonres=_BLS381_Q - 1,
    b2=(4, 4),
    g2=(
        (0x024AA2B2F08F0A91260805272DC51051C6E47AD4FA403B02B4510B647AE3D1770BAC0326A805BBEFD48056C8C121BDB8,
         0x13E02B6052719F607DACD3A088274F65596BD0D09920B61AB5DA61BBDC7F5049334CF11213945D57E5AC7D055D042B7E),
        (0x0CE5D527727D6E118CC9CDC6DA2E351AADFD9BAA8CBDD3A76D429A695160D12C923AC9CC3BACA289E193548608B82801,
         0x0606C4A02EA734CC32ACD2B02BC28B99CB3E287E85A763AF267492AB572E99AB3F370D275CEC1DA1AAA9075FF05F79BE),
    ),
    fq_limbs64=6,
)

_BLS377_R = 8444461749428370424248824938781546531375899335154063827935233455917409239041
_BLS377_Q = 0x01AE3A4617C510EAC63B05C06CA1493B1A22D9F300F5138F1EF3622FBA094800170B5D44300000008508C00000000001

BLS12_377 = Curve(
    "bls12_377",
    r=_BLS377_R,
    r_gen=22,
    q=_BLS377_Q,
    b1=1,
    g1=(
        81937999373150964239938255573465948239988671502647976594219695644855304257327692006745978603320413799295628339695,
        241266749859715473739788878240585681733927191168601896383759122102112907357779751001206799952863815012735208165030,
    ),
    nonres=_BLS377_Q - 5,
    b2=None,   # G2 of BLS12-377 is not exercised by the reference's hot path tests
    g2=None,
    fq_limbs64=6,
)

CURVES = {c.name: c for c in (BN254, BLS12_381, BLS12_377)}

# values recorded in SURVEY.md 8c (checked in tests/test_oracle_fields.py)
SURVEY_TWO_ADIC = {
    "bn254": (28, 19103219067921713944291392827692070036145651957329286315305642004821462161904),
    "bls12_381": (32, 10238227357739495823651030575849232062558860180284477541189508159991286009131),
    "bls12_377": (47, 8065159656716812877374967518403273466521432693661810619979959746626482506078),
}
