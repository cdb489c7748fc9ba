"""Documented counter-based PRNG shared by oracle, C restatement and the HIP path.

TEST INFRASTRUCTURE ONLY (see oracle/params.py).

The reference draws share randomness from ``ark_std::test_rng()`` /
``thread_rng()`` (dist-primitives/src/dfft/mod.rs:251, utils/pack.rs:14,
utils/deg_red.rs:108), whose streams cannot be replayed without Rust
(SURVEY.md 8c).  Reconstructed results do not depend on that randomness
(SURVEY.md F6); to make even the *shares* comparable bit-for-bit between the
oracle and the GPU, every implementation in this repo uses the generator below.

``rand_fp(seed, idx, p)``: SplitMix64 stream whose initial state is
``mix(seed ^ mix(idx + C0))``; draw ceil(bits(p)/64) little-endian u64 limbs,
mask the top limb to bits(p) bits, reject and redraw (continuing the stream)
while the value is >= p.
"""

M64 = (1 << 64) - 1
GOLDEN = 0x9E3779B97F4A7C15
C0 = 0x632BE59BD9B4E019


def mix64(z):
    z &= M64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
    return z ^ (z >> 31)


class SplitMix64:
    def __init__(self, state):
        self.state = state & M64

    def next(self):
        self.state = (self.state + GOLDEN) & M64
        return mix64(self.state)


def stream(seed, idx):
    return SplitMix64(mix64((seed & M64) ^ mix64((idx + C0) & M64)))


def rand_fp(seed, idx, p):
    s = stream(seed, idx)
    nbits = p.bit_length()
    nl = (nbits + 63) // 64
    top_mask = (1 << (nbits - 64 * (nl - 1))) - 1
    while True:
        v = 0
        for i in range(nl):
            limb = s.next()
            if i == nl - 1:
                limb &= top_mask
            v |= limb << (64 * i)
        if v < p:
            return v


def rand_vec(seed, n, p, base=0):
    return [rand_fp(seed, base + i, p) for i in range(n)]
