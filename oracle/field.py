"""Prime-field helpers and a restatement of ark-poly's Radix2EvaluationDomain.

TEST INFRASTRUCTURE ONLY (see oracle/params.py).

ark-poly ^0.4.0 is a third-party dependency not present under /root/reference
(SURVEY.md F2); the behaviours restated here are the ones the reference's call
sites rely on (secret-sharing/src/pss.rs:44-52,81-84,104,119,129-132,145,160;
dist-primitives/src/dfft/mod.rs:49,121,159,279):

* ``Radix2EvaluationDomain::new(k)``: size = next power of two >= k,
  group_gen = TWO_ADIC_ROOT^(2^(s - log2 size)), offset = 1.
* ``get_coset(g)``: same subgroup, evaluation points g*w^i.
* ``fft_in_place``: resize (zero-pad OR truncate) to the domain size, multiply
  coefficient i by offset^i, evaluate at w^i, natural output order.
* ``ifft_in_place``: resize, interpolate, multiply by size^-1 and coefficient
  i by offset^-i.
* ``distribute_powers(v, g)``: v[i] *= g^i.
* ``element(i)`` = offset * w^i.

All coefficient types are handled through a tiny "ops" adapter so that the same
FFT works over Fr elements and over curve points (ark-poly ``DomainCoeff``).
"""


def inv_mod(a, p):
    return pow(a, p - 2, p)


class FieldOps:
    """DomainCoeff adapter for elements of F_p themselves."""

    def __init__(self, p):
        self.p = p
        self.zero = 0

    def add(self, a, b):
        return (a + b) % self.p

    def sub(self, a, b):
        return (a - b) % self.p

    def mul(self, a, k):  # a * scalar
        return a * k % self.p

    def eq(self, a, b):
        return a % self.p == b % self.p


def log2_ceil(n):
    """ark_std::log2: ceil(log2(n)) with log2(0)=log2(1)=0."""
    if n <= 1:
        return 0
    return (n - 1).bit_length()


def bitrev_permute(data):
    """dist-primitives/src/dfft/mod.rs:322-335 fft_in_place_rearrange (in place)."""
    n = len(data)
    target = 0
    for pos in range(n):
        if target > pos:
            data[target], data[pos] = data[pos], data[target]
        mask = n >> 1
        while target & mask != 0:
            target &= ~mask
            mask >>= 1
        target |= mask
    return data


class Domain:
    """Radix2EvaluationDomain<F> (optionally a coset)."""

    def __init__(self, curve, num_coeffs, offset=1):
        self.curve = curve
        self.p = curve.r
        size = 1 << log2_ceil(num_coeffs)
        self.size = size
        self.log_size = log2_ceil(size)
        if self.log_size > curve.two_adicity:
            raise ValueError("domain too large")
        self.group_gen = pow(curve.two_adic_root, 1 << (curve.two_adicity - self.log_size), self.p)
        self.group_gen_inv = inv_mod(self.group_gen, self.p)
        self.size_inv = inv_mod(size % self.p, self.p)
        self.offset = offset % self.p
        self.offset_inv = inv_mod(self.offset, self.p)

    def get_coset(self, offset):
        return Domain(self.curve, self.size, offset)

    def coset_offset(self):
        return self.offset

    def coset_offset_inv(self):
        return self.offset_inv

    def element(self, i):
        return self.offset * pow(self.group_gen, i, self.p) % self.p

    def elements(self):
        out, cur = [], self.offset
        for _ in range(self.size):
            out.append(cur)
            cur = cur * self.group_gen % self.p
        return out

    def evaluate_vanishing_polynomial(self, tau):
        # z(tau) = tau^size - offset^size
        return (pow(tau, self.size, self.p) - pow(self.offset, self.size, self.p)) % self.p

    # -- core transform (natural in, natural out) ---------------------------------
    def _ntt(self, vals, root, ops):
        n = self.size
        a = list(vals)
        bitrev_permute(a)
        length = 2
        while length <= n:
            wlen = pow(root, n // length, self.p)
            half = length // 2
            tw = [1] * half
            for k in range(1, half):
                tw[k] = tw[k - 1] * wlen % self.p
            for start in range(0, n, length):
                for k in range(half):
                    u = a[start + k]
                    v = ops.mul(a[start + k + half], tw[k]) if k else a[start + k + half]
                    a[start + k] = ops.add(u, v)
                    a[start + k + half] = ops.sub(u, v)
            length *= 2
        return a

    @staticmethod
    def distribute_powers(vals, g, ops, p, c=1):
        """v[i] *= c * g^i (ark-poly distribute_powers_and_mul_by_const)."""
        out, cur = [], c % p
        for v in vals:
            out.append(ops.mul(v, cur))
            cur = cur * g % p
        return out

    def _resize(self, vals, ops):
        vals = list(vals)[: self.size]
        return vals + [ops.zero] * (self.size - len(vals))

    def fft(self, coeffs, ops=None):
        ops = ops or FieldOps(self.p)
        c = self._resize(coeffs, ops)
        if self.offset != 1:
            c = self.distribute_powers(c, self.offset, ops, self.p)
        return self._ntt(c, self.group_gen, ops)

    def ifft(self, evals, ops=None):
        ops = ops or FieldOps(self.p)
        e = self._resize(evals, ops)
        c = self._ntt(e, self.group_gen_inv, ops)
        return self.distribute_powers(c, self.offset_inv, ops, self.p, self.size_inv)
