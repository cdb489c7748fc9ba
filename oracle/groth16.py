"""groth16/ restated: QAP, witness extension (circom_h / libsnark_h), proof elements, CRS packing.

TEST INFRASTRUCTURE ONLY (see oracle/params.py).

Follows groth16/src/qap.rs, ext_wit.rs, prove.rs, proving_key.rs and
groth16/examples/sha256.rs.  ark-groth16 ^0.4.0 / ark-circom (git, unpinned) are
third-party and absent; what they contribute on this path is
restated from their published algorithm and anchored as SURVEY.md 8c describes:
``circom_ref``/``libsnark_ref`` are in-tree in the reference
(ext_wit.rs:204-285), the prover assembly is prove.rs, and the trapdoor setup is
checked in closed form (the Groth16 verification equation evaluated in the
exponent), so no pairing is needed.
"""

from .field import Domain, FieldOps, bitrev_permute, inv_mod
from .prng import rand_fp
from . import dist
from .dist import transpose


class R1CS:
    """ark_relations ConstraintMatrices as used by qap.rs:42-89.

    a, b, c: list (one per constraint) of lists of (coeff, variable index).
    Variable 0 is the constant 1; the first ``num_instance`` variables are public.
    """

    def __init__(self, num_instance, num_witness, a, b, c):
        self.num_instance_variables = num_instance
        self.num_witness_variables = num_witness
        self.a, self.b, self.c = a, b, c
        self.num_constraints = len(a)
        assert len(b) == len(a) and len(c) == len(a)

    @property
    def num_variables(self):
        return self.num_instance_variables + self.num_witness_variables


def evaluate_constraint(row, assignment, p):
    """ark_groth16::r1cs_to_qap::evaluate_constraint (qap.rs:63-64)."""
    acc = 0
    for coeff, idx in row:
        acc += coeff * assignment[idx]
    return acc % p


def is_satisfied(r1cs, w, p):
    for ra, rb, rc in zip(r1cs.a, r1cs.b, r1cs.c):
        if evaluate_constraint(ra, w, p) * evaluate_constraint(rb, w, p) % p != evaluate_constraint(rc, w, p):
            return False
    return True


class QAP:
    def __init__(self, num_inputs, num_constraints, a, b, c, domain):
        self.num_inputs, self.num_constraints = num_inputs, num_constraints
        self.a, self.b, self.c, self.domain = a, b, c, domain

    def pss(self, pp, seed=0):
        """qap.rs:91-135: bit-reverse, stride-pack, deal. Returns per-party (a,b,c)."""
        out = []
        for k, v in enumerate((self.a, self.b, self.c)):
            x = list(v)
            bitrev_permute(x)
            out.append(transpose(dist.stride_pack(x, pp, seed + k)))
        return [(out[0][i], out[1][i], out[2][i]) for i in range(pp.n)]


def qap(curve, r1cs, full_assignment):
    """qap.rs:42-89."""
    p = curve.r
    ni, nc = r1cs.num_instance_variables, r1cs.num_constraints
    domain = Domain(curve, nc + ni)
    m = domain.size
    a, b, c = [0] * m, [0] * m, [0] * m
    for i in range(nc):
        a[i] = evaluate_constraint(r1cs.a[i], full_assignment, p)
        b[i] = evaluate_constraint(r1cs.b[i], full_assignment, p)
    a[nc:nc + ni] = [x % p for x in full_assignment[:ni]]
    for i in range(nc):
        c[i] = a[i] * b[i] % p
    return QAP(ni, nc, a, b, c, domain)


def _root_of_unity_2m(curve, domain):
    """ext_wit.rs:120-125: Radix2EvaluationDomain::new(2m).element(1)."""
    return Domain(curve, 2 * domain.size).element(1)


def circom_ref(a, b, c, domain):
    """ext_wit.rs:239-285 (== ark-circom CircomReduction::witness_map_from_matrices)."""
    p = domain.p
    fo = FieldOps(p)
    w2m = _root_of_unity_2m(domain.curve, domain)
    out = []
    for v in (a, b, c):
        v = domain.ifft(v)
        v = Domain.distribute_powers(v, w2m, fo, p)
        out.append(domain.fft(v))
    return [(x * y - z) % p for x, y, z in zip(*out)]


def libsnark_ref(a, b, c, domain):
    """ext_wit.rs:204-237."""
    p = domain.p
    coset = domain.get_coset(domain.curve.r_gen)
    ev = [coset.fft(domain.ifft(v)) for v in (a, b, c)]
    zinv = inv_mod(domain.evaluate_vanishing_polynomial(domain.curve.r_gen), p)
    ab = [(x * y - z) % p * zinv % p for x, y, z in zip(*ev)]
    return coset.ifft(ab)


def circom_h(qap_shares, fft_masks, degred_masks, pp, domain, seed=0, parties=None):
    """ext_wit.rs:104-181 for all parties.

    qap_shares[i] = (a, b, c) of party i; fft_masks[k][i] for k in 0..5;
    degred_masks[i].
    """
    p = pp.p
    n = pp.n
    w2m = _root_of_unity_2m(domain.curve, domain)
    coeff = []
    for k in range(3):
        coeff.append(dist.d_ifft([qap_shares[i][k] for i in range(n)], fft_masks[k], True, domain, w2m, pp,
                                 seed + k, parties))
    evals = []
    for k in range(3):
        evals.append(dist.d_fft(coeff[k], fft_masks[3 + k], False, domain, pp, seed + 3 + k, parties))
    h_eval = [[(x * y - z) % p for x, y, z in zip(evals[0][i], evals[1][i], evals[2][i])] for i in range(n)]
    return dist.deg_red(h_eval, degred_masks, pp, seed + 6, parties=parties)


def libsnark_h(qap_shares, fft_masks, pp, domain, seed=0, parties=None):
    """ext_wit.rs:14-102 for all parties (7 masks)."""
    p = pp.p
    n = pp.n
    coset = domain.get_coset(domain.curve.r_gen)
    coeff = [dist.d_ifft([qap_shares[i][k] for i in range(n)], fft_masks[k], True, domain, coset.offset, pp,
                         seed + k, parties) for k in range(3)]
    evals = [dist.d_fft(coeff[k], fft_masks[3 + k], True, domain, pp, seed + 3 + k, parties) for k in range(3)]
    zinv = inv_mod(domain.evaluate_vanishing_polynomial(domain.curve.r_gen), p)
    h_eval = [[(x * y - z) % p * zinv % p for x, y, z in zip(evals[0][i], evals[1][i], evals[2][i])]
              for i in range(n)]
    return dist.d_ifft(h_eval, fft_masks[6], False, domain, coset.offset_inv, pp, seed + 6, parties)


# ---------------------------------------------------------------------------- setup (trapdoor)
class Trapdoor:
    def __init__(self, alpha, beta, gamma, delta, tau):
        self.alpha, self.beta, self.gamma, self.delta, self.tau = alpha, beta, gamma, delta, tau

    @staticmethod
    def from_seed(seed, p):
        v = [rand_fp(seed, i, p) for i in range(5)]
        return Trapdoor(*v)


def lagrange_coeffs_at(domain, tau):
    """EvaluationDomain::evaluate_all_lagrange_coefficients(tau) for tau outside the domain."""
    p = domain.p
    m = domain.size
    zt = (pow(tau, m, p) - 1) % p
    assert zt != 0
    out = []
    w = 1
    for _ in range(m):
        # L_i(tau) = Z(tau) * w^i / (m * (tau - w^i))
        out.append(zt * w % p * inv_mod(m * (tau - w) % p, p) % p)
        w = w * domain.group_gen % p
    return out


class ProvingKeyScalars:
    """Discrete logs of every CRS element (what a trapdoor holder knows)."""


def setup_scalars(curve, r1cs, td):
    """ark-groth16 generate_parameters with CircomReduction, in the exponent.

    a/b/c = LibsnarkReduction::instance_map_with_evaluation; h = CircomReduction::h_query_scalars.
    """
    p = curve.r
    ni, nc = r1cs.num_instance_variables, r1cs.num_constraints
    domain = Domain(curve, nc + ni)
    m = domain.size
    u = lagrange_coeffs_at(domain, td.tau)
    nv = r1cs.num_variables
    a, b, c = [0] * nv, [0] * nv, [0] * nv
    for i in range(ni):
        a[i] = u[nc + i]
    for i in range(nc):
        ui = u[i]
        for coeff, idx in r1cs.a[i]:
            a[idx] = (a[idx] + ui * coeff) % p
        for coeff, idx in r1cs.b[i]:
            b[idx] = (b[idx] + ui * coeff) % p
        for coeff, idx in r1cs.c[i]:
            c[idx] = (c[idx] + ui * coeff) % p
    ginv, dinv = inv_mod(td.gamma, p), inv_mod(td.delta, p)
    abc = [(td.beta * x + td.alpha * y + z) % p for x, y, z in zip(a, b, c)]
    k = ProvingKeyScalars()
    k.domain = domain
    k.a_query, k.b_query = a, b
    k.gamma_abc = [x * ginv % p for x in abc[:ni]]
    k.l_query = [x * dinv % p for x in abc[ni:]]
    # CircomReduction::h_query_scalars(m-1, tau, zt, delta_inverse)
    scalars, t = [], dinv
    for _ in range(2 * (m - 1) + 1):
        scalars.append(t)
        t = t * td.tau % p
    d2 = Domain(curve, len(scalars))
    k.h_query = d2.ifft(scalars)[1::2]
    k.td = td
    return k


# ---------------------------------------------------------------------------- prover
def prove_scalars(curve, r1cs, key, w, r, s):
    """Discrete logs (A, B, C) of the proof: closed form a trapdoor holder can evaluate."""
    p = curve.r
    td = key.td
    ni = r1cs.num_instance_variables
    q = qap(curve, r1cs, w)
    h = circom_ref(q.a, q.b, q.c, q.domain)
    A = (td.alpha + sum(x * y for x, y in zip(key.a_query, w)) + r * td.delta) % p
    B = (td.beta + sum(x * y for x, y in zip(key.b_query, w)) + s * td.delta) % p
    C = (s * A + r * B - r * s % p * td.delta
         + sum(x * y for x, y in zip(key.l_query, w[ni:]))
         + sum(x * y for x, y in zip(key.h_query, h))) % p
    return A, B, C


def verify_scalars(curve, r1cs, key, w, proof_scalars):
    """Groth16 verification equation e(A,B) = e(alpha,beta) e(IC,gamma) e(C,delta), in the exponent."""
    p = curve.r
    td = key.td
    A, B, C = proof_scalars
    ic = sum(x * y for x, y in zip(key.gamma_abc, w[:r1cs.num_instance_variables])) % p
    return A * B % p == (td.alpha * td.beta + ic * td.gamma + C * td.delta) % p


class ProvingKey:
    """ark_groth16::ProvingKey fields used on the path (affine points)."""


def proving_key_points(key, G1, G2):
    def mul1(s):
        return G1.to_affine(G1.mul(G1.from_affine(G1.gen), s))

    def mul2(s):
        return G2.to_affine(G2.mul(G2.from_affine(G2.gen), s))

    pk = ProvingKey()
    td = key.td
    pk.a_query = [mul1(s) for s in key.a_query]
    pk.b_g1_query = [mul1(s) for s in key.b_query]
    pk.b_g2_query = [mul2(s) for s in key.b_query]
    pk.h_query = [mul1(s) for s in key.h_query]
    pk.l_query = [mul1(s) for s in key.l_query]
    pk.alpha_g1, pk.beta_g1, pk.delta_g1 = mul1(td.alpha), mul1(td.beta), mul1(td.delta)
    pk.beta_g2, pk.delta_g2 = mul2(td.beta), mul2(td.delta)
    return pk


def create_proof_local(curve, r1cs, pk, G1, G2, w, r, s):
    """ark_groth16 create_proof_with_reduction_and_matrices (sha256.rs:191-199), Jacobian outputs."""
    ni = r1cs.num_instance_variables
    q = qap(curve, r1cs, w)
    h = circom_ref(q.a, q.b, q.c, q.domain)
    assignment = w[1:]
    h_acc = G1.msm(pk.h_query, h)
    l_aux = G1.msm(pk.l_query, w[ni:])
    fa = G1.from_affine
    d1 = fa(pk.delta_g1)
    g_a = G1.sum([G1.mul(d1, r), fa(pk.a_query[0]), G1.msm(pk.a_query[1:], assignment), fa(pk.alpha_g1)])
    g1_b = G1.identity if r % curve.r == 0 else G1.sum(
        [G1.mul(d1, s), fa(pk.b_g1_query[0]), G1.msm(pk.b_g1_query[1:], assignment), fa(pk.beta_g1)])
    fb = G2.from_affine
    g2_b = G2.sum([G2.mul(fb(pk.delta_g2), s), fb(pk.b_g2_query[0]), G2.msm(pk.b_g2_query[1:], assignment),
                   fb(pk.beta_g2)])
    g_c = G1.sum([G1.mul(g_a, s), G1.mul(g1_b, r), G1.neg(G1.mul(d1, r * s % curve.r)), l_aux, h_acc])
    return g_a, g2_b, g_c


class PackedProvingKeyShare:
    """groth16/src/proving_key.rs:18-37."""


def _det_pack_points(points, pp, gops, identity):
    chunks = []
    for j in range(0, len(points), pp.l):
        ch = list(points[j:j + pp.l])
        # cfg_chunks! yields a short last chunk; det_pack's resize(t) zero-pads it (pss.rs:78)
        chunks.append(pp.det_pack(ch, gops))
    return chunks


def pack_proving_key(pk, pp, G1, G2, g1ops, g2ops):
    """proving_key.rs:47-123; returns n PackedProvingKeyShare with affine vectors."""
    def pack(points, G, ops):
        jac = [G.from_affine(x) for x in points]
        packed = _det_pack_points(jac, pp, ops, G.identity)
        return [[G.to_affine(packed[j][i]) for j in range(len(packed))] for i in range(pp.n)]

    s = pack(pk.a_query[1:], G1, g1ops)
    u = pack(pk.h_query, G1, g1ops)
    w = pack(pk.l_query, G1, g1ops)
    h = pack(pk.b_g1_query[1:], G1, g1ops)
    v = pack(pk.b_g2_query[1:], G2, g2ops)
    out = []
    for i in range(pp.n):
        sh = PackedProvingKeyShare()
        sh.s, sh.u, sh.w, sh.h, sh.v = s[i], u[i], w[i], h[i], v[i]
        sh.a_query0, sh.b_g1_query0, sh.b_g2_query0 = pk.a_query[0], pk.b_g1_query[0], pk.b_g2_query[0]
        sh.delta_g1, sh.delta_g2 = pk.delta_g1, pk.delta_g2
        sh.alpha_g1, sh.beta_g1, sh.beta_g2 = pk.alpha_g1, pk.beta_g1, pk.beta_g2
        out.append(sh)
    return out


def pack_from_witness(pp, assignment, seed):
    """groth16/examples/sha256.rs:131-156 (last chunk zero-padded)."""
    a = list(assignment)
    if len(a) % pp.l:
        a += [0] * (pp.l - len(a) % pp.l)
    return transpose(dist.pack_vec(a, pp, seed))


def dist_prove(curve, pp, crs_shares, qap_shares, a_shares, ax_shares, r, s, fft_masks, degred_masks,
               g1_masks, g2_mask, domain, G1, G2, g1ops, g2ops, seed=0):
    """groth16/examples/sha256.rs:32-129 (dsha256) for all parties; r_share = r, s_share = s.

    Returns per-party (pi_a, pi_b_g2, pi_c) Jacobian shares.
    """
    n = pp.n
    h_shares = circom_h(qap_shares, fft_masks, degred_masks, pp, domain, seed)
    fa, fb = G1.from_affine, G2.from_affine

    def dm(G, ops, bases, scalars, masks):
        return dist.d_msm(bases, scalars, masks, pp, G, ops)

    # prove.rs:28-58
    prod = dm(G1, g1ops, [c.s for c in crs_shares], a_shares, g1_masks[0])
    pi_a = [G1.sum([fa(c.a_query0), G1.mul(fa(c.delta_g1), r), prod[i], fa(c.alpha_g1)])
            for i, c in enumerate(crs_shares)]
    # prove.rs:81-112
    if r % curve.r == 0:
        pi_b1 = [G1.identity] * n
    else:
        prod = dm(G1, g1ops, [c.h for c in crs_shares], a_shares, g1_masks[1])
        pi_b1 = [G1.sum([fa(c.b_g1_query0), G1.mul(fa(c.delta_g1), s), prod[i], fa(c.beta_g1)])
                 for i, c in enumerate(crs_shares)]
    # prove.rs:134-160
    prod = dm(G2, g2ops, [c.v for c in crs_shares], a_shares, g2_mask)
    pi_b2 = [G2.sum([fb(c.b_g2_query0), G2.mul(fb(c.delta_g2), s), prod[i], fb(c.beta_g2)])
             for i, c in enumerate(crs_shares)]
    # prove.rs:195-237
    w = dm(G1, g1ops, [c.w for c in crs_shares], ax_shares, g1_masks[2])
    u = dm(G1, g1ops, [c.u for c in crs_shares], h_shares, g1_masks[3])
    pi_c = []
    for i, c in enumerate(crs_shares):
        rsd = G1.mul(fa(c.delta_g1), r * s % curve.r)
        pi_c.append(G1.sum([G1.mul(pi_a[i], s), G1.mul(pi_b1[i], r), G1.neg(rsd), w[i], u[i]]))
    return list(zip(pi_a, pi_b2, pi_c))
