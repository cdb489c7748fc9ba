// Cold-path kernels over vectors of group elements:
//   * points_lincomb_kernel -- small fixed linear maps over points: the king's unpack2 and pack of `deg_red` when
//     T is a curve point (dist-primitives/src/utils/deg_red.rs:80-126 with T: DomainCoeff<F> = G, secret-sharing/
//     src/pss.rs:90-166 over group elements);
//   * point (de)compression of whole vectors in ark-serialize's compressed form -- what CRS shares and MSM results look
//     like when they arrive from a stock mpc-net peer (mpc-net/src/ser_net.rs:111-120): batched square roots on the
//     device instead of one Python big-integer exponentiation per point.
#pragma once
#include "ec.hpp"
#include "ntt.hpp"

namespace zk {
#if defined(__HIPCC__)

// One input group of a linear map: `count` points per chunk at in[j * cs + i * is], using coefficient columns
// off .. off + count - 1.
template <class Fld>
struct PtGroup {
  const Affine<Fld>* in;
  size_t cs, is;
  int count, off;
};

// dst[i] = a[i] + b[i] over affine points ((0,0) = identity): the mask additions of deg_red over group elements
// (deg_red.rs:94-100, :121-125 with T = G).  One mixed addition and one inversion per point (round 4 went through
// points_lincomb_kernel with the coefficient 1: a 256-step double-and-add walk per point).  dst may alias a or b.
template <class Fld>
__global__ __launch_bounds__(128) void points_add_kernel(const Affine<Fld>* a, const Affine<Fld>* b, size_t count,
                                                        Affine<Fld>* dst) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  const Affine<Fld> pa = a[i], pb = b[i];
  const bool ia = pa.x.is_zero() && pa.y.is_zero(), ib = pb.x.is_zero() && pb.y.is_zero();
  Affine<Fld> r;
  if (ia) r = pb;
  else if (ib) r = pa;
  else r = xyzz_to_affine(xyzz_madd(XYZZ<Fld>::from_affine(pa), pb.x, pb.y));
  dst[i] = r;
}

// out[r * out_rs + j * out_cs] = sum over groups, i: coef[r][off + i] * in[j, i]   (+ addend[r * add_rs + j])
// coef: canonical (non-Montgomery) scalars [rows][ncoef].  One lane per (chunk j, row r): MSB-first double-and-add
// over all inputs at once (one shared doubling chain), affine output (one inversion per output).
template <class FrP, class Fld>
__global__ __launch_bounds__(128) void points_lincomb_kernel(PtGroup<Fld> g0, PtGroup<Fld> g1, int ngroups,
                                                            const Fp<FrP>* __restrict__ coef, int ncoef, int rows,
                                                            size_t nchunks, const Affine<Fld>* __restrict__ addend,
                                                            size_t add_rs, Affine<Fld>* __restrict__ out, size_t out_rs,
                                                            size_t out_cs) {
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nchunks * (size_t)rows) return;
  const size_t j = t % nchunks;
  const int r = (int)(t / nchunks);
  const Fp<FrP>* k = coef + (size_t)r * ncoef;
  XYZZ<Fld> acc = XYZZ<Fld>::identity();
  constexpr int N = FrP::N;
  for (int w = N - 1; w >= 0; w--)
    for (int b = 31; b >= 0; b--) {
      acc = xyzz_dbl(acc);
      for (int gi = 0; gi < ngroups; gi++) {
        const PtGroup<Fld>& g = gi ? g1 : g0;
        for (int i = 0; i < g.count; i++) {
          if (!((k[g.off + i].v[w] >> b) & 1u)) continue;
          Affine<Fld> p = load_elem(g.in + j * g.cs + (size_t)i * g.is);
          if (!p.is_identity()) acc = xyzz_madd(acc, p.x, p.y);
        }
      }
    }
  if (addend) {
    Affine<Fld> p = load_elem(addend + (size_t)r * add_rs + j);
    if (!p.is_identity()) acc = xyzz_madd(acc, p.x, p.y);
  }
  store_elem(out + (size_t)r * out_rs + j * out_cs, xyzz_to_affine(acc));
}

// ---- square roots for q = 3 mod 4 (BN254, BLS12-381)
template <class P>
ZK_D bool fq_sqrt(const Fp<P>& a, Fp<P>* out) {        // a^((q+1)/4), checked
  uint32_t e[P::N];
  uint32_t carry = 1;
#pragma unroll
  for (int i = 0; i < P::N; i++) {        // q + 1
    uint64_t t = (uint64_t)P::MOD[i] + carry;
    e[i] = (uint32_t)t;
    carry = (uint32_t)(t >> 32);
  }
#pragma unroll
  for (int i = 0; i < P::N; i++) e[i] = (e[i] >> 2) | (i + 1 < P::N ? e[i + 1] << 30 : carry << 30);
  Fp<P> r = a.pow(e, P::N);
  *out = r;
  return Fp<P>::mul_ni(r, r) == a;
}
// ---- square roots for any odd q: Tonelli-Shanks (BLS12-377: q = 1 mod 4, q - 1 = 2^46 t).  The parameters are built
// on the host once per context (Engine::ts_params): z = c^t for a quadratic non-residue c (a generator of the 2^s-th
// roots of unity), e = (t - 1) / 2.
template <class P>
struct TsParams {
  Fp<P> z;
  uint32_t e[P::N];
  int s;
};
template <class P>
ZK_D bool fq_sqrt_ts(const Fp<P>& a, const TsParams<P>& ts, Fp<P>* out) {
  using F = Fp<P>;
  if (a.is_zero()) {
    *out = a;
    return true;
  }
  F w = a.pow(ts.e, P::N);            // a^((t-1)/2)
  F x = F::mul_ni(a, w);              // a^((t+1)/2)
  F b = F::mul_ni(x, w);              // a^t
  F z = ts.z;
  int v = ts.s;
  while (!(b == F::one())) {
    int k = 0;
    F b2k = b;
    while (!(b2k == F::one())) {      // least k with b^(2^k) = 1
      b2k = F::mul_ni(b2k, b2k);
      k++;
      if (k >= v) return false;       // a is not a square
    }
    F wz = z;
    for (int j = 0; j < v - k - 1; j++) wz = F::mul_ni(wz, wz);
    z = F::mul_ni(wz, wz);
    b = F::mul_ni(b, z);
    x = F::mul_ni(x, wz);
    v = k;
  }
  *out = x;
  return F::mul_ni(x, x) == a;
}

template <class F2>
ZK_D F2 f2_pow(const F2& a, const uint32_t* e, int nl) {
  F2 r = F2::one();
  bool started = false;
  for (int i = nl - 1; i >= 0; i--)
    for (int b = 31; b >= 0; b--) {
      if (started) r = F2::mul_ni(r, r);
      if ((e[i] >> b) & 1u) {
        r = started ? F2::mul_ni(r, a) : a;
        started = true;
      }
    }
  return r;
}
// Fq2 = Fq[u]/(u^2 + 1), q = 3 mod 4: Adj & Rodriguez-Henriquez, "Square root computation over even extension
// fields", Algorithm 9
template <class P, bool I>
ZK_D bool fq_sqrt(const Fp2T<P, I>& a, Fp2T<P, I>* out) {
  using F2 = Fp2T<P, I>;
  using B = Fp<P>;
  if (a.is_zero()) {
    *out = a;
    return true;
  }
  uint32_t e1[P::N], e2[P::N];          // (q - 3) / 4 and (q - 1) / 2
  {
    uint32_t borrow = 3;
    uint32_t t3[P::N];
#pragma unroll
    for (int i = 0; i < P::N; i++) {
      uint64_t t = (uint64_t)P::MOD[i] - borrow;
      t3[i] = (uint32_t)t;
      borrow = (uint32_t)(t >> 63);
    }
#pragma unroll
    for (int i = 0; i < P::N; i++) e1[i] = (t3[i] >> 2) | (i + 1 < P::N ? t3[i + 1] << 30 : 0u);
#pragma unroll
    for (int i = 0; i < P::N; i++) e2[i] = (P::MOD[i] >> 1) | (i + 1 < P::N ? P::MOD[i + 1] << 31 : 0u);   // (q-1)/2, q odd
  }
  F2 a1 = f2_pow(a, e1, P::N);
  F2 alpha = F2::mul_ni(F2::mul_ni(a1, a1), a);
  F2 conj = {alpha.c0, alpha.c1.neg()};               // alpha^q
  F2 a0 = F2::mul_ni(conj, alpha);
  F2 minus_one = {B::one().neg(), B::zero()};
  if (a0 == minus_one) return false;
  F2 x0 = F2::mul_ni(a1, a);
  F2 x;
  if (alpha == minus_one) {
    x = {x0.c1.neg(), x0.c0};                         // u * x0
  } else {
    F2 b = f2_pow(alpha + F2::one(), e2, P::N);
    x = F2::mul_ni(b, x0);
  }
  *out = x;
  return F2::mul_ni(x, x) == a;
}

// canonical "y is the lexicographically larger of (y, -y)": integers for Fq; Fq2 compares c1 first, then c0
template <class P>
ZK_D bool canon_gt(const Fp<P>& a, const Fp<P>& b) {      // both canonical (non-Montgomery)
  for (int i = P::N - 1; i >= 0; i--) {
    if (a.v[i] != b.v[i]) return a.v[i] > b.v[i];
  }
  return false;
}
template <class P>
ZK_D bool y_is_larger(const Fp<P>& y) {
  return canon_gt(y.from_mont(), y.neg().from_mont());
}
template <class P, bool I>
ZK_D bool y_is_larger(const Fp2T<P, I>& y) {
  Fp<P> a1 = y.c1.from_mont(), b1 = y.c1.neg().from_mont();
  if (a1 != b1) return canon_gt(a1, b1);
  return canon_gt(y.c0.from_mont(), y.c0.neg().from_mont());
}

// base-field element <-> its canonical bytes at `p`; ZCASH: big-endian, else little-endian.  `top_mask` clears the flag
// bits in the most significant byte on read.
template <class P>
ZK_D Fp<P> fq_from_bytes(const uint8_t* p, bool zcash, bool strip_flags, bool* ok) {
  constexpr int NB = (P::BITS + 7) / 8;
  Fp<P> r = Fp<P>::zero();
  for (int i = 0; i < NB; i++) {
    uint32_t byte = zcash ? p[NB - 1 - i] : p[i];
    if (i == NB - 1 && strip_flags) byte &= zcash ? 0x1fu : 0x3fu;
    r.v[i / 4] |= byte << (8 * (i % 4));
  }
  if (!r.is_canonical()) *ok = false;
  return r.to_mont();
}
template <class P>
ZK_D void fq_to_bytes(const Fp<P>& v, uint8_t* p, bool zcash) {
  constexpr int NB = (P::BITS + 7) / 8;
  Fp<P> c = v.from_mont();
  for (int i = 0; i < NB; i++) {
    uint8_t byte = (uint8_t)(c.v[i / 4] >> (8 * (i % 4)));
    if (zcash) p[NB - 1 - i] = byte;
    else p[i] = byte;
  }
}

template <class Fld>
struct PointCodec;
template <class P>
struct PointCodec<Fp<P>> {
  static constexpr int NB = (P::BITS + 7) / 8, SIZE = NB;
  static ZK_D Fp<P> read_x(const uint8_t* p, bool zcash, bool* ok) { return fq_from_bytes<P>(p, zcash, true, ok); }
  static ZK_D void write_x(const Fp<P>& x, uint8_t* p, bool zcash) { fq_to_bytes<P>(x, p, zcash); }
};
template <class P, bool I>
struct PointCodec<Fp2T<P, I>> {
  static constexpr int NB = (P::BITS + 7) / 8, SIZE = 2 * NB;
  static ZK_D Fp2T<P, I> read_x(const uint8_t* p, bool zcash, bool* ok) {
    // arkworks default: c0 || c1 little-endian, flags in the last byte (of c1); zcash: c1 || c0 big-endian, flags in
    // the first byte (of c1)
    Fp<P> c0 = fq_from_bytes<P>(zcash ? p + NB : p, zcash, false, ok);
    Fp<P> c1 = fq_from_bytes<P>(zcash ? p : p + NB, zcash, true, ok);
    return {c0, c1};
  }
  static ZK_D void write_x(const Fp2T<P, I>& x, uint8_t* p, bool zcash) {
    fq_to_bytes<P>(x.c0, zcash ? p + NB : p, zcash);
    fq_to_bytes<P>(x.c1, zcash ? p : p + NB, zcash);
  }
};

// bytes [len][SIZE] -> affine Montgomery points; err[0] = 1 + index of the first offending element (atomicMin-free:
// any offender is reported)
// square root as the decompression kernel calls it: the (q + 1) / 4 power when q = 3 mod 4, Tonelli-Shanks otherwise
struct NoTs {};
template <class Fld>
ZK_D bool codec_sqrt(const Fld& a, const NoTs&, Fld* out) {
  return fq_sqrt(a, out);
}
template <class P>
ZK_D bool codec_sqrt(const Fp<P>& a, const TsParams<P>& ts, Fp<P>* out) {
  return fq_sqrt_ts(a, ts, out);
}

template <class Fld, class Sqrt>
__global__ __launch_bounds__(128) void points_decompress_kernel(const uint8_t* __restrict__ bytes, size_t len, Fld b,
                                                               int zcash, Sqrt sq, Affine<Fld>* __restrict__ out,
                                                               uint32_t* __restrict__ err) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= len) return;
  using C = PointCodec<Fld>;
  const uint8_t* p = bytes + i * C::SIZE;
  const uint8_t fb = zcash ? p[0] : p[C::SIZE - 1];
  const bool inf = (fb & 0x40) != 0, larger = zcash ? (fb & 0x20) != 0 : (fb & 0x80) != 0;
  bool ok = true;
  if (zcash && !(fb & 0x80)) ok = false;                    // compression flag missing
  if (!zcash && (fb & 0xC0) == 0xC0) ok = false;            // invalid flag combination
  Fld x = C::read_x(p, zcash != 0, &ok);
  Affine<Fld> r{Fld::zero(), Fld::zero()};
  if (inf) {
    if (!x.is_zero() || (zcash && larger)) ok = false;
  } else if (ok) {
    Fld y;
    Fld rhs = x.sqr() * x + b;
    if (!codec_sqrt(rhs, sq, &y)) ok = false;
    else {
      if (y_is_larger(y) != larger) y = y.neg();
      r = {x, y};
    }
  }
  if (!ok) atomicMax(err, (uint32_t)(i + 1));
  store_elem(out + i, r);
}

template <class Fld>
__global__ __launch_bounds__(128) void points_compress_kernel(const Affine<Fld>* __restrict__ pts, size_t len, int zcash,
                                                             uint8_t* __restrict__ bytes) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= len) return;
  using C = PointCodec<Fld>;
  uint8_t* p = bytes + i * C::SIZE;
  Affine<Fld> a = load_elem(pts + i);
  if (a.is_identity()) {
    for (int k = 0; k < C::SIZE; k++) p[k] = 0;
    if (zcash) p[0] = 0xC0;
    else p[C::SIZE - 1] = 0x40;
    return;
  }
  C::write_x(a.x, p, zcash != 0);
  const bool larger = y_is_larger(a.y);
  if (zcash) p[0] |= 0x80 | (larger ? 0x20 : 0);
  else if (larger) p[C::SIZE - 1] |= 0x80;
}

#endif  // __HIPCC__
}  // namespace zk
