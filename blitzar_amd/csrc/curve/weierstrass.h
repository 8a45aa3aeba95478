// Short-Weierstrass curves y^2 = x^3 + b (a = 0) in homogeneous projective coordinates with the
// complete formulas of Renes-Costello-Batina 2015 (Alg. 7 add, Alg. 8 mixed add, Alg. 9 double).
// One template serves the three Weierstrass groups of the Blitzar C ABI:
//   bn254 G1      b = 3    (reference sxt/curve_bng1/operation/add.h:37-96, double.cc:43,
//                           mul_by_3b.h:32-41)
//   bls12-381 G1  b = 4    (reference sxt/curve_g1/...)
//   grumpkin      b = -17  (reference sxt/curve_gk/..., mul_by_3b.h:33-37)
// Identity is (0 : 1 : 0).  Results are only observable through canonical encodings (affine
// Montgomery limbs / 48-byte compressed), so any complete formula gives reference-identical
// output; the formulas are written from the paper.
#pragma once

#include "blitzar_amd/csrc/field/mont.h"

namespace bz {

template <int N> struct sw_point {
  fe_mont<N> X, Y, Z;
};

// affine addend; identity is flagged out of band
template <int N> struct sw_affine {
  fe_mont<N> x, y;
};

struct bn254_g1_params {
  using F = bn254_fq;
  // 3b = 9
  BZ_HD static F::fe mul_3b(const F::fe& a) {
    F::fe a8 = F::dbl(F::dbl(F::dbl(a)));
    return F::add(a8, a);
  }
};

struct bls12_381_g1_params {
  using F = bls12_381_fp;
  // 3b = 12
  BZ_HD static F::fe mul_3b(const F::fe& a) {
    F::fe a4 = F::dbl(F::dbl(a));
    return F::add(F::dbl(a4), a4);
  }
};

struct grumpkin_params {
  using F = grumpkin_fq;
  // 3b = -51 = -(32 + 16 + 2 + 1)
  BZ_HD static F::fe mul_3b(const F::fe& a) {
    F::fe a2 = F::dbl(a);
    F::fe a16 = F::dbl(F::dbl(F::dbl(a2)));
    F::fe a32 = F::dbl(a16);
    return F::neg(F::add(F::add(a32, a16), F::add(a2, a)));
  }
};

template <class C> struct sw {
  using F = typename C::F;
  using fe = typename F::fe;
  static constexpr int N = F::N;
  using point = sw_point<N>;
  using affine = sw_affine<N>;

  BZ_HD static point identity() { return {F::zero(), F::one(), F::zero()}; }

  BZ_HD static bool is_identity(const point& p) { return F::is_zero(p.Z); }

  BZ_HD static point from_affine(const affine& a, bool infinity) {
    if (infinity) return identity();
    return {a.x, a.y, F::one()};
  }

  BZ_HD static point neg(const point& p) { return {p.X, F::neg(p.Y), p.Z}; }

  BZ_HD static point cneg(const point& p, bool b) { return {p.X, F::cneg(p.Y, b), p.Z}; }

  // complete addition, 12M + 2 m3b
  BZ_HD static point add(const point& p, const point& q) {
    fe t0 = F::mul(p.X, q.X);
    fe t1 = F::mul(p.Y, q.Y);
    fe t2 = F::mul(p.Z, q.Z);
    fe t3 = F::mul(F::add(p.X, p.Y), F::add(q.X, q.Y));
    t3 = F::sub(t3, F::add(t0, t1)); // X1Y2 + X2Y1
    fe t4 = F::mul(F::add(p.Y, p.Z), F::add(q.Y, q.Z));
    t4 = F::sub(t4, F::add(t1, t2)); // Y1Z2 + Y2Z1
    fe y3 = F::mul(F::add(p.X, p.Z), F::add(q.X, q.Z));
    y3 = F::sub(y3, F::add(t0, t2)); // X1Z2 + X2Z1
    t0 = F::add(F::dbl(t0), t0);     // 3 X1X2
    t2 = C::mul_3b(t2);
    fe z3 = F::add(t1, t2);
    t1 = F::sub(t1, t2);
    y3 = C::mul_3b(y3);
    point r;
    r.X = F::sub(F::mul(t3, t1), F::mul(t4, y3));
    r.Y = F::add(F::mul(t1, z3), F::mul(y3, t0));
    r.Z = F::add(F::mul(z3, t4), F::mul(t0, t3));
    return r;
  }

  // complete mixed addition p + (x2, y2, 1), 11M + 2 m3b; q must not be the identity
  BZ_HD static point add_mixed(const point& p, const affine& q) {
    fe t0 = F::mul(p.X, q.x);
    fe t1 = F::mul(p.Y, q.y);
    fe t3 = F::mul(F::add(q.x, q.y), F::add(p.X, p.Y));
    t3 = F::sub(t3, F::add(t0, t1));
    fe t4 = F::add(F::mul(q.y, p.Z), p.Y);
    fe y3 = F::add(F::mul(q.x, p.Z), p.X);
    t0 = F::add(F::dbl(t0), t0);
    fe t2 = C::mul_3b(p.Z);
    fe z3 = F::add(t1, t2);
    t1 = F::sub(t1, t2);
    y3 = C::mul_3b(y3);
    point r;
    r.X = F::sub(F::mul(t3, t1), F::mul(t4, y3));
    r.Y = F::add(F::mul(t1, z3), F::mul(y3, t0));
    r.Z = F::add(F::mul(z3, t4), F::mul(t0, t3));
    return r;
  }

  // exception-free doubling, 6M + 2S + 1 m3b
  BZ_HD static point dbl(const point& p) {
    fe t0 = F::sqr(p.Y);
    fe z3 = F::dbl(F::dbl(F::dbl(t0))); // 8 Y^2
    fe t1 = F::mul(p.Y, p.Z);
    fe t2 = C::mul_3b(F::sqr(p.Z));
    point r;
    r.X = F::mul(t2, z3);
    r.Y = F::add(t0, t2);
    r.Z = F::mul(t1, z3);
    t1 = F::dbl(t2);
    t2 = F::add(t1, t2);
    t0 = F::sub(t0, t2);
    r.Y = F::add(F::mul(t0, r.Y), r.X);
    t1 = F::mul(p.X, p.Y);
    r.X = F::dbl(F::mul(t0, t1));
    return r;
  }

  BZ_HD static point dbl_n(point p, int k) {
    for (int i = 0; i < k; ++i) p = dbl(p);
    return p;
  }

  // canonical affine coordinates (Montgomery limbs); identity -> (0, R, infinity = true), which
  // is what the reference's to_element_affine returns (curve_bng1/type/conversion_utility.h:45-62)
  BZ_HD static bool to_affine(affine& a, const point& p) {
    const bool inf = F::is_zero(p.Z);
    fe zinv = F::invert(p.Z);
    a.x = F::mul(p.X, zinv);
    a.y = F::mul(p.Y, zinv);
    if (inf) {
      a.x = F::zero();
      a.y = F::one();
    }
    return inf;
  }

  BZ_HD static bool equal(const point& a, const point& b) {
    const bool ai = is_identity(a), bi = is_identity(b);
    if (ai || bi) return ai == bi;
    return F::equal(F::mul(a.X, b.Z), F::mul(b.X, a.Z)) &&
           F::equal(F::mul(a.Y, b.Z), F::mul(b.Y, a.Z));
  }
};

using bn254_g1 = sw<bn254_g1_params>;
using bls12_381_g1 = sw<bls12_381_g1_params>;
using grumpkin_g = sw<grumpkin_params>;

// 48-byte compressed bls12-381 G1 encoding (zkcrypto serialization; reference
// sxt/curve_g1/operation/compression.cc:34-59): big-endian x with flag bits
// 7 = compressed, 6 = infinity, 5 = y lexicographically largest.
BZ_HD void bls12_381_g1_compress_affine(u8 out[48], const bls12_381_g1::affine& a, bool inf) {
  using F = bls12_381_fp;
  F::fe x = inf ? F::zero() : F::from_mont(a.x);
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j < 8; ++j) out[8 * (5 - i) + (7 - j)] = static_cast<u8>(x.v[i] >> (8 * j));
  out[0] |= 0x80;
  if (inf) {
    out[0] |= 0x40;
    return;
  }
  // y > (p - 1) / 2  <=>  y - ((p - 1) / 2 + 1) does not borrow
  constexpr u64 half_p1[6] = {0xdcff7fffffffd556ULL, 0x0f55ffff58a9ffffULL, 0xb39869507b587b12ULL,
                              0xb23ba5c279c2895fULL, 0x258dd3db21a5d66bULL, 0x0d0088f51cbff34dULL};
  F::fe y = F::from_mont(a.y);
  u64 borrow = 0;
  for (int i = 0; i < 6; ++i) {
    u128 d = static_cast<u128>(y.v[i]) - half_p1[i] - borrow;
    borrow = static_cast<u64>(d >> 64) & 1;
  }
  if (borrow == 0) out[0] |= 0x20;
}

BZ_HD void bls12_381_g1_compress(u8 out[48], const bls12_381_g1::point& p) {
  bls12_381_g1::affine a;
  const bool inf = bls12_381_g1::to_affine(a, p);
  bls12_381_g1_compress_affine(out, a, inf);
}
} // namespace bz
