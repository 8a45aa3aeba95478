// Short-Weierstrass curves y^2 = x^3 + b (a = 0) on the unsaturated-limb Montgomery fields of
// field/mont29.h: the group law the gfx950 MSM kernels execute for bn254 G1, grumpkin and
// bls12-381 G1.  Homogeneous projective coordinates, complete formulas of Renes-Costello-Batina
// 2015 (Alg. 7 add, Alg. 8 mixed add, Alg. 9 doubling) -- the same law as the reference's
// sxt/curve_bng1/operation/add.h:37-96 / double.cc:43, which is only observable through canonical
// encodings.  curve/weierstrass.h keeps the ABI-form (saturated 64-bit Montgomery) version used
// for encodings, the host partition-table code and as the conversion target.
//
// Bound bookkeeping (see field/mont29.h): every coordinate of a point handed between functions is
// normalised (B <= 1) with V < 6; the comments give (B, V) of intermediates for the tightest
// field, bn254 (max_v = 169, products need V_a V_b <= ~169 and B_a B_b <= 6).  Host builds with
// BZ_MONT29_CHECK assert the limb-level contracts at run time (tests/test_host_arith.py).
#pragma once

#include "blitzar_amd/csrc/curve/weierstrass.h"
#include "blitzar_amd/csrc/field/mont29.h"

namespace bz {

template <int N> struct sw29_point {
  fe29m<N> X, Y, Z;
};

// affine addend, normalised, V <= 1.01; (0, 0) marks the identity
template <int N> struct sw29_affine {
  fe29m<N> x, y;
};

// the addend as it sits in HBM: both coordinates as canonical integers of the mont29 form, packed
// into 64-bit words -- 64 bytes (bn254, grumpkin: one sector per gather instead of 72 bytes
// straddling two or three) or 96 bytes (bls12-381).  (0, 0) marks the identity.
template <int N64> struct sw29_affine_packed {
  u64 x[N64], y[N64];
};

// P: field F (mont29<...>), the ABI-form curve G64 (sw<...>), |3b| and its sign
template <class P> struct sw29 {
  using F = typename P::F;
  using fe = typename F::fe;
  using G64 = typename P::G64;
  static constexpr int N = F::N;
  static constexpr int N64 = F::N64;
  using point = sw29_point<N>;
  using affine = sw29_affine<N>;
  static constexpr bool b3_negative = P::b3_negative;
  static constexpr u32 b3_abs = P::b3_abs;

  BZ_HD static point identity() { return {F::zero(), F::one(), F::zero()}; }

  // |3b| * x: normalised, V < 4.  Input: any B <= 4, V < 8.
  BZ_HD static fe mul_b3(const fe& x) {
    if constexpr (P::b3_abs * 8 <= F::max_v) {
      return F::reduce(F::mul_small(x, P::b3_abs));
    } else {
      // grumpkin: 51 = 3 * 17, reducing in between keeps every intermediate below max_v p
      const fe r1 = F::reduce(F::norm(x));
      const fe r3 = F::reduce(F::mul_small(r1, 3));
      return F::reduce(F::mul_small(r3, 17));
    }
  }

  // the pair (t1 + 3b z, t1 - 3b z) shared by Alg. 7 and 8, u = |3b| z with V < 4, t1 V < 1.3:
  //   plus  = lazy (B 2, V < 5.3),  minus = normalised (V < 9.3)
  struct pm {
    fe plus, minus;
  };
  template <int K = 8> BZ_HD static pm plus_minus(const fe& t1, const fe& u) {
    pm r;
    r.plus = F::add(t1, u);
    r.minus = F::norm(F::template sub<K>(t1, u));
    return r;
  }

  // tail shared by Alg. 7 and 8.  In: t0 = 3 X1X2 (B 3, V < 3.7), t1 = Y1Y2 (V < 1.3),
  // u2 = |3b| Z1Z2 (V < 4), t3 = X1Y2 + X2Y1 (B 1, V < 5.9), t4 = Y1Z2 + Y2Z1 (B <= 2, V < 7.1),
  // u3 = |3b| (X1Z2 + X2Z1) (V < 4).
  // Fast: the pinned product-scanning products of field/mont29.h (k_accumulate)
  template <bool Fast>
  BZ_HD static fe prod(const fe& a, const fe& b) {
    if constexpr (Fast) {
      return F::mul_pinned(a, b);
    } else {
      return F::mul(a, b);
    }
  }
  template <bool Fast>
  BZ_HD static fe prod2(const fe& a, const fe& b, const fe& c, const fe& d) {
    if constexpr (Fast) {
      return F::mul2_pinned(a, b, c, d);
    } else {
      return F::mul2(a, b, c, d);
    }
  }

  template <bool Fast = false, int KMinus = 8>
  BZ_HD static point finish(const fe& t0, const fe& t1, const fe& u2, const fe& t3, const fe& t4,
                            const fe& u3) {
    const pm s = plus_minus<KMinus>(t1, u2);
    point r;
    // every coordinate is a sum of two products: one Montgomery reduction each (F::mul2)
    if constexpr (!P::b3_negative) {
      // z3 = t1 + 3b Z, t1' = t1 - 3b Z, y3 = 3b (...)
      const fe z3 = F::norm(s.plus);                              // B 1, V < 5.3
      const fe& t1m = s.minus;                                    // B 1, V < 9.3
      const fe t4n = F::template neg<8>(t4);                      // B <= 3, V < 8
      r.X = prod2<Fast>(t3, t1m, t4n, u3);                            // B 1*1 + 3*1, V 55 + 32
      r.Y = prod2<Fast>(t1m, z3, u3, t0);                             // B 1*1 + 1*3
      r.Z = prod2<Fast>(z3, t4, t0, t3);                              // B 1*2 + 3*1
    } else {
      // 3b = -|3b|: z3 = t1 - u2, t1' = t1 + u2, y3 = -u3
      const fe& z3 = s.minus;                                     // B 1, V < 9.3
      const fe& t1m = s.plus;                                     // B 2, V < 5.3
      const fe t0r = F::norm(t0);                                 // B 1, V < 3.7
      const fe t0n = F::template neg<4>(t0r);                     // B <= 3, V < 4
      r.X = prod2<Fast>(t3, t1m, t4, u3);                             // B 1*2 + 2*1
      r.Y = prod2<Fast>(t1m, z3, u3, t0n);                            // B 2*1 + 1*3
      r.Z = prod2<Fast>(z3, t4, t0r, t3);                             // B 1*2 + 1*1
    }
    return r;
  }

  // p + q (q negated when `negate`), Alg. 8; q must not be the identity
  template <bool Fast = false>
  BZ_HD static point add_mixed(const point& p, const affine& q, bool negate) {
    const fe y2 = F::select(q.y, F::norm(F::template neg<2>(q.y)), negate); // V <= 2
    fe t0 = prod<Fast>(p.X, q.x);                                            // V < 1.1
    const fe t1 = prod<Fast>(p.Y, y2);                                       // V < 1.1
    fe t3 = prod<Fast>(F::add(q.x, y2), F::add(p.X, p.Y));                   // B 2*2, V 3*12 -> < 1.3
    t3 = F::norm(F::template sub<4>(t3, F::add(t0, t1)));                    // V < 5.3
    const fe t4 = F::add(prod<Fast>(y2, p.Z), p.Y);                          // B 2, V < 7.1
    const fe y3 = F::add(prod<Fast>(q.x, p.Z), p.X);                         // B 2, V < 7.1
    t0 = F::add(F::add(t0, t0), t0);                                         // B 3, V < 3.3
    return finish<Fast>(t0, t1, mul_b3(p.Z), t3, t4, mul_b3(y3));
  }

  // The same addition for k_accumulate's bucket accumulators.  There p is the identity or a result
  // of this very function, so its coordinates obey a much tighter invariant than the V < 6 of the
  // general contract, (V_X, V_Y, V_Z) <= P::acc_v, and most of the partial reductions around the
  // multiplications by |3b| can go (a `reduce` is ~50 instructions at N = 9):
  //   bn254 (|3b| = 9, max_v 169), invariant (5, 4, 1.6): no reduce at all --
  //     u2 = 9 Z1 < 14.4 (so t1 - u2 takes 16 p), u3 = 9 (x2 Z1 + X1) < 54.1, and with t3 < 5.16,
  //     t4 < 5.02, 3 t0 < 3.09, t1 + u2 < 15.45, t1 - u2 + 16 p < 17.05:
  //     X3 < (5.16 * 17.05 + 8 * 54.1) / 169 + 1 = 4.08, Y3 < (17.05 * 15.45 + 54.1 * 3.09) / 169 + 1
  //     = 3.55, Z3 < (15.45 * 5.02 + 3.09 * 5.16) / 169 + 1 = 1.56;
  //   bls12-381 (|3b| = 12, max_v 2520), invariant (2, 2, 1.1): no reduce -- u2 < 13.2, u3 < 36.1,
  //     X3 < 1.15, Y3 < 1.14, Z3 < 1.03;
  //   grumpkin (|3b| = 51, max_v 169), invariant (1.5, 2, 1.5) (V_Y = 2 only on entry, for a negated
  //     affine point lifted by lift_acc): one reduce per product by 51
  //     instead of three -- 51 Z1 < 76.5 and 51 (x2 Z1 + X1) < 128.1 both fit below max_v p, so
  //     u2, u3 < 4 after one reduce: X3 < 1.21, Y3 < 1.37, Z3 < 1.23.
  // Every output also satisfies the general contract (normalised, V < 6).  Host builds with
  // BZ_MONT29_CHECK assert the invariant on entry and exit; tests/test_host_arith.py re-derives the
  // three fixed points with exact fractions.
  BZ_HD static fe mul_b3_acc(const fe& x) {
    if constexpr (P::acc_reduce_b3) {
      return F::reduce(F::mul_small(x, P::b3_abs));
    } else {
      return F::mul_small(x, P::b3_abs);
    }
  }
  BZ_HD static void check_acc_invariant(const point& p) {
    BZ_M29_ASSERT(F::v_below(p.X, P::acc_vx) && F::v_below(p.Y, P::acc_vy) &&
                      F::v_below(p.Z, P::acc_vz),
                  "add_mixed_acc: accumulator outside its invariant");
    (void)p;
  }
  template <bool Fast = false>
  BZ_HD static point add_mixed_acc(const point& p, const affine& q, bool negate) {
    check_acc_invariant(p);
    const fe y2 = F::select(q.y, F::norm(F::template neg<2>(q.y)), negate); // V <= 2
    fe t0 = prod<Fast>(p.X, q.x);
    const fe t1 = prod<Fast>(p.Y, y2);
    fe t3 = prod<Fast>(F::add(q.x, y2), F::add(p.X, p.Y));
    t3 = F::norm(F::template sub<4>(t3, F::add(t0, t1)));
    const fe t4 = F::add(prod<Fast>(y2, p.Z), p.Y);                          // B 2
    const fe y3 = F::add(prod<Fast>(q.x, p.Z), p.X);                         // B 2
    t0 = F::add(F::add(t0, t0), t0);                                         // B 3
    const point r =
        finish<Fast, P::acc_minus_k>(t0, t1, mul_b3_acc(p.Z), t3, t4, mul_b3_acc(y3));
    check_acc_invariant(r);
    return r;
  }

  // identity + q (q negated when `negate`; not the identity) as an accumulator of add_mixed_acc:
  // the affine point with Z = 1, inside the invariant (V_X <= 1, V_Y < 2, V_Z <= 1)
  BZ_HD static point lift_acc(const affine& q, bool negate) {
    const point r{q.x, F::select(q.y, F::norm(F::template neg<2>(q.y)), negate), F::one()};
    check_acc_invariant(r);
    return r;
  }

  // p + q, Alg. 7
  BZ_HD static point add(const point& p, const point& q) {
    fe t0 = F::mul(p.X, q.X);                                                // V < 1.3
    const fe t1 = F::mul(p.Y, q.Y);
    const fe t2 = F::mul(p.Z, q.Z);
    fe t3 = F::mul(F::add(p.X, p.Y), F::add(q.X, q.Y));                      // V 12*12 -> < 1.9
    t3 = F::norm(F::template sub<4>(t3, F::add(t0, t1)));                    // V < 5.9
    fe t4 = F::mul(F::add(p.Y, p.Z), F::add(q.Y, q.Z));
    t4 = F::norm(F::template sub<4>(t4, F::add(t1, t2)));                    // V < 5.9
    fe y3 = F::mul(F::add(p.X, p.Z), F::add(q.X, q.Z));
    y3 = F::norm(F::template sub<4>(y3, F::add(t0, t2)));                    // V < 5.9
    t0 = F::add(F::add(t0, t0), t0);                                         // B 3, V < 3.7
    return finish(t0, t1, mul_b3(t2), t3, t4, mul_b3(y3));
  }

  // 2p, Alg. 9
  BZ_HD static point dbl(const point& p) {
    const fe t0 = F::mul(p.Y, p.Y);                                          // V < 1.3
    const fe z3 = F::mul_small(t0, 8);                                       // V < 10
    const fe t1 = F::mul(p.Y, p.Z);
    const fe u = mul_b3(F::mul(p.Z, p.Z));                                   // |3b| Z^2, V < 4
    const fe xy = F::mul(p.X, p.Y);
    point r;
    r.Z = F::mul(t1, z3);                                                    // V < 1.1
    if constexpr (!P::b3_negative) {
      const fe y3 = F::add(t0, u);                                           // B 2, V < 5.3
      const fe u3 = F::norm(F::add(F::add(u, u), u));                        // V < 12
      const fe t0m = F::norm(F::template sub<16>(t0, u3));                   // V < 17.3
      r.Y = F::mul2(u, z3, t0m, y3);                                         // B 1*1 + 1*2, V < 1.8
      const fe x = F::mul(t0m, xy);
      r.X = F::norm(F::add(x, x));                                           // V < 2.3
    } else {
      // t2 = -u: X3' = -u z3, Y3' = t0 - u, t0' = t0 + 3u
      const fe y3 = F::norm(F::template sub<8>(t0, u));                      // V < 9.3
      const fe t0m = F::add(t0, F::add(F::add(u, u), u));                    // B 4, V < 13.3
      r.Y = F::norm(F::template sub<4>(F::mul(t0m, y3), F::mul(u, z3)));     // V < 5.8
      const fe x = F::mul(t0m, xy);
      r.X = F::norm(F::add(x, x));
    }
    return r;
  }

  BZ_HD static point dbl_n(point p, int k) {
    for (int i = 0; i < k; ++i) p = dbl(p);
    return p;
  }

  BZ_HD static point neg(const point& p) {
    return {p.X, F::reduce(F::norm(F::template neg<8>(p.Y))), p.Z};
  }

  //------------------------------------------------------------------------------------------------
  // conversions from / to the ABI form (canonical 64-bit Montgomery limbs, curve/weierstrass.h)
  //------------------------------------------------------------------------------------------------
  BZ_HD static point from_point64(const typename G64::point& p) {
    return {F::from_mont64(p.X.v), F::from_mont64(p.Y.v), F::from_mont64(p.Z.v)};
  }

  BZ_HD static typename G64::point to_point64(const point& p) {
    typename G64::point r;
    F::to_mont64(r.X.v, p.X);
    F::to_mont64(r.Y.v, p.Y);
    F::to_mont64(r.Z.v, p.Z);
    return r;
  }

  // canonical affine coordinates in ABI form; identity -> (0, R) and true, as the reference's
  // to_element_affine (curve_bng1/type/conversion_utility.h:45-62).  One inversion, done on the
  // unsaturated limbs (a^(p-2): ~1.5 N LB products).
  BZ_HD static bool to_affine64(typename G64::affine& a, const point& p) {
    const bool inf = F::is_zero(p.Z);
    const fe zinv = F::invert(p.Z);
    F::to_mont64(a.x.v, F::mul(p.X, zinv));
    F::to_mont64(a.y.v, F::mul(p.Y, zinv));
    if (inf) {
      a.x = G64::F::zero();
      a.y = G64::F::one();
    }
    return inf;
  }

  // affine ABI coordinates -> addend; identity -> (0, 0)
  BZ_HD static affine affine_from_mont64(const u64* x, const u64* y, bool infinity) {
    affine a;
    a.x = F::from_mont64(x);
    a.y = F::from_mont64(y);
    if (infinity) {
      a.x = F::zero();
      a.y = F::zero();
    }
    return a;
  }

  using packed = sw29_affine_packed<N64>;
  BZ_HD static packed pack(const affine& a) {
    packed m;
    F::to_words(m.x, F::canonical(F::norm(a.x)));
    F::to_words(m.y, F::canonical(F::norm(a.y)));
    return m;
  }
  BZ_HD static affine unpack(const packed& m) { return {F::from_words(m.x), F::from_words(m.y)}; }
  BZ_HD static bool is_identity_addend(const packed& m) {
    u64 acc = 0;
#pragma unroll
    for (int i = 0; i < N64; ++i) acc |= m.x[i] | m.y[i];
    return acc == 0;
  }

  // exact test on stored addends (the identity is written as all-zero limbs)
  BZ_HD static bool is_identity_addend(const affine& a) {
    u32 acc = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) acc |= a.x.v[i] | a.y.v[i];
    return acc == 0;
  }
};

// acc_*: the bucket-accumulation form (add_mixed_acc): reduce after the products by |3b|?, the
// multiple of p that t1 - |3b| Z1 takes, and the invariant of the accumulator's coordinates
struct bn254_g1_29_params {
  using F = bn254_fq29;
  using G64 = bn254_g1;
  static constexpr u32 b3_abs = 9;
  static constexpr bool b3_negative = false;
  static constexpr bool acc_reduce_b3 = false;
  static constexpr int acc_minus_k = 16;
  static constexpr double acc_vx = 5, acc_vy = 4, acc_vz = 1.6;
};
struct grumpkin_29_params {
  using F = grumpkin_fq29;
  using G64 = grumpkin_g;
  static constexpr u32 b3_abs = 51;
  static constexpr bool b3_negative = true;
  static constexpr bool acc_reduce_b3 = true;
  static constexpr int acc_minus_k = 8;
  static constexpr double acc_vx = 1.5, acc_vy = 2, acc_vz = 1.5;
};
struct bls12_381_g1_28_params {
  using F = bls12_381_fp28;
  using G64 = bls12_381_g1;
  static constexpr u32 b3_abs = 12;
  static constexpr bool b3_negative = false;
  static constexpr bool acc_reduce_b3 = false;
  static constexpr int acc_minus_k = 16;
  static constexpr double acc_vx = 2, acc_vy = 2, acc_vz = 1.1;
};

using bn254_g1_29 = sw29<bn254_g1_29_params>;
using grumpkin_29 = sw29<grumpkin_29_params>;
using bls12_381_g1_28 = sw29<bls12_381_g1_28_params>;
} // namespace bz
