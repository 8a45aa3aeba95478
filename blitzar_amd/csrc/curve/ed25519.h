// edwards25519 in extended twisted-Edwards coordinates (a = -1) + the ristretto255 encoding and
// the deterministic generator derivation used by Blitzar's built-in Pedersen generators.
//
// Reference behaviour restated here (file:line under /root/reference):
//   group law      sxt/curve21/operation/add.cc:41-55 (p3 + cached -> p1p1), add.h:44-51,
//                  sxt/curve21/type/double_impl.cc:43-58, conversion_utility.h (p1p1 -> p3/p2)
//   ristretto      sxt/ristretto/base/byte_conversion.cc:74-129 (encode), :134-190 (decode),
//                  sqrt_ratio_m1.cc:37-74, elligator.cc:47-93, point_formation.cc:29-35
//   generators     sxt/seqcommit/generator/base_element.cc:30-35,
//                  sxt/base/num/fast_random_number_generator.h:29-46 (xorshift128+),
//                  sxt/field51/random/element.cc:30-37
// The *sequence* of field operations in the generator path mirrors the reference because raw
// 5x51 limbs of generators are observable through `sxt_ristretto255_get_generators`
// (unreduced adds make the limb pattern depend on the operation order).  The MSM kernels are free
// to use any correct formula: their results are only observable through canonical encodings.
#pragma once

#include "blitzar_amd/csrc/field/f51.h"

namespace bz {

// (X:Y:Z:T), x = X/Z, y = Y/Z, XY = ZT.  Layout == sxt_ristretto255 (160 bytes).
struct ed_point {
  fe51 X, Y, Z, T;
};

// addend precomputed for repeated additions: (Y+X, Y-X, Z, 2dT)
struct ed_cached {
  fe51 YpX, YmX, Z, T2d;
};

// completed point ((X:Z),(Y:T))
struct ed_p1p1 {
  fe51 X, Y, Z, T;
};

namespace ed {
BZ_HD ed_point identity() { return {f51::zero(), f51::one(), f51::one(), f51::zero()}; }

BZ_HD ed_cached to_cached(const ed_point& p) {
  return {f51::add(p.Y, p.X), f51::sub(p.Y, p.X), p.Z, f51::mul(p.T, f51::const_2d())};
}

BZ_HD ed_point to_point(const ed_p1p1& c) {
  return {f51::mul(c.X, c.T), f51::mul(c.Y, c.Z), f51::mul(c.Z, c.T), f51::mul(c.X, c.Y)};
}

// p + q  (q negated when `negate`); unified: valid for doubling and the identity as well
BZ_HD ed_p1p1 add_cached(const ed_point& p, const ed_cached& q) {
  ed_p1p1 r;
  fe51 ypx = f51::add(p.Y, p.X);
  fe51 ymx = f51::sub(p.Y, p.X);
  fe51 a = f51::mul(ypx, q.YpX);
  fe51 b = f51::mul(ymx, q.YmX);
  fe51 c = f51::mul(q.T2d, p.T);
  fe51 zz = f51::mul(p.Z, q.Z);
  fe51 d = f51::add(zz, zz);
  r.X = f51::sub(a, b);
  r.Y = f51::add(a, b);
  r.Z = f51::add(d, c);
  r.T = f51::sub(d, c);
  return r;
}

BZ_HD ed_p1p1 sub_cached(const ed_point& p, const ed_cached& q) {
  ed_p1p1 r;
  fe51 ypx = f51::add(p.Y, p.X);
  fe51 ymx = f51::sub(p.Y, p.X);
  fe51 a = f51::mul(ypx, q.YmX);
  fe51 b = f51::mul(ymx, q.YpX);
  fe51 c = f51::mul(q.T2d, p.T);
  fe51 zz = f51::mul(p.Z, q.Z);
  fe51 d = f51::add(zz, zz);
  r.X = f51::sub(a, b);
  r.Y = f51::add(a, b);
  r.Z = f51::sub(d, c);
  r.T = f51::add(d, c);
  return r;
}

BZ_HD ed_point add(const ed_point& p, const ed_point& q) {
  return to_point(add_cached(p, to_cached(q)));
}

BZ_HD ed_p1p1 dbl_p1p1(const fe51& X, const fe51& Y, const fe51& Z) {
  ed_p1p1 r;
  fe51 xx = f51::sq(X);
  fe51 yy = f51::sq(Y);
  fe51 zz2 = f51::sq2(Z);
  fe51 s = f51::sq(f51::add(X, Y));
  r.Y = f51::add(yy, xx);
  r.Z = f51::sub(yy, xx);
  r.X = f51::sub(s, r.Y);
  r.T = f51::sub(zz2, r.Z);
  return r;
}

BZ_HD ed_point dbl(const ed_point& p) { return to_point(dbl_p1p1(p.X, p.Y, p.Z)); }

// 2^k * p, skipping the T coordinate on intermediate doublings
BZ_HD ed_point dbl_n(const ed_point& p, int k) {
  if (k <= 0) return p;
  fe51 X = p.X, Y = p.Y, Z = p.Z;
  for (int i = 0; i + 1 < k; ++i) {
    ed_p1p1 c = dbl_p1p1(X, Y, Z);
    X = f51::mul(c.X, c.T);
    Y = f51::mul(c.Y, c.Z);
    Z = f51::mul(c.Z, c.T);
  }
  return to_point(dbl_p1p1(X, Y, Z));
}

BZ_HD ed_point neg(const ed_point& p) { return {f51::neg(p.X), p.Y, p.Z, f51::neg(p.T)}; }

BZ_HD ed_point cneg(const ed_point& p, bool b) {
  return {f51::cneg(p.X, b), p.Y, p.Z, f51::cneg(p.T, b)};
}

// projective equality (cross-multiplied), sxt/curve21/type/element_p3.cc:56-66
BZ_HD bool equal(const ed_point& a, const ed_point& b) {
  fe51 l = f51::sub(f51::mul(a.X, b.Z), f51::mul(b.X, a.Z));
  fe51 r = f51::sub(f51::mul(a.Y, b.Z), f51::mul(b.Y, a.Z));
  return f51::is_zero(l) && f51::is_zero(r);
}
} // namespace ed

namespace ristretto {
// x = sqrt(u/v) (or sqrt(i*u/v) when u/v is a non-residue); returns whether u/v was a square
BZ_HD bool sqrt_ratio_m1(fe51& x, const fe51& u, const fe51& v) {
  const fe51 sqrtm1 = f51::const_sqrtm1();
  fe51 v3 = f51::mul(f51::sq(v), v);
  x = f51::mul(f51::mul(f51::sq(v3), u), v); // u v^7
  x = f51::pow22523(x);
  x = f51::mul(f51::mul(x, v3), u); // u v^3 (u v^7)^((p-5)/8)

  fe51 vxx = f51::mul(f51::sq(x), v);
  fe51 m_root_check = f51::sub(vxx, u);
  fe51 p_root_check = f51::add(vxx, u);
  fe51 f_root_check = f51::add(vxx, f51::mul(u, sqrtm1));
  bool has_m_root = f51::is_zero(m_root_check);
  bool has_p_root = f51::is_zero(p_root_check);
  bool has_f_root = f51::is_zero(f_root_check);
  fe51 x_sqrtm1 = f51::mul(x, sqrtm1);
  f51::cmov(x, x_sqrtm1, has_p_root | has_f_root);
  x = f51::abs(x);
  return has_m_root | has_p_root;
}

// canonical 32-byte ristretto255 encoding as four little-endian words
BZ_HD void encode_words(u64 out[4], const ed_point& p) {
  const fe51 one = f51::one();
  fe51 u1 = f51::mul(f51::add(p.Z, p.Y), f51::sub(p.Z, p.Y));
  fe51 u2 = f51::mul(p.X, p.Y);
  fe51 u1_u2u2 = f51::mul(u1, f51::sq(u2));
  fe51 inv_sqrt;
  (void)sqrt_ratio_m1(inv_sqrt, one, u1_u2u2);
  fe51 den1 = f51::mul(inv_sqrt, u1);
  fe51 den2 = f51::mul(inv_sqrt, u2);
  fe51 z_inv = f51::mul(f51::mul(den1, den2), p.T);
  fe51 ix = f51::mul(p.X, f51::const_sqrtm1());
  fe51 iy = f51::mul(p.Y, f51::const_sqrtm1());
  fe51 eden = f51::mul(den1, f51::const_invsqrtamd());
  bool rotate = f51::is_negative(f51::mul(p.T, z_inv));
  fe51 x = p.X, y = p.Y, den_inv = den2;
  f51::cmov(x, iy, rotate);
  f51::cmov(y, ix, rotate);
  f51::cmov(den_inv, eden, rotate);
  y = f51::cneg(y, f51::is_negative(f51::mul(x, z_inv)));
  fe51 s = f51::abs(f51::mul(den_inv, f51::sub(p.Z, y)));
  f51::to_words(out, s);
}

BZ_HD void encode(u8 out[32], const ed_point& p) {
  u64 w[4];
  encode_words(w, p);
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 8; ++j) out[8 * i + j] = static_cast<u8>(w[i] >> (8 * j));
}

// decode; returns false when `s` is not a canonical encoding of a ristretto255 element
BZ_HD bool decode(ed_point& p, const u8 s[32]) {
  // canonical: s < p and s even (non-negative)
  fe51 s_ = f51::from_bytes(s);
  u8 chk[32];
  f51::to_bytes(chk, s_);
  bool canonical = (s[0] & 1) == 0;
  for (int i = 0; i < 32; ++i) canonical = canonical && (chk[i] == s[i]);
  const fe51 one = f51::one();
  fe51 ss = f51::sq(s_);
  fe51 u1 = f51::sub(one, ss);
  fe51 u1u1 = f51::sq(u1);
  fe51 u2 = f51::add(one, ss);
  fe51 u2u2 = f51::sq(u2);
  fe51 v = f51::sub(f51::neg(f51::mul(f51::const_d(), u1u1)), u2u2); // -(d u1^2) - u2^2
  fe51 v_u2u2 = f51::mul(v, u2u2);
  fe51 inv_sqrt;
  bool was_square = sqrt_ratio_m1(inv_sqrt, one, v_u2u2);
  fe51 x = f51::mul(inv_sqrt, u2);
  fe51 y = f51::mul(f51::mul(inv_sqrt, x), v);
  x = f51::mul(x, s_);
  x = f51::abs(f51::add(x, x));
  y = f51::mul(u1, y);
  p.X = x;
  p.Y = y;
  p.Z = one;
  p.T = f51::mul(x, y);
  bool y_zero = f51::is_zero(y);
  bool t_neg = f51::is_negative(p.T);
  return canonical && was_square && !t_neg && !y_zero;
}

// elligator2-based map field element -> curve point (ristretto255 MAP)
BZ_HD ed_point elligator(const fe51& t) {
  const fe51 one = f51::one();
  const fe51 d = f51::const_d();
  fe51 r = f51::mul(f51::const_sqrtm1(), f51::sq(t));
  fe51 u = f51::mul(f51::add(r, one), f51::const_onemsqd());
  fe51 c = f51::neg(one);
  fe51 rpd = f51::add(r, d);
  fe51 v = f51::mul(f51::sub(c, f51::mul(r, d)), rpd);
  fe51 s;
  bool wasnt_square = !sqrt_ratio_m1(s, u, v);
  fe51 s_prime = f51::neg(f51::abs(f51::mul(s, t)));
  f51::cmov(s, s_prime, wasnt_square);
  f51::cmov(c, r, wasnt_square);
  fe51 n = f51::sub(f51::mul(f51::mul(f51::sub(r, one), c), f51::const_sqdmone()), v);
  fe51 w0 = f51::mul(f51::add(s, s), v);
  fe51 w1 = f51::mul(n, f51::const_sqrtadm1());
  fe51 ss = f51::sq(s);
  fe51 w2 = f51::sub(one, ss);
  fe51 w3 = f51::add(one, ss);
  return {f51::mul(w0, w3), f51::mul(w2, w1), f51::mul(w1, w3), f51::mul(w0, w2)};
}
} // namespace ristretto

// xorshift128+ exactly as the reference's fast_random_number_generator
struct xorshift128p {
  u64 a, b;
  BZ_HD u64 next() {
    u64 t = a;
    const u64 s = b;
    a = s;
    t ^= t << 23;
    t ^= t >> 17;
    t ^= s ^ (s >> 26);
    b = t;
    return t + s;
  }
};

namespace ed {
// built-in Pedersen generator g_index
BZ_HD ed_point base_element(u64 index) {
  xorshift128p rng{index + 1, index + 2};
  u64 w0[4], w1[4];
  for (int i = 0; i < 4; ++i) w0[i] = rng.next();
  for (int i = 0; i < 4; ++i) w1[i] = rng.next();
  fe51 r0 = f51::from_words(w0);
  fe51 r1 = f51::from_words(w1);
  ed_point p0 = ristretto::elligator(r0);
  ed_point p1 = ristretto::elligator(r1);
  return add(p1, p0);
}
} // namespace ed
} // namespace bz
