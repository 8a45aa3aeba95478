// Arithmetic modulo the group order l = 2^252 + 27742317777372353535851937790883648493 of
// curve25519 (reference: sxt/scalar25, a port of libsodium's sc25519), for the inner-product
// argument.  Scalars cross the C ABI as 32 little-endian bytes (`sxt_curve25519_scalar`); inputs
// need not be reduced, everything this code returns is the canonical representative in [0, l) --
// the same bytes the reference's s25o::mul / muladd / inv / reduce32 produce.
//
// Representation: the unsaturated-limb Montgomery arithmetic of field/mont29.h (9 x 29 bits,
// R = 2^261) instantiated for l.  `mul(u, v)` is u v / R, so:
//   * a Montgomery-form constant times a plain value is their plain product (folds),
//   * plain times plain accumulates sum a_i b_i / R and one product with R^2 finishes it
//     (inner products): no per-element conversions either way.
#pragma once

#include <cstring>

#include "blitzar_amd/csrc/field/mont29.h"

namespace bz {
using scalar25_field = mont29<scalar25_29_params>;

namespace s25 {
using F = scalar25_field;
using fe = F::fe;

// plain integer value of 32 little-endian bytes (up to 2^256 - 1: V < 16)
BZ_HD fe load(const u8* bytes) {
  u64 w[4];
  for (int k = 0; k < 4; ++k) {
    u64 v = 0;
    for (int j = 7; j >= 0; --j) v = (v << 8) | bytes[8 * k + j];
    w[k] = v;
  }
  return F::from_words(w);
}
BZ_HD fe load_words(const u64* w) { return F::from_words(w); }

// canonical words / bytes of a plain value (any normalised element)
BZ_HD void store_words(u64* w, const fe& plain) { F::to_words(w, F::canonical(plain)); }
BZ_HD void store(u8* bytes, const fe& plain) {
  u64 w[4];
  store_words(w, plain);
  for (int k = 0; k < 4; ++k)
    for (int j = 0; j < 8; ++j) bytes[8 * k + j] = static_cast<u8>(w[k] >> (8 * j));
}

BZ_HD fe r2() {
  fe c;
  for (int i = 0; i < F::N; ++i) c.v[i] = scalar25_29_params::r2(i);
  return c;
}
BZ_HD fe plain_one() {
  fe c = F::zero();
  c.v[0] = 1;
  return c;
}
BZ_HD fe to_mont(const fe& plain) { return F::mul(plain, r2()); }
BZ_HD fe from_mont(const fe& m) { return F::mul(m, plain_one()); }
// lazy sum kept in contract for the next product: normalised, V < 4
BZ_HD fe add(const fe& a, const fe& b) { return F::reduce(F::norm(F::add(a, b))); }
BZ_HD fe neg(const fe& a) { return F::reduce(F::norm(F::template neg<8>(a))); } // a: V < 8

// host-side value type: Montgomery form, normalised, V < 4
struct scalar {
  fe m;
  static scalar from_bytes(const u8* bytes) { return {to_mont(load(bytes))}; }
  void to_bytes(u8* bytes) const { store(bytes, from_mont(m)); }
  friend scalar operator*(const scalar& a, const scalar& b) { return {F::mul(a.m, b.m)}; }
  friend scalar operator+(const scalar& a, const scalar& b) { return {add(a.m, b.m)}; }
  scalar operator-() const { return {neg(m)}; }
  scalar inverse() const { return {F::invert(m)}; }
};
} // namespace s25
} // namespace bz
