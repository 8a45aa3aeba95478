// GF(2^255 - 19) in radix 2^51 (five unsaturated 64-bit limbs).
//
// This is the *observable* representation of curve25519 coordinates in the Blitzar C ABI:
// `sxt_ristretto255 {X,Y,Z,T}[5]` (cbindings/blitzar_api.h:66-71) carries raw 5x51 limbs, and
// `sxt_ristretto255_get_generators` / `sxt_curve25519_get_one_commit` hand them back to callers.
// To be a byte-for-byte drop-in, every operation below produces the same limb values as the
// reference's libsodium-derived field (sxt/field51): same column sums, same partial carry chain
// r0->r1->r2->r3->r4->(x19)r0->r1->r2 after a product (sxt/field51/operation/mul.cc:57-94,
// sq.cc:36-95), limb-wise unreduced add (operation/add.h:39-52), `(f + 2p) - carry(g)`
// subtraction (operation/sub.cc:24-56) and the three-pass freeze used by to_bytes
// (base/reduce.cc:24-100, base/byte_conversion.cc:57-73).
//
// Written from the algorithm, not transliterated: one generic column-sum routine serves both
// mul and sq, carries are loops.
#pragma once

#include "blitzar_amd/csrc/base/macros.h"

namespace bz {

struct fe51 {
  u64 v[5];
};

namespace f51 {
constexpr u64 kMask = (u64{1} << 51) - 1;

BZ_HD fe51 zero() { return {{0, 0, 0, 0, 0}}; }
BZ_HD fe51 one() { return {{1, 0, 0, 0, 0}}; }

// h = f + g, limb-wise, no carry (inputs must leave headroom; same contract as the reference)
BZ_HD fe51 add(const fe51& f, const fe51& g) {
  fe51 h;
  for (int i = 0; i < 5; ++i) h.v[i] = f.v[i] + g.v[i];
  return h;
}

// h = f - g computed as (f + 2p) - carry_propagated(g)
BZ_HD fe51 sub(const fe51& f, const fe51& g) {
  u64 t[5];
  for (int i = 0; i < 5; ++i) t[i] = g.v[i];
  for (int i = 0; i < 4; ++i) {
    t[i + 1] += t[i] >> 51;
    t[i] &= kMask;
  }
  t[0] += 19 * (t[4] >> 51);
  t[4] &= kMask;
  fe51 h;
  h.v[0] = (f.v[0] + 0xfffffffffffdaULL) - t[0]; // 2p limb 0 = 2^52 - 38
  for (int i = 1; i < 5; ++i) h.v[i] = (f.v[i] + 0xffffffffffffeULL) - t[i]; // 2^52 - 2
  return h;
}

BZ_HD fe51 neg(const fe51& f) { return sub(zero(), f); }

// shared tail of mul/sq: five 128-bit column sums -> five ~51-bit limbs
BZ_HD fe51 carry_columns(u128 r0, u128 r1, u128 r2, u128 r3, u128 r4) {
  fe51 h;
  r1 += static_cast<u64>(r0 >> 51);
  r2 += static_cast<u64>(r1 >> 51);
  r3 += static_cast<u64>(r2 >> 51);
  r4 += static_cast<u64>(r3 >> 51);
  u64 h0 = static_cast<u64>(r0) & kMask;
  u64 h1 = static_cast<u64>(r1) & kMask;
  u64 h2 = static_cast<u64>(r2) & kMask;
  h.v[3] = static_cast<u64>(r3) & kMask;
  h.v[4] = static_cast<u64>(r4) & kMask;
  h0 += 19 * static_cast<u64>(r4 >> 51);
  h1 += h0 >> 51;
  h0 &= kMask;
  h2 += h1 >> 51;
  h1 &= kMask;
  h.v[0] = h0;
  h.v[1] = h1;
  h.v[2] = h2;
  return h;
}

BZ_HD fe51 mul(const fe51& f, const fe51& g) {
  const u128 f0 = f.v[0], f1 = f.v[1], f2 = f.v[2], f3 = f.v[3], f4 = f.v[4];
  const u128 g0 = g.v[0], g1 = g.v[1], g2 = g.v[2], g3 = g.v[3], g4 = g.v[4];
  // limbs that wrap past 2^255 pick up the factor 19
  const u128 f1w = 19 * f1, f2w = 19 * f2, f3w = 19 * f3, f4w = 19 * f4;
  u128 r0 = f0 * g0 + f1w * g4 + f2w * g3 + f3w * g2 + f4w * g1;
  u128 r1 = f0 * g1 + f1 * g0 + f2w * g4 + f3w * g3 + f4w * g2;
  u128 r2 = f0 * g2 + f1 * g1 + f2 * g0 + f3w * g4 + f4w * g3;
  u128 r3 = f0 * g3 + f1 * g2 + f2 * g1 + f3 * g0 + f4w * g4;
  u128 r4 = f0 * g4 + f1 * g3 + f2 * g2 + f3 * g1 + f4 * g0;
  return carry_columns(r0, r1, r2, r3, r4);
}

BZ_HD fe51 sq(const fe51& f) {
  const u128 f0 = f.v[0], f1 = f.v[1], f2 = f.v[2], f3 = f.v[3], f4 = f.v[4];
  const u128 d0 = 2 * f0, d1 = 2 * f1;
  const u128 f3w = 19 * f3, f4w = 19 * f4;
  u128 r0 = f0 * f0 + 2 * (19 * f1) * f4 + 2 * (19 * f2) * f3;
  u128 r1 = d0 * f1 + 2 * (19 * f2) * f4 + f3w * f3;
  u128 r2 = d0 * f2 + f1 * f1 + 2 * f3w * f4;
  u128 r3 = d0 * f3 + d1 * f2 + f4w * f4;
  u128 r4 = d0 * f4 + d1 * f3 + f2 * f2;
  return carry_columns(r0, r1, r2, r3, r4);
}

// 2 * f^2 (reference sq2: column sums doubled before the carry chain, sq.cc:103-170)
BZ_HD fe51 sq2(const fe51& f) {
  const u128 f0 = f.v[0], f1 = f.v[1], f2 = f.v[2], f3 = f.v[3], f4 = f.v[4];
  const u128 d0 = 2 * f0, d1 = 2 * f1;
  const u128 f3w = 19 * f3, f4w = 19 * f4;
  u128 r0 = f0 * f0 + 2 * (19 * f1) * f4 + 2 * (19 * f2) * f3;
  u128 r1 = d0 * f1 + 2 * (19 * f2) * f4 + f3w * f3;
  u128 r2 = d0 * f2 + f1 * f1 + 2 * f3w * f4;
  u128 r3 = d0 * f3 + d1 * f2 + f4w * f4;
  u128 r4 = d0 * f4 + d1 * f3 + f2 * f2;
  return carry_columns(r0 << 1, r1 << 1, r2 << 1, r3 << 1, r4 << 1);
}

BZ_HD fe51 sqn(fe51 f, int n) {
  for (int i = 0; i < n; ++i) f = sq(f);
  return f;
}

// canonical representative in [0, p)
BZ_HD fe51 freeze(const fe51& f) {
  u128 t[5];
  for (int i = 0; i < 5; ++i) t[i] = f.v[i];
  auto sweep = [&]() {
    for (int i = 0; i < 4; ++i) {
      t[i + 1] += t[i] >> 51;
      t[i] &= kMask;
    }
    t[0] += 19 * (t[4] >> 51);
    t[4] &= kMask;
  };
  sweep();
  sweep();
  // now in [0, 2^255): add 19 so that values >= p wrap past 2^255 ...
  t[0] += 19;
  sweep();
  // ... then add 2^255 - 19 and drop bit 255
  t[0] += (u64{1} << 51) - 19;
  for (int i = 1; i < 5; ++i) t[i] += (u64{1} << 51) - 1;
  for (int i = 0; i < 4; ++i) {
    t[i + 1] += t[i] >> 51;
    t[i] &= kMask;
  }
  t[4] &= kMask;
  fe51 h;
  for (int i = 0; i < 5; ++i) h.v[i] = static_cast<u64>(t[i]);
  return h;
}

BZ_HD void to_bytes(u8 s[32], const fe51& f) {
  fe51 t = freeze(f);
  u64 w[4];
  w[0] = t.v[0] | (t.v[1] << 51);
  w[1] = (t.v[1] >> 13) | (t.v[2] << 38);
  w[2] = (t.v[2] >> 26) | (t.v[3] << 25);
  w[3] = (t.v[3] >> 39) | (t.v[4] << 12);
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 8; ++j) s[8 * i + j] = static_cast<u8>(w[i] >> (8 * j));
}

// 4 x u64 little-endian words of the canonical value (device-friendly variant of to_bytes)
BZ_HD void to_words(u64 w[4], const fe51& f) {
  fe51 t = freeze(f);
  w[0] = t.v[0] | (t.v[1] << 51);
  w[1] = (t.v[1] >> 13) | (t.v[2] << 38);
  w[2] = (t.v[2] >> 26) | (t.v[3] << 25);
  w[3] = (t.v[3] >> 39) | (t.v[4] << 12);
}

// little-endian 255-bit load; bit 255 is ignored (sxt/field51/base/byte_conversion.cc:34-52)
BZ_HD fe51 from_words(const u64 w[4]) {
  fe51 h;
  h.v[0] = w[0] & kMask;
  h.v[1] = ((w[0] >> 51) | (w[1] << 13)) & kMask;
  h.v[2] = ((w[1] >> 38) | (w[2] << 26)) & kMask;
  h.v[3] = ((w[2] >> 25) | (w[3] << 39)) & kMask;
  h.v[4] = (w[3] >> 12) & kMask;
  return h;
}

BZ_HD fe51 from_bytes(const u8 s[32]) {
  u64 w[4];
  for (int i = 0; i < 4; ++i) {
    w[i] = 0;
    for (int j = 0; j < 8; ++j) w[i] |= static_cast<u64>(s[8 * i + j]) << (8 * j);
  }
  return from_words(w);
}

BZ_HD bool is_negative(const fe51& f) { return (freeze(f).v[0] & 1) != 0; }

BZ_HD bool is_zero(const fe51& f) {
  fe51 t = freeze(f);
  return (t.v[0] | t.v[1] | t.v[2] | t.v[3] | t.v[4]) == 0;
}

// f <- g when b
BZ_HD void cmov(fe51& f, const fe51& g, bool b) {
  const u64 m = b ? ~u64{0} : 0;
  for (int i = 0; i < 5; ++i) f.v[i] ^= m & (f.v[i] ^ g.v[i]);
}

BZ_HD fe51 cneg(const fe51& f, bool b) {
  fe51 h = f;
  cmov(h, neg(f), b);
  return h;
}

BZ_HD fe51 abs(const fe51& f) { return cneg(f, is_negative(f)); }

// z^(2^250 - 1) and z^11, the shared prefix of the inversion / (p-5)/8 ladders
BZ_HD void pow_2_250_m1(fe51& z250, fe51& z11, const fe51& z) {
  fe51 z2 = sq(z);
  fe51 z9 = mul(z, sqn(z2, 2));
  z11 = mul(z2, z9);
  fe51 z2_5 = mul(z9, sq(z11));                   // 2^5 - 1
  fe51 z2_10 = mul(sqn(z2_5, 5), z2_5);           // 2^10 - 1
  fe51 z2_20 = mul(sqn(z2_10, 10), z2_10);        // 2^20 - 1
  fe51 z2_40 = mul(sqn(z2_20, 20), z2_20);        // 2^40 - 1
  fe51 z2_50 = mul(sqn(z2_40, 10), z2_10);        // 2^50 - 1
  fe51 z2_100 = mul(sqn(z2_50, 50), z2_50);       // 2^100 - 1
  fe51 z2_200 = mul(sqn(z2_100, 100), z2_100);    // 2^200 - 1
  z250 = mul(sqn(z2_200, 50), z2_50);             // 2^250 - 1
}

// z^(p-2)
BZ_HD fe51 invert(const fe51& z) {
  fe51 z250, z11;
  pow_2_250_m1(z250, z11, z);
  return mul(sqn(z250, 5), z11); // 2^255 - 21
}

// z^((p-5)/8) = z^(2^252 - 3)
BZ_HD fe51 pow22523(const fe51& z) {
  fe51 z250, z11;
  pow_2_250_m1(z250, z11, z);
  return mul(sqn(z250, 2), z);
}

// curve / ristretto constants, canonical limbs (values are derived and cross-checked against the
// reference headers sxt/field51/constant/*.h by tests/test_field_constants.py)
BZ_HD fe51 const_d() {
  return {{929955233495203ULL, 466365720129213ULL, 1662059464998953ULL, 2033849074728123ULL,
           1442794654840575ULL}};
}
BZ_HD fe51 const_2d() {
  return {{1859910466990425ULL, 932731440258426ULL, 1072319116312658ULL, 1815898335770999ULL,
           633789495995903ULL}};
}
// 1 / d (the extended coordinate 2T of a cached addend is its 2dT times this)
BZ_HD fe51 const_dinv() {
  return {{266592072628291ULL, 853561038980284ULL, 1943101592401754ULL, 2007251003935334ULL,
           1135829554646364ULL}};
}
BZ_HD fe51 const_sqrtm1() {
  return {{1718705420411056ULL, 234908883556509ULL, 2233514472574048ULL, 2117202627021982ULL,
           765476049583133ULL}};
}
BZ_HD fe51 const_invsqrtamd() {
  return {{278908739862762ULL, 821645201101625ULL, 8113234426968ULL, 1777959178193151ULL,
           2118520810568447ULL}};
}
BZ_HD fe51 const_onemsqd() {
  return {{1136626929484150ULL, 1998550399581263ULL, 496427632559748ULL, 118527312129759ULL,
           45110755273534ULL}};
}
BZ_HD fe51 const_sqdmone() {
  return {{1507062230895904ULL, 1572317787530805ULL, 683053064812840ULL, 317374165784489ULL,
           1572899562415810ULL}};
}
BZ_HD fe51 const_sqrtadm1() {
  return {{2241493124984347ULL, 425987919032274ULL, 2207028919301688ULL, 1220490630685848ULL,
           974799131293748ULL}};
}
} // namespace f51
} // namespace bz
