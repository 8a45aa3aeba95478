// GF(2^255 - 19) for the gfx950 MSM kernels: nine unsaturated 29-bit limbs in 32-bit registers.
//
// Why not the reference's 5 x 51-bit limbs (sxt/field51, kept in field/f51.h for everything whose
// raw limbs are observable through the C ABI)?  gfx950 has no 64 x 64 multiplier: hipcc lowers
// every 51 x 51 -> 102-bit product to four v_mad_u64_u32 plus ~8 moves / 64-bit adds (measured:
// ~92 mads and ~370 VALU instructions per field product).  v_mad_u64_u32 (32 x 32 + 64 -> 64) is
// the one wide primitive the SIMDs have (5.6 cycles per wave-instruction, tools/ubench), so the
// native representation is the one that spends mads on nothing else:
//
//   value = sum_i v[i] * 2^(29 i)  (mod p),   v[i] < 2^32 ("loose"), normally < 2^29 + eps
//
// 9 x 9 = 81 mads for the schoolbook columns, 64-bit column accumulators that cannot overflow
// (9 products < 2^58 each for inputs below ~2.4 * 2^29), the carry between columns rides for free
// in the mad addend, and the wrap 2^261 = 64 * 2^255 = 1216 (mod p) costs 10 more mads.
//
// Results of the MSM are only observable through canonical encodings (ristretto bytes), so any
// correct representation gives reference-identical output.  Conversions
// from / to the ABI's radix-2^51 limbs happen once per generator and once per commitment.
//
// Bounds contract (B(x) = max limb / 2^29):
//   mul(f, g), sq(f): need B(f) * B(g) <= 6; output limbs < 2^29 + 2^18   (B ~ 1)
//   add: B adds up; sub(f, g): needs B(g) < 1.99, output B <= B(f) + 2
#pragma once

#include "blitzar_amd/csrc/field/f51.h"

namespace bz {

struct fe29 {
  u32 v[9];
};

namespace f29 {
constexpr u32 kMask = (1u << 29) - 1;
constexpr u32 kWrap = 1216; // 2^261 mod p

BZ_HD fe29 zero() { return {{0, 0, 0, 0, 0, 0, 0, 0, 0}}; }
BZ_HD fe29 one() { return {{1, 0, 0, 0, 0, 0, 0, 0, 0}}; }

// a * b + c in one v_mad_u64_u32.  BZ_F29_MAD_MODE=1 passes the result through an empty asm
// statement so that hipcc cannot reassociate the column sums (left alone it restarts every column
// from 0 and joins the carry with an extra 64-bit add, ~4 cycles each, 112 per point addition;
// pinned, it pads the dependent mads with s_nop instead, which other waves fill).  Measured inside
// the real kernels on MI355X (A/B on one box, profiles/round2_ab_mad_mode.log): k_accumulate 0.723 ->
// 0.688 ms at 3 waves per SIMD, the lone-wave kernels mixed (k_reduce 0.203 -> 0.209, k_horner
// 0.204 -> 0.197): config 2 1.404 -> 1.364 ms, so mode 1 is the default.  (A micro-benchmark of a
// dependent chain of additions, tools/ubench/fmul_rates.hip, had favoured mode 0.)
#ifndef BZ_F29_MAD_MODE
#define BZ_F29_MAD_MODE 1
#endif
BZ_HD u64 mad(u32 a, u32 b, u64 c) {
  u64 d = static_cast<u64>(a) * b + c;
#if defined(__HIP_DEVICE_COMPILE__) && BZ_F29_MAD_MODE == 1
  asm("" : "+v"(d));
#endif
  return d;
}

// materialise the limbs here (device code): an empty volatile asm per limb keeps hipcc from sinking
// the computation of f to its uses, so whatever f was computed from is dead at this point
BZ_HD void pin(fe29& f) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
  for (int i = 0; i < 9; ++i) asm volatile("" : "+v"(f.v[i]));
#else
  (void)f;
#endif
}

BZ_HD fe29 add(const fe29& f, const fe29& g) {
  fe29 h;
#pragma unroll
  for (int i = 0; i < 9; ++i) h.v[i] = f.v[i] + g.v[i];
  return h;
}

// f - g + 2 * (2^261 - 1216): limb-wise, no borrow as long as every limb of g is below
// 2^30 - 2432, i.e. g is a mul / sq / weak_reduce output or a sum with B(g) < 1.99
BZ_HD fe29 sub(const fe29& f, const fe29& g) {
  fe29 h;
  h.v[0] = f.v[0] + ((1u << 30) - 2 * kWrap) - g.v[0];
#pragma unroll
  for (int i = 1; i < 9; ++i) h.v[i] = f.v[i] + ((1u << 30) - 2) - g.v[i];
  return h;
}

BZ_HD fe29 neg(const fe29& f) { return sub(zero(), f); }

// b ? neg(f) : f without a select: K - t = (t ^ ~0) + (K + 1) in 32-bit arithmetic, so both cases
// are (t ^ m) + c with m = b ? ~0 : 0 and c = b ? K + 1 : 0 -- one v_xad_u32 per limb
BZ_HD fe29 cneg_xad(const fe29& f, bool b) {
  const u32 m = b ? ~0u : 0u;
  const u32 c0 = b ? ((1u << 30) - 2 * kWrap) + 1 : 0u;
  const u32 c = b ? ((1u << 30) - 2) + 1 : 0u;
  fe29 h;
  h.v[0] = (f.v[0] ^ m) + c0;
#pragma unroll
  for (int i = 1; i < 9; ++i) h.v[i] = (f.v[i] ^ m) + c;
  return h;
}

// limbs back below 2^29 + eps (one carry sweep with the wrap folded into limb 0, two steps)
BZ_HD fe29 weak_reduce(const fe29& f) {
  fe29 h;
  u32 c = 0;
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    const u32 t = f.v[i] + c; // f.v[i] <= 2^32 - 2^3 keeps this in range (c < 2^3)
    h.v[i] = t & kMask;
    c = t >> 29;
  }
  const u32 t0 = h.v[0] + c * kWrap; // < 2^29 + 2^14
  h.v[0] = t0 & kMask;
  h.v[1] += t0 >> 29;
  return h;
}

// High columns 9..16 and the fold 2^261 = 1216 (mod p): every high column keeps its own 64-bit sum
// c = lo + 2^32 hi and is folded as it stands, 1216 lo into low column k and 8 * 1216 hi into low
// column k + 1 (2^32 = 8 * 2^29): 81 + 17 mads, no mask / shift on the high columns.  A low column
// receives at most 9 * 6 * 2^58 (products, B(f) B(g) <= 6) + 2^43 + 2^46 + 2^35 (carry) < 2^64.
// (Carrying the high columns into 29-bit limbs first -- 81 + 10 mads but a mask and a 64-bit shift
// per column -- measured 3 % slower at config 2, profiles/round2_ab_f29_fold_waves.log.)
BZ_HD fe29 mul(const fe29& f, const fe29& g) {
  fe29 h;
  u64 acc = 0;
  u64 c[8];
#pragma unroll
  for (int k = 9; k <= 16; ++k) {
    u64 t = 0;
#pragma unroll
    for (int i = k - 8; i <= 8; ++i) t = mad(f.v[i], g.v[k - i], t);
    c[k - 9] = t;
  }
#pragma unroll
  for (int k = 0; k <= 8; ++k) {
    if (k < 8) acc = mad(static_cast<u32>(c[k]), kWrap, acc);
    if (k > 0) acc = mad(static_cast<u32>(c[k - 1] >> 32), 8 * kWrap, acc);
#pragma unroll
    for (int i = 0; i <= k; ++i) acc = mad(f.v[i], g.v[k - i], acc);
    h.v[k] = static_cast<u32>(acc) & kMask;
    acc >>= 29;
  }
  // acc < 2^35 sits at 2^261: fold once more, then one carry step
  const u64 t0 = mad(static_cast<u32>(acc), kWrap, h.v[0]) +
                 (static_cast<u64>(static_cast<u32>(acc >> 32) * kWrap) << 32);
  h.v[0] = static_cast<u32>(t0) & kMask;
  h.v[1] += static_cast<u32>(t0 >> 29);
  return h;
}

BZ_HD fe29 sq(const fe29& f) {
  u32 d[9]; // doubled limbs (B(f) <= 2.4 keeps them in 32 bits)
#pragma unroll
  for (int i = 0; i < 9; ++i) d[i] = 2 * f.v[i];
  fe29 h;
  u64 acc = 0;
  u64 c[8];
#pragma unroll
  for (int k = 9; k <= 16; ++k) {
    u64 t = 0;
#pragma unroll
    for (int i = k - 8; 2 * i < k; ++i) t = mad(d[i], f.v[k - i], t);
    if (k % 2 == 0) t = mad(f.v[k / 2], f.v[k / 2], t);
    c[k - 9] = t;
  }
#pragma unroll
  for (int k = 0; k <= 8; ++k) {
    if (k < 8) acc = mad(static_cast<u32>(c[k]), kWrap, acc);
    if (k > 0) acc = mad(static_cast<u32>(c[k - 1] >> 32), 8 * kWrap, acc);
#pragma unroll
    for (int i = 0; 2 * i < k; ++i) acc = mad(d[i], f.v[k - i], acc);
    if (k % 2 == 0) acc = mad(f.v[k / 2], f.v[k / 2], acc);
    h.v[k] = static_cast<u32>(acc) & kMask;
    acc >>= 29;
  }
  const u64 t0 = mad(static_cast<u32>(acc), kWrap, h.v[0]) +
                 (static_cast<u64>(static_cast<u32>(acc >> 32) * kWrap) << 32);
  h.v[0] = static_cast<u32>(t0) & kMask;
  h.v[1] += static_cast<u32>(t0 >> 29);
  return h;
}

BZ_HD fe29 sqn(fe29 f, int n) {
  for (int i = 0; i < n; ++i) f = sq(f);
  return f;
}

// f * small constant (k < 2^32 / (B 2^29) ... used with k <= 4)
BZ_HD fe29 mul_small(const fe29& f, u32 k) {
  fe29 h;
#pragma unroll
  for (int i = 0; i < 9; ++i) h.v[i] = f.v[i] * k;
  return h;
}

//--------------------------------------------------------------------------------------------------
// conversions
//--------------------------------------------------------------------------------------------------
// radix-2^51 limbs (any loose fe51 the reference arithmetic produces) -> fe29
BZ_HD fe29 from_fe51(const fe51& a) {
  // tighten: every limb < 2^51 + eps, then regroup the 255-bit string
  u64 t[5];
  for (int i = 0; i < 5; ++i) t[i] = a.v[i];
  for (int i = 0; i < 4; ++i) {
    t[i + 1] += t[i] >> 51;
    t[i] &= f51::kMask;
  }
  t[0] += 19 * (t[4] >> 51);
  t[4] &= f51::kMask;
  t[1] += t[0] >> 51;
  t[0] &= f51::kMask;
  // words of the (at most 255-bit + eps) integer; t[1] may equal 2^51 + eps after the last carry
  // step, so the limbs are added, not OR-ed
  u64 w[4];
  const u128 lo = static_cast<u128>(t[0]) + (static_cast<u128>(t[1]) << 51);
  w[0] = static_cast<u64>(lo);
  const u128 mid = (lo >> 64) + (static_cast<u128>(t[2]) << 38);
  w[1] = static_cast<u64>(mid);
  const u128 hi = (mid >> 64) + (static_cast<u128>(t[3]) << 25);
  w[2] = static_cast<u64>(hi);
  w[3] = static_cast<u64>(hi >> 64) + (t[4] << 12);
  fe29 h;
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    const int bit = 29 * i;
    const int word = bit >> 6, sh = bit & 63;
    u64 v = w[word] >> sh;
    if (sh > 35 && word < 3) v |= w[word + 1] << (64 - sh);
    h.v[i] = static_cast<u32>(v) & kMask;
  }
  // limb 8 covers bits 232..260: keep whatever is there (the value is < 2^256)
  h.v[8] = static_cast<u32>(w[3] >> 40);
  return h;
}

// canonical representative as four little-endian 64-bit words
BZ_HD void to_words(u64 w[4], const fe29& f) {
  // exact carry sweep
  u64 t[9];
  u64 c = 0;
  for (int i = 0; i < 9; ++i) {
    const u64 x = static_cast<u64>(f.v[i]) + c;
    t[i] = x & kMask;
    c = x >> 29;
  }
  // bits >= 255: limb 8 holds bits 232..260 (23 bits below 255), c sits at 2^261
  u64 top = (t[8] >> 23) + (c << 6);
  t[8] &= (1u << 23) - 1;
  t[0] += 19 * top;
  for (int i = 0; i < 8; ++i) {
    t[i + 1] += t[i] >> 29;
    t[i] &= kMask;
  }
  top = t[8] >> 23;
  t[8] &= (1u << 23) - 1;
  t[0] += 19 * top; // cannot carry far: value now < 2^255 + small
  for (int i = 0; i < 8; ++i) {
    t[i + 1] += t[i] >> 29;
    t[i] &= kMask;
  }
  // now 0 <= value < 2^255 (+ at most a tiny excess already folded); conditional subtract p:
  // value >= p  <=>  value + 19 >= 2^255
  u64 s[9];
  u64 carry = 19;
  for (int i = 0; i < 9; ++i) {
    const u64 x = t[i] + carry;
    s[i] = x & kMask;
    carry = x >> 29;
  }
  const bool ge = (s[8] >> 23) != 0;
  s[8] &= (1u << 23) - 1;
  u64 r[9];
  for (int i = 0; i < 9; ++i) r[i] = ge ? s[i] : t[i];
  // pack 9 x 29 bits (last 23) into 4 words
  w[0] = r[0] | (r[1] << 29) | (r[2] << 58);
  w[1] = (r[2] >> 6) | (r[3] << 23) | (r[4] << 52);
  w[2] = (r[4] >> 12) | (r[5] << 17) | (r[6] << 46);
  w[3] = (r[6] >> 18) | (r[7] << 11) | (r[8] << 40);
}

// tight radix-2^51 limbs of the canonical value
BZ_HD fe51 to_fe51(const fe29& f) {
  u64 w[4];
  to_words(w, f);
  return f51::from_words(w);
}

BZ_HD bool is_negative(const fe29& f) {
  u64 w[4];
  to_words(w, f);
  return (w[0] & 1) != 0;
}

BZ_HD bool is_zero(const fe29& f) {
  u64 w[4];
  to_words(w, f);
  return (w[0] | w[1] | w[2] | w[3]) == 0;
}

BZ_HD void cmov(fe29& f, const fe29& g, bool b) {
#pragma unroll
  for (int i = 0; i < 9; ++i) f.v[i] = b ? g.v[i] : f.v[i];
}

BZ_HD fe29 select(const fe29& f, const fe29& g, bool pick_g) {
  fe29 h;
#pragma unroll
  for (int i = 0; i < 9; ++i) h.v[i] = pick_g ? g.v[i] : f.v[i];
  return h;
}

BZ_HD fe29 cneg(const fe29& f, bool b) { return select(f, neg(f), b); }

BZ_HD fe29 abs(const fe29& f) { return cneg(f, is_negative(f)); }

// z^(2^250 - 1), z^11
BZ_HD void pow_2_250_m1(fe29& z250, fe29& z11, const fe29& z) {
  fe29 z2 = sq(z);
  fe29 z9 = mul(z, sqn(z2, 2));
  z11 = mul(z2, z9);
  fe29 z2_5 = mul(z9, sq(z11));
  fe29 z2_10 = mul(sqn(z2_5, 5), z2_5);
  fe29 z2_20 = mul(sqn(z2_10, 10), z2_10);
  fe29 z2_40 = mul(sqn(z2_20, 20), z2_20);
  fe29 z2_50 = mul(sqn(z2_40, 10), z2_10);
  fe29 z2_100 = mul(sqn(z2_50, 50), z2_50);
  fe29 z2_200 = mul(sqn(z2_100, 100), z2_100);
  z250 = mul(sqn(z2_200, 50), z2_50);
}

BZ_HD fe29 invert(const fe29& z) {
  fe29 z250, z11;
  pow_2_250_m1(z250, z11, z);
  return mul(sqn(z250, 5), z11);
}

BZ_HD fe29 pow22523(const fe29& z) {
  fe29 z250, z11;
  pow_2_250_m1(z250, z11, z);
  return mul(sqn(z250, 2), z);
}

// curve constants, converted from the canonical radix-2^51 values of field/f51.h
BZ_HD fe29 const_2d() { return from_fe51(f51::const_2d()); }
BZ_HD fe29 const_dinv() { return from_fe51(f51::const_dinv()); }
BZ_HD fe29 const_sqrtm1() { return from_fe51(f51::const_sqrtm1()); }
BZ_HD fe29 const_invsqrtamd() { return from_fe51(f51::const_invsqrtamd()); }
} // namespace f29
} // namespace bz
