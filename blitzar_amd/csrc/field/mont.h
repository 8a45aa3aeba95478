// Prime fields in Montgomery form over N saturated 64-bit limbs -- one template for the three
// Weierstrass base fields of the Blitzar C ABI:
//   bn254 Fq    (reference sxt/field25,  constants sxt/field25/base/constants.h:39-72)
//   grumpkin Fq (reference sxt/fieldgk,  = bn254 Fr; sxt/fieldgk/base/constants.h:30-75)
//   bls12-381 Fp (reference sxt/field12, constants sxt/field12/base/constants.h:30-68)
//
// Values are always fully reduced, so the limb pattern of every result is unique and equals the
// reference's regardless of how the product/reduction is scheduled (reference: full product +
// HAC 14.32 reduction + conditional subtract, sxt/field25/operation/mul.cc:37-69,
// base/reduce.h:44-84).  Here: coarsely-integrated operand scanning (CIOS), written for N limbs.
#pragma once

#include "blitzar_amd/csrc/base/macros.h"

namespace bz {

template <int N> struct fe_mont {
  u64 v[N];
};

// P supplies: static constexpr int N; BZ_HD static constexpr u64 p(int), r(int), r2(int); inv
template <class P> struct mont {
  static constexpr int N = P::N;
  using fe = fe_mont<N>;

  BZ_HD static fe zero() {
    fe h;
#pragma unroll
    for (int i = 0; i < N; ++i) h.v[i] = 0;
    return h;
  }

  BZ_HD static fe one() {
    fe h;
#pragma unroll
    for (int i = 0; i < N; ++i) h.v[i] = P::r(i);
    return h;
  }

  BZ_HD static fe modulus() {
    fe h;
#pragma unroll
    for (int i = 0; i < N; ++i) h.v[i] = P::p(i);
    return h;
  }

  BZ_HD static bool is_zero(const fe& a) {
    u64 acc = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) acc |= a.v[i];
    return acc == 0;
  }

  BZ_HD static bool equal(const fe& a, const fe& b) {
    u64 acc = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) acc |= a.v[i] ^ b.v[i];
    return acc == 0;
  }

  // h = (t >= p) ? t - p : t, with `hi` the carry word above t
  BZ_HD static fe cond_sub_p(const u64 t[N], u64 hi) {
    u64 d[N];
    u64 borrow = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
      u128 x = static_cast<u128>(t[i]) - P::p(i) - borrow;
      d[i] = static_cast<u64>(x);
      borrow = static_cast<u64>(x >> 64) & 1;
    }
    // keep the difference when no net borrow (t + hi*2^(64N) >= p)
    const bool ge = hi != 0 || borrow == 0;
    fe h;
#pragma unroll
    for (int i = 0; i < N; ++i) h.v[i] = ge ? d[i] : t[i];
    return h;
  }

  BZ_HD static fe add(const fe& a, const fe& b) {
    u64 t[N];
    u64 carry = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
      u128 x = static_cast<u128>(a.v[i]) + b.v[i] + carry;
      t[i] = static_cast<u64>(x);
      carry = static_cast<u64>(x >> 64);
    }
    return cond_sub_p(t, carry);
  }

  BZ_HD static fe dbl(const fe& a) { return add(a, a); }

  BZ_HD static fe sub(const fe& a, const fe& b) {
    u64 t[N];
    u64 borrow = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
      u128 x = static_cast<u128>(a.v[i]) - b.v[i] - borrow;
      t[i] = static_cast<u64>(x);
      borrow = static_cast<u64>(x >> 64) & 1;
    }
    const u64 m = borrow ? ~u64{0} : 0;
    fe h;
    u64 carry = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
      u128 x = static_cast<u128>(t[i]) + (P::p(i) & m) + carry;
      h.v[i] = static_cast<u64>(x);
      carry = static_cast<u64>(x >> 64);
    }
    return h;
  }

  BZ_HD static fe neg(const fe& a) { return sub(zero(), a); }

  BZ_HD static fe cneg(const fe& a, bool b) { return b ? neg(a) : a; }

  BZ_HD static fe mul(const fe& a, const fe& b) {
    u64 t[N + 2];
#pragma unroll
    for (int i = 0; i < N + 2; ++i) t[i] = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
      u64 c = 0;
#pragma unroll
      for (int j = 0; j < N; ++j) {
        u128 x = static_cast<u128>(a.v[j]) * b.v[i] + t[j] + c;
        t[j] = static_cast<u64>(x);
        c = static_cast<u64>(x >> 64);
      }
      u128 y = static_cast<u128>(t[N]) + c;
      t[N] = static_cast<u64>(y);
      t[N + 1] = static_cast<u64>(y >> 64);

      const u64 m = t[0] * P::inv;
      u128 x = static_cast<u128>(m) * P::p(0) + t[0];
      c = static_cast<u64>(x >> 64);
#pragma unroll
      for (int j = 1; j < N; ++j) {
        x = static_cast<u128>(m) * P::p(j) + t[j] + c;
        t[j - 1] = static_cast<u64>(x);
        c = static_cast<u64>(x >> 64);
      }
      y = static_cast<u128>(t[N]) + c;
      t[N - 1] = static_cast<u64>(y);
      t[N] = t[N + 1] + static_cast<u64>(y >> 64);
    }
    return cond_sub_p(t, t[N]);
  }

  BZ_HD static fe sqr(const fe& a) { return mul(a, a); }

  // plain integer (little-endian limbs, < p) -> Montgomery form
  BZ_HD static fe to_mont(const fe& a) {
    fe r2;
#pragma unroll
    for (int i = 0; i < N; ++i) r2.v[i] = P::r2(i);
    return mul(a, r2);
  }

  // Montgomery form -> plain integer
  BZ_HD static fe from_mont(const fe& a) {
    fe u;
#pragma unroll
    for (int i = 0; i < N; ++i) u.v[i] = i == 0 ? 1 : 0;
    return mul(a, u);
  }

  // a^(p-2); returns zero for zero.  The exponent is a public constant.
  BZ_HD_NOINLINE static fe invert(const fe& a) {
    // p - 2 never borrows past limb 0 for the supported moduli (p(0) >= 2)
    fe acc = one();
    for (int i = N - 1; i >= 0; --i) {
      const u64 e = i == 0 ? P::p(0) - 2 : P::p(i);
      for (int b = 63; b >= 0; --b) {
        acc = sqr(acc);
        if ((e >> b) & 1) acc = mul(acc, a);
      }
    }
    return acc;
  }
};

//--------------------------------------------------------------------------------------------------
// field parameter packs
//--------------------------------------------------------------------------------------------------
#define BZ_LIMB_FN(name, ...)                                                                      \
  BZ_HD static constexpr u64 name(int i) {                                                        \
    constexpr u64 t[] = {__VA_ARGS__};                                                             \
    return t[i];                                                                                   \
  }

// bn254 base field Fq
struct bn254_fq_params {
  static constexpr int N = 4;
  BZ_LIMB_FN(p, 0x3c208c16d87cfd47ULL, 0x97816a916871ca8dULL, 0xb85045b68181585dULL,
             0x30644e72e131a029ULL)
  BZ_LIMB_FN(r, 0xd35d438dc58f0d9dULL, 0x0a78eb28f5c70b3dULL, 0x666ea36f7879462cULL,
             0x0e0a77c19a07df2fULL)
  BZ_LIMB_FN(r2, 0xf32cfc5b538afa89ULL, 0xb5e71911d44501fbULL, 0x47ab1eff0a417ff6ULL,
             0x06d89f71cab8351fULL)
  static constexpr u64 inv = 0x87d20782e4866389ULL;
};

// grumpkin base field = bn254 scalar field Fr
struct grumpkin_fq_params {
  static constexpr int N = 4;
  BZ_LIMB_FN(p, 0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL,
             0x30644e72e131a029ULL)
  BZ_LIMB_FN(r, 0xac96341c4ffffffbULL, 0x36fc76959f60cd29ULL, 0x666ea36f7879462eULL,
             0x0e0a77c19a07df2fULL)
  BZ_LIMB_FN(r2, 0x1bb8e645ae216da7ULL, 0x53fe3ab1e35c59e3ULL, 0x8c49833d53bb8085ULL,
             0x0216d0b17f4e44a5ULL)
  static constexpr u64 inv = 0xc2e1f593efffffffULL;
};

// bls12-381 base field Fp
struct bls12_381_fp_params {
  static constexpr int N = 6;
  BZ_LIMB_FN(p, 0xb9feffffffffaaabULL, 0x1eabfffeb153ffffULL, 0x6730d2a0f6b0f624ULL,
             0x64774b84f38512bfULL, 0x4b1ba7b6434bacd7ULL, 0x1a0111ea397fe69aULL)
  BZ_LIMB_FN(r, 0x760900000002fffdULL, 0xebf4000bc40c0002ULL, 0x5f48985753c758baULL,
             0x77ce585370525745ULL, 0x5c071a97a256ec6dULL, 0x15f65ec3fa80e493ULL)
  BZ_LIMB_FN(r2, 0xf4df1f341c341746ULL, 0x0a76e6a609d104f1ULL, 0x8de5476c4c95b6d5ULL,
             0x67eb88a9939d83c0ULL, 0x9a793e85b519952dULL, 0x11988fe592cae3aaULL)
  static constexpr u64 inv = 0x89f3fffcfffcfffdULL;
};

#undef BZ_LIMB_FN

using bn254_fq = mont<bn254_fq_params>;
using grumpkin_fq = mont<grumpkin_fq_params>;
using bls12_381_fp = mont<bls12_381_fp_params>;
} // namespace bz
