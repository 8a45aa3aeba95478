// Montgomery arithmetic on unsaturated 29-bit (28-bit for BLS12-381) limbs in 32-bit registers:
// the representation the gfx950 kernels compute in for the three Weierstrass base fields.
//
// Why not the ABI's saturated 64-bit limbs (field/mont.h, kept for everything observable)?  gfx950
// has one wide multiplier primitive, v_mad_u64_u32 (32 x 32 + 64 -> 64, 5.6 cycles per
// wave-instruction).  With saturated limbs every partial product needs its carry moved by hand:
// hipcc's lowering of the 4 x 64-bit CIOS costs ~130 mads plus ~870 v_addc / v_mov / s_nop per
// product inside k_accumulate.  With LB-bit limbs (LB < 32) the 64-bit column accumulators absorb
// 2N products without overflowing, so a product is 2 N^2 mads and ~N (4 VALU) of bookkeeping:
//   bn254 / grumpkin  N = 9,  LB = 29, R = 2^261:  162 mads, ~260 instructions
//   bls12-381         N = 14, LB = 28, R = 2^392:  392 mads, ~560 instructions
// and additions are N carry-less v_add_u32 (values are reduced lazily).
//
// Values: an element x is held as an integer X = x * R mod p *plus a multiple of p*,
//   X = sum_i v[i] * 2^(LB i),   0 <= X < V p.
// Two bounds are tracked in the comments of every caller (curve/sw29.h):
//   B = max limb / 2^LB   "normalised" means B <= 1 (top limb: whatever the value needs)
//   V = X / p             P::max_v = floor(R / p) is the largest representable
// Contracts (checked at run time in host builds with BZ_MONT29_CHECK, see tests/):
//   mul(a, b):   N (B_a B_b + 1) <= 64  (i.e. B_a B_b <= 6 for N = 9, <= 3.5 for N = 14);
//                V_a V_b <= max_v^2-ish such that the result fits; result normalised,
//                V < V_a V_b / max_v + 1
//   mul2(a,b,c,d): a b + c d with one reduction; N (B_a B_b + B_c B_d + 1) <= 2^(64 - 2 LB)
//   add(a, b):   limb-wise (B and V add up)
//   sub<K>(a,b): a + K p - b with the limbs of K p inflated; needs every limb of b <= 2^(LB+1) - 2
//                and V_b < K - 0.01; result B <= B_a + 3, V < V_a + K
//   norm(a):     carry sweep, B <= 1, value unchanged
//   reduce(a):   normalised in, normalised out with V < 4
// Results only leave through canonical encodings (from / to the ABI's R = 2^(64 N64) form), so
// this representation is unobservable.
#pragma once

#include "blitzar_amd/csrc/field/mont29_params.h"

#ifndef BZ_MONT29_MAD_MODE
#define BZ_MONT29_MAD_MODE 0
#endif

#if defined(BZ_MONT29_CHECK) && !defined(__HIP_DEVICE_COMPILE__)
#include <cstdio>
#include <cstdlib>
#define BZ_M29_ASSERT(cond, what)                                                                  \
  do {                                                                                             \
    if (!(cond)) {                                                                                 \
      std::fprintf(stderr, "mont29 bound violated: %s (%s:%d)\n", what, __FILE__, __LINE__);      \
      std::abort();                                                                                \
    }                                                                                              \
  } while (0)
#else
#define BZ_M29_ASSERT(cond, what) ((void)0)
#endif

namespace bz {

template <int N> struct fe29m {
  u32 v[N];
};

template <class P> struct mont29 {
  static constexpr int N = P::N;
  static constexpr int LB = P::LB;
  static constexpr int N64 = P::N64;
  static constexpr u32 kMask = (1u << LB) - 1;
  static constexpr u32 max_v = P::max_v;
  using fe = fe29m<N>;
  using params = P;

  BZ_HD static fe zero() {
    fe h;
#pragma unroll
    for (int i = 0; i < N; ++i) h.v[i] = 0;
    return h;
  }

  BZ_HD static fe one() {
    fe h;
#pragma unroll
    for (int i = 0; i < N; ++i) h.v[i] = P::one(i);
    return h;
  }

  // materialise the limbs here (device code; see f29::pin)
  BZ_HD static void pin(fe& a) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
    for (int i = 0; i < N; ++i) asm volatile("" : "+v"(a.v[i]));
#else
    (void)a;
#endif
  }
#if defined(BZ_MONT29_CHECK) && !defined(__HIP_DEVICE_COMPILE__)
  // X < bound * p ?  (check builds only; the ratio through long doubles, exact to ~2^-60)
  static bool v_below(const fe& a, double bound) {
    long double x = 0, p = 0;
    for (int i = N - 1; i >= 0; --i) {
      x = x * static_cast<long double>(u64{1} << LB) + a.v[i];
      p = p * static_cast<long double>(u64{1} << LB) + P::p(i);
    }
    return x < static_cast<long double>(bound) * p;
  }
#endif
  BZ_HD static fe add(const fe& a, const fe& b) {
    fe h;
#pragma unroll
    for (int i = 0; i < N; ++i) {
      BZ_M29_ASSERT(static_cast<u64>(a.v[i]) + b.v[i] < (u64{1} << 32), "add overflows a limb");
      h.v[i] = a.v[i] + b.v[i];
    }
    return h;
  }

  // carry sweep: every limb but the top below 2^LB; the value is unchanged
  BZ_HD static fe norm(const fe& a) {
    fe h;
    u32 c = 0;
#pragma unroll
    for (int i = 0; i < N - 1; ++i) {
      const u64 t = static_cast<u64>(a.v[i]) + c;
      h.v[i] = static_cast<u32>(t) & kMask;
      c = static_cast<u32>(t >> LB);
    }
    BZ_M29_ASSERT(static_cast<u64>(a.v[N - 1]) + c < (u64{1} << 32), "norm overflows the top limb");
    h.v[N - 1] = a.v[N - 1] + c;
    return h;
  }

  template <int K> BZ_HD static constexpr u32 bias(int i) {
    if constexpr (K == 2) return P::bias2(i);
    if constexpr (K == 4) return P::bias4(i);
    if constexpr (K == 8) return P::bias8(i);
    if constexpr (K == 16) return P::bias16(i);
    if constexpr (K == 32) return P::bias32(i);
    if constexpr (K == 64) return P::bias64(i);
    return 0;
  }

  // a - b + K p
  template <int K> BZ_HD static fe sub(const fe& a, const fe& b) {
    fe h;
#pragma unroll
    for (int i = 0; i < N; ++i) {
      BZ_M29_ASSERT(b.v[i] <= bias<K>(i), "sub: a limb of b exceeds the bias");
      BZ_M29_ASSERT(static_cast<u64>(a.v[i]) + bias<K>(i) < (u64{1} << 32), "sub overflows a limb");
      h.v[i] = a.v[i] + bias<K>(i) - b.v[i];
    }
    return h;
  }

  template <int K = 2> BZ_HD static fe neg(const fe& a) { return sub<K>(zero(), a); }

  BZ_HD static fe select(const fe& a, const fe& b, bool pick_b) {
    fe h;
#pragma unroll
    for (int i = 0; i < N; ++i) h.v[i] = pick_b ? b.v[i] : a.v[i];
    return h;
  }

  // a * b + c in one v_mad_u64_u32; BZ_MONT29_MAD_MODE=1 pins the accumulation order (see
  // field/f29.h BZ_F29_MAD_MODE: hipcc otherwise splits the column sums and joins them with extra
  // 64-bit adds)
  BZ_HD static u64 mad(u32 a, u32 b, u64 c) {
    u64 d = static_cast<u64>(a) * b + c;
#if defined(__HIP_DEVICE_COMPILE__) && BZ_MONT29_MAD_MODE == 1
    asm("" : "+v"(d));
#endif
    return d;
  }

#if defined(BZ_MONT29_CHECK) && !defined(__HIP_DEVICE_COMPILE__)
  static void check_mul_operands(const fe& a, const fe& b) {
    u64 ma = 0, mb = 0;
    for (int i = 0; i < N; ++i) {
      if (a.v[i] > ma) ma = a.v[i];
      if (b.v[i] > mb) mb = b.v[i];
    }
    // every accumulator takes at most N products a_j b_i, N products m p_j and a carry
    const unsigned __int128 worst =
        static_cast<unsigned __int128>(N) * ma * mb +
        static_cast<unsigned __int128>(N) * (u64{1} << LB) * (u64{1} << LB) + (u64{1} << 40);
    BZ_M29_ASSERT(worst < (static_cast<unsigned __int128>(1) << 64), "mul: column accumulator overflow");
  }
  static void check_mul2_operands(const fe& a, const fe& b, const fe& c, const fe& d) {
    u64 ma = 0, mb = 0, mc = 0, md = 0;
    for (int i = 0; i < N; ++i) {
      if (a.v[i] > ma) ma = a.v[i];
      if (b.v[i] > mb) mb = b.v[i];
      if (c.v[i] > mc) mc = c.v[i];
      if (d.v[i] > md) md = d.v[i];
    }
    const unsigned __int128 worst =
        static_cast<unsigned __int128>(N) * ma * mb + static_cast<unsigned __int128>(N) * mc * md +
        static_cast<unsigned __int128>(N) * (u64{1} << LB) * (u64{1} << LB) + (u64{1} << 40);
    BZ_M29_ASSERT(worst < (static_cast<unsigned __int128>(1) << 64), "mul2: column accumulator overflow");
  }
#endif

  // Montgomery product a b / R (mod p), coarsely integrated operand scanning on 64-bit column
  // accumulators: per limb of b, N mads for a * b_i, one v_mul_lo for the quotient digit, N mads
  // for m * p, and one shift that retires the (now zero) lowest column.
  BZ_HD static fe mul(const fe& a, const fe& b) {
#if defined(BZ_MONT29_CHECK) && !defined(__HIP_DEVICE_COMPILE__)
    check_mul_operands(a, b);
#endif
    u64 t[N];
#pragma unroll
    for (int j = 0; j < N; ++j) t[j] = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
#pragma unroll
      for (int j = 0; j < N; ++j) t[j] = mad(a.v[j], b.v[i], t[j]);
      const u32 m = (static_cast<u32>(t[0]) * P::inv) & kMask;
#pragma unroll
      for (int j = 0; j < N; ++j) t[j] = mad(m, P::p(j), t[j]);
      // t[0] is now a multiple of 2^LB: retire it
      const u64 carry = t[0] >> LB;
#pragma unroll
      for (int j = 0; j < N - 1; ++j) t[j] = t[j + 1];
      t[0] += carry;
      t[N - 1] = 0;
    }
    fe h;
#pragma unroll
    for (int j = 0; j < N - 1; ++j) {
      h.v[j] = static_cast<u32>(t[j]) & kMask;
      t[j + 1] += t[j] >> LB;
    }
    BZ_M29_ASSERT(t[N - 1] < (u64{1} << LB), "mul: result does not fit (V_a V_b too large)");
    h.v[N - 1] = static_cast<u32>(t[N - 1]);
    return h;
  }

  BZ_HD static fe sqr(const fe& a) { return mul(a, a); }

  // (a b + c d) / R (mod p) with ONE Montgomery reduction: the two products share the column
  // accumulators, so the sum costs 3 N^2 mads instead of the 4 N^2 of two products.  Contract:
  // N (B_a B_b + B_c B_d + 1) <= 2^(64 - 2 LB); result normalised, V < (V_a V_b + V_c V_d) / max_v + 1.
  BZ_HD static fe mul2(const fe& a, const fe& b, const fe& c, const fe& d) {
#if defined(BZ_MONT29_CHECK) && !defined(__HIP_DEVICE_COMPILE__)
    check_mul2_operands(a, b, c, d);
#endif
    u64 t[N];
#pragma unroll
    for (int j = 0; j < N; ++j) t[j] = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
#pragma unroll
      for (int j = 0; j < N; ++j) t[j] = mad(a.v[j], b.v[i], t[j]);
#pragma unroll
      for (int j = 0; j < N; ++j) t[j] = mad(c.v[j], d.v[i], t[j]);
      const u32 m = (static_cast<u32>(t[0]) * P::inv) & kMask;
#pragma unroll
      for (int j = 0; j < N; ++j) t[j] = mad(m, P::p(j), t[j]);
      const u64 carry = t[0] >> LB;
#pragma unroll
      for (int j = 0; j < N - 1; ++j) t[j] = t[j + 1];
      t[0] += carry;
      t[N - 1] = 0;
    }
    fe h;
#pragma unroll
    for (int j = 0; j < N - 1; ++j) {
      h.v[j] = static_cast<u32>(t[j]) & kMask;
      t[j + 1] += t[j] >> LB;
    }
    BZ_M29_ASSERT(t[N - 1] < (u64{1} << LB), "mul2: result does not fit (V products too large)");
    h.v[N - 1] = static_cast<u32>(t[N - 1]);
    return h;
  }

  // The same products by PRODUCT scanning: one 64-bit accumulator walks the 2N - 1 columns, the
  // carry into the next column rides in the accumulator (no per-row carry join, no final carry
  // sweep: ~17 fewer 64-bit operations per product than the operand-scanning form above), the
  // quotient digit m_k is fixed when column k is complete.  The accumulation order is pinned
  // (an empty asm on every step) -- hipcc would otherwise split each column into independent
  // partial sums and join them again.  Same operand contract, bit-identical results.  A single
  // dependent chain: for kernels with several waves per SIMD (k_accumulate); a lone wave is
  // better served by the N independent accumulators of `mul`.
  BZ_HD static u64 mad_pinned(u32 a, u32 b, u64 c) {
    u64 d = static_cast<u64>(a) * b + c;
#if defined(__HIP_DEVICE_COMPILE__)
    asm("" : "+v"(d));
#endif
    return d;
  }

  template <bool Two>
  BZ_HD static fe mul_scan(const fe& a, const fe& b, const fe& c, const fe& d) {
#if defined(BZ_MONT29_CHECK) && !defined(__HIP_DEVICE_COMPILE__)
    if constexpr (Two) {
      check_mul2_operands(a, b, c, d);
    } else {
      check_mul_operands(a, b);
    }
#endif
    u32 m[N];
    fe h;
    u64 acc = 0;
#pragma unroll
    for (int k = 0; k < 2 * N - 1; ++k) {
#pragma unroll
      for (int i = 0; i < N; ++i) {
        const int j = k - i;
        if (j < 0 || j >= N) continue;
        acc = mad_pinned(a.v[i], b.v[j], acc);
        if constexpr (Two) acc = mad_pinned(c.v[i], d.v[j], acc);
      }
#pragma unroll
      for (int i = 0; i < N; ++i) {
        const int j = k - i;
        if (j < 0 || j >= N || i >= k) continue; // m_i exists for i < k (m_k joins below)
        acc = mad_pinned(m[i], P::p(j), acc);
      }
      if (k < N) {
        m[k] = (static_cast<u32>(acc) * P::inv) & kMask;
        acc = mad_pinned(m[k], P::p(0), acc); // the column's low LB bits are now zero
      } else {
        h.v[k - N] = static_cast<u32>(acc) & kMask;
      }
      acc >>= LB;
    }
    BZ_M29_ASSERT(acc < (u64{1} << LB), "mul_scan: result does not fit (V products too large)");
    h.v[N - 1] = static_cast<u32>(acc);
    return h;
  }
  BZ_HD static fe mul_pinned(const fe& a, const fe& b) { return mul_scan<false>(a, b, a, b); }
  BZ_HD static fe mul2_pinned(const fe& a, const fe& b, const fe& c, const fe& d) {
    return mul_scan<true>(a, b, c, d);
  }

  // a * c for a small constant c, normalised (V grows by the factor c)
  BZ_HD static fe mul_small(const fe& a, u32 c) {
    fe h;
    u64 acc = 0;
#pragma unroll
    for (int i = 0; i < N - 1; ++i) {
      acc = mad(a.v[i], c, acc);
      h.v[i] = static_cast<u32>(acc) & kMask;
      acc >>= LB;
    }
    acc = mad(a.v[N - 1], c, acc);
    BZ_M29_ASSERT(acc < (u64{1} << LB), "mul_small: result does not fit");
    h.v[N - 1] = static_cast<u32>(acc);
    return h;
  }

  // normalised a (top limb < 2^LB) -> normalised, same residue, V < 4
  BZ_HD static fe reduce(const fe& a) {
    BZ_M29_ASSERT(a.v[N - 1] < (1u << LB), "reduce: input not normalised");
    // q underestimates floor(a / p) by at most 2
    const u32 q = static_cast<u32>((static_cast<u64>(a.v[N - 1]) * P::top_magic) >> 32);
    fe h;
    i64 c = 0;
#pragma unroll
    for (int i = 0; i < N - 1; ++i) {
      const i64 t = static_cast<i64>(a.v[i]) + c - static_cast<i64>(static_cast<u64>(q) * P::p(i));
      h.v[i] = static_cast<u32>(t) & kMask;
      c = t >> LB;
    }
    const i64 t = static_cast<i64>(a.v[N - 1]) + c - static_cast<i64>(static_cast<u64>(q) * P::p(N - 1));
    BZ_M29_ASSERT(t >= 0 && t < (i64{1} << LB), "reduce: quotient estimate out of range");
    h.v[N - 1] = static_cast<u32>(t);
    return h;
  }

  // the unique representative in [0, p) (input: any normalised element)
  BZ_HD static fe canonical(const fe& a) {
    fe r = reduce(a); // < 4 p
#pragma unroll
    for (int round = 0; round < 3; ++round) {
      fe d;
      i64 c = 0;
#pragma unroll
      for (int i = 0; i < N - 1; ++i) {
        const i64 t = static_cast<i64>(r.v[i]) + c - static_cast<i64>(P::p(i));
        d.v[i] = static_cast<u32>(t) & kMask;
        c = t >> LB;
      }
      const i64 t = static_cast<i64>(r.v[N - 1]) + c - static_cast<i64>(P::p(N - 1));
      d.v[N - 1] = static_cast<u32>(t);
      r = select(r, d, t >= 0);
    }
    return r;
  }

  BZ_HD static bool is_zero(const fe& a) {
    const fe r = canonical(norm(a));
    u32 acc = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) acc |= r.v[i];
    return acc == 0;
  }

  // limbs of an integer < 2^(64 N64) given as little-endian 64-bit words
  BZ_HD static fe from_words(const u64* w) {
    fe h;
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const int bit = LB * i;
      const int word = bit >> 6, sh = bit & 63;
      u64 v = word < N64 ? w[word] >> sh : 0;
      if (sh + LB > 64 && word + 1 < N64) v |= w[word + 1] << (64 - sh);
      h.v[i] = i == N - 1 ? static_cast<u32>(v) : static_cast<u32>(v) & kMask;
    }
    return h;
  }

  // little-endian 64-bit words of a canonical (fully reduced, normalised) element
  BZ_HD static void to_words(u64* w, const fe& a) {
#pragma unroll
    for (int k = 0; k < N64; ++k) w[k] = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const int bit = LB * i;
      const int word = bit >> 6, sh = bit & 63;
      if (word < N64) w[word] |= static_cast<u64>(a.v[i]) << sh;
      if (sh + LB > 64 && word + 1 < N64) w[word + 1] |= static_cast<u64>(a.v[i]) >> (64 - sh);
    }
  }

  // ABI Montgomery form (x * 2^(64 N64) mod p, canonical 64-bit limbs) -> this representation
  BZ_HD static fe from_mont64(const u64* w) {
    fe c;
#pragma unroll
    for (int i = 0; i < N; ++i) c.v[i] = P::c_in(i);
    return mul(from_words(w), c);
  }

  // this representation -> ABI Montgomery form, canonical
  BZ_HD static void to_mont64(u64* w, const fe& a) {
    fe c;
#pragma unroll
    for (int i = 0; i < N; ++i) c.v[i] = P::c_out(i);
    to_words(w, canonical(mul(a, c)));
  }

  // 1 / a (zero for zero); `a` normalised with V < 8.
  // Kaliski's almost Montgomery inverse on the canonical integer in 64-bit words, with the shifts
  // taken in bulk: a subtraction of the two odd numbers, then every trailing zero of the
  // difference at once (count-trailing-zeros, one multi-word shift of u or v and one of s or r --
  // no halving modulo p anywhere).  ~0.7 log2(p) rounds of ~100 instructions instead of the
  // 2 log2(p) halvings + 0.7 log2(p) modular subtractions of the binary extended Euclid of rounds
  // 1-3 (bls12-381, one lane: 0.15 -> ~0.09 ms), itself ~9x fewer instructions than the
  // 1.5 log2(p) Montgomery products of a^(p-2).  It is one dependent chain either way (one
  // inversion per output column, on one lane), so instructions are what counts.
  // With A = a R the loop yields A^-1 2^k mod p (log2 p <= k <= 2 log2 p) as a plain integer; two
  // products by powers of two take the 2^k out (x 2^m / R each, m1 + m2 = 2 LB N - k), one with R^3
  // makes it a^-1 R.
  BZ_HD_NOINLINE static fe invert(const fe& a) {
    const fe ac = canonical(norm(a));
    u32 any = 0;
    for (int i = 0; i < N; ++i) any |= ac.v[i];
    if (any == 0) return zero();
    fe pf;
    for (int i = 0; i < N; ++i) pf.v[i] = P::p(i);
    u64 p[N64], u[N64], v[N64], r[N64], s[N64];
    to_words(p, pf);
    to_words(v, ac);
    for (int k = 0; k < N64; ++k) {
      u[k] = p[k];
      r[k] = 0;
      s[k] = k == 0 ? 1 : 0;
    }
    // trailing zeros of a non-zero multi-word number, at most 63 per step
    auto zeros = [](u64 w0) -> unsigned {
      if (w0 == 0) return 63;
      const unsigned z = static_cast<unsigned>(__builtin_ctzll(w0));
      return z > 63 ? 63 : z;
    };
    auto shr = [](u64* w, unsigned z) { // 1 <= z <= 63
      for (int k = 0; k < N64 - 1; ++k) w[k] = (w[k] >> z) | (w[k + 1] << (64 - z));
      w[N64 - 1] >>= z;
    };
    auto shl = [](u64* w, unsigned z) { // 1 <= z <= 63; r, s stay below 2 p < 2^(64 N64)
      for (int k = N64 - 1; k > 0; --k) w[k] = (w[k] << z) | (w[k - 1] >> (64 - z));
      w[0] <<= z;
    };
    auto sub = [](u64* w, const u64* z) { // w -= z, w >= z
      u64 borrow = 0;
      for (int k = 0; k < N64; ++k) {
        const u64 d = w[k] - z[k];
        const u64 b1 = w[k] < z[k];
        w[k] = d - borrow;
        borrow = b1 | (d < borrow);
      }
    };
    auto add = [](u64* w, const u64* z) {
      u64 carry = 0;
      for (int k = 0; k < N64; ++k) {
        const u64 t = w[k] + carry;
        const u64 c1 = t < carry;
        w[k] = t + z[k];
        carry = c1 | (w[k] < t);
      }
    };
    auto greater = [](const u64* w, const u64* z) { // w > z
      for (int k = N64 - 1; k >= 0; --k) {
        if (w[k] != z[k]) return w[k] > z[k];
      }
      return false;
    };
    auto is_zero_words = [](const u64* w) {
      u64 acc = 0;
      for (int k = 0; k < N64; ++k) acc |= w[k];
      return acc == 0;
    };
    // invariants (Kaliski 1995): p = u s + v r with u, v > 0 until the last step; r, s <= 2 p
    unsigned k = 0;
    for (;;) {
      if ((u[0] & 1) == 0) {
        const unsigned z = zeros(u[0]);
        shr(u, z);
        shl(s, z);
        k += z;
      } else if ((v[0] & 1) == 0) {
        const unsigned z = zeros(v[0]);
        shr(v, z);
        shl(r, z);
        k += z;
      } else if (greater(u, v)) {
        sub(u, v);
        add(r, s);
      } else {
        sub(v, u);
        add(s, r);
        if (is_zero_words(v)) { // u = gcd = 1: the last halving step of v, r doubles
          shl(r, 1);
          k += 1;
          break;
        }
      }
    }
    // A^-1 2^k = p - r  (r < 2 p)
    if (!greater(p, r)) sub(r, p);
    u64 x[N64];
    for (int i = 0; i < N64; ++i) x[i] = p[i];
    sub(x, r);
    const unsigned total = 2u * static_cast<unsigned>(LB * N) - k; // > 0: k <= 2 bits(p) < 2 LB N
    const unsigned m1 = total < static_cast<unsigned>(LB * N) ? total : static_cast<unsigned>(LB * N) - 1;
    const unsigned m2 = total - m1;
    auto pow2 = [](unsigned m) {
      fe h;
      for (int i = 0; i < N; ++i) {
        h.v[i] = static_cast<unsigned>(i) == m / LB ? (1u << (m % LB)) : 0u;
      }
      return h;
    };
    fe r3;
    for (int i = 0; i < N; ++i) r3.v[i] = P::r3(i);
    return mul(mul(mul(from_words(x), pow2(m1)), pow2(m2)), r3);
  }
};

using bn254_fq29 = mont29<bn254_fq29_params>;
using grumpkin_fq29 = mont29<grumpkin_fq29_params>;
using bls12_381_fp28 = mont29<bls12_381_fp28_params>;
} // namespace bz
