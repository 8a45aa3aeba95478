// Latency-oriented arithmetic for the Weierstrass curves (bn254 G1, grumpkin, bls12-381 G1) on the one
// dependent chain at the end of an MSM -- Horner over the windows of a column: c doublings and one
// addition per window that nothing can run beside.  ONE projective point spread over a whole
// wavefront, the counterpart of curve/ed16_wave.h for the Montgomery fields.
//
// Layout: the wavefront is 4 DPP rows of 16 lanes.  A field element keeps the limbs of the engine's
// unsaturated form (field/mont29.h: N limbs of LB bits) and gets one more: limb j in lane j of a row,
// NW = N + 1 lanes in use (bn254 / grumpkin 10 x 29 bits, bls12-381 15 x 28 bits).  The extra limb
// makes the Montgomery radix Rw = 2^(LB NW) 2^36 .. 2^39 times the modulus, so values may grow to
// thousands of p between products -- sums, limb-wise subtractions against a multiple of p, the
// multiplications by |3b| = 9 / 51 / 12 -- and every product still returns less than 1.001 p: no
// reduction other than the products themselves.  A point of the engine enters limb for limb: its
// coordinates a R29 read as (a R29 / Rw) Rw are all scaled by the same factor, which a projective
// point does not notice (the formulas are homogeneous in each operand and use no constant but the
// integer |3b|).
//
// A product u v / Rw mod p is computed by the lanes of a row together, lane j owning columns j and
// j + NW of u v (u: NW limbs re-read from LDS as broadcast 128-bit loads; v: this lane's limb, moved
// along the row by zero-filling DPP shifts, one v_mov_b32_dpp per term), then the Montgomery
// reduction in the same shape: q = (low half) (-1/p) mod Rw as NW more columns, q p as 2 NW, and the
// one carry that has to cross from the low half into the high one read off the two top lanes of the
// low half (it is an exact multiple of Rw).  bls12-381: 71 v_mad_u64_u32 and ~130 other
// instructions per lane instead of the 392 + 170 of a lone lane.  The four rows run four products at
// once: a doubling (Renes-Costello-Batina 2015, Alg. 9) is two rounds, a complete addition (Alg. 7)
// three; between rounds every lane fetches limb j of the four results with one 128-bit LDS load.
//
// The lane program and its bounds -- 32-bit limbs, 32-bit carries, 64-bit column sums, the limb-wise
// multiple of p -- are restated and checked by tools/models/sw_wave_model.py (run by
// tests/test_sw_wave_model.py); names here follow the model.
//
// Device only, gfx950.
#pragma once

#include <utility>

#include "blitzar_amd/csrc/curve/sw29.h"

#if defined(__HIPCC__)
namespace bz {
namespace sww {

// LDS scratch of the wavefront running the chain
struct scratch {
  alignas(16) u32 bcast[64]; // [row][limb]: the u operands of the products in flight
  alignas(16) u32 xch[64];   // [limb][row]: results of a round, read back as one 128-bit load
  alignas(16) u32 io[64];    // [row][limb]: hand-over of a point held in registers
};

// orders this wavefront's LDS traffic (lanes read what other lanes of the wave wrote)
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

template <int N, class F> __device__ __forceinline__ void unroll(F&& f) {
  [&]<int... K>(std::integer_sequence<int, K...>) {
    (f(std::integral_constant<int, K>{}), ...);
  }(std::make_integer_sequence<int, N>{});
}

// DPP row shifts, zero filled.  shr<I>: lane j receives lane j - I; shl<I>: lane j receives lane j + I
// (lanes without a source lane in their row get 0).  The empty asm statement keeps the move a plain
// v_mov_b32_dpp: hipcc's folding of DPP moves into VOP2 instructions miscomputed in
// curve/sw29_coop.h.
template <int I> __device__ __forceinline__ u32 shr(u32 x) {
  static_assert(I >= 1 && I <= 15);
  u32 v = static_cast<u32>(
      __builtin_amdgcn_update_dpp(0, static_cast<int>(x), 0x110 + I, 0xf, 0xf, true));
  asm volatile("" : "+v"(v));
  return v;
}
template <int I> __device__ __forceinline__ u32 shl(u32 x) {
  static_assert(I >= 1 && I <= 15);
  u32 v = static_cast<u32>(
      __builtin_amdgcn_update_dpp(0, static_cast<int>(x), 0x100 + I, 0xf, 0xf, true));
  asm volatile("" : "+v"(v));
  return v;
}
template <int I> __device__ __forceinline__ u64 shr64(u64 x) {
  return static_cast<u64>(shr<I>(static_cast<u32>(x))) |
         (static_cast<u64>(shr<I>(static_cast<u32>(x >> 32))) << 32);
}
template <int I> __device__ __forceinline__ u64 shl64(u64 x) {
  return static_cast<u64>(shl<I>(static_cast<u32>(x))) |
         (static_cast<u64>(shl<I>(static_cast<u32>(x >> 32))) << 32);
}

// G: sw29<...> (curve/sw29.h)
template <class G> struct wave {
  using F = typename G::F;
  using P = typename F::params;
  using point = typename G::point;
  static constexpr int N = F::N;
  static constexpr int NW = N + 1;
  static constexpr int LB = F::LB;
  static constexpr u32 kMask = (1u << LB) - 1;
  static_assert(NW <= 16 && 3 * LB >= 64, "carry3 splits a 64-bit column into three pieces");

  struct ctx {
    u32 lane, row, j;
    u32 live;  // all ones in lanes j < NW, zero above: operands must be zero there (shl moves them down)
    u32 top;   // all ones in lane NW - 1
    u32 bias;  // limb j of the multiple of p that subtractions add (P::wave_bias)
    scratch* lds;
  };

  __device__ static __forceinline__ ctx make_ctx(scratch* lds) {
    ctx c;
    c.lane = threadIdx.x & 63;
    c.row = c.lane >> 4;
    c.j = c.lane & 15;
    c.live = c.j < NW ? ~0u : 0u;
    c.top = c.j == NW - 1 ? ~0u : 0u;
    u32 b = 0;
    unroll<NW>([&](auto k) {
      constexpr int i = decltype(k)::value;
      b = c.j == static_cast<u32>(i) ? P::wave_bias(i) : b;
    });
    c.bias = b;
    c.lds = lds;
    return c;
  }

  __device__ static __forceinline__ u32 by_row(const ctx& c, u32 r0, u32 r1, u32 r2, u32 r3) {
    const u32 lo = (c.row & 1) ? r1 : r0;
    const u32 hi = (c.row & 1) ? r3 : r2;
    return (c.row & 2) ? hi : lo;
  }

  // ---- carries (model: carry3 / carry1 / scale / sub) ----
  // 64-bit column -> limb < 2^LB + 2; what leaves lane NW - 1 is dropped
  __device__ static __forceinline__ u32 carry3(u64 x) {
    const u32 m0 = static_cast<u32>(x) & kMask;
    const u32 m1 = static_cast<u32>(x >> LB) & kMask;
    const u32 m2 = static_cast<u32>(x >> (2 * LB));
    const u32 a1 = m0 + shr<1>(m1) + shr<2>(m2);
    return (a1 & kMask) + shr<1>(a1 >> LB);
  }
  __device__ static __forceinline__ u32 carry1(u32 x) { return (x & kMask) + shr<1>(x >> LB); }
  // c x, carried: limb < 2^LB + c + 1
  template <u32 C> __device__ static __forceinline__ u32 scale(u32 x) {
    const u64 wide = static_cast<u64>(x) * C;
    return (static_cast<u32>(wide) & kMask) + shr<1>(static_cast<u32>(wide >> LB));
  }
  // a - b + (multiple of p), b a sum of at most three carried elements
  __device__ static __forceinline__ u32 sub(const ctx& c, u32 a, u32 b) { return a + c.bias - b; }

  // ---- the product (model: Wave.mul) ----
  // ui: the NW limbs of u (any lane may hold any row's); v: this lane's limb of the other operand
  __device__ static __forceinline__ u32 mul(const ctx& c, const u32 (&ui)[NW], u32 v_in) {
    const u32 v = v_in & c.live;
    u64 lo = static_cast<u64>(ui[0]) * v, hi = 0;
    unroll<NW - 1>([&](auto k) {
      constexpr int i = decltype(k)::value + 1;
      lo += static_cast<u64>(ui[i]) * shr<i>(v);
      hi += static_cast<u64>(ui[i]) * shl<NW - i>(v);
    });
    const u32 l = carry3(lo);
    u64 d = static_cast<u64>(P::wave_ninv(0)) * l;
    unroll<NW - 1>([&](auto k) {
      constexpr int i = decltype(k)::value + 1;
      d += static_cast<u64>(P::wave_ninv(i)) * shr<i>(l);
    });
    const u32 q = carry3(d) & c.live;
    u64 lo2 = static_cast<u64>(P::p(0)) * q, hi2 = 0;
    unroll<N - 1>([&](auto k) {
      constexpr int i = decltype(k)::value + 1;
      lo2 += static_cast<u64>(P::p(i)) * shr<i>(q);
      hi2 += static_cast<u64>(P::p(i)) * shl<NW - i>(q);
    });
    // the low half is k Rw: k from its two top lanes (meaningful in lane NW - 1 only)
    const u64 s = lo + lo2;
    const u64 sp = shr64<1>(s);
    const u64 t = s + (sp >> LB);
    const u32 frac = (static_cast<u32>(t) | static_cast<u32>(sp)) & kMask;
    const u64 kq = (t >> LB) + (frac != 0 ? 1u : 0u);
    const u64 kk = kq & (static_cast<u64>(c.top) | (static_cast<u64>(c.top) << 32));
    return carry3(hi + hi2 + shl64<NW - 1>(kk));
  }

  // the NW limbs of row `src_row`, as last written to lds->bcast
  __device__ static __forceinline__ void broadcast_load(const ctx& c, u32 src_row, u32 (&ui)[NW]) {
    const uint4* up = reinterpret_cast<const uint4*>(&c.lds->bcast[src_row * 16]);
    const uint4 q0 = up[0], q1 = up[1], q2 = up[2], q3 = up[3];
    const u32 all[16] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w,
                         q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w};
#pragma unroll
    for (int i = 0; i < NW; ++i) ui[i] = all[i];
  }

  // per row: u * v / Rw
  __device__ static __forceinline__ u32 fmul(const ctx& c, u32 u, u32 v) {
    c.lds->bcast[c.lane] = u;
    wave_lds_sync();
    u32 ui[NW];
    broadcast_load(c, c.row, ui);
    return mul(c, ui, v);
  }

  // limb j of the four rows' values
  __device__ static __forceinline__ uint4 exchange(const ctx& c, u32 v) {
    c.lds->xch[c.j * 4 + c.row] = v;
    wave_lds_sync();
    return *reinterpret_cast<const uint4*>(&c.lds->xch[c.j * 4]);
  }

  // ---- group law: state = this lane's limb of (X | Y | Z | - by row), carried ----
  // 2 P, Alg. 9 with a = 0 (model: dbl)
  __device__ static __forceinline__ u32 dbl(const ctx& c, u32 st) {
    // round 1: Y Y | Y Z | Z Z | X Y
    c.lds->bcast[c.lane] = st;
    wave_lds_sync();
    u32 ui[NW];
    broadcast_load(c, c.row == 3 ? 0u : (c.row == 2 ? 2u : 1u), ui);
    const u32 y = c.lds->bcast[16 + c.j], z = c.lds->bcast[32 + c.j];
    const uint4 m = exchange(c, mul(c, ui, by_row(c, y, z, z, y)));
    const u32 t0 = m.x, t1 = m.y, zz = m.z, xy = m.w;
    const u32 z3 = scale<8>(t0);
    const u32 ub = scale<G::b3_abs>(zz);
    const u32 ub3 = scale<3 * G::b3_abs>(zz);
    u32 y3, t0m;
    if constexpr (!G::b3_negative) {
      y3 = t0 + ub;
      t0m = carry1(sub(c, t0, ub3));
    } else {
      y3 = carry1(sub(c, t0, ub));
      t0m = t0 + ub3;
    }
    // round 2: t1 z3 (Z3) | u z3 | t0m y3 | t0m 2xy (X3)
    const uint4 h = exchange(c, fmul(c, by_row(c, t1, ub, t0m, t0m), by_row(c, z3, z3, y3, 2 * xy)));
    u32 ny;
    if constexpr (!G::b3_negative) {
      ny = carry1(h.y + h.z);
    } else {
      ny = carry1(sub(c, h.z, h.y));
    }
    return by_row(c, h.w, ny, h.x, h.x);
  }

  // P + Q, complete (Alg. 7 with a = 0; model: add); q: Q in the same layout
  __device__ static __forceinline__ u32 add(const ctx& c, u32 st, u32 q) {
    const uint4 a = exchange(c, st); // X1 Y1 Z1
    const uint4 b = exchange(c, q);  // X2 Y2 Z2
    // round 1: X1 X2 | Y1 Y2 | Z1 Z2 | (X1 + Y1)(X2 + Y2)
    const uint4 r1 = exchange(c, fmul(c, by_row(c, a.x, a.y, a.z, a.x + a.y),
                                      by_row(c, b.x, b.y, b.z, b.x + b.y)));
    const u32 t0 = r1.x, t1 = r1.y, t2 = r1.z;
    const u32 ub = scale<G::b3_abs>(t2);
    const u32 t00 = 3 * t0;
    const u32 t3 = carry1(sub(c, r1.w, t0 + t1));
    u32 z3, t1m;
    if constexpr (!G::b3_negative) {
      z3 = t1 + ub;
      t1m = carry1(sub(c, t1, ub));
    } else {
      z3 = carry1(sub(c, t1, ub));
      t1m = t1 + ub;
    }
    // round 2: (Y1 + Z1)(Y2 + Z2) | (X1 + Z1)(X2 + Z2) | t1m z3 | 3 t0 t3
    const uint4 r2 = exchange(c, fmul(c, by_row(c, a.y + a.z, a.x + a.z, t1m, t00),
                                      by_row(c, b.y + b.z, b.x + b.z, z3, t3)));
    const u32 t4 = carry1(sub(c, r2.x, t1 + t2));
    const u32 t5 = carry1(sub(c, r2.y, t0 + t2));
    const u32 y3 = scale<G::b3_abs>(t5);
    // round 3: t3 t1m | t4 y3 | y3 3t0 | z3 t4
    const uint4 r3 = exchange(c, fmul(c, by_row(c, t3, t4, y3, z3), by_row(c, t1m, y3, t00, t4)));
    u32 nx, ny;
    if constexpr (!G::b3_negative) {
      nx = carry1(sub(c, r3.x, r3.y));
      ny = carry1(r2.z + r3.z);
    } else {
      nx = carry1(r3.x + r3.y);
      ny = carry1(sub(c, r2.z, r3.z));
    }
    const u32 nz = carry1(r3.w + r2.w);
    return by_row(c, nx, ny, nz, nz);
  }

  // ---- hand-over ----
  // a point of the engine (N limbs per coordinate, in LDS) -> this lane's limb
  __device__ static __forceinline__ u32 load_point(const ctx& c, const point* p) {
    const u32* words = reinterpret_cast<const u32*>(p);
    return (c.row < 3 && c.j < N) ? words[c.row * N + c.j] : 0u;
  }
  // the same for a point held in registers (every lane has it)
  __device__ static __forceinline__ u32 load_point_value(const ctx& c, const point& p) {
    if (c.lane == 0) {
#pragma unroll
      for (int i = 0; i < N; ++i) {
        c.lds->io[i] = p.X.v[i];
        c.lds->io[16 + i] = p.Y.v[i];
        c.lds->io[32 + i] = p.Z.v[i];
      }
    }
    wave_lds_sync();
    const u32 v = (c.row < 3 && c.j < N) ? c.lds->io[c.row * 16 + c.j] : 0u;
    wave_lds_sync();
    return v;
  }
  // the point, in every lane, in the engine's form: a last product by Rw mod p leaves every
  // coordinate below 1.001 p with a zero top lane, the carry sweep of F::norm makes the limbs exact
  __device__ static __forceinline__ point store_point(const ctx& c, u32 st) {
    u32 one = 0;
    unroll<NW>([&](auto k) {
      constexpr int i = decltype(k)::value;
      one = c.j == static_cast<u32>(i) ? P::wave_one(i) : one;
    });
    const u32 r = fmul(c, st, one);
    c.lds->io[c.lane] = r;
    wave_lds_sync();
    point p;
#pragma unroll
    for (int i = 0; i < N; ++i) {
      p.X.v[i] = c.lds->io[i];
      p.Y.v[i] = c.lds->io[16 + i];
      p.Z.v[i] = c.lds->io[32 + i];
    }
    wave_lds_sync();
    p.X = F::norm(p.X);
    p.Y = F::norm(p.Y);
    p.Z = F::norm(p.Z);
    return p;
  }

  // acc = 2^window_bits acc + window_sums[i * stride] for i = num_windows - 1 .. 0 (k_horner's chain;
  // window_sums in LDS).  Every lane of the wavefront calls it and receives the result.
  __device__ static point horner(scratch* lds, const point& acc, bool have_acc, const point* window_sums,
                                 u32 stride, u32 num_windows, u32 window_bits) {
    const ctx c = make_ctx(lds);
    u32 i = num_windows;
    u32 st;
    if (have_acc) {
      st = load_point_value(c, acc);
    } else {
      st = load_point(c, window_sums + (num_windows - 1) * stride);
      i = num_windows - 1;
    }
    while (i-- > 0) {
#pragma unroll 1
      for (u32 k = 0; k < window_bits; ++k) st = dbl(c, st);
      st = add(c, st, load_point(c, window_sums + i * stride));
    }
    return store_point(c, st);
  }
};

// the wavefront's scratch (one per workgroup: a single wavefront runs the chain)
__device__ __forceinline__ scratch* wave_scratch() {
  __shared__ scratch lds;
  return &lds;
}
} // namespace sww
} // namespace bz
#endif
