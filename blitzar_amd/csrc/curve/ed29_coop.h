// Latency-oriented edwards25519 arithmetic for the single dependent chain at the end of an MSM
// (Horner over windows: ~250 doublings that nothing can run in parallel with).
//
// A lone lane needs 4 squarings + 4 products per doubling, ~4700 cycles back to back.  Here the
// four lanes of a DPP quad hold the same point and split each phase: four squarings at once, then
// four products at once, with `v_mov_b32 ... quad_perm` (full-rate VALU, no LDS) redistributing the
// results.  A doubling becomes one squaring + one product + ~70 DPP moves and selects: ~2.2x
// shorter.  All 16 quads of the wavefront compute the same thing; that costs nothing, the SIMD
// would idle otherwise.
//
// Device only.  Formulas and limb bounds are those of curve/ed29.h.
#pragma once

#include "blitzar_amd/csrc/curve/ed29.h"

#if defined(__HIPCC__)
namespace bz {
namespace ed29 {

// value of lane K of the caller's quad (lanes 4q .. 4q+3)
// (the empty asm keeps the move a plain v_mov_b32_dpp: hipcc's fold of a DPP move into
// v_subrev_u32_dpp was observed to miscompute, see curve/sw29_coop.h)
template <int K> __device__ __forceinline__ u32 quad_get(u32 v) {
  u32 r = static_cast<u32>(
      __builtin_amdgcn_update_dpp(0, static_cast<int>(v), K * 0x55, 0xf, 0xf, true));
  asm volatile("" : "+v"(r));
  return r;
}

template <int K> __device__ __forceinline__ fe29 quad_get(const fe29& f) {
  fe29 h;
#pragma unroll
  for (int i = 0; i < 9; ++i) h.v[i] = quad_get<K>(f.v[i]);
  return h;
}

__device__ __forceinline__ fe29 select4(u32 role, const fe29& a, const fe29& b, const fe29& c,
                                        const fe29& d) {
  fe29 h;
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    const u32 lo = (role & 1) ? b.v[i] : a.v[i];
    const u32 hi = (role & 1) ? d.v[i] : c.v[i];
    h.v[i] = (role & 2) ? hi : lo;
  }
  return h;
}

// second phase shared by doubling and addition: from (ex, ey, ez, et) the four products
// X3 = ex et, Y3 = ey ez, Z3 = ez et, T3 = ex ey, one per lane, then broadcast
__device__ __forceinline__ ed29_point finish_coop4(u32 role, const fe29& ex, const fe29& ey,
                                                   const fe29& ez, const fe29& et) {
  const fe29 f = select4(role, ex, ey, ez, ex);
  const fe29 g = select4(role, et, ez, et, ey);
  const fe29 h = f29::mul(f, g);
  return {quad_get<0>(h), quad_get<1>(h), quad_get<2>(h), quad_get<3>(h)};
}

// 2p; p replicated in the quad, role = lane & 3
__device__ __forceinline__ ed29_point dbl_coop4(const ed29_point& p, u32 role) {
  const fe29 in = select4(role, p.X, p.Y, p.Z, f29::add(p.X, p.Y));
  const fe29 sq = f29::sq(in); // xx | yy | zz | (x + y)^2
  const fe29 xx = quad_get<0>(sq), yy = quad_get<1>(sq), zz = quad_get<2>(sq), s = quad_get<3>(sq);
  const fe29 ey = f29::add(yy, xx);                                           // B 2
  const fe29 ez = f29::sub(yy, xx);                                           // B 3
  const fe29 ex = f29::weak_reduce(f29::sub(f29::sub(s, yy), xx));            // B 1
  const fe29 et = f29::weak_reduce(f29::sub(f29::add(f29::add(zz, zz), xx), yy)); // B 1
  return finish_coop4(role, ex, ey, ez, et);
}

// p + q for a cached addend q; p and q replicated in the quad
__device__ __forceinline__ ed29_point add_cached_coop4(const ed29_point& p, const ed29_cached& q,
                                                       u32 role) {
  const fe29 f = select4(role, f29::add(p.Y, p.X), f29::sub(p.Y, p.X), p.T, p.Z); // B 2 | 3 | 1 | 1
  const fe29 g = select4(role, q.YpX, q.YmX, q.T2d, q.Z);
  const fe29 m = f29::mul(f, g); // a | b | c | zz
  const fe29 a = quad_get<0>(m), b = quad_get<1>(m), c = quad_get<2>(m), zz = quad_get<3>(m);
  const fe29 d = f29::add(zz, zz);
  const fe29 ez = f29::add(d, c);                   // B 3
  const fe29 et = f29::weak_reduce(f29::sub(d, c)); // B 1
  const fe29 ex = f29::sub(a, b);                   // B 3
  const fe29 ey = f29::add(a, b);                   // B 2
  return finish_coop4(role, ex, ey, ez, et);
}
} // namespace ed29

} // namespace bz
#endif
