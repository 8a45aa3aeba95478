// Latency-oriented doubling for the Weierstrass curves: a lone lane needs the eight field products
// of a doubling back to back.  Here the four lanes of a DPP quad hold the same point and split the two rounds of four independent field
// products of RCB15 Alg. 9, so a doubling on the Horner chain of a single column costs two product
// latencies instead of eight.  Device only; formulas and bounds are those of sw29::dbl.
//
// Compiler note (ROCm 7.2 hipcc): a DPP move whose result only feeds a subtraction gets folded into
// `v_subrev_u32_dpp`, and that form was observed to produce wrong values in every lane but the
// source lane (grumpkin's `sub<4>(quad_get<2>(h), quad_get<1>(h))`).  The
// value is therefore passed through an empty asm statement, which keeps it a plain
// `v_mov_b32_dpp`.
#pragma once

#include "blitzar_amd/csrc/curve/sw29.h"

#if defined(__HIPCC__)
namespace bz {
namespace sw29_coop {

template <int K, int N> __device__ __forceinline__ fe29m<N> quad_get(const fe29m<N>& f) {
  fe29m<N> h;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    u32 v = static_cast<u32>(
        __builtin_amdgcn_update_dpp(0, static_cast<int>(f.v[i]), K * 0x55, 0xf, 0xf, true));
    asm volatile("" : "+v"(v)); // keep it a plain v_mov_b32_dpp (see the note above)
    h.v[i] = v;
  }
  return h;
}

template <int N>
__device__ __forceinline__ fe29m<N> select4(u32 role, const fe29m<N>& a, const fe29m<N>& b,
                                            const fe29m<N>& c, const fe29m<N>& d) {
  fe29m<N> h;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const u32 lo = (role & 1) ? b.v[i] : a.v[i];
    const u32 hi = (role & 1) ? d.v[i] : c.v[i];
    h.v[i] = (role & 2) ? hi : lo;
  }
  return h;
}

// 2p; p replicated in the quad, role = lane & 3
template <class G> __device__ __forceinline__ typename G::point dbl_coop4(const typename G::point& p, u32 role) {
  using F = typename G::F;
  using fe = typename F::fe;
  constexpr int N = F::N;
  // round 1: Y^2 | Y Z | Z^2 | X Y
  const fe m = F::mul(select4<N>(role, p.Y, p.Y, p.Z, p.X), select4<N>(role, p.Y, p.Z, p.Z, p.Y));
  const fe t0 = quad_get<0, N>(m), t1 = quad_get<1, N>(m), zz = quad_get<2, N>(m),
           xy = quad_get<3, N>(m);
  const fe z3 = F::mul_small(t0, 8); // V < 10
  const fe u = G::mul_b3(zz);        // |3b| Z^2, V < 4
  typename G::point r;
  if constexpr (!G::b3_negative) {
    const fe y3 = F::add(t0, u);                               // B 2
    const fe u3 = F::norm(F::add(F::add(u, u), u));            // V < 12
    const fe t0m = F::norm(F::template sub<16>(t0, u3));       // V < 17.3
    // round 2: t1 z3 | u z3 | t0m y3 | t0m xy
    const fe h = F::mul(select4<N>(role, t1, u, t0m, t0m), select4<N>(role, z3, z3, y3, xy));
    r.Z = quad_get<0, N>(h);
    r.Y = F::norm(F::add(quad_get<1, N>(h), quad_get<2, N>(h)));
    const fe x = quad_get<3, N>(h);
    r.X = F::norm(F::add(x, x));
  } else {
    const fe y3 = F::norm(F::template sub<8>(t0, u));          // V < 9.3
    const fe t0m = F::add(t0, F::add(F::add(u, u), u));        // B 4, V < 13.3
    const fe h = F::mul(select4<N>(role, t1, u, t0m, t0m), select4<N>(role, z3, z3, y3, xy));
    r.Z = quad_get<0, N>(h);
    r.Y = F::norm(F::template sub<4>(quad_get<2, N>(h), quad_get<1, N>(h)));
    const fe x = quad_get<3, N>(h);
    r.X = F::norm(F::add(x, x));
  }
  return r;
}
} // namespace sw29_coop
} // namespace bz
#endif
