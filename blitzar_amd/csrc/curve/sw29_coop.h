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

// p + q (Alg. 7, complete), both replicated in the quad, role = lane & 3: the six products of
// sw29::add in two rounds (4 + 2), the two multiplications by |3b| side by side, the three
// two-product coordinates of sw29::finish on three lanes -- the latency of about four field
// products instead of twelve (k_horner's window fold: the levels that have at most one addition per
// quad of the workgroup).  Same operations on the same values as sw29::add, so its bounds apply.
template <class G>
__device__ __forceinline__ typename G::point add_coop4(const typename G::point& p,
                                                       const typename G::point& q, u32 role) {
  using F = typename G::F;
  using fe = typename F::fe;
  constexpr int N = F::N;
  // round 1: X1 X2 | Y1 Y2 | Z1 Z2 | (X1 + Y1)(X2 + Y2)
  const fe m1 = F::mul(select4<N>(role, p.X, p.Y, p.Z, F::add(p.X, p.Y)),
                       select4<N>(role, q.X, q.Y, q.Z, F::add(q.X, q.Y)));
  fe t0 = quad_get<0, N>(m1);
  const fe t1 = quad_get<1, N>(m1), t2 = quad_get<2, N>(m1);
  const fe t3 = F::norm(F::template sub<4>(quad_get<3, N>(m1), F::add(t0, t1)));
  // round 2: (Y1 + Z1)(Y2 + Z2) | (X1 + Z1)(X2 + Z2) (lanes 2, 3 repeat lanes 0, 1)
  const fe m2 = F::mul(F::add((role & 1) ? p.X : p.Y, p.Z), F::add((role & 1) ? q.X : q.Y, q.Z));
  const fe t4 = F::norm(F::template sub<4>(quad_get<0, N>(m2), F::add(t1, t2)));
  const fe y3 = F::norm(F::template sub<4>(quad_get<1, N>(m2), F::add(t0, t2)));
  // |3b| t2 on the even lanes, |3b| y3 on the odd ones
  const fe ub = G::mul_b3((role & 1) ? y3 : t2);
  const fe u2 = quad_get<0, N>(ub), u3 = quad_get<1, N>(ub);
  t0 = F::add(F::add(t0, t0), t0);
  // round 3: the coordinates of sw29::finish, X | Y | Z | (X again)
  const typename G::pm s = G::template plus_minus<8>(t1, u2);
  fe a, b, c, d;
  if constexpr (!G::b3_negative) {
    const fe z3 = F::norm(s.plus);
    const fe& t1m = s.minus;
    const fe t4n = F::template neg<8>(t4);
    a = select4<N>(role, t3, t1m, z3, t3);
    b = select4<N>(role, t1m, z3, t4, t1m);
    c = select4<N>(role, t4n, u3, t0, t4n);
    d = select4<N>(role, u3, t0, t3, u3);
  } else {
    const fe& z3 = s.minus;
    const fe& t1m = s.plus;
    const fe t0r = F::norm(t0);
    const fe t0n = F::template neg<4>(t0r);
    a = select4<N>(role, t3, t1m, z3, t3);
    b = select4<N>(role, t1m, z3, t4, t1m);
    c = select4<N>(role, t4, u3, t0r, t4);
    d = select4<N>(role, u3, t0n, t3, u3);
  }
  const fe h = F::mul2(a, b, c, d);
  return {quad_get<0, N>(h), quad_get<1, N>(h), quad_get<2, N>(h)};
}
} // namespace sw29_coop
} // namespace bz
#endif
