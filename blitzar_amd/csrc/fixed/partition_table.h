// On-disk interoperability with the reference's fixed-base "partition table"
// (sxt/multiexp/pippenger2): a file is `u32 window_width` followed by, for every window of
// `window_width` consecutive generators (the generator list padded with identities to a multiple
// of the width), the 2^w subset sums  entry[m] = sum_{bit i of m} g_{k w + i}  stored as the
// curve's compact (affine) element.  Sources: partition_table.h:36-98 (entry recurrence),
// in_memory_partition_table_accessor.h:42-59,98-105 (file layout),
// in_memory_partition_table_accessor_utility.h:41-58 (identity padding).
//
// The native engine never needs the 2^w-entry table -- it keeps one resident addend per
// generator -- so the table only exists transiently while a handle is written to a file, one
// window slice (2^w entries) at a time, and reading a file only picks the single-generator
// entries m = 2^i back out.
//
// Compact element layouts (must match byte for byte):
//   curve25519     {X, Y, T} 5x51 limbs, Z = 1   (sxt/curve21/type/compact_element.h:30-38);
//                  limbs are *not* canonical, so the recurrence and the add_inplace operation order
//                  of the reference are reproduced exactly (partition_table.h:52-66,
//                  sxt/curve21/operation/add.h:54-80, element_p3.cc:33-41)
//   Weierstrass    {X, Y} Montgomery limbs, identity = {X[N-1] = 2^64-1, Y = R}
//                  (sxt/curve_bng1/type/compact_element.h:26-38); canonical by construction.
#pragma once

#include <cstdio>
#include <vector>

#include "blitzar_amd/csrc/msm/curve_traits.h"

namespace bz {

struct ed_compact {
  fe51 X, Y, T;
};

template <int N> struct sw_compact {
  fe_mont<N> X, Y;
};

template <class C> struct compact_ops;

template <> struct compact_ops<ed25519_msm> {
  using compact = ed_compact;
  using point = ed_point;
  static point identity() { return ed::identity(); }
  static point expand(const compact& c) { return {c.X, c.Y, f51::one(), c.T}; }
  static compact shrink(const point& p) {
    fe51 zinv = f51::invert(p.Z);
    compact c;
    c.X = f51::mul(p.X, zinv);
    c.Y = f51::mul(p.Y, zinv);
    c.T = f51::mul(c.X, c.Y);
    return c;
  }
  // p + q with the operation order of the reference's add_inplace (limb-exact)
  static point add(point p, point q) {
    q.X = f51::add(q.Y, q.X);
    q.Y = f51::add(q.Y, q.Y);
    q.Y = f51::sub(q.Y, q.X);
    q.T = f51::mul(q.T, f51::const_2d());
    p.X = f51::add(p.Y, p.X);
    p.Y = f51::add(p.Y, p.Y);
    p.Y = f51::sub(p.Y, p.X);
    p.Y = f51::mul(p.Y, q.Y);
    p.T = f51::mul(q.T, p.T);
    p.X = f51::mul(p.X, q.X);
    q.X = f51::sub(p.X, p.Y);
    q.Y = f51::add(p.X, p.Y);
    p.Z = f51::mul(p.Z, q.Z);
    q.T = f51::add(p.Z, p.Z);
    q.Z = f51::add(q.T, p.T);
    q.T = f51::sub(q.T, p.T);
    point r;
    r.X = f51::mul(q.X, q.T);
    r.Y = f51::mul(q.Y, q.Z);
    r.Z = f51::mul(q.Z, q.T);
    r.T = f51::mul(q.X, q.Y);
    return r;
  }
};

// ABI-form (saturated 64-bit Montgomery) arithmetic: table entries are canonical by construction
template <class C> struct sw_compact_ops {
  using G = typename C::G64;
  using F = typename G::F;
  static constexpr int N = F::N;
  using compact = sw_compact<N>;
  using point = typename G::point;
  static point identity() { return G::identity(); }
  static bool is_identity(const compact& c) { return c.X.v[N - 1] == ~u64{0}; }
  static point expand(const compact& c) {
    if (is_identity(c)) return G::identity();
    return {c.X, c.Y, F::one()};
  }
  static compact shrink(const point& p) {
    compact c;
    if (F::is_zero(p.Z)) {
      c.X = F::zero();
      c.X.v[N - 1] = ~u64{0};
      c.Y = F::one();
      return c;
    }
    typename G::affine a;
    (void)G::to_affine(a, p);
    c.X = a.x;
    c.Y = a.y;
    return c;
  }
  static point add(const point& p, const point& q) { return G::add(p, q); }
};
template <> struct compact_ops<bn254_msm> : sw_compact_ops<bn254_msm> {};
template <> struct compact_ops<grumpkin_msm> : sw_compact_ops<grumpkin_msm> {};
template <> struct compact_ops<bls12_381_msm> : sw_compact_ops<bls12_381_msm> {};

// one window slice: sums[m] for all 2^w masks; gens holds w projective generators
template <class C>
void partition_table_slice(typename compact_ops<C>::compact* sums, unsigned w,
                           const typename compact_ops<C>::point* gens) {
  using ops = compact_ops<C>;
  sums[0] = ops::shrink(ops::identity());
  for (unsigned i = 0; i < w; ++i) sums[1u << i] = ops::shrink(gens[i]);
  const u64 count = u64{1} << w;
  // entry m (>= 2 bits set) = entry[m without its lowest set bit] + entry[lowest set bit];
  // ascending m visits every dependency first
  for (u64 m = 3; m < count; ++m) {
    const u64 rest = m & (m - 1);
    if (rest == 0) continue;
    const u64 low = m ^ rest;
    sums[m] = ops::shrink(ops::add(ops::expand(sums[rest]), ops::expand(sums[low])));
  }
}

// returns false on a short write
template <class C>
bool write_partition_table(std::FILE* f, unsigned w, const void* projective_generators, u64 n) {
  using ops = compact_ops<C>;
  using point = typename ops::point; // the ABI's projective element
  const point* g = static_cast<const point*>(projective_generators);
  const u32 w32 = w;
  if (std::fwrite(&w32, sizeof(w32), 1, f) != 1) return false;
  const u64 windows = (n + w - 1) / w;
  std::vector<typename ops::compact> sums(u64{1} << w);
  std::vector<point> slice(w);
  for (u64 k = 0; k < windows; ++k) {
    for (unsigned i = 0; i < w; ++i) {
      const u64 idx = k * w + i;
      slice[i] = idx < n ? g[idx] : ops::identity();
    }
    partition_table_slice<C>(sums.data(), w, slice.data());
    if (std::fwrite(sums.data(), sizeof(typename ops::compact), sums.size(), f) != sums.size()) {
      return false;
    }
  }
  return true;
}

// the first n generators as compact elements (what the reference's accessor.copy_generators
// yields: table entry 2^i of each window, in_memory_partition_table_accessor.h:69-82)
template <class C>
void write_compact_generators(std::FILE* f, const void* projective_generators, u64 n) {
  using ops = compact_ops<C>;
  const auto* g = static_cast<const typename ops::point*>(projective_generators);
  for (u64 i = 0; i < n; ++i) {
    const typename ops::compact c = ops::shrink(g[i]);
    std::fwrite(&c, sizeof(c), 1, f);
  }
}

// returns false on a malformed file; generators (projective) are appended to `out`
template <class C>
bool read_partition_generators(std::FILE* f, unsigned& w, std::vector<u8>& out, u64& n) {
  using ops = compact_ops<C>;
  using point = typename ops::point;
  u32 w32 = 0;
  if (std::fread(&w32, sizeof(w32), 1, f) != 1 || w32 == 0 || w32 > 32) return false;
  w = w32;
  std::fseek(f, 0, SEEK_END);
  const long size = std::ftell(f);
  const u64 entry = sizeof(typename ops::compact);
  const u64 slice_bytes = entry << w;
  if (size < 4 || (static_cast<u64>(size) - 4) % slice_bytes != 0) return false;
  const u64 windows = (static_cast<u64>(size) - 4) / slice_bytes;
  n = windows * w;
  out.resize(n * sizeof(point));
  point* g = reinterpret_cast<point*>(out.data());
  for (u64 k = 0; k < windows; ++k) {
    for (unsigned i = 0; i < w; ++i) {
      typename ops::compact c;
      std::fseek(f, static_cast<long>(4 + k * slice_bytes + (entry << i)), SEEK_SET);
      if (std::fread(&c, entry, 1, f) != 1) return false;
      g[k * w + i] = ops::expand(c);
    }
  }
  return true;
}
} // namespace bz
