// Per-curve traits consumed by the MSM kernels (csrc/msm/*.h).  A trait fixes
//   * the accumulator type (`point`) and the resident addend type (`addend`) of the curve,
//   * how a caller-supplied generator in C-ABI layout becomes an addend,
//   * the canonical output encoding written to `commitments[i]`
// for the four curves selectable through the reference's `curve_id`
// (cbindings/blitzar_api.h:28-31, sxt/cbindings/base/curve_id_utility.h:44-61).
#pragma once

#include "blitzar_amd/csrc/curve/ed25519.h"
#include "blitzar_amd/csrc/curve/weierstrass.h"

namespace bz {

// C-ABI affine layouts of the reference: {X, Y, u8 infinity} with the struct
// padded to 8 bytes.  bn254/grumpkin: 72 bytes; bls12-381: the reference reads a 104-byte
// stride (cbindings/pedersen.cc:212-217), infinity at byte 96.
template <int N> struct sw_api_affine {
  u64 X[N];
  u64 Y[N];
  u8 infinity;
};
static_assert(sizeof(sw_api_affine<4>) == 72);
static_assert(sizeof(sw_api_affine<6>) == 104);

struct ed25519_msm {
  static constexpr unsigned curve_id = 0;
  using point = ed_point;
  using addend = ed_cached;
  static constexpr size_t api_generator_size = 160; // sxt_ristretto255
  static constexpr size_t output_size = 32;         // sxt_ristretto255_compressed
  static constexpr size_t projective_size = 160;    // element_p3 (fixed-base results)

  BZ_HD static point identity() { return ed::identity(); }
  BZ_HD static point add(const point& a, const point& b) { return ed::add(a, b); }
  BZ_HD static point dbl_n(const point& a, int k) { return ed::dbl_n(a, k); }
  BZ_HD static point neg(const point& a) { return ed::neg(a); }
  BZ_HD static void accumulate(point& acc, const addend& q, bool negate) {
    acc = ed::to_point(negate ? ed::sub_cached(acc, q) : ed::add_cached(acc, q));
  }
  BZ_HD static addend make_addend(const void* api_generators, u64 i) {
    const ed_point* g = static_cast<const ed_point*>(api_generators);
    return ed::to_cached(g[i]);
  }
  BZ_HD static addend addend_from_point(const point& p) { return ed::to_cached(p); }
  BZ_HD static point point_from_api(const void* api_generators, u64 i) {
    return static_cast<const ed_point*>(api_generators)[i];
  }
  BZ_HD static void encode(u8* out, const point& p) { ristretto::encode(out, p); }
  BZ_HD static void store_projective(u8* out, const point& p) {
    *reinterpret_cast<point*>(out) = p;
  }
};

template <class G, unsigned CurveId> struct sw_msm_base {
  static constexpr unsigned curve_id = CurveId;
  static constexpr int N = G::N;
  using F = typename G::F;
  using point = typename G::point;
  using addend = typename G::affine; // (0, 0) marks the identity (never on y^2 = x^3 + b, b != 0)
  using api_affine = sw_api_affine<N>;
  static constexpr size_t api_generator_size = sizeof(api_affine);
  static constexpr size_t projective_size = sizeof(point);

  BZ_HD static point identity() { return G::identity(); }
  BZ_HD static point add(const point& a, const point& b) { return G::add(a, b); }
  BZ_HD static point dbl_n(const point& a, int k) { return G::dbl_n(a, k); }
  BZ_HD static point neg(const point& a) { return G::neg(a); }
  BZ_HD static void accumulate(point& acc, const addend& q, bool negate) {
    if (F::is_zero(q.x) && F::is_zero(q.y)) return;
    addend t = q;
    t.y = F::cneg(q.y, negate);
    acc = G::add_mixed(acc, t);
  }
  BZ_HD static addend make_addend(const void* api_generators, u64 i) {
    const api_affine& g = static_cast<const api_affine*>(api_generators)[i];
    addend a;
    for (int k = 0; k < N; ++k) {
      a.x.v[k] = g.infinity ? 0 : g.X[k];
      a.y.v[k] = g.infinity ? 0 : g.Y[k];
    }
    return a;
  }
  BZ_HD static addend addend_from_point(const point& p) {
    addend a;
    if (G::to_affine(a, p)) {
      a.x = F::zero();
      a.y = F::zero();
    }
    return a;
  }
  BZ_HD static point point_from_api(const void* api_generators, u64 i) {
    const api_affine& g = static_cast<const api_affine*>(api_generators)[i];
    if (g.infinity) return G::identity();
    point p;
    for (int k = 0; k < N; ++k) {
      p.X.v[k] = g.X[k];
      p.Y.v[k] = g.Y[k];
    }
    p.Z = F::one();
    return p;
  }
  BZ_HD static void store_projective(u8* out, const point& p) {
    *reinterpret_cast<point*>(out) = p;
  }
  // {X, Y Montgomery, u8 infinity}; identity = {0, R, 1}
  BZ_HD static void encode_affine(u8* out, const point& p) {
    typename G::affine a;
    const bool inf = G::to_affine(a, p);
    u64* o = reinterpret_cast<u64*>(out);
    for (int k = 0; k < N; ++k) {
      o[k] = a.x.v[k];
      o[N + k] = a.y.v[k];
    }
    o[2 * N] = inf ? 1 : 0; // infinity byte + zeroed struct padding
  }
};

struct bn254_msm : sw_msm_base<bn254_g1, 2> {
  static constexpr size_t output_size = 72; // sxt_bn254_g1
  BZ_HD static void encode(u8* out, const point& p) { encode_affine(out, p); }
};

struct grumpkin_msm : sw_msm_base<grumpkin_g, 3> {
  static constexpr size_t output_size = 72; // sxt_grumpkin
  BZ_HD static void encode(u8* out, const point& p) { encode_affine(out, p); }
};

struct bls12_381_msm : sw_msm_base<bls12_381_g1, 1> {
  static constexpr size_t output_size = 48; // sxt_bls12_381_g1_compressed
  BZ_HD static void encode(u8* out, const point& p) { bls12_381_g1_compress(out, p); }
};
} // namespace bz
