// Per-curve traits consumed by the MSM kernels (csrc/msm/*.h).  A trait fixes
//   * the accumulator type (`point`) and the resident addend type (`addend`) of the curve,
//   * how a caller-supplied generator in C-ABI layout becomes an addend,
//   * the canonical output encoding written to `commitments[i]`
// for the four curves selectable through the reference's `curve_id`
// (cbindings/blitzar_api.h:28-31, sxt/cbindings/base/curve_id_utility.h:44-61).
#pragma once

#include "blitzar_amd/csrc/curve/ed25519.h"
#include "blitzar_amd/csrc/curve/ed29.h"
#include "blitzar_amd/csrc/curve/ed29_coop.h"
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

// curve25519: the kernels compute on the 9 x 29-bit representation (field/f29.h, curve/ed29.h);
// the ABI's radix-2^51 element_p3 only appears where generators enter (make_addend) and where a
// result leaves (encode / store_projective).
struct ed25519_msm {
  static constexpr unsigned curve_id = 0;
  using point = ed29_point;
  using addend = ed29_cached;
  using api_projective = ed_point; // sxt_ristretto255 / c21t::element_p3
  static constexpr size_t api_generator_size = 160; // sxt_ristretto255
  static constexpr size_t output_size = 32;         // sxt_ristretto255_compressed
  static constexpr size_t projective_size = 160;    // element_p3 (fixed-base results)
  // register budget of k_accumulate: 3 waves per SIMD = at most 168 VGPRs (accumulator, current
  // addend, prefetched next addend, product temporaries)
  static constexpr int accumulate_waves_per_simd = 3;

  BZ_HD static point identity() { return ed29::identity(); }
  BZ_HD static point add(const point& a, const point& b) { return ed29::add(a, b); }
  BZ_HD static point dbl_n(const point& a, int k) { return ed29::dbl_n(a, k); }
  BZ_HD static point neg(const point& a) { return ed29::neg(a); }
  BZ_HD static void accumulate(point& acc, const addend& q, bool negate) {
    acc = ed29::add_cached(acc, q, negate);
  }
  BZ_HD static addend make_addend(const void* api_generators, u64 i) {
    return ed29::cached_from_ed(static_cast<const ed_point*>(api_generators)[i]);
  }
  // handle generators arrive as element_p3 too
  BZ_HD static addend addend_from_api_projective(const void* projective, u64 i) {
    return ed29::cached_from_ed(static_cast<const ed_point*>(projective)[i]);
  }
  BZ_HD static void encode(u8* out, const point& p) { ristretto29::encode(out, p); }
  BZ_HD static void store_projective(u8* out, const point& p) {
    *reinterpret_cast<ed_point*>(out) = ed29::to_ed(p);
  }
  // ABI projective element -> engine point (fold of row-sharded partials)
  BZ_HD static point point_from_api_projective(const void* projective, u64 i) {
    return ed29::from_ed(static_cast<const ed_point*>(projective)[i]);
  }
  // k_combine's dependent chain, run by all 64 lanes of one wavefront with the four lanes of
  // every DPP quad sharing each doubling / addition (curve/ed29_coop.h):
  //   sum_w 2^(c w) * window_sums[w * stride]
  static constexpr bool has_wave_horner = true;
#if defined(__HIPCC__)
  __device__ static point wave_horner(const point* window_sums, u32 stride, u32 num_windows,
                                      u32 window_bits) {
    const u32 role = threadIdx.x & 3;
    point acc = window_sums[(num_windows - 1) * stride];
    for (u32 wi = num_windows - 1; wi-- > 0;) {
      for (u32 k = 0; k < window_bits; ++k) acc = ed29::dbl_coop4(acc, role);
      acc = ed29::add_cached_coop4(acc, ed29::to_cached(window_sums[wi * stride]), role);
    }
    return acc;
  }
#endif
};

template <class G, unsigned CurveId> struct sw_msm_base {
  static constexpr unsigned curve_id = CurveId;
  static constexpr int N = G::N;
  using F = typename G::F;
  using point = typename G::point;
  using addend = typename G::affine; // (0, 0) marks the identity (never on y^2 = x^3 + b, b != 0)
  using api_projective = point;      // sxt_*_p2 / element_p2
  using api_affine = sw_api_affine<N>;
  static constexpr size_t api_generator_size = sizeof(api_affine);
  static constexpr size_t projective_size = sizeof(point);
  static constexpr int accumulate_waves_per_simd = 1;
  static constexpr bool has_wave_horner = false;

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
  // the ABI's projective element (element_p2) is the engine's point type for these curves
  BZ_HD static addend addend_from_api_projective(const void* projective, u64 i) {
    return addend_from_point(static_cast<const point*>(projective)[i]);
  }
  BZ_HD static point point_from_api_projective(const void* projective, u64 i) {
    return static_cast<const point*>(projective)[i];
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
