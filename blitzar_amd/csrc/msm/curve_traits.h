// Per-curve traits consumed by the MSM kernels (csrc/msm/*.h).  A trait fixes
//   * the accumulator type (`point`) and the resident addend type (`addend`) of the curve,
//   * how a caller-supplied generator in C-ABI layout becomes an addend,
//   * the canonical output encoding written to `commitments[i]`
// for the four curves selectable through the reference's `curve_id`
// (cbindings/blitzar_api.h:28-31, sxt/cbindings/base/curve_id_utility.h:44-61).
#pragma once

#include "blitzar_amd/csrc/curve/ed25519.h"
#include "blitzar_amd/csrc/curve/ed29.h"
#include "blitzar_amd/csrc/curve/ed16_wave.h"
#include "blitzar_amd/csrc/curve/sw29.h"
#include "blitzar_amd/csrc/curve/sw29_coop.h"
#include "blitzar_amd/csrc/curve/sw_wave.h"
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
  using addend = ed29_cached_packed; // (Y+X, Y-X, Z, 2dT), 4 x 256 bits
  using api_projective = ed_point; // sxt_ristretto255 / c21t::element_p3
  // Itanium-mangled names of the reference types (meta.txt of a BLITZAR_DUMP_DIR recording)
  static constexpr const char* reference_element_name = "N3sxt4c21t10element_p3E";
  static constexpr const char* reference_compact_name = "N3sxt4c21t15compact_elementE";
  static constexpr size_t api_generator_size = 160; // sxt_ristretto255
  static constexpr size_t output_size = 32;         // sxt_ristretto255_compressed
  static constexpr size_t projective_size = 160;    // element_p3 (fixed-base results)
  // register budget of k_accumulate: 3 waves per SIMD = at most 168 VGPRs (accumulator, current
  // addend, prefetched next addend, product temporaries)
  // (measured on MI355X at config 2: 2 / 3 / 4 waves per SIMD -> 0.862 / 0.860 / 1.13 ms, the last
  // one spills: the kernel is issue-bound, not latency-bound)
  static constexpr int accumulate_waves_per_simd = 3;
  static constexpr bool has_batched_prepare = false;
  static constexpr double call_table_entry_cost = 1.0;

  BZ_HD static point identity() { return ed29::identity(); }
  BZ_HD static point add(const point& a, const point& b) { return ed29::add(a, b); }
  BZ_HD static point dbl_n(const point& a, int k) { return ed29::dbl_n(a, k); }
  BZ_HD static point neg(const point& a) { return ed29::neg(a); }
  BZ_HD static void accumulate(point& acc, const addend& q, bool negate) {
    acc = ed29::add_cached(acc, ed29::unpack(q), negate);
  }
  // k_accumulate's pipeline: the gathered row is unpacked at the top of the iteration (`stage`,
  // pinned: the staging registers are free for the next gather), the addition consumes limbs
  using operand = ed29_cached;
  BZ_HD static operand stage(const addend& q) {
    operand o = ed29::unpack(q);
    f29::pin(o.YpX);
    f29::pin(o.YmX);
    f29::pin(o.Z);
    f29::pin(o.T2d);
    return o;
  }
  BZ_HD static void accumulate(point& acc, const operand& q, bool negate) {
    acc = ed29::add_cached(acc, q, negate);
  }
  // k_accumulate's gather: the row's first two 32-byte pieces (Y+X | Y-X) arrive exchanged when
  // the digit is negative -- by address, two 16-byte loads each -- so the addition needs no selects
  // on them (18 v_cndmask per addition) and only conditions 2dT (`accumulate_gathered`)
  static constexpr bool has_signed_gather = true;
  BZ_HD static addend gather(const addend* table, u32 row, bool negate) {
    return ed29::gather_signed(table, row, negate);
  }
  BZ_HD static void accumulate_gathered(point& acc, const operand& q, bool negate) {
    acc = ed29::add_cached_presigned(acc, q, negate);
  }
  // identity + (+-q) of a segment's first entry: loaded, not added (k_accumulate)
  // the addition in two halves with no operand live between them (k_accumulate, BZ_ACCUMULATE_DIRECT=2)
  static constexpr bool has_split_add = true;
  using completed = ed29::ed29_completed;
  BZ_HD static completed add_front(const point& acc, const operand& q, bool negate) {
    completed m = ed29::add_cached_presigned_front(acc, q, negate);
    f29::pin(m.ex);
    f29::pin(m.ey);
    f29::pin(m.ez);
    f29::pin(m.et);
    return m;
  }
  BZ_HD static point add_back(const completed& m) { return ed29::add_cached_back(m); }
  static constexpr bool first_pinned = true; // (see k_accumulate)
  BZ_HD static point first_gathered(const operand& q, bool negate) {
    return ed29::from_cached_presigned(q, negate);
  }
  BZ_HD static addend make_addend(const void* api_generators, u64 i) {
    return ed29::pack(ed29::cached_from_ed(static_cast<const ed_point*>(api_generators)[i]));
  }
  // handle generators arrive as element_p3 too
  BZ_HD static addend addend_from_api_projective(const void* projective, u64 i) {
    return make_addend(projective, i);
  }
  BZ_HD static void encode(u8* out, const point& p) { ristretto29::encode(out, p); }
  BZ_HD static void store_projective(u8* out, const point& p) {
    *reinterpret_cast<ed_point*>(out) = ed29::to_ed(p);
  }
  // ABI projective element -> engine point (fold of row-sharded partials)
  BZ_HD static point point_from_api_projective(const void* projective, u64 i) {
    return ed29::from_ed(static_cast<const ed_point*>(projective)[i]);
  }
  // engine point -> caller generator layout (sxt_ristretto255)
  BZ_HD static void store_api_generator(u8* out, const point& p) { store_projective(out, p); }
  // caller generator (sxt_ristretto255 = projective element_p3) -> engine point
  BZ_HD static point point_from_api_generator(const void* api_generators, u64 i) {
    return point_from_api_projective(api_generators, i);
  }
  static constexpr u32 batch_points_per_lane = 4;
  // shared inversions (Montgomery's trick over a workgroup: msm/kernels.h tree_products /
  // tree_inverses; k_prepare_addends_batched, fixed/partition_table_device.h)
  using batch_fe = fe29;
  BZ_HD static fe29 batch_one() { return f29::one(); }
  BZ_HD static fe29 batch_mul(const fe29& a, const fe29& b) { return f29::mul(a, b); }
  // the coordinate to invert for the affine form; edwards points never have Z = 0
  BZ_HD static fe29 batch_z(const point& p, bool& is_identity) {
    is_identity = false;
    return p.Z;
  }
  // compact table entry {x, y, x y} as 5 x 51-bit limbs (sxt/curve21/type/compact_element.h:
  // 30-38); the limbs the reference's field products leave are the canonical digits of the value
  // (checked against the reference's own tables, tests/test_oracle.py)
  static constexpr size_t compact_size = 120;
  BZ_HD static void store_compact(u8* out, const point& p, const fe29& zinv, bool) {
    const fe29 x = f29::mul(p.X, zinv), y = f29::mul(p.Y, zinv);
    const fe51 c[3] = {f29::to_fe51(x), f29::to_fe51(y), f29::to_fe51(f29::mul(x, y))};
    u64* o = reinterpret_cast<u64*>(out);
    for (int k = 0; k < 3; ++k)
      for (int i = 0; i < 5; ++i) o[5 * k + i] = c[k].v[i];
  }
#if defined(__HIPCC__)
  // 1 / z by the 64 lanes of one wavefront (every lane passes the same z and gets the result):
  // z^(p - 2) = (z^(2^252 - 3))^8 * z^3, the long exponentiation row-parallel (curve/ed16_wave.h)
  __device__ static fe29 batch_wave_invert(const fe29& z) {
    const ed16w::lane_ctx c = ed16w::make_ctx(ed16w::wave_scratch());
    const fe29 t = ed16w::pow22523(c, z);
    return f29::mul(f29::sqn(t, 3), f29::mul(f29::sq(z), z));
  }
#endif
  // k_horner's dependent chain, run by one wavefront that holds the single accumulator spread over
  // its 64 lanes (curve/ed16_wave.h: a row of 16 lanes per coordinate, a limb per lane):
  //   2^(c n) * acc + sum_{w < n} 2^(c w) * window_sums[w * stride]   (acc absent: top window first)
  // The window sums are rewritten in place to their packed cached form (128 of the slot's 144 bytes).
  static constexpr bool has_wave_horner = true;
#if defined(__HIPCC__)
  __device__ static point wave_horner(const point& acc, bool have_acc, point* window_sums,
                                      u32 stride, u32 num_windows, u32 window_bits) {
    const ed16w::lane_ctx c = ed16w::make_ctx(ed16w::wave_scratch());
    for (u32 w = c.lane; w < num_windows; w += 64) {
      point* slot = window_sums + static_cast<size_t>(w) * stride;
      const ed29_cached_packed q = ed29::pack(ed29::to_cached(*slot));
      *reinterpret_cast<ed29_cached_packed*>(slot) = q;
    }
    ed16w::wave_lds_sync();
    u32 state = ed16w::identity(c);
    if (have_acc) state = ed16w::load_point(c, acc);
    for (u32 i = num_windows; i-- > 0;) {
      if (have_acc || i + 1 != num_windows) {
        for (u32 k = 0; k < window_bits; ++k) state = ed16w::dbl(c, state);
      }
      const u32* q = reinterpret_cast<const u32*>(window_sums + static_cast<size_t>(i) * stride);
      state = ed16w::add_cached(c, state, ed16w::load_words(c, q));
    }
    return ed16w::store_point(c, state);
  }
  // out[w * stride] = 2^(bits w) g for w < windows by one wavefront (every lane passes the same g):
  // the doubling chain of a per-call window table (curve_tu.h, k_chain_points_wave), ~6x shorter
  // than a lane walking it alone
  __device__ static void wave_chain(point* out, u64 stride, const point& g, u32 windows, u32 bits) {
    const ed16w::lane_ctx c = ed16w::make_ctx(ed16w::wave_scratch());
    u32 state = ed16w::load_point(c, g);
#pragma unroll 1
    for (u32 w = 0;; ++w) {
      const point p = ed16w::store_point(c, state);
      if (c.lane == 0) out[static_cast<u64>(w) * stride] = p;
      if (w + 1 >= windows) break;
#pragma unroll 1
      for (u32 k = 0; k < bits; ++k) state = ed16w::dbl(c, state);
    }
  }
  // v + m * s (m != 0) by a whole wavefront: every lane passes the same points and m
  static constexpr bool has_wave_add_multiple = true;
  static constexpr bool reduce_scan_few_columns_only = false;
  __device__ static point wave_add_multiple(const point& v, const point& s, u32 m) {
    __shared__ ed29_cached_packed addend[2];
    const ed16w::lane_ctx c = ed16w::make_ctx(ed16w::wave_scratch());
    if (c.lane < 2) addend[c.lane] = ed29::pack(ed29::to_cached(c.lane == 0 ? s : v));
    ed16w::wave_lds_sync();
    const u32 sq = ed16w::load_words(c, addend[0].w), vq = ed16w::load_words(c, addend[1].w);
    u32 state = ed16w::identity(c);
    for (int bit = 31 - __builtin_clz(m); bit >= 0; --bit) {
      state = ed16w::dbl(c, state);
      if ((m >> bit) & 1) state = ed16w::add_cached(c, state, sq);
    }
    state = ed16w::add_cached(c, state, vq);
    return ed16w::store_point(c, state);
  }
  // canonical encoding by a whole wavefront (every lane passes the same point, lane 0 writes): the
  // inverse square root's 252 squarings run lane-parallel
  static constexpr bool has_wave_encode = true;
  static constexpr bool has_coop_add = false;
  __device__ static void wave_encode(u8* out, const point& p) {
    const ed16w::lane_ctx c = ed16w::make_ctx(ed16w::wave_scratch());
    u64 w[4];
    ristretto29::encode_words(w, p, [&c](const fe29& z) { return ed16w::pow22523(c, z); });
    if (c.lane == 0) {
      for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 8; ++j) out[8 * i + j] = static_cast<u8>(w[i] >> (8 * j));
    }
  }
#endif
};

// curve25519 against a *resident* generator set (registered once: built-in generators,
// bzamd_generators, handles): Z = 1 addends of 128 bytes, 7 instead of 8 products per addition
struct ed25519_niels_msm : ed25519_msm {
  using addend = ed29_niels;
  BZ_HD static void accumulate(point& acc, const addend& q, bool negate) {
    acc = ed29::add_niels(acc, q, negate);
  }
  static constexpr bool has_split_add = false;
  BZ_HD static point first(const addend& q, bool negate) { return ed29::from_niels(q, negate); }
  static constexpr bool has_signed_gather = false; // 36-byte limb pieces: no aligned exchange
  // an accumulated entry against Z = 1 addends relative to the per-call (Y+X, Y-X, Z, 2dT) form
  // (7 of 8 field products, nothing to unpack: plan.h, choose_call_table)
  static constexpr double call_table_entry_cost = 0.88;
  using operand = ed29_niels; // stored as limbs: nothing to unpack
  BZ_HD static operand stage(const addend& q) {
    operand o = q;
    f29::pin(o.YpX);
    f29::pin(o.YmX);
    f29::pin(o.T2d);
    return o;
  }
  BZ_HD static addend make_addend(const void* api_generators, u64 i) {
    return ed29::to_niels(ed29::from_ed(static_cast<const ed_point*>(api_generators)[i]));
  }
  BZ_HD static addend addend_from_api_projective(const void* projective, u64 i) {
    return make_addend(projective, i);
  }
  // batched normalisation (k_prepare_addends_batched): the inversions of a workgroup's generators
  // share one exponentiation
  static constexpr bool has_batched_prepare = true;
  BZ_HD static fe29 batch_load_z(const void* api_generators, u64 i) {
    return f29::from_fe51(static_cast<const ed_point*>(api_generators)[i].Z);
  }
  BZ_HD static addend batch_make_addend(const void* api_generators, u64 i, const fe29& zinv) {
    const ed_point& g = static_cast<const ed_point*>(api_generators)[i];
    return niels_of(f29::from_fe51(g.X), f29::from_fe51(g.Y), zinv);
  }
  // engine point and 1 / Z -> Z = 1 addend (window tables)
  BZ_HD static addend addend_from_point(const point& p, const fe29& zinv, bool) {
    return niels_of(p.X, p.Y, zinv);
  }
  BZ_HD static addend niels_of(const fe29& big_x, const fe29& big_y, const fe29& zinv) {
    const fe29 x = f29::mul(big_x, zinv);
    const fe29 y = f29::mul(big_y, zinv);
    ed29_niels n;
    n.YpX = f29::weak_reduce(f29::add(y, x));
    n.YmX = f29::weak_reduce(f29::sub(y, x));
    n.T2d = f29::mul(f29::mul(x, y), f29::const_2d());
    for (int k = 0; k < 5; ++k) n.pad[k] = 0;
    return n;
  }
};

// Weierstrass curves: the kernels compute on the unsaturated-limb Montgomery representation
// (field/mont29.h, curve/sw29.h); the ABI's saturated 64-bit Montgomery limbs only appear where
// generators enter and where a result leaves (conversions + the existing ABI-form encoders).
template <class G29, unsigned CurveId> struct sw_msm_base {
  static constexpr unsigned curve_id = CurveId;
  static constexpr double call_table_entry_cost = 1.0; // (per-call and resident addends are the same)
  using G64 = typename G29::G64;       // ABI-form curve (curve/weierstrass.h)
  using F64 = typename G64::F;
  static constexpr int N64 = F64::N;
  using point = typename G29::point;
  using addend = typename G29::packed; // affine (x, y); (0, 0) marks the identity (never on the curve)
  using api_projective = typename G64::point; // sxt_*_p2 / element_p2
  using api_affine = sw_api_affine<N64>;
  static constexpr size_t api_generator_size = sizeof(api_affine);
  static constexpr size_t projective_size = sizeof(api_projective);
#ifndef BZ_ACC_WAVES_SW9
#define BZ_ACC_WAVES_SW9 3
#endif
  static constexpr int accumulate_waves_per_simd = G29::N <= 9 ? BZ_ACC_WAVES_SW9 : 2;
  // k_accumulate's additions use the pinned product-scanning field products (field/mont29.h) on the
  // 9-limb curves: config 4 accumulate 348.3 -> 338.8 ms, config 5 138.1 -> 135.1 (A/B on one box,
  // profiles/round2_ab_pinned_products.log).  With 14 limbs the quotient digits and the single
  // chain spill (bls12-381: 11.3 -> 170 ms); there the operand-scanning product keeps its N
  // accumulators and only pins their order (msm_bls12_381.hip).
#ifndef BZ_SW_ACCUMULATE_PINNED
#define BZ_SW_ACCUMULATE_PINNED 1
#endif
  static constexpr bool accumulate_pinned = BZ_SW_ACCUMULATE_PINNED != 0 && G29::N <= 9;
  static constexpr bool has_batched_prepare = false;
  static constexpr bool has_signed_gather = false;
  static constexpr bool has_wave_encode = false;
  // k_reduce's lane weights as a suffix scan over the lanes (8 additions per lane instead of a
  // 15-bit double-and-add); the one multiple left per workgroup -- the block's first bucket index
  // times the block's plain sum -- by a whole wavefront on the lane-spread form (curve/sw_wave.h).
  // BZ_SW_REDUCE_SCAN=0: the per-lane double-and-add of rounds 1-3.
#ifndef BZ_SW_REDUCE_SCAN
#define BZ_SW_REDUCE_SCAN 1
#endif
  static constexpr bool has_wave_add_multiple = BZ_SW_REDUCE_SCAN != 0;
  // only for launches of few columns (engine.h; see k_reduce): many columns keep the per-lane form
  static constexpr bool reduce_scan_few_columns_only = true;
#if defined(__HIPCC__)
  // v + m * s (m != 0) by a whole wavefront: every lane passes the same points and m
  __device__ static point wave_add_multiple(const point& v, const point& s, u32 m) {
    using W = sww::wave<G29>;
    const typename W::ctx c = W::make_ctx(sww::wave_scratch());
    const u32 sq = W::load_point_value(c, s), vq = W::load_point_value(c, v);
    u32 st = sq;
    for (int bit = 30 - __builtin_clz(m); bit >= 0; --bit) {
      st = W::dbl(c, st);
      if ((m >> bit) & 1) st = W::add(c, st, sq);
    }
    return W::store_point(c, W::add(c, st, vq));
  }
#endif
  // k_horner's dependent chain on one wavefront: the point spread over the wavefront, limb j of a
  // coordinate in lane j of a DPP row, four field products at once (curve/sw_wave.h).
  // BZ_SW_WAVE_HORNER=0: the form of rounds 2-3 -- doublings split over the lanes of each DPP quad
  // (curve/sw29_coop.h), the one addition per window computed redundantly by every lane.
#ifndef BZ_SW_WAVE_HORNER
#define BZ_SW_WAVE_HORNER 1
#endif
  static constexpr bool has_wave_horner = true;
#if defined(__HIPCC__)
  // out[w * stride] = 2^(bits w) g for w < windows by one wavefront (see ed25519_msm::wave_chain)
  __device__ static void wave_chain(point* out, u64 stride, const point& g, u32 windows, u32 bits) {
    using W = sww::wave<G29>;
    const typename W::ctx c = W::make_ctx(sww::wave_scratch());
    u32 st = W::load_point_value(c, g);
#pragma unroll 1
    for (u32 w = 0;; ++w) {
      const point p = W::store_point(c, st);
      if (c.lane == 0) out[static_cast<u64>(w) * stride] = p;
      if (w + 1 >= windows) break;
#pragma unroll 1
      for (u32 k = 0; k < bits; ++k) st = W::dbl(c, st);
    }
  }
  __device__ static point wave_horner(point acc, bool have_acc, point* window_sums,
                                      u32 stride, u32 num_windows, u32 window_bits) {
#if BZ_SW_WAVE_HORNER
    return sww::wave<G29>::horner(sww::wave_scratch(), acc, have_acc, window_sums, stride,
                                  num_windows, window_bits);
#else
    const u32 role = threadIdx.x & 3;
    u32 i = num_windows;
    if (!have_acc) {
      acc = window_sums[(num_windows - 1) * stride];
      i = num_windows - 1;
    }
    while (i-- > 0) {
      for (u32 k = 0; k < window_bits; ++k) acc = sw29_coop::dbl_coop4<G29>(acc, role);
      acc = G29::add(acc, window_sums[i * stride]);
    }
    return acc;
#endif
  }
#endif

  // k_horner's window fold: an addition split over the four lanes of a DPP quad (curve/sw29_coop.h)
  static constexpr bool has_coop_add = true;
#if defined(__HIPCC__)
  __device__ static point add_coop4(const point& a, const point& b, u32 role) {
    return sw29_coop::add_coop4<G29>(a, b, role);
  }
#endif

  // shared inversions (Montgomery's trick over a workgroup)
  using F29 = typename G29::F;
  using batch_fe = typename G29::fe;
  BZ_HD static batch_fe batch_one() { return F29::one(); }
  BZ_HD static batch_fe batch_mul(const batch_fe& a, const batch_fe& b) { return F29::mul(a, b); }
  BZ_HD static batch_fe batch_z(const point& p, bool& is_identity) {
    is_identity = F29::is_zero(p.Z);
    return is_identity ? F29::one() : p.Z;
  }
  // every lane inverts the same value: uniform control flow, the cost of one lane
  BZ_HD static batch_fe batch_wave_invert(const batch_fe& z) { return F29::invert(z); }
  // compact table entry {X, Y} in the ABI's Montgomery form, identity = {X[N-1] = 2^64 - 1, Y = R}
  // (sxt/curve_bng1/type/compact_element.h:26-38)
  static constexpr size_t compact_size = 16 * N64;
  BZ_HD static void store_compact(u8* out, const point& p, const batch_fe& zinv, bool is_identity) {
    u64* o = reinterpret_cast<u64*>(out);
    if (is_identity) {
      const typename G64::F::fe one = G64::F::one();
      for (int k = 0; k < N64; ++k) {
        o[k] = k == N64 - 1 ? ~u64{0} : 0;
        o[N64 + k] = one.v[k];
      }
      return;
    }
    F29::to_mont64(o, F29::mul(p.X, zinv));
    F29::to_mont64(o + N64, F29::mul(p.Y, zinv));
  }

  BZ_HD static point identity() { return G29::identity(); }
  BZ_HD static point add(const point& a, const point& b) { return G29::add(a, b); }
  BZ_HD static point dbl_n(const point& a, int k) { return G29::dbl_n(a, k); }
  BZ_HD static void accumulate(point& acc, const addend& q, bool negate) {
    if (G29::is_identity_addend(q)) return;
    acc = G29::template add_mixed<accumulate_pinned>(acc, G29::unpack(q), negate);
  }
  // k_accumulate's pipeline (see ed25519_msm::stage); the identity test is folded to one word here,
  // so nothing of the packed row is read after this point
  struct operand {
    typename G29::affine a;
    u32 nonzero; // 0: the identity's all-zero row
  };
  BZ_HD static operand stage(const addend& q) {
    u64 any = 0;
#pragma unroll
    for (int i = 0; i < N64; ++i) any |= q.x[i] | q.y[i];
    operand o{G29::unpack(q), static_cast<u32>(any) | static_cast<u32>(any >> 32)};
    F29::pin(o.a.x);
    F29::pin(o.a.y);
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(o.nonzero));
#endif
    return o;
  }
  // (k_accumulate only: `acc` is the identity or a result of this function, see add_mixed_acc)
  BZ_HD static void accumulate(point& acc, const operand& q, bool negate) {
    if (q.nonzero == 0) return;
    acc = G29::template add_mixed_acc<accumulate_pinned>(acc, q.a, negate);
  }
  // identity + (+-q) of a segment's first entry: the affine point itself (k_accumulate)
  static constexpr bool first_pinned = false;
  static constexpr bool has_split_add = false;
  BZ_HD static point first(const operand& q, bool negate) {
    if (q.nonzero == 0) return G29::identity();
    return G29::lift_acc(q.a, negate);
  }
  BZ_HD static addend make_addend(const void* api_generators, u64 i) {
    const api_affine& g = static_cast<const api_affine*>(api_generators)[i];
    return G29::pack(G29::affine_from_mont64(g.X, g.Y, g.infinity != 0));
  }
  // handle generators: ABI projective element -> affine (one inversion, ABI-form arithmetic)
  BZ_HD static addend addend_from_api_projective(const void* projective, u64 i) {
    typename G64::affine a;
    const bool inf = G64::to_affine(a, static_cast<const api_projective*>(projective)[i]);
    return G29::pack(G29::affine_from_mont64(a.x.v, a.y.v, inf));
  }
  BZ_HD static point point_from_api_projective(const void* projective, u64 i) {
    return G29::from_point64(static_cast<const api_projective*>(projective)[i]);
  }
  // caller generator ({X, Y, infinity} affine) -> engine point
  BZ_HD static point point_from_api_generator(const void* api_generators, u64 i) {
    const api_affine& g = static_cast<const api_affine*>(api_generators)[i];
    if (g.infinity != 0) return G29::identity();
    return {F29::from_mont64(g.X), F29::from_mont64(g.Y), F29::one()};
  }
  // engine point and 1 / Z -> affine addend; the identity is the all-zero addend (window tables)
  BZ_HD static addend addend_from_point(const point& p, const batch_fe& zinv, bool is_identity) {
    typename G29::affine a{F29::mul(p.X, zinv), F29::mul(p.Y, zinv)};
    if (is_identity) a = {F29::zero(), F29::zero()};
    return G29::pack(a);
  }
  static constexpr u32 batch_points_per_lane = G29::N <= 9 ? 4 : 2;
  BZ_HD static void store_projective(u8* out, const point& p) {
    *reinterpret_cast<api_projective*>(out) = G29::to_point64(p);
  }
  // engine point -> caller generator layout ({X, Y, u8 infinity} at the C-ABI stride)
  BZ_HD static void store_api_generator(u8* out, const point& p) { encode_affine(out, p); }
  // {X, Y Montgomery, u8 infinity}; identity = {0, R, 1}
  BZ_HD static void encode_affine(u8* out, const point& p) {
    typename G64::affine a;
    const bool inf = G29::to_affine64(a, p);
    u64* o = reinterpret_cast<u64*>(out);
    for (int k = 0; k < N64; ++k) {
      o[k] = a.x.v[k];
      o[N64 + k] = a.y.v[k];
    }
    o[2 * N64] = inf ? 1 : 0; // infinity byte + zeroed struct padding
  }
};

struct bn254_msm : sw_msm_base<bn254_g1_29, 2> {
  static constexpr const char* reference_element_name = "N3sxt4cn1t10element_p2E";
  static constexpr const char* reference_compact_name = "N3sxt4cn1t15compact_elementE";
  static constexpr size_t output_size = 72; // sxt_bn254_g1
  BZ_HD static void encode(u8* out, const point& p) { encode_affine(out, p); }
};

struct grumpkin_msm : sw_msm_base<grumpkin_29, 3> {
  static constexpr const char* reference_element_name = "N3sxt4cgkt10element_p2E";
  static constexpr const char* reference_compact_name = "N3sxt4cgkt15compact_elementE";
  static constexpr size_t output_size = 72; // sxt_grumpkin
  BZ_HD static void encode(u8* out, const point& p) { encode_affine(out, p); }
};

struct bls12_381_msm : sw_msm_base<bls12_381_g1_28, 1> {
  static constexpr const char* reference_element_name = "N3sxt4cg1t10element_p2E";
  static constexpr const char* reference_compact_name = "N3sxt4cg1t15compact_elementE";
  static constexpr size_t output_size = 48; // sxt_bls12_381_g1_compressed
  BZ_HD static void encode(u8* out, const point& p) {
    bls12_381_g1::affine a;
    const bool inf = bls12_381_g1_28::to_affine64(a, p);
    bls12_381_g1_compress_affine(out, a, inf);
  }
};
} // namespace bz
