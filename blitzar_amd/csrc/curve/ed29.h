// edwards25519 group law on the 9 x 29-bit field (field/f29.h): the representation the gfx950
// MSM kernels compute in.  Formulas: extended twisted-Edwards coordinates, a = -1, unified
// addition with a precomputed addend (Hisil-Wong-Carter-Dawson 2008, section 3.1; the same law as the
// reference's sxt/curve21/operation/add.cc:41-55, which only ever becomes observable through
// canonical ristretto bytes).  Every function documents the limb bound B (max limb / 2^29) of
// what it returns so that f29::mul's contract (B(f) B(g) <= 6) can be checked by reading.
#pragma once

#include "blitzar_amd/csrc/curve/ed25519.h"
#include "blitzar_amd/csrc/field/f29.h"

namespace bz {

// (X:Y:Z:T) with XY = ZT; all coordinates B ~ 1
struct ed29_point {
  fe29 X, Y, Z, T;
};

// resident addend (Y+X, Y-X, Z, 2dT), all B ~ 1 (weakly reduced when built); 144 bytes
struct ed29_cached {
  fe29 YpX, YmX, Z, T2d;
};

// resident addend of a generator set that is registered once and reused (built-in generators,
// bzamd_generators, sxt_multiexp_handle): normalised to Z = 1, (y+x, y-x, 2dxy), padded to one
// 128-byte line.  One field product and 80 bytes of gather traffic less per addition than
// ed29_cached; the normalisation costs an inversion per generator, once.
struct alignas(16) ed29_niels {
  fe29 YpX, YmX, T2d;
  u32 pad[5];
};
static_assert(sizeof(ed29_niels) == 128);

// ed29_cached as it sits in HBM: the four coordinates as canonical 256-bit little-endian integers,
// 128 bytes = exactly one line per gather.  (144-byte limb rows need nothing unpacked -- 1250 -> 1211
// VALU instructions per addition -- but are two line requests per gather: measured in round 5,
// k_accumulate 0.636 -> 0.940 ms at 2^20 rows, profiles/round5_ab_limb_addends.log.)
struct alignas(16) ed29_cached_packed {
  u32 w[32];
};
static_assert(sizeof(ed29_cached_packed) == 128);

namespace ed29 {
BZ_HD void pack_words(u32* w, const fe29& f) {
  u64 q[4];
  f29::to_words(q, f);
  for (int i = 0; i < 4; ++i) {
    w[2 * i] = static_cast<u32>(q[i]);
    w[2 * i + 1] = static_cast<u32>(q[i] >> 32);
  }
}

// eight 32-bit words of a value < 2^256 -> nine 29-bit limbs (all below 2^29)
BZ_HD fe29 unpack_words(const u32* w) {
  fe29 h;
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    const int bit = 29 * i, j = bit >> 5, sh = bit & 31;
    u32 v = w[j] >> sh;
    if (sh > 3 && j + 1 < 8) v |= w[j + 1] << (32 - sh);
    h.v[i] = v & f29::kMask;
  }
  return h;
}

BZ_HD ed29_point identity() { return {f29::zero(), f29::one(), f29::one(), f29::zero()}; }

BZ_HD ed29_point from_ed(const ed_point& p) {
  return {f29::from_fe51(p.X), f29::from_fe51(p.Y), f29::from_fe51(p.Z), f29::from_fe51(p.T)};
}

BZ_HD ed_point to_ed(const ed29_point& p) {
  return {f29::to_fe51(p.X), f29::to_fe51(p.Y), f29::to_fe51(p.Z), f29::to_fe51(p.T)};
}

BZ_HD ed29_cached to_cached(const ed29_point& p) {
  ed29_cached c;
  c.YpX = f29::weak_reduce(f29::add(p.Y, p.X));
  c.YmX = f29::weak_reduce(f29::sub(p.Y, p.X));
  c.Z = p.Z;
  c.T2d = f29::mul(p.T, f29::const_2d());
  return c;
}

BZ_HD ed29_cached cached_from_ed(const ed_point& p) { return to_cached(from_ed(p)); }

BZ_HD ed29_cached_packed pack(const ed29_cached& c) {
  ed29_cached_packed m;
  pack_words(m.w, c.YpX);
  pack_words(m.w + 8, c.YmX);
  pack_words(m.w + 16, c.Z);
  pack_words(m.w + 24, c.T2d);
  return m;
}

// row `row` of a table of packed addends with its first two 32-byte pieces (Y+X | Y-X) exchanged
// when `negate`: by address, as 16-byte loads (k_accumulate's gather)
struct alignas(16) u32x4 {
  u32 v[4];
};
BZ_HD ed29_cached_packed gather_signed(const ed29_cached_packed* table, u32 row, bool negate) {
  const u32x4* src = reinterpret_cast<const u32x4*>(table + row);
  const u32 a = negate ? 2u : 0u;
  ed29_cached_packed q;
  u32x4* dst = reinterpret_cast<u32x4*>(&q);
  dst[0] = src[a];
  dst[1] = src[a + 1];
  dst[2] = src[2 - a];
  dst[3] = src[3 - a];
#pragma unroll
  for (int i = 4; i < 8; ++i) dst[i] = src[i];
  return q;
}

BZ_HD ed29_cached unpack(const ed29_cached_packed& m) {
  return {unpack_words(m.w), unpack_words(m.w + 8), unpack_words(m.w + 16), unpack_words(m.w + 24)};
}

// p + q, or p - q when `negate` (one code path: the sign only selects operands, so lanes of a
// wavefront with different digit signs do not diverge)
BZ_HD ed29_point add_cached(const ed29_point& p, const ed29_cached& q, bool negate) {
  // -q = (Y-X, Y+X, Z, -2dT)
  const fe29 qa = f29::select(q.YpX, q.YmX, negate);
  const fe29 qb = f29::select(q.YmX, q.YpX, negate);
  const fe29 qt = f29::cneg_xad(q.T2d, negate);                // B 2
  const fe29 ypx = f29::add(p.Y, p.X);               // B 2
  const fe29 ymx = f29::sub(p.Y, p.X);               // B 3
  const fe29 a = f29::mul(ypx, qa);                  // 2 * 1
  const fe29 b = f29::mul(ymx, qb);                  // 3 * 1
  const fe29 c = f29::mul(p.T, qt);                  // 1 * 2
  const fe29 zz = f29::mul(p.Z, q.Z);                // 1 * 1
  const fe29 d = f29::add(zz, zz);                   // B 2
  const fe29 ez = f29::add(d, c);                    // B 3
  const fe29 et = f29::weak_reduce(f29::sub(d, c));  // B 4 -> 1
  const fe29 ex = f29::sub(a, b);                    // B 3
  const fe29 ey = f29::add(a, b);                    // B 2
  ed29_point r;
  r.X = f29::mul(ex, et); // 3 * 1
  r.Y = f29::mul(ey, ez); // 2 * 3
  r.Z = f29::mul(ez, et); // 3 * 1
  r.T = f29::mul(ex, ey); // 3 * 2
  return r;
}

// the same with q.YpX / q.YmX already exchanged when `negate` (k_accumulate gathers the two 32-byte
// pieces of the packed row in the order the digit's sign asks for): only 2dT still depends on it
BZ_HD ed29_point add_cached_presigned(const ed29_point& p, const ed29_cached& q, bool negate) {
  const fe29 qt = f29::cneg_xad(q.T2d, negate);      // B 2
  const fe29 ypx = f29::add(p.Y, p.X);               // B 2
  const fe29 ymx = f29::sub(p.Y, p.X);               // B 3
  const fe29 a = f29::mul(ypx, q.YpX);               // 2 * 1
  const fe29 b = f29::mul(ymx, q.YmX);               // 3 * 1
  const fe29 c = f29::mul(p.T, qt);                  // 1 * 2
  const fe29 zz = f29::mul(p.Z, q.Z);                // 1 * 1
  const fe29 d = f29::add(zz, zz);                   // B 2
  const fe29 ez = f29::add(d, c);                    // B 3
  const fe29 et = f29::weak_reduce(f29::sub(d, c));  // B 4 -> 1
  const fe29 ex = f29::sub(a, b);                    // B 3
  const fe29 ey = f29::add(a, b);                    // B 2
  ed29_point r;
  r.X = f29::mul(ex, et); // 3 * 1
  r.Y = f29::mul(ey, ez); // 2 * 3
  r.Z = f29::mul(ez, et); // 3 * 1
  r.T = f29::mul(ex, ey); // 3 * 2
  return r;
}

// add_cached_presigned in two halves: everything that reads p and q (four products), then the four
// products of the result.  Between them neither operand is live (k_accumulate issues the next gather
// there).
struct ed29_completed {
  fe29 ex, ey, ez, et;
};
BZ_HD ed29_completed add_cached_presigned_front(const ed29_point& p, const ed29_cached& q,
                                                bool negate) {
  const fe29 qt = f29::cneg_xad(q.T2d, negate);      // B 2
  const fe29 ypx = f29::add(p.Y, p.X);               // B 2
  const fe29 ymx = f29::sub(p.Y, p.X);               // B 3
  const fe29 a = f29::mul(ypx, q.YpX);               // 2 * 1
  const fe29 b = f29::mul(ymx, q.YmX);               // 3 * 1
  const fe29 c = f29::mul(p.T, qt);                  // 1 * 2
  const fe29 zz = f29::mul(p.Z, q.Z);                // 1 * 1
  const fe29 d = f29::add(zz, zz);                   // B 2
  ed29_completed m;
  m.ez = f29::add(d, c);                             // B 3
  m.et = f29::weak_reduce(f29::sub(d, c));           // B 4 -> 1
  m.ex = f29::sub(a, b);                             // B 3
  m.ey = f29::add(a, b);                             // B 2
  return m;
}
BZ_HD ed29_point add_cached_back(const ed29_completed& m) {
  ed29_point r;
  r.X = f29::mul(m.ex, m.et); // 3 * 1
  r.Y = f29::mul(m.ey, m.ez); // 2 * 3
  r.Z = f29::mul(m.ez, m.et); // 3 * 1
  r.T = f29::mul(m.ex, m.ey); // 3 * 2
  return r;
}

// identity + q (q.YpX / q.YmX already exchanged when `negate`, as add_cached_presigned takes them)
// without an addition: (2X : 2Y : 2Z : 2T) with 2T = (2dT) / d -- one field product instead of
// eight.  k_accumulate's first entry of a segment: every lane of the wavefront holds the identity
// there.  All coordinates B ~ 1.
BZ_HD ed29_point from_cached_presigned(const ed29_cached& q, bool negate) {
  ed29_point r;
  r.X = f29::weak_reduce(f29::sub(q.YpX, q.YmX));                     // B 3 -> 1
  r.Y = f29::weak_reduce(f29::add(q.YpX, q.YmX));                     // B 2 -> 1
  r.Z = f29::weak_reduce(f29::add(q.Z, q.Z));                         // B 2 -> 1
  r.T = f29::mul(f29::cneg_xad(q.T2d, negate), f29::const_dinv());    // 2 * 1
  return r;
}

BZ_HD ed29_point add(const ed29_point& p, const ed29_point& q) {
  return add_cached(p, to_cached(q), false);
}

BZ_HD ed29_niels to_niels(const ed29_point& p) {
  const fe29 zinv = f29::invert(p.Z);
  const fe29 x = f29::mul(p.X, zinv);
  const fe29 y = f29::mul(p.Y, zinv);
  ed29_niels n;
  n.YpX = f29::weak_reduce(f29::add(y, x));
  n.YmX = f29::weak_reduce(f29::sub(y, x));
  n.T2d = f29::mul(f29::mul(x, y), f29::const_2d());
  for (int i = 0; i < 5; ++i) n.pad[i] = 0;
  return n;
}

// p + q (q negated when `negate`) for a Z = 1 addend: 7 field products
BZ_HD ed29_point add_niels(const ed29_point& p, const ed29_niels& q, bool negate) {
  const fe29 qa = f29::select(q.YpX, q.YmX, negate);
  const fe29 qb = f29::select(q.YmX, q.YpX, negate);
  const fe29 qt = f29::cneg_xad(q.T2d, negate);                // B 2
  const fe29 a = f29::mul(f29::add(p.Y, p.X), qa);             // 2 * 1
  const fe29 b = f29::mul(f29::sub(p.Y, p.X), qb);             // 3 * 1
  const fe29 c = f29::mul(p.T, qt);                            // 1 * 2
  const fe29 d = f29::add(p.Z, p.Z);                           // B 2
  const fe29 ez = f29::add(d, c);                              // B 3
  const fe29 et = f29::weak_reduce(f29::sub(d, c));            // B 4 -> 1
  const fe29 ex = f29::sub(a, b);                              // B 3
  const fe29 ey = f29::add(a, b);                              // B 2
  ed29_point r;
  r.X = f29::mul(ex, et);
  r.Y = f29::mul(ey, ez);
  r.Z = f29::mul(ez, et);
  r.T = f29::mul(ex, ey);
  return r;
}

// identity + q (q negated when `negate`) for a Z = 1 addend: (2x : 2y : 2 : 2xy)
BZ_HD ed29_point from_niels(const ed29_niels& q, bool negate) {
  const fe29 qa = f29::select(q.YpX, q.YmX, negate);
  const fe29 qb = f29::select(q.YmX, q.YpX, negate);
  ed29_point r;
  r.X = f29::weak_reduce(f29::sub(qa, qb));
  r.Y = f29::weak_reduce(f29::add(qa, qb));
  r.Z = f29::zero();
  r.Z.v[0] = 2;
  r.T = f29::mul(f29::cneg_xad(q.T2d, negate), f29::const_dinv());
  return r;
}

// 2p; T is only produced when `want_t` (intermediate doublings of a 2^k chain do not need it)
BZ_HD ed29_point dbl(const ed29_point& p, bool want_t = true) {
  const fe29 xx = f29::sq(p.X);
  const fe29 yy = f29::sq(p.Y);
  const fe29 zz = f29::sq(p.Z);
  const fe29 s = f29::sq(f29::add(p.X, p.Y));                              // (B 2)^2
  const fe29 ey = f29::add(yy, xx);                                        // B 2   yy + xx
  const fe29 ez = f29::sub(yy, xx);                                        // B 3   yy - xx
  const fe29 ex = f29::weak_reduce(f29::sub(f29::sub(s, yy), xx));         // B 5 -> 1
  const fe29 zz2x = f29::add(f29::add(zz, zz), xx);                        // B 3
  const fe29 et = f29::weak_reduce(f29::sub(zz2x, yy));                    // 2 zz - (yy - xx), B 5 -> 1
  ed29_point r;
  r.X = f29::mul(ex, et); // 1 * 1
  r.Y = f29::mul(ey, ez); // 2 * 3
  r.Z = f29::mul(ez, et); // 3 * 1
  r.T = want_t ? f29::mul(ex, ey) : f29::zero(); // 1 * 2
  return r;
}

// 2^k p
BZ_HD ed29_point dbl_n(ed29_point p, int k) {
  for (int i = 0; i < k; ++i) p = dbl(p, i + 1 == k);
  return p;
}

BZ_HD ed29_point neg(const ed29_point& p) {
  return {f29::weak_reduce(f29::neg(p.X)), p.Y, p.Z, f29::weak_reduce(f29::neg(p.T))};
}
} // namespace ed29

// canonical ristretto255 encoding computed on the 29-bit field (same algorithm as
// ristretto::encode_words in curve/ed25519.h, reference sxt/ristretto/base/byte_conversion.cc:
// 74-129); the output bytes are canonical, hence identical.
namespace ristretto29 {
// `pow22523`: z -> z^((p - 5) / 8); a parameter so that k_horner can run the 252 dependent
// squarings lane-parallel (curve/ed16_wave.h)
template <class Pow>
BZ_HD bool sqrt_ratio_m1(fe29& x, const fe29& u, const fe29& v, Pow&& pow22523) {
  const fe29 sqrtm1 = f29::const_sqrtm1();
  const fe29 v3 = f29::mul(f29::sq(v), v);
  x = f29::mul(f29::mul(f29::sq(v3), u), v);
  x = pow22523(x);
  x = f29::mul(f29::mul(x, v3), u);
  const fe29 vxx = f29::mul(f29::sq(x), v);
  const fe29 m_root_check = f29::sub(vxx, u);
  const fe29 p_root_check = f29::add(vxx, u);
  const fe29 f_root_check = f29::add(vxx, f29::mul(u, sqrtm1));
  const bool has_m_root = f29::is_zero(m_root_check);
  const bool has_p_root = f29::is_zero(p_root_check);
  const bool has_f_root = f29::is_zero(f_root_check);
  const fe29 x_sqrtm1 = f29::mul(x, sqrtm1);
  f29::cmov(x, x_sqrtm1, has_p_root | has_f_root);
  x = f29::abs(x);
  return has_m_root | has_p_root;
}

BZ_HD bool sqrt_ratio_m1(fe29& x, const fe29& u, const fe29& v) {
  return sqrt_ratio_m1(x, u, v, [](const fe29& z) { return f29::pow22523(z); });
}

template <class Pow> BZ_HD void encode_words(u64 out[4], const ed29_point& p, Pow&& pow22523) {
  const fe29 one = f29::one();
  const fe29 u1 = f29::mul(f29::add(p.Z, p.Y), f29::sub(p.Z, p.Y)); // 2 * 3
  const fe29 u2 = f29::mul(p.X, p.Y);
  const fe29 u1_u2u2 = f29::mul(u1, f29::sq(u2));
  fe29 inv_sqrt;
  (void)sqrt_ratio_m1(inv_sqrt, one, u1_u2u2, pow22523);
  const fe29 den1 = f29::mul(inv_sqrt, u1);
  const fe29 den2 = f29::mul(inv_sqrt, u2);
  const fe29 z_inv = f29::mul(f29::mul(den1, den2), p.T);
  const fe29 ix = f29::mul(p.X, f29::const_sqrtm1());
  const fe29 iy = f29::mul(p.Y, f29::const_sqrtm1());
  const fe29 eden = f29::mul(den1, f29::const_invsqrtamd());
  const bool rotate = f29::is_negative(f29::mul(p.T, z_inv));
  fe29 x = p.X, y = p.Y, den_inv = den2;
  f29::cmov(x, iy, rotate);
  f29::cmov(y, ix, rotate);
  f29::cmov(den_inv, eden, rotate);
  y = f29::cneg(y, f29::is_negative(f29::mul(x, z_inv))); // B <= 2
  const fe29 s = f29::abs(f29::mul(den_inv, f29::sub(p.Z, f29::weak_reduce(y))));
  f29::to_words(out, s);
}

BZ_HD void encode_words(u64 out[4], const ed29_point& p) {
  encode_words(out, p, [](const fe29& z) { return f29::pow22523(z); });
}
} // namespace ristretto29

// bytes of the canonical encoding
namespace ristretto29 {
BZ_HD void encode(u8 out[32], const ed29_point& p) {
  u64 w[4];
  encode_words(w, p);
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 8; ++j) out[8 * i + j] = static_cast<u8>(w[i] >> (8 * j));
}
} // namespace ristretto29
} // namespace bz
