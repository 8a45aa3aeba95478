// Latency-oriented edwards25519 arithmetic for the one dependent chain at the end of an MSM
// (Horner over windows, ~250 doublings that nothing can run beside): ONE point spread over a whole
// wavefront.
//
// Layout: the wavefront is 4 DPP rows of 16 lanes.  A field element is 16 limbs of 16 bits
// (radix 2^16, 2^256 = 38 mod p), limb j in lane j of a row; the four rows hold the coordinates
// X, Y, Z, T of the point.  A field product is computed by all 16 lanes of a row at once, lane j
// owning column j of the cyclic convolution
//     c_j = sum_i u_i * v'_(j-i),          v' = v with the wrapped limbs multiplied by 38
//   * u_i: the row's 16 limbs, re-read from LDS as broadcast 128-bit loads (1 write + 4 reads),
//   * v'_(j-i): v rotated by i lanes, `v_mul_u32_u24 ... row_ror:i` by a per-lane constant (38 in
//     the lanes the rotation wrapped into, 1 elsewhere): 15 independent instructions,
//   * 16 v_mad_u64_u32 per lane, then three carry rounds (shift, row_ror:1, x38 into lane 0).
// The four rows run four products at once, which is exactly the shape of the extended-coordinate
// formulas: a doubling is 4 squarings then 4 products, an addition 4 + 4 products; between the two
// rounds every lane fetches limb j of all four results with one 128-bit LDS load and the linear
// combinations cost one instruction per limb.  A doubling is ~150 instructions per lane instead of
// the ~1300 of a lone lane (config 2, MI355X: k_horner 0.41 -> 0.21 ms together with the
// lane-parallel inverse square root of the encoding below).  What one instruction costs when a
// single wavefront runs: tools/ubench/tail_latency.hip (v_mad_u64_u32 4.0 ns, an LDS write ->
// read round trip 53 ns).
//
// Bounds (checked by interval propagation in tools/models/ed16_wave_model.py, run by
// tests/test_ed16_wave_model.py): products leave
// limbs < 2^16 + 64; the rotated operand must stay < 2^24 / 38 = 2^18.75 and column sums < 2^48.
// Subtractions add the limb-wise multiple 3p (every limb >= 98301).
//
// Device only, gfx950.
#pragma once

#include "blitzar_amd/csrc/curve/ed29.h"

#if defined(__HIPCC__)
namespace bz {
namespace ed16w {

// LDS scratch of the wavefront running the chain
struct scratch {
  alignas(16) u32 bcast[64]; // [row][limb]: the u operands of the products in flight
  alignas(16) u32 xch[64];   // [limb][row]: results of a round, read back as one 128-bit load
  alignas(16) u32 io[64];    // [row][limb]: hand-over to / from the 9 x 29-bit form
};

// orders this wavefront's LDS traffic (lanes read what other lanes of the wave wrote)
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// value of the previous lane of the row (lane 0 gets lane 15's).  Written as assembly so that
// the DPP operand stays where it is put (hipcc's own folding of DPP moves into VOP2 instructions
// miscomputed in curve/sw29_coop.h); the s_nop covers the VALU-write -> DPP-read hazard, which the
// compiler does not track through inline assembly.
__device__ __forceinline__ u32 row_ror1(u32 v) {
  u32 r;
  asm("s_nop 1\n\tv_mov_b32_dpp %0, %1 row_ror:1 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v));
  return r;
}

// row_ror1(v) * m as one v_mul_u32_u24 with a DPP source; v and m below 2^24
__device__ __forceinline__ u32 row_ror1_mul24(u32 v, u32 m) {
  u32 r;
  asm("s_nop 1\n\tv_mul_u32_u24_dpp %0, %1, %2 row_ror:1 row_mask:0xf bank_mask:0xf"
      : "=v"(r)
      : "v"(v), "v"(m));
  return r;
}

struct lane_ctx {
  u32 lane, row, j;
  u32 m38;    // 38 in lane 0 of a row, 1 elsewhere: the 2^256 = 38 wrap of a carry entering lane 0
  u32 m[16];  // m[i] = 38 in lanes j < i (where a rotation by i wrapped), 1 elsewhere
  u32 p3;     // limb j of 3p
  scratch* lds;
};

__device__ __forceinline__ lane_ctx make_ctx(scratch* lds) {
  lane_ctx c;
  c.lane = threadIdx.x & 63;
  c.row = c.lane >> 4;
  c.j = c.lane & 15;
  c.m38 = c.j == 0 ? 38u : 1u;
#pragma unroll
  for (u32 i = 0; i < 16; ++i) c.m[i] = c.j < i ? 38u : 1u;
  const u32 pl = c.j == 0 ? 0xffedu : (c.j == 15 ? 0x7fffu : 0xffffu);
  c.p3 = 3 * pl;
  c.lds = lds;
  return c;
}

// vs[i] = v'_(j-i) for i = 1..15: fifteen independent DPP multiplies (two blocks: an asm statement
// takes at most 30 operands)
__device__ __forceinline__ void rotations(const lane_ctx& c, u32 v, u32 vs[16]) {
  vs[0] = v;
  u32 r1, r2, r3, r4, r5, r6, r7, r8, r9, r10, r11, r12, r13, r14, r15;
  asm("s_nop 1\n\t"
      "v_mul_u32_u24_dpp %0, %[v], %[m1] row_ror:1 row_mask:0xf bank_mask:0xf\n\t"
      "v_mul_u32_u24_dpp %1, %[v], %[m2] row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
      "v_mul_u32_u24_dpp %2, %[v], %[m3] row_ror:3 row_mask:0xf bank_mask:0xf\n\t"
      "v_mul_u32_u24_dpp %3, %[v], %[m4] row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
      "v_mul_u32_u24_dpp %4, %[v], %[m5] row_ror:5 row_mask:0xf bank_mask:0xf\n\t"
      "v_mul_u32_u24_dpp %5, %[v], %[m6] row_ror:6 row_mask:0xf bank_mask:0xf\n\t"
      "v_mul_u32_u24_dpp %6, %[v], %[m7] row_ror:7 row_mask:0xf bank_mask:0xf\n\t"
      "v_mul_u32_u24_dpp %7, %[v], %[m8] row_ror:8 row_mask:0xf bank_mask:0xf"
      : "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5), "=&v"(r6), "=&v"(r7), "=&v"(r8)
      : [v] "v"(v), [m1] "v"(c.m[1]), [m2] "v"(c.m[2]), [m3] "v"(c.m[3]), [m4] "v"(c.m[4]),
        [m5] "v"(c.m[5]), [m6] "v"(c.m[6]), [m7] "v"(c.m[7]), [m8] "v"(c.m[8]));
  asm("s_nop 1\n\t"
      "v_mul_u32_u24_dpp %0, %[v], %[m9] row_ror:9 row_mask:0xf bank_mask:0xf\n\t"
      "v_mul_u32_u24_dpp %1, %[v], %[m10] row_ror:10 row_mask:0xf bank_mask:0xf\n\t"
      "v_mul_u32_u24_dpp %2, %[v], %[m11] row_ror:11 row_mask:0xf bank_mask:0xf\n\t"
      "v_mul_u32_u24_dpp %3, %[v], %[m12] row_ror:12 row_mask:0xf bank_mask:0xf\n\t"
      "v_mul_u32_u24_dpp %4, %[v], %[m13] row_ror:13 row_mask:0xf bank_mask:0xf\n\t"
      "v_mul_u32_u24_dpp %5, %[v], %[m14] row_ror:14 row_mask:0xf bank_mask:0xf\n\t"
      "v_mul_u32_u24_dpp %6, %[v], %[m15] row_ror:15 row_mask:0xf bank_mask:0xf"
      : "=&v"(r9), "=&v"(r10), "=&v"(r11), "=&v"(r12), "=&v"(r13), "=&v"(r14), "=&v"(r15)
      : [v] "v"(v), [m9] "v"(c.m[9]), [m10] "v"(c.m[10]), [m11] "v"(c.m[11]), [m12] "v"(c.m[12]),
        [m13] "v"(c.m[13]), [m14] "v"(c.m[14]), [m15] "v"(c.m[15]));
  vs[1] = r1, vs[2] = r2, vs[3] = r3, vs[4] = r4, vs[5] = r5, vs[6] = r6, vs[7] = r7, vs[8] = r8;
  vs[9] = r9, vs[10] = r10, vs[11] = r11, vs[12] = r12, vs[13] = r13, vs[14] = r14, vs[15] = r15;
}

// the 16 limbs of row `src_row`, as last written with `broadcast_store`
__device__ __forceinline__ void broadcast_store(const lane_ctx& c, u32 u) {
  c.lds->bcast[c.lane] = u;
  wave_lds_sync();
}
__device__ __forceinline__ void broadcast_load(const lane_ctx& c, u32 src_row, u32 ui[16]) {
  const uint4* up = reinterpret_cast<const uint4*>(&c.lds->bcast[src_row * 16]);
  const uint4 q0 = up[0], q1 = up[1], q2 = up[2], q3 = up[3];
  ui[0] = q0.x, ui[1] = q0.y, ui[2] = q0.z, ui[3] = q0.w;
  ui[4] = q1.x, ui[5] = q1.y, ui[6] = q1.z, ui[7] = q1.w;
  ui[8] = q2.x, ui[9] = q2.y, ui[10] = q2.z, ui[11] = q2.w;
  ui[12] = q3.x, ui[13] = q3.y, ui[14] = q3.z, ui[15] = q3.w;
}

// column j of u * v mod p, carried.  u limbs < 2^19.3, v limbs < 2^18.75; result limbs < 2^16 + 64
__device__ __forceinline__ u32 mul_columns(const lane_ctx& c, const u32 ui[16], u32 v) {
  u32 vs[16];
  rotations(c, v, vs);
  u64 acc = static_cast<u64>(ui[0]) * vs[0];
#pragma unroll
  for (int i = 1; i < 16; ++i) acc += static_cast<u64>(ui[i]) * vs[i];
  // carries: column sums < 2^48
  const u32 lo = static_cast<u32>(acc) & 0xffffu;
  const u32 hi = static_cast<u32>(acc >> 16);
  const u64 x = static_cast<u64>(row_ror1(hi)) * c.m38 + lo;
  const u32 lo2 = static_cast<u32>(x) & 0xffffu;
  const u32 hi2 = static_cast<u32>(x >> 16); // < 2^24
  const u32 y = lo2 + row_ror1_mul24(hi2, c.m38);
  return (y & 0xffffu) + row_ror1_mul24(y >> 16, c.m38);
}

// per row: u * v mod p
__device__ __forceinline__ u32 fmul(const lane_ctx& c, u32 u, u32 v) {
  broadcast_store(c, u);
  u32 ui[16];
  broadcast_load(c, c.row, ui);
  return mul_columns(c, ui, v);
}

// limb j of the four rows' values
__device__ __forceinline__ uint4 exchange(const lane_ctx& c, u32 v) {
  c.lds->xch[c.j * 4 + c.row] = v;
  wave_lds_sync();
  return *reinterpret_cast<const uint4*>(&c.lds->xch[c.j * 4]);
}

__device__ __forceinline__ u32 by_row(const lane_ctx& c, u32 r0, u32 r1, u32 r2, u32 r3) {
  const u32 lo = (c.row & 1) ? r1 : r0;
  const u32 hi = (c.row & 1) ? r3 : r2;
  return (c.row & 2) ? hi : lo;
}

// state: this lane's limb of (X | Y | Z | T by row).  Returns 2P up to the projective factor -1.
__device__ __forceinline__ u32 dbl(const lane_ctx& c, u32 state) {
  // round 1: rows 0..2 square their coordinate, row 3 forms X * Y (T is not an input)
  broadcast_store(c, state);
  u32 ui[16];
  broadcast_load(c, c.row == 3 ? 0 : c.row, ui);
  const u32 y_limb = c.lds->bcast[16 + c.j];
  const uint4 s = exchange(c, mul_columns(c, ui, c.row == 3 ? y_limb : state)); // X^2 Y^2 Z^2 XY
  const u32 e = 2 * s.w;                    // 2 X Y
  const u32 h = s.x + s.y;                  // X^2 + Y^2
  const u32 g = s.y + c.p3 - s.x;           // Y^2 - X^2
  const u32 f = s.x + 2 * s.z + c.p3 - s.y; // X^2 + 2 Z^2 - Y^2
  // (-X3, -Y3, -Z3, -T3) = (e f, g h, f g, e h)
  return fmul(c, by_row(c, e, g, f, e), by_row(c, f, h, g, h));
}

// P + Q, `cached` = this lane's limb of Q as (Y+X | Y-X | Z | 2dT by row), limbs < 2^16
__device__ __forceinline__ u32 add_cached(const lane_ctx& c, u32 state, u32 cached) {
  const uint4 p = exchange(c, state); // X Y Z T
  const u32 u = by_row(c, p.y + p.x, p.y + c.p3 - p.x, p.z, p.w);
  const uint4 m = exchange(c, fmul(c, u, cached)); // a b zz c
  const u32 d = 2 * m.z;
  const u32 ez = d + m.w, et = d + c.p3 - m.w, ex = m.x + c.p3 - m.y, ey = m.x + m.y;
  // (X3, Y3, Z3, T3) = (ex et, ey ez, ez et, ex ey)
  return fmul(c, by_row(c, ex, ey, ez, ex), by_row(c, et, ez, et, ey));
}

// this lane's limb of four 256-bit values stored as 8 little-endian words each
__device__ __forceinline__ u32 load_words(const lane_ctx& c, const u32* words) {
  return (words[8 * c.row + (c.j >> 1)] >> (16 * (c.j & 1))) & 0xffffu;
}

__device__ __forceinline__ u32 load_point(const lane_ctx& c, const ed29_point& p) {
  if (c.lane == 0) {
    ed29::pack_words(c.lds->io, p.X);
    ed29::pack_words(c.lds->io + 8, p.Y);
    ed29::pack_words(c.lds->io + 16, p.Z);
    ed29::pack_words(c.lds->io + 24, p.T);
  }
  wave_lds_sync();
  const u32 v = load_words(c, c.lds->io);
  wave_lds_sync();
  return v;
}

__device__ __forceinline__ u32 identity(const lane_ctx& c) {
  return (c.j == 0 && (c.row == 1 || c.row == 2)) ? 1u : 0u; // (0, 1, 1, 0)
}

// 16 limbs (each < 2^17) of one row -> 9 x 29-bit form
__device__ __forceinline__ fe29 gather_row(const u32* limbs) {
  u32 w[8];
  u64 carry = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const u64 t = carry + limbs[2 * k] + (static_cast<u64>(limbs[2 * k + 1]) << 16);
    w[k] = static_cast<u32>(t);
    carry = t >> 32;
  }
  fe29 h = ed29::unpack_words(w);
  h.v[0] += 38 * static_cast<u32>(carry); // 2^256 = 38
  return h;
}

// the point, in every lane
__device__ __forceinline__ ed29_point store_point(const lane_ctx& c, u32 state) {
  c.lds->io[c.lane] = state;
  wave_lds_sync();
  ed29_point p;
  p.X = gather_row(c.lds->io);
  p.Y = gather_row(c.lds->io + 16);
  p.Z = gather_row(c.lds->io + 32);
  p.T = gather_row(c.lds->io + 48);
  wave_lds_sync();
  return p;
}

// z^((p - 5) / 8), the exponentiation inside the ristretto inverse square root: 252 dependent
// squarings, each one row-parallel product (the four rows compute the same thing).  Every lane of
// the wavefront passes the same z and receives the result.
__device__ __forceinline__ u32 sqn(const lane_ctx& c, u32 x, int n) {
  for (int i = 0; i < n; ++i) x = fmul(c, x, x);
  return x;
}

__device__ __forceinline__ fe29 pow22523(const lane_ctx& c, const fe29& z_in) {
  if (c.lane == 0) ed29::pack_words(c.lds->io, z_in);
  wave_lds_sync();
  const u32 z = (c.lds->io[c.j >> 1] >> (16 * (c.j & 1))) & 0xffffu;
  wave_lds_sync();
  // the chain of f29::pow_2_250_m1 / f29::pow22523
  const u32 z2 = fmul(c, z, z);
  const u32 z9 = fmul(c, z, sqn(c, z2, 2));
  const u32 z11 = fmul(c, z2, z9);
  const u32 z2_5 = fmul(c, z9, fmul(c, z11, z11));
  const u32 z2_10 = fmul(c, sqn(c, z2_5, 5), z2_5);
  const u32 z2_20 = fmul(c, sqn(c, z2_10, 10), z2_10);
  const u32 z2_40 = fmul(c, sqn(c, z2_20, 20), z2_20);
  const u32 z2_50 = fmul(c, sqn(c, z2_40, 10), z2_10);
  const u32 z2_100 = fmul(c, sqn(c, z2_50, 50), z2_50);
  const u32 z2_200 = fmul(c, sqn(c, z2_100, 100), z2_100);
  const u32 z2_250 = fmul(c, sqn(c, z2_200, 50), z2_50);
  const u32 r = fmul(c, sqn(c, z2_250, 2), z);
  c.lds->io[c.lane] = r;
  wave_lds_sync();
  const fe29 out = gather_row(c.lds->io);
  wave_lds_sync();
  return out;
}

// the wavefront's scratch (one per workgroup: a single wavefront runs the chain)
__device__ __forceinline__ scratch* wave_scratch() {
  __shared__ scratch lds;
  return &lds;
}
} // namespace ed16w
} // namespace bz
#endif
