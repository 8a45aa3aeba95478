// TEST INFRASTRUCTURE ONLY -- never linked into or called from the product (blitzar_amd/).
//
// Thin extern "C" driver over the *reference's own* CPU code, compiled from the sources where
// they lie under /root/reference (nothing is copied into this repo).  It mirrors what
// `cpu_backend::compute_commitments` does (sxt/cbindings/backend/cpu_backend.cc:117-152):
//   mtxcrv::compute_multiexponentiation<T>  (sxt/multiexp/curve/multiexponentiation.h:128-142)
//   followed by the host canonicaliser (rsto::batch_compress / cg1o::batch_compress /
//   batch_to_element_affine).
// sxt/multiexp/curve/multiexponentiation.h itself cannot be included (it drags in the CUDA
// bucket methods, :40-47), so the three lines of its CPU entry point are restated in `msm<T>`.
//
// Built by oracle/ref/build_ref.py into oracle/_ref/libblitzar_ref.so.
#include <cstdint>
#include <cstring>
#include <vector>

#include "sxt/base/container/span.h"
#include "sxt/base/num/fast_random_number_generator.h"
#include "sxt/curve21/operation/add.h"
#include "sxt/curve21/operation/double.h"
#include "sxt/curve21/operation/neg.h"
#include "sxt/curve21/type/element_p3.h"
#include "sxt/curve_bng1/operation/add.h"
#include "sxt/curve_bng1/operation/double.h"
#include "sxt/curve_bng1/operation/neg.h"
#include "sxt/curve_bng1/random/element_affine.h"
#include "sxt/curve_bng1/type/conversion_utility.h"
#include "sxt/curve_bng1/type/element_affine.h"
#include "sxt/curve_bng1/type/element_p2.h"
#include "sxt/curve_g1/operation/add.h"
#include "sxt/curve_g1/operation/compression.h"
#include "sxt/curve_g1/operation/double.h"
#include "sxt/curve_g1/operation/neg.h"
#include "sxt/curve_g1/random/element_affine.h"
#include "sxt/curve_g1/type/compressed_element.h"
#include "sxt/curve_g1/type/conversion_utility.h"
#include "sxt/curve_g1/type/element_affine.h"
#include "sxt/curve_g1/type/element_p2.h"
#include "sxt/curve_gk/operation/add.h"
#include "sxt/curve_gk/operation/double.h"
#include "sxt/curve_gk/operation/neg.h"
#include "sxt/curve_gk/random/element_affine.h"
#include "sxt/curve_gk/type/conversion_utility.h"
#include "sxt/curve_gk/type/element_affine.h"
#include "sxt/curve_gk/type/element_p2.h"
#include "sxt/field12/operation/mul.h"
#include "sxt/field25/operation/mul.h"
#include "sxt/field51/operation/mul.h"
#include "sxt/field51/operation/sq.h"
#include "sxt/field51/operation/sub.h"
#include "sxt/field51/operation/invert.h"
#include "sxt/fieldgk/operation/mul.h"
#include "sxt/memory/management/managed_array.h"
#include "sxt/multiexp/base/exponent_sequence.h"
#include "sxt/multiexp/curve/multiexponentiation_cpu_driver.h"
#include "sxt/multiexp/curve/pippenger_multiproduct_solver.h"
#include "sxt/multiexp/pippenger/multiexponentiation.h"
#include "sxt/multiexp/pippenger2/partition_table.h"
#include "sxt/ristretto/base/byte_conversion.h"
#include "sxt/ristretto/operation/compression.h"
#include "sxt/ristretto/type/compressed_element.h"
#include "sxt/seqcommit/generator/base_element.h"

using namespace sxt;

namespace {
// same field order / offsets as sxt_sequence_descriptor (cbindings/blitzar_api.h:115-131)
struct seq_desc {
  uint8_t element_nbytes;
  uint64_t n;
  const uint8_t* data;
  int is_signed;
};

template <class T>
memmg::managed_array<T> msm(const T* generators, uint64_t num_generators, const seq_desc* descs,
                            uint32_t num_sequences) {
  std::vector<mtxb::exponent_sequence> seqs(num_sequences);
  for (uint32_t i = 0; i < num_sequences; ++i) {
    seqs[i].element_nbytes = descs[i].element_nbytes;
    seqs[i].n = descs[i].n;
    seqs[i].data = descs[i].data;
    seqs[i].is_signed = descs[i].is_signed;
  }
  mtxcrv::pippenger_multiproduct_solver<T> solver;
  mtxcrv::multiexponentiation_cpu_driver<T> driver{&solver};
  return mtxpi::compute_multiexponentiation(
             driver, {static_cast<const void*>(generators), num_generators, sizeof(T)},
             {seqs.data(), seqs.size()})
      .value()
      .template as_array<T>();
}

uint64_t max_n(const seq_desc* descs, uint32_t num_sequences) {
  uint64_t n = 0;
  for (uint32_t i = 0; i < num_sequences; ++i) {
    n = descs[i].n > n ? descs[i].n : n;
  }
  return n;
}
} // namespace

extern "C" {
//--------------------------------------------------------------------------------------------------
// curve25519 / ristretto255
//--------------------------------------------------------------------------------------------------
void ref_c25519_base_element(uint64_t* out20, uint64_t index) {
  c21t::element_p3 g;
  sqcgn::compute_base_element(g, index);
  std::memcpy(out20, &g, 160);
}

void ref_c25519_base_elements(uint64_t* out, uint64_t first, uint64_t n) {
  for (uint64_t i = 0; i < n; ++i) {
    ref_c25519_base_element(out + 20 * i, first + i);
  }
}

// one_commit(n) = g_0 + ... + g_{n-1}, accumulated left to right from the identity exactly as
// sqcgn::cpu_get_one_commit does (sxt/seqcommit/generator/cpu_one_commitments.cc:45-59; that TU
// itself pulls in the CUDA generator kernel, so its 4-line loop is restated here).
void ref_c25519_one_commit(uint64_t* out20, uint64_t n) {
  auto r = c21t::element_p3::identity();
  for (uint64_t i = 0; i < n; ++i) {
    c21t::element_p3 g;
    sqcgn::compute_base_element(g, i);
    c21o::add(r, r, g);
  }
  std::memcpy(out20, &r, 160);
}

void ref_c25519_compress(uint8_t* out32, const uint64_t* p20) {
  c21t::element_p3 p;
  std::memcpy(&p, p20, 160);
  rstb::to_bytes(out32, p);
}

int ref_c25519_decompress(uint64_t* out20, const uint8_t* s32) {
  c21t::element_p3 p;
  int rc = rstb::from_bytes(p, s32);
  std::memcpy(out20, &p, 160);
  return rc;
}

void ref_c25519_add(uint64_t* out20, const uint64_t* a20, const uint64_t* b20) {
  c21t::element_p3 a, b, r;
  std::memcpy(&a, a20, 160);
  std::memcpy(&b, b20, 160);
  c21o::add(r, a, b);
  std::memcpy(out20, &r, 160);
}

void ref_c25519_double(uint64_t* out20, const uint64_t* a20) {
  c21t::element_p3 a, r;
  std::memcpy(&a, a20, 160);
  c21o::double_element(r, a);
  std::memcpy(out20, &r, 160);
}

void ref_c25519_neg(uint64_t* out20, const uint64_t* a20) {
  c21t::element_p3 a, r;
  std::memcpy(&a, a20, 160);
  c21o::neg(r, a);
  std::memcpy(out20, &r, 160);
}

// raw (projective, non-canonical) MSM result, element_p3 per sequence
void ref_c25519_msm_p3(uint64_t* out, uint32_t num_sequences, const seq_desc* descs,
                       const uint64_t* generators) {
  auto res = msm<c21t::element_p3>(reinterpret_cast<const c21t::element_p3*>(generators),
                                   max_n(descs, num_sequences), descs, num_sequences);
  std::memcpy(out, res.data(), 160 * num_sequences);
}

// == cpu_backend::compute_commitments (cpu_backend.cc:117-122)
void ref_c25519_commit(uint8_t* out, uint32_t num_sequences, const seq_desc* descs,
                       const uint64_t* generators) {
  auto res = msm<c21t::element_p3>(reinterpret_cast<const c21t::element_p3*>(generators),
                                   max_n(descs, num_sequences), descs, num_sequences);
  std::vector<rstt::compressed_element> c(num_sequences);
  rsto::batch_compress(c, res);
  for (uint32_t i = 0; i < num_sequences; ++i) {
    std::memcpy(out + 32 * i, c[i].data(), 32);
  }
}

// f51 limb-level ops (raw 5x51 limbs in/out) for limb-for-limb checks
void ref_f51_mul(uint64_t* h, const uint64_t* f, const uint64_t* g) {
  f51t::element a{f[0], f[1], f[2], f[3], f[4]}, b{g[0], g[1], g[2], g[3], g[4]}, r;
  f51o::mul(r, a, b);
  std::memcpy(h, r.data(), 40);
}
void ref_f51_sq(uint64_t* h, const uint64_t* f) {
  f51t::element a{f[0], f[1], f[2], f[3], f[4]}, r;
  f51o::sq(r, a);
  std::memcpy(h, r.data(), 40);
}
void ref_f51_sub(uint64_t* h, const uint64_t* f, const uint64_t* g) {
  f51t::element a{f[0], f[1], f[2], f[3], f[4]}, b{g[0], g[1], g[2], g[3], g[4]}, r;
  f51o::sub(r, a, b);
  std::memcpy(h, r.data(), 40);
}
void ref_f51_invert(uint64_t* h, const uint64_t* f) {
  f51t::element a{f[0], f[1], f[2], f[3], f[4]}, r;
  f51o::invert(r, a);
  std::memcpy(h, r.data(), 40);
}

//--------------------------------------------------------------------------------------------------
// short-Weierstrass curves: one macro-free template per curve family
//--------------------------------------------------------------------------------------------------
#define REF_WEIERSTRASS(PFX, TNS, ONS, RNS, FNS, FOPS, NL)                                         \
  void ref_##PFX##_random_affine(void* out, uint64_t seed1, uint64_t seed2) {                      \
    basn::fast_random_number_generator rng{seed1, seed2};                                          \
    TNS::element_affine a;                                                                         \
    RNS::generate_random_element(a, rng);                                                          \
    std::memset(out, 0, sizeof(a));                                                                \
    std::memcpy(out, &a.X, sizeof(a.X));                                                           \
    std::memcpy(static_cast<uint8_t*>(out) + sizeof(a.X), &a.Y, sizeof(a.Y));                      \
    static_cast<uint8_t*>(out)[2 * sizeof(a.X)] = a.infinity;                                      \
  }                                                                                                \
  void ref_##PFX##_add_p2(uint64_t* out, const uint64_t* a, const uint64_t* b) {                   \
    TNS::element_p2 x, y, r;                                                                       \
    std::memcpy(&x, a, sizeof(x));                                                                 \
    std::memcpy(&y, b, sizeof(y));                                                                 \
    ONS::add(r, x, y);                                                                             \
    std::memcpy(out, &r, sizeof(r));                                                               \
  }                                                                                                \
  void ref_##PFX##_double_p2(uint64_t* out, const uint64_t* a) {                                   \
    TNS::element_p2 x, r;                                                                          \
    std::memcpy(&x, a, sizeof(x));                                                                 \
    ONS::double_element(r, x);                                                                     \
    std::memcpy(out, &r, sizeof(r));                                                               \
  }                                                                                                \
  void ref_##PFX##_to_affine(void* out, const uint64_t* a) {                                       \
    TNS::element_p2 x;                                                                             \
    std::memcpy(&x, a, sizeof(x));                                                                 \
    TNS::element_affine r;                                                                         \
    TNS::to_element_affine(r, x);                                                                  \
    std::memset(out, 0, sizeof(r));                                                                \
    std::memcpy(out, &r.X, sizeof(r.X));                                                           \
    std::memcpy(static_cast<uint8_t*>(out) + sizeof(r.X), &r.Y, sizeof(r.Y));                      \
    static_cast<uint8_t*>(out)[2 * sizeof(r.X)] = r.infinity;                                      \
  }                                                                                                \
  /* out[i] = affine(start + (i + 1) * step), i < count: the chain g_i = g_{i-1} + g_0 of big   */ \
  /* synthetic generator sets -- the reference's add and to_element_affine in a    */ \
  /* loop; `start` is overwritten with the last element so that a caller can go on               */ \
  void ref_##PFX##_generator_chain(void* out, uint64_t* start, const uint64_t* step,               \
                                   uint64_t count) {                                               \
    TNS::element_p2 acc, g;                                                                        \
    std::memcpy(&acc, start, sizeof(acc));                                                         \
    std::memcpy(&g, step, sizeof(g));                                                              \
    auto dst = static_cast<uint8_t*>(out);                                                         \
    for (uint64_t i = 0; i < count; ++i, dst += sizeof(TNS::element_affine)) {                     \
      TNS::element_p2 r;                                                                           \
      ONS::add(r, acc, g);                                                                         \
      TNS::element_affine a;                                                                       \
      TNS::to_element_affine(a, r);                                                                \
      std::memset(dst, 0, sizeof(a));                                                              \
      std::memcpy(dst, &a.X, sizeof(a.X));                                                         \
      std::memcpy(dst + sizeof(a.X), &a.Y, sizeof(a.Y));                                           \
      dst[2 * sizeof(a.X)] = a.infinity;                                                           \
      acc = r;                                                                                     \
    }                                                                                              \
    std::memcpy(start, &acc, sizeof(acc));                                                         \
  }                                                                                                \
  void ref_##PFX##_msm_p2(uint64_t* out, uint32_t num_sequences, const seq_desc* descs,            \
                          const void* generators_affine) {                                         \
    /* host loop affine -> projective as cbindings/pedersen.cc:126-128 / :157-159 / :188-190 */    \
    auto n = max_n(descs, num_sequences);                                                          \
    auto gens = static_cast<const TNS::element_affine*>(generators_affine);                        \
    std::vector<TNS::element_p2> gp(n);                                                            \
    for (uint64_t i = 0; i < n; ++i) {                                                             \
      TNS::to_element_p2(gp[i], gens[i]);                                                          \
    }                                                                                              \
    auto res = msm<TNS::element_p2>(gp.data(), n, descs, num_sequences);                           \
    std::memcpy(out, res.data(), sizeof(TNS::element_p2) * num_sequences);                         \
  }                                                                                                \
  void ref_##PFX##_field_mul(uint64_t* h, const uint64_t* f, const uint64_t* g) {                  \
    FNS::element a, b, r;                                                                          \
    std::memcpy(&a, f, 8 * NL);                                                                    \
    std::memcpy(&b, g, 8 * NL);                                                                    \
    FOPS::mul(r, a, b);                                                                            \
    std::memcpy(h, &r, 8 * NL);                                                                    \
  }

REF_WEIERSTRASS(bn254, cn1t, cn1o, cn1rn, f25t, f25o, 4)
REF_WEIERSTRASS(grumpkin, cgkt, cgko, cgkrn, fgkt, fgko, 4)
REF_WEIERSTRASS(bls12_381, cg1t, cg1o, cg1rn, f12t, f12o, 6)

// == cpu_backend::compute_commitments for bn254 / grumpkin (cpu_backend.cc:137-152): affine out,
// element_affine is {X, Y, u8 infinity} with sizeof == 72
#define REF_COMMIT_AFFINE(PFX, TNS)                                                                \
  void ref_##PFX##_commit(void* out, uint32_t num_sequences, const seq_desc* descs,                \
                          const void* generators_affine) {                                         \
    std::vector<uint64_t> p2(12 * num_sequences);                                                  \
    ref_##PFX##_msm_p2(p2.data(), num_sequences, descs, generators_affine);                        \
    std::vector<TNS::element_affine> res(num_sequences);                                           \
    TNS::batch_to_element_affine(                                                                  \
        res, {reinterpret_cast<const TNS::element_p2*>(p2.data()), num_sequences});                \
    static_assert(sizeof(TNS::element_affine) == 72);                                              \
    std::memset(out, 0, 72 * num_sequences);                                                       \
    for (uint32_t i = 0; i < num_sequences; ++i) {                                                 \
      auto o = static_cast<uint8_t*>(out) + 72 * i;                                                \
      std::memcpy(o, &res[i].X, 32);                                                               \
      std::memcpy(o + 32, &res[i].Y, 32);                                                          \
      o[64] = res[i].infinity;                                                                     \
    }                                                                                              \
  }
REF_COMMIT_AFFINE(bn254, cn1t)
REF_COMMIT_AFFINE(grumpkin, cgkt)

// == cpu_backend::compute_commitments for bls12-381 (cpu_backend.cc:127-132): 48-byte compressed.
// `generators_affine` uses the reference's 104-byte element_affine stride.
void ref_bls12_381_commit(uint8_t* out, uint32_t num_sequences, const seq_desc* descs,
                          const void* generators_affine) {
  static_assert(sizeof(cg1t::element_affine) == 104);
  std::vector<uint64_t> p2(18 * num_sequences);
  ref_bls12_381_msm_p2(p2.data(), num_sequences, descs, generators_affine);
  std::vector<cg1t::compressed_element> c(num_sequences);
  cg1o::batch_compress(c, {reinterpret_cast<const cg1t::element_p2*>(p2.data()), num_sequences});
  for (uint32_t i = 0; i < num_sequences; ++i) {
    std::memcpy(out + 48 * i, c[i].data(), 48);
  }
}

void ref_bls12_381_compress(uint8_t* out48, const uint64_t* p18) {
  cg1t::element_p2 p;
  std::memcpy(&p, p18, sizeof(p));
  cg1t::compressed_element c;
  cg1o::compress(c, p);
  std::memcpy(out48, c.data(), 48);
}

//--------------------------------------------------------------------------------------------------
// fixed-base partition tables (sxt/multiexp/pippenger2/partition_table.h:36-98)
// `sums` receives (n / w) * 2^w compact elements U; generators are projective T, n % w == 0.
//--------------------------------------------------------------------------------------------------
void ref_c25519_partition_table(void* sums, unsigned w, const uint64_t* gens, unsigned n) {
  using U = c21t::compact_element;
  using T = c21t::element_p3;
  size_t m = (static_cast<size_t>(n) / w) << w;
  mtxpp2::compute_partition_table<U, T>({static_cast<U*>(sums), m}, w,
                                        {reinterpret_cast<const T*>(gens), n});
}
#define REF_PARTITION_TABLE(PFX, TNS)                                                              \
  void ref_##PFX##_partition_table(void* sums, unsigned w, const uint64_t* gens, unsigned n) {     \
    using U = TNS::compact_element;                                                                \
    using T = TNS::element_p2;                                                                     \
    size_t m = (static_cast<size_t>(n) / w) << w;                                                  \
    mtxpp2::compute_partition_table<U, T>({static_cast<U*>(sums), m}, w,                           \
                                          {reinterpret_cast<const T*>(gens), n});                  \
  }
REF_PARTITION_TABLE(bn254, cn1t)
REF_PARTITION_TABLE(grumpkin, cgkt)
REF_PARTITION_TABLE(bls12_381, cg1t)
} // extern "C"
