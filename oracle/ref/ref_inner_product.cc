// TEST INFRASTRUCTURE ONLY -- never linked into or called from the product (blitzar_amd/).
//
// Inner-product argument oracle: the reference's OWN prover / verifier,
//   prfip::prove_inner_product / verify_inner_product (sxt/proof/inner_product/proof_computation.cc),
// its transcript (sxt/proof/transcript/*), scalar field (sxt/scalar25/*), scalar / generator folds
// (fold.cc, generator_fold.cc) and verification exponents (verification_computation.cc), compiled
// from /root/reference.  Only the computational `driver` is restated: the reference's
// cpu_driver.cc includes sxt/multiexp/curve/multiexponentiation.h, which drags in the CUDA bucket
// methods (:40-47), and s25o::inner_product's translation unit is CUDA too.  `oracle_driver` below
// follows cpu_driver.cc function by function (lines cited) with `msm<T>` -- the same three-line
// restatement of mtxcrv::compute_multiexponentiation as ref_driver.cc -- in place of that header,
// and s25o::inner_product is given its host definition (inner_product.cc:62-70) here.
//
// C entry points mirror cbindings/inner_product_proof.cc:96-167 (generators = built-in
// compute_base_element(offset + i), Q = generator np).
#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>

#include "sxt/base/container/span.h"
#include "sxt/base/num/ceil_log2.h"
#include "sxt/curve21/operation/add.h"
#include "sxt/curve21/operation/double.h"
#include "sxt/curve21/operation/neg.h"
#include "sxt/curve21/type/element_p3.h"
#include "sxt/execution/async/future.h"
#include "sxt/execution/schedule/scheduler.h"
#include "sxt/memory/management/managed_array.h"
#include "sxt/multiexp/base/exponent_sequence.h"
#include "sxt/multiexp/curve/multiexponentiation_cpu_driver.h"
#include "sxt/multiexp/curve/pippenger_multiproduct_solver.h"
#include "sxt/multiexp/pippenger/multiexponentiation.h"
#include "sxt/proof/inner_product/driver.h"
#include "sxt/proof/inner_product/fold.h"
#include "sxt/proof/inner_product/generator_fold.h"
#include "sxt/proof/inner_product/proof_computation.h"
#include "sxt/proof/inner_product/proof_descriptor.h"
#include "sxt/proof/inner_product/verification_computation.h"
#include "sxt/proof/inner_product/workspace.h"
#include "sxt/proof/transcript/transcript.h"
#include "sxt/ristretto/operation/compression.h"
#include "sxt/ristretto/type/compressed_element.h"
#include "sxt/scalar25/constant/max_bits.h"
#include "sxt/scalar25/operation/inv.h"
#include "sxt/scalar25/operation/mul.h"
#include "sxt/scalar25/operation/muladd.h"
#include "sxt/scalar25/type/element.h"
#include "sxt/seqcommit/generator/base_element.h"

using namespace sxt;

namespace sxt::s25o {
// host definition of s25o::inner_product (sxt/scalar25/operation/inner_product.cc:62-70; that TU
// also holds the CUDA reduction and cannot be compiled here)
void inner_product(s25t::element& res, basct::cspan<s25t::element> lhs,
                   basct::cspan<s25t::element> rhs) noexcept {
  auto n = std::min(lhs.size(), rhs.size());
  s25o::mul(res, lhs[0], rhs[0]);
  for (size_t i = 1; i < n; ++i) {
    s25o::muladd(res, lhs[i], rhs[i], res);
  }
}
} // namespace sxt::s25o

// the four symbols of skipped translation units (oracle/ref/build_ref.py SKIP) that the host-only
// scheduler and the logger reference: one host "device", a silent log sink
namespace sxt::basdv {
unsigned get_num_devices() noexcept { return 1; }
int get_device() noexcept { return 0; }
void set_device(int) noexcept {}
} // namespace sxt::basdv
namespace sxt::basl {
void info_impl(std::string_view) noexcept {}
} // namespace sxt::basl

namespace {
using p3 = c21t::element_p3;

// mtxcrv::compute_multiexponentiation<element_p3> for one 32-byte sequence
// (sxt/multiexp/curve/multiexponentiation.h:128-142)
p3 msm1(basct::cspan<p3> g_vector, basct::cspan<s25t::element> x_vector) {
  auto n = std::min(g_vector.size(), x_vector.size());
  mtxb::exponent_sequence seq{.element_nbytes = 32,
                              .n = n,
                              .data = reinterpret_cast<const uint8_t*>(x_vector.data()),
                              .is_signed = 0};
  mtxcrv::pippenger_multiproduct_solver<p3> solver;
  mtxcrv::multiexponentiation_cpu_driver<p3> driver{&solver};
  auto res = mtxpi::compute_multiexponentiation(
                 driver, {static_cast<const void*>(g_vector.data()), n, sizeof(p3)}, {&seq, 1})
                 .value()
                 .template as_array<p3>();
  return res[0];
}

// restatement of prfip::cpu_driver (sxt/proof/inner_product/cpu_driver.cc)
class oracle_driver final : public prfip::driver {
public:
  // cpu_driver.cc:104-117
  std::unique_ptr<prfip::workspace>
  make_workspace(const prfip::proof_descriptor& descriptor,
                 basct::cspan<s25t::element> a_vector) const noexcept override {
    auto res = std::make_unique<prfip::workspace>();
    res->descriptor = &descriptor;
    res->a_vector0 = a_vector;
    prfip::init_workspace(*res);
    return res;
  }

  // cpu_driver.cc:122-171
  xena::future<void> commit_to_fold(rstt::compressed_element& l_value,
                                    rstt::compressed_element& r_value,
                                    prfip::workspace& work) const noexcept override {
    basct::cspan<p3> g_vector;
    basct::cspan<s25t::element> a_vector, b_vector;
    select(g_vector, a_vector, b_vector, work);
    auto mid = g_vector.size() / 2;
    auto a_low = a_vector.subspan(0, mid);
    auto a_high = a_vector.subspan(mid);
    auto b_low = b_vector.subspan(0, mid);
    auto b_high = b_vector.subspan(mid);
    auto g_low = g_vector.subspan(0, mid);
    auto g_high = g_vector.subspan(mid);
    s25t::element c_values[2];
    s25o::inner_product(c_values[0], a_low, b_high);
    s25o::inner_product(c_values[1], a_high, b_low);
    p3 c_commits[2];
    for (int k = 0; k < 2; ++k) {
      c_commits[k] = msm1({work.descriptor->q_value, 1}, {&c_values[k], 1});
    }
    p3 l_value_p = msm1(g_high, a_low);
    c21o::add(l_value_p, l_value_p, c_commits[0]);
    rsto::compress(l_value, l_value_p);
    p3 r_value_p = msm1(g_low, a_high);
    c21o::add(r_value_p, r_value_p, c_commits[1]);
    rsto::compress(r_value, r_value_p);
    return xena::make_ready_future();
  }

  // cpu_driver.cc:176-213
  xena::future<void> fold(prfip::workspace& work, const s25t::element& x) const noexcept override {
    basct::cspan<p3> g_vector;
    basct::cspan<s25t::element> a_vector, b_vector;
    select(g_vector, a_vector, b_vector, work);
    auto mid = g_vector.size() / 2;
    ++work.round_index;
    s25t::element x_inv;
    s25o::inv(x_inv, x);
    prfip::fold_scalars(work.a_vector, a_vector, x, x_inv, mid);
    if (mid == 1) {
      return xena::make_ready_future();
    }
    prfip::fold_scalars(work.b_vector, b_vector, x_inv, x, mid);
    // cpu_driver.cc:44-55
    unsigned data[s25cn::max_bits_v];
    basct::span<unsigned> decomposition{data};
    prfip::decompose_generator_fold(decomposition, x_inv, x);
    for (size_t i = 0; i < mid; ++i) {
      prfip::fold_generators(work.g_vector[i], decomposition, g_vector[i], g_vector[mid + i]);
    }
    work.g_vector = work.g_vector.subspan(0, mid);
    return xena::make_ready_future();
  }

  // cpu_driver.cc:218-256
  xena::future<void> compute_expected_commitment(
      rstt::compressed_element& commit, const prfip::proof_descriptor& descriptor,
      basct::cspan<rstt::compressed_element> l_vector,
      basct::cspan<rstt::compressed_element> r_vector, basct::cspan<s25t::element> x_vector,
      const s25t::element& ap_value) const noexcept override {
    auto num_rounds = l_vector.size();
    auto np = descriptor.g_vector.size();
    auto num_exponents = 1 + np + 2 * num_rounds;
    std::vector<s25t::element> exponents(num_exponents);
    prfip::compute_verification_exponents(exponents, x_vector, ap_value, descriptor.b_vector);
    std::vector<p3> generators(num_exponents);
    auto iter = generators.data();
    *iter++ = *descriptor.q_value;
    iter = std::copy(descriptor.g_vector.begin(), descriptor.g_vector.end(), iter);
    for (auto& li : l_vector) {
      rsto::decompress(*iter++, li);
    }
    for (auto& ri : r_vector) {
      rsto::decompress(*iter++, ri);
    }
    p3 res = msm1(generators, exponents);
    rsto::compress(commit, res);
    return xena::make_ready_future();
  }

private:
  static void select(basct::cspan<p3>& g_vector, basct::cspan<s25t::element>& a_vector,
                     basct::cspan<s25t::element>& b_vector, const prfip::workspace& work) {
    if (work.round_index == 0) {
      g_vector = work.descriptor->g_vector;
      a_vector = work.a_vector0;
      b_vector = work.descriptor->b_vector;
    } else {
      g_vector = work.g_vector;
      a_vector = work.a_vector;
      b_vector = work.b_vector;
    }
  }
};

std::vector<p3> builtin_generators(uint64_t count, uint64_t offset) {
  std::vector<p3> g(count);
  for (uint64_t i = 0; i < count; ++i) {
    sqcgn::compute_base_element(g[i], offset + i);
  }
  return g;
}
} // namespace

extern "C" {
// prft::transcript{label}: the 203-byte Merlin state a caller hands to the C API
void ref_transcript_new(uint8_t* out203, const char* label, uint64_t label_len) {
  static_assert(sizeof(prft::transcript) == 203);
  prft::transcript t{std::string_view{label, label_len}};
  std::memcpy(out203, &t, 203);
}

// == sxt_curve25519_prove_inner_product (cbindings/inner_product_proof.cc:96-126) on the cpu backend
void ref_ip_prove(uint8_t* l_vector, uint8_t* r_vector, uint8_t* ap_value, uint8_t* transcript203,
                  uint64_t n, uint64_t generators_offset, const uint8_t* a_vector,
                  const uint8_t* b_vector) {
  auto n_lg2 = static_cast<size_t>(basn::ceil_log2(n));
  auto np = 1ull << n_lg2;
  auto gens = builtin_generators(np + 1, generators_offset);
  prfip::proof_descriptor descriptor{
      .b_vector = {reinterpret_cast<const s25t::element*>(b_vector), n},
      .g_vector = {gens.data(), np},
      .q_value = gens.data() + np};
  oracle_driver drv;
  auto fut = prfip::prove_inner_product(
      {reinterpret_cast<rstt::compressed_element*>(l_vector), n_lg2},
      {reinterpret_cast<rstt::compressed_element*>(r_vector), n_lg2},
      *reinterpret_cast<s25t::element*>(ap_value),
      *reinterpret_cast<prft::transcript*>(transcript203), drv, descriptor,
      {reinterpret_cast<const s25t::element*>(a_vector), n});
  xens::get_scheduler().run();
}

// == sxt_curve25519_verify_inner_product (cbindings/inner_product_proof.cc:131-167)
int ref_ip_verify(uint8_t* transcript203, uint64_t n, uint64_t generators_offset,
                  const uint8_t* b_vector, const uint8_t* product, const uint64_t* a_commit20,
                  const uint8_t* l_vector, const uint8_t* r_vector, const uint8_t* ap_value) {
  auto n_lg2 = static_cast<size_t>(basn::ceil_log2(n));
  auto np = 1ull << n_lg2;
  auto gens = builtin_generators(np + 1, generators_offset);
  prfip::proof_descriptor descriptor{
      .b_vector = {reinterpret_cast<const s25t::element*>(b_vector), n},
      .g_vector = {gens.data(), np},
      .q_value = gens.data() + np};
  oracle_driver drv;
  p3 a_commit;
  std::memcpy(&a_commit, a_commit20, 160);
  auto fut = prfip::verify_inner_product(
      *reinterpret_cast<prft::transcript*>(transcript203), drv, descriptor,
      *reinterpret_cast<const s25t::element*>(product), a_commit,
      {reinterpret_cast<const rstt::compressed_element*>(l_vector), n_lg2},
      {reinterpret_cast<const rstt::compressed_element*>(r_vector), n_lg2},
      *reinterpret_cast<const s25t::element*>(ap_value));
  xens::get_scheduler().run();
  return fut.value() ? 1 : 0;
}

// scalar25 helpers for building test inputs: res = <a, b> mod l (canonical)
void ref_s25_inner_product(uint8_t* res32, const uint8_t* a, const uint8_t* b, uint64_t n) {
  s25o::inner_product(*reinterpret_cast<s25t::element*>(res32),
                      {reinterpret_cast<const s25t::element*>(a), n},
                      {reinterpret_cast<const s25t::element*>(b), n});
}
} // extern "C"
