// TEST INFRASTRUCTURE ONLY -- never linked into or called from the product (blitzar_amd/).
//
// Sumcheck oracle: the reference's own prover prfsk::prove_sum with its own cpu_driver<T>
// (sxt/proof/sumcheck/{proof_computation,cpu_driver,polynomial_utility}.h, header templates,
// host-clean) over its own field types, driven exactly as cpu_backend::prove_sumcheck does
// (sxt/cbindings/backend/cpu_backend.cc:73-112): a callback transcript, field selected by id
// (0 = curve25519 scalar field s25t::element, 1 = Grumpkin base field fgkt::element).
#include <cstdint>
#include <utility>

#include "sxt/base/container/span.h"
#include "sxt/base/num/ceil_log2.h"
#include "sxt/execution/async/future.h"
#include "sxt/fieldgk/realization/field.h"
#include "sxt/proof/sumcheck/cpu_driver.h"
#include "sxt/proof/sumcheck/proof_computation.h"
#include "sxt/proof/sumcheck/sumcheck_transcript.h"
#include "sxt/scalar25/realization/field.h"

using namespace sxt;

namespace {
// sxt/cbindings/backend/callback_sumcheck_transcript.h:27-45 (that header is only a few lines
// over the abstract transcript; restated to keep the cbindings backend out of the closure)
template <class T> class callback_transcript final : public prfsk::sumcheck_transcript<T> {
public:
  using callback_t = void (*)(T* r, void* context, const T* polynomial, unsigned polynomial_len);
  callback_transcript(callback_t f, void* context) noexcept : f_{f}, context_{context} {}
  void init(size_t, size_t) noexcept override {}
  void round_challenge(T& r, basct::cspan<T> polynomial) noexcept override {
    f_(&r, context_, polynomial.data(), static_cast<unsigned>(polynomial.size()));
  }

private:
  callback_t f_;
  void* context_;
};

template <class T>
void prove(void* polynomials, void* evaluation_point, const void* mles, const void* product_table,
           const unsigned* product_terms, unsigned n, unsigned num_mles, unsigned num_products,
           unsigned num_product_terms, unsigned round_degree, void* callback, void* context) {
  auto num_variables = static_cast<size_t>(std::max(basn::ceil_log2(n), 1));
  callback_transcript<T> transcript{
      reinterpret_cast<typename callback_transcript<T>::callback_t>(callback), context};
  basct::span<T> polynomials_span{static_cast<T*>(polynomials), (round_degree + 1u) * num_variables};
  basct::span<T> evaluation_point_span{static_cast<T*>(evaluation_point), num_variables};
  basct::cspan<T> mles_span{static_cast<const T*>(mles), static_cast<size_t>(n) * num_mles};
  basct::cspan<std::pair<T, unsigned>> product_table_span{
      static_cast<const std::pair<T, unsigned>*>(product_table), num_products};
  basct::cspan<unsigned> product_terms_span{product_terms, num_product_terms};
  prfsk::cpu_driver<T> drv;
  auto fut = prfsk::prove_sum<T>(polynomials_span, evaluation_point_span, transcript, drv, mles_span,
                                 product_table_span, product_terms_span, n);
  (void)fut;
}
} // namespace

extern "C" {
// == sxt_prove_sumcheck on the cpu backend; sizeof(std::pair<T, unsigned>) is the stride of
// product_table (40 bytes for both fields)
void ref_prove_sumcheck(void* polynomials, void* evaluation_point, unsigned field_id,
                        const void* mles, const void* product_table, const unsigned* product_terms,
                        unsigned n, unsigned num_mles, unsigned num_products,
                        unsigned num_product_terms, unsigned round_degree, void* callback,
                        void* context) {
  if (field_id == 0) {
    prove<s25t::element>(polynomials, evaluation_point, mles, product_table, product_terms, n,
                         num_mles, num_products, num_product_terms, round_degree, callback, context);
  } else {
    prove<fgkt::element>(polynomials, evaluation_point, mles, product_table, product_terms, n,
                         num_mles, num_products, num_product_terms, round_degree, callback, context);
  }
}
unsigned ref_sumcheck_product_stride(unsigned field_id) {
  return field_id == 0 ? sizeof(std::pair<s25t::element, unsigned>)
                       : sizeof(std::pair<fgkt::element, unsigned>);
}
} // extern "C"
