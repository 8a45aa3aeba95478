// Sumcheck prover behind sxt_prove_sumcheck (proof/sumcheck.hip).
#pragma once

#include "blitzar_amd/csrc/api/state.h"

namespace bz::proof {
// the fields of `struct sumcheck_descriptor` (cbindings/blitzar_api.h:147-181)
struct sumcheck_inputs {
  const void* mles;              // n x num_mles field elements, column-major
  const void* product_table;     // num_products x {element multiplier; unsigned product_length}
  const unsigned* product_terms; // MLE indices of every product, back to back
  unsigned n, num_mles, num_products, num_product_terms, round_degree;
};
// runs on st.backend under the api lock held by the caller; `callback` has the signature
// void (FIELD* r, void* context, const FIELD* polynomial, unsigned polynomial_length)
void prove_sumcheck(api_state& st, void* polynomials, void* evaluation_point, unsigned field_id,
                    const sumcheck_inputs& inputs, void* callback, void* context);
} // namespace bz::proof
