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
// Runs on st.backend; GPU backend: on devices[0], whose lease the caller holds and passes in -- it
// is given up around every call of `callback` (the caller's transcript may call back into the
// library) and the proof's tables live in device memory of the call's own.  `callback` has the
// signature void (FIELD* r, void* context, const FIELD* polynomial, unsigned polynomial_length)
void prove_sumcheck(api_state& st, void* polynomials, void* evaluation_point, unsigned field_id,
                    const sumcheck_inputs& inputs, void* callback, void* context,
                    api_state::device_lease* lease = nullptr);
} // namespace bz::proof
