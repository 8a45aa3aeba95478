// Inner-product argument over curve25519 (Bulletproofs protocol 2 with a public b vector), the
// next consumer of the MSM engine after the commitment calls.
// Reference: cbindings/inner_product_proof.cc:96-167, sxt/proof/inner_product/
// {proof_computation,cpu_driver,gpu_driver,fold,generator_fold,verification_computation}.cc.
#pragma once

#include <cstdint>

#include "blitzar_amd/csrc/api/state.h"

namespace bz::proof {
// Both run under the api lock of `st` (taken by the caller) on st.backend.
//   l_vector / r_vector: ceil(log2 n) compressed ristretto points each; transcript: 203 bytes,
//   updated in place; a_vector / b_vector: n scalars of 32 bytes; generators are the built-in ones
//   [offset, offset + np) with Q = generator offset + np, np = 2^ceil(log2 n).
void prove_inner_product(api_state& st, u8* l_vector, u8* r_vector, u8* ap_value, void* transcript,
                         u64 n, u64 generators_offset, const u8* a_vector, const u8* b_vector);
bool verify_inner_product(api_state& st, void* transcript, u64 n, u64 generators_offset,
                          const u8* b_vector, const u8* product, const void* a_commit,
                          const u8* l_vector, const u8* r_vector, const u8* ap_value);

// services of api/capi.hip the prover needs
// generators [offset, offset + n) as raw extended coordinates on the host (cache + derivation)
void host_builtin_generators_unlocked(api_state& st, ed_point* out, u64 n, u64 offset);
// canonical commitment of ONE 32-byte column against HOST generators in ABI layout (the backend's
// normal Pedersen path, sharding included); the api lock is held by the caller
void commit_column_unlocked(api_state& st, u8* out32, const u8* scalars, u64 n,
                            const ed_point* generators);
} // namespace bz::proof
