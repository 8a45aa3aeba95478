// TEST INFRASTRUCTURE.  What a binding generator (bindgen for blitzar-sys, rust/blitzar-sys/build.rs)
// sees of a header: every struct's size, alignment and field offsets, every enumerator, and the exact
// type of every exported function -- printed, so that the same program compiled once against the
// REFERENCE's header (-I/root/reference, BZ_API_HEADER="cbindings/blitzar_api.h") and once against
// include/blitzar_api.h must print the same lines (tests/test_boundary_pins.py).  Taking each
// function's address also makes the link fail if libblitzar_amd.so lacks a symbol.
#include <cstddef>
#include <cstdio>
#include <typeinfo>

#include BZ_API_HEADER

#define SHOW_STRUCT(T) std::printf("struct %s size %zu align %zu\n", #T, sizeof(T), alignof(T))
#define SHOW_FIELD(T, f)                                                                           \
  std::printf("  %s.%s offset %zu size %zu\n", #T, #f, offsetof(T, f), sizeof(((T*)nullptr)->f))
#define SHOW_FN(f)                                                                                 \
  std::printf("fn %s : %s%s\n", #f, typeid(decltype(&f)).name(),                                   \
              reinterpret_cast<const void*>(&f) != nullptr ? "" : " (null)")

int main() {
  std::printf("SXT_CPU_BACKEND %d SXT_GPU_BACKEND %d\n", SXT_CPU_BACKEND, SXT_GPU_BACKEND);
  std::printf("curves %d %d %d %d\n", SXT_CURVE_RISTRETTO255, SXT_CURVE_BLS_381, SXT_CURVE_BN_254,
              SXT_CURVE_GRUMPKIN);
  std::printf("fields %d %d\n", SXT_FIELD_SCALAR255, SXT_FIELD_GRUMPKIN);
  SHOW_STRUCT(sxt_config);
  SHOW_FIELD(sxt_config, backend);
  SHOW_FIELD(sxt_config, num_precomputed_generators);
  SHOW_STRUCT(sxt_sequence_descriptor);
  SHOW_FIELD(sxt_sequence_descriptor, element_nbytes);
  SHOW_FIELD(sxt_sequence_descriptor, n);
  SHOW_FIELD(sxt_sequence_descriptor, data);
  SHOW_FIELD(sxt_sequence_descriptor, is_signed);
  SHOW_STRUCT(sxt_ristretto255_compressed);
  SHOW_FIELD(sxt_ristretto255_compressed, ristretto_bytes);
  SHOW_STRUCT(sxt_bls12_381_g1_compressed);
  SHOW_FIELD(sxt_bls12_381_g1_compressed, g1_bytes);
  SHOW_STRUCT(sxt_curve25519_scalar);
  SHOW_FIELD(sxt_curve25519_scalar, bytes);
  SHOW_STRUCT(sxt_transcript);
  SHOW_FIELD(sxt_transcript, bytes);
  SHOW_STRUCT(sxt_ristretto255);
  SHOW_FIELD(sxt_ristretto255, X);
  SHOW_FIELD(sxt_ristretto255, Y);
  SHOW_FIELD(sxt_ristretto255, Z);
  SHOW_FIELD(sxt_ristretto255, T);
  SHOW_STRUCT(sxt_bls12_381_g1);
  SHOW_FIELD(sxt_bls12_381_g1, X);
  SHOW_FIELD(sxt_bls12_381_g1, Y);
  SHOW_STRUCT(sxt_bls12_381_g1_p2);
  SHOW_FIELD(sxt_bls12_381_g1_p2, X);
  SHOW_FIELD(sxt_bls12_381_g1_p2, Y);
  SHOW_FIELD(sxt_bls12_381_g1_p2, Z);
  SHOW_STRUCT(sxt_bn254_g1);
  SHOW_FIELD(sxt_bn254_g1, X);
  SHOW_FIELD(sxt_bn254_g1, Y);
  SHOW_FIELD(sxt_bn254_g1, infinity);
  SHOW_STRUCT(sxt_bn254_g1_p2);
  SHOW_FIELD(sxt_bn254_g1_p2, X);
  SHOW_FIELD(sxt_bn254_g1_p2, Y);
  SHOW_FIELD(sxt_bn254_g1_p2, Z);
  SHOW_STRUCT(sxt_grumpkin);
  SHOW_FIELD(sxt_grumpkin, X);
  SHOW_FIELD(sxt_grumpkin, Y);
  SHOW_FIELD(sxt_grumpkin, infinity);
  SHOW_STRUCT(sxt_grumpkin_p2);
  SHOW_FIELD(sxt_grumpkin_p2, X);
  SHOW_FIELD(sxt_grumpkin_p2, Y);
  SHOW_FIELD(sxt_grumpkin_p2, Z);
  SHOW_STRUCT(sumcheck_descriptor);
  SHOW_FIELD(sumcheck_descriptor, mles);
  SHOW_FIELD(sumcheck_descriptor, product_table);
  SHOW_FIELD(sumcheck_descriptor, product_terms);
  SHOW_FIELD(sumcheck_descriptor, n);
  SHOW_FIELD(sumcheck_descriptor, num_mles);
  SHOW_FIELD(sumcheck_descriptor, num_products);
  SHOW_FIELD(sumcheck_descriptor, num_product_terms);
  SHOW_FIELD(sumcheck_descriptor, round_degree);
  SHOW_FN(sxt_init);
  SHOW_FN(sxt_curve25519_compute_pedersen_commitments);
  SHOW_FN(sxt_curve25519_compute_pedersen_commitments_with_generators);
  SHOW_FN(sxt_bls12_381_g1_compute_pedersen_commitments_with_generators);
  SHOW_FN(sxt_bn254_g1_uncompressed_compute_pedersen_commitments_with_generators);
  SHOW_FN(sxt_grumpkin_uncompressed_compute_pedersen_commitments_with_generators);
  SHOW_FN(sxt_ristretto255_get_generators);
  SHOW_FN(sxt_curve25519_get_one_commit);
  SHOW_FN(sxt_curve25519_prove_inner_product);
  SHOW_FN(sxt_curve25519_verify_inner_product);
  SHOW_FN(sxt_multiexp_handle_new);
  SHOW_FN(sxt_multiexp_handle_new_from_file);
  SHOW_FN(sxt_multiexp_handle_write_to_file);
  SHOW_FN(sxt_multiexp_handle_free);
  SHOW_FN(sxt_fixed_multiexponentiation);
  SHOW_FN(sxt_fixed_packed_multiexponentiation);
  SHOW_FN(sxt_fixed_vlen_multiexponentiation);
  SHOW_FN(sxt_prove_sumcheck);
  return 0;
}
