/* Drop-in C ABI of the MI355X-native MSM / Pedersen-commitment engine.
 *
 * Every declaration below replaces the declaration of the same name in the reference's
 * `cbindings/blitzar_api.h` (cited per entry as `ref:<line>`); struct layouts, argument order and
 * error behaviour are identical so that `blitzar-sys` (bindgen over the reference header,
 * rust/blitzar-sys/src/lib.rs:1-5) and C callers (example/cbindings1/main.cc) link against
 * `libblitzar_amd.so` unchanged.  Only the MSM / commitment path is implemented natively
 *  together with its two consumers in the same header, the inner-product
 * argument and the sumcheck prover.
 *
 * Error convention (reference: SXT_RELEASE_ASSERT -> std::abort, sxt/base/error/assert.h:51-58):
 * functions returning `void` abort the process on misuse; there is no errno and no exception
 * crosses the ABI.  All calls are blocking; a single caller thread is assumed.
 */
#ifndef BLITZAR_AMD_BLITZAR_API_H
#define BLITZAR_AMD_BLITZAR_API_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ref:25-34 */
#define SXT_CPU_BACKEND 1
#define SXT_GPU_BACKEND 2

#define SXT_CURVE_RISTRETTO255 0
#define SXT_CURVE_BLS_381 1
#define SXT_CURVE_BN_254 2
#define SXT_CURVE_GRUMPKIN 3

#define SXT_FIELD_SCALAR255 0
#define SXT_FIELD_GRUMPKIN 1

/* ref:37-40 */
struct sxt_config {
  int backend;
  uint64_t num_precomputed_generators;
};

/* ref:43-45  canonical ristretto255 encoding */
struct sxt_ristretto255_compressed {
  uint8_t ristretto_bytes[32];
};

/* ref:48-50  big-endian x, flag bits 7 compressed / 6 infinity / 5 y-largest */
struct sxt_bls12_381_g1_compressed {
  uint8_t g1_bytes[48];
};

/* ref:56-58 */
struct sxt_curve25519_scalar {
  uint8_t bytes[32];
};

/* ref:61-63 */
struct sxt_transcript {
  uint8_t bytes[203];
};

/* ref:66-71  edwards25519 extended coordinates, raw radix-2^51 limbs */
struct sxt_ristretto255 {
  uint64_t X[5];
  uint64_t Y[5];
  uint64_t Z[5];
  uint64_t T[5];
};

/* ref:74-77  affine, Montgomery limbs.  NOTE: the reference reads these arrays with the 104-byte
 * stride of its internal {X, Y, u8 infinity} type (cbindings/pedersen.cc:212-217); so do we. */
struct sxt_bls12_381_g1 {
  uint64_t X[6];
  uint64_t Y[6];
};

/* ref:80-84 */
struct sxt_bls12_381_g1_p2 {
  uint64_t X[6];
  uint64_t Y[6];
  uint64_t Z[6];
};

/* ref:87-91  affine, Montgomery limbs; identity = {0, R, infinity = 1} */
struct sxt_bn254_g1 {
  uint64_t X[4];
  uint64_t Y[4];
  uint8_t infinity;
};

/* ref:94-98 */
struct sxt_bn254_g1_p2 {
  uint64_t X[4];
  uint64_t Y[4];
  uint64_t Z[4];
};

/* ref:101-105 */
struct sxt_grumpkin {
  uint64_t X[4];
  uint64_t Y[4];
  uint8_t infinity;
};

/* ref:108-112 */
struct sxt_grumpkin_p2 {
  uint64_t X[4];
  uint64_t Y[4];
  uint64_t Z[4];
};

/* ref:115-131  one column of scalars: `n` little-endian integers of `element_nbytes` bytes
 * (1..32), row-major; two's complement when `is_signed` (then element_nbytes <= 16). */
struct sxt_sequence_descriptor {
  uint8_t element_nbytes;
  uint64_t n;
  const uint8_t* data;
  int is_signed;
};

/* ref:147-181  sumcheck polynomial  sum_p mult_p * prod_{j in terms_p} f_j(X_1 .. X_r):
 * mles = n x num_mles FIELD elements, column-major; product_table = num_products entries
 * {FIELD multiplier; unsigned product_length} (36 bytes for SXT_FIELD_SCALAR255, 40 for
 * SXT_FIELD_GRUMPKIN: the reference's std::pair<FIELD, unsigned>); product_terms = MLE indices */
struct sumcheck_descriptor {
  const void* mles;
  const void* product_table;
  const unsigned* product_terms;
  unsigned n;
  unsigned num_mles;
  unsigned num_products;
  unsigned num_product_terms;
  unsigned round_degree;
};

/* ref:184  precomputed state for multiexponentiations with fixed generators */
struct sxt_multiexp_handle;

/* ref:200  select backend (env BLITZAR_BACKEND=cpu|gpu overrides), precompute the first
 * `num_precomputed_generators` built-in generators.  0 on success, 1 for an unknown backend;
 * aborts on a null config, on re-initialisation, or when the GPU backend finds no device. */
int sxt_init(const struct sxt_config* config);

/* ref:243  commitments[i] = sum_j a_ij * g_{offset_generators + j} with the built-in generators */
void sxt_curve25519_compute_pedersen_commitments(struct sxt_ristretto255_compressed* commitments,
                                                 uint32_t num_sequences,
                                                 const struct sxt_sequence_descriptor* descriptors,
                                                 uint64_t offset_generators);

/* ref:284  same with caller generators (null generators = built-in generators at offset 0) */
void sxt_curve25519_compute_pedersen_commitments_with_generators(
    struct sxt_ristretto255_compressed* commitments, uint32_t num_sequences,
    const struct sxt_sequence_descriptor* descriptors, const struct sxt_ristretto255* generators);

/* ref:324 */
void sxt_bls12_381_g1_compute_pedersen_commitments_with_generators(
    struct sxt_bls12_381_g1_compressed* commitments, uint32_t num_sequences,
    const struct sxt_sequence_descriptor* descriptors, const struct sxt_bls12_381_g1* generators);

/* ref:364 */
void sxt_bn254_g1_uncompressed_compute_pedersen_commitments_with_generators(
    struct sxt_bn254_g1* commitments, uint32_t num_sequences,
    const struct sxt_sequence_descriptor* descriptors, const struct sxt_bn254_g1* generators);

/* ref:404 */
void sxt_grumpkin_uncompressed_compute_pedersen_commitments_with_generators(
    struct sxt_grumpkin* commitments, uint32_t num_sequences,
    const struct sxt_sequence_descriptor* descriptors, const struct sxt_grumpkin* generators);

/* ref:440  NOTE the reference *implementation* takes (generators, num_generators,
 * offset_generators) positionally (cbindings/get_generators.cc:32-33) although its header names
 * the 2nd/3rd parameters the other way round; positional behaviour is what callers rely on
 * (cbindings/get_generators.t.cc:42,62,69), so the 2nd argument is the COUNT here too.
 * Returns 1 when num_generators > 0 and generators is null, else 0. */
int sxt_ristretto255_get_generators(struct sxt_ristretto255* generators, uint64_t num_generators,
                                    uint64_t offset_generators);

/* ref:477  one_commit = g_0 + ... + g_{n-1} (identity for n = 0) */
int sxt_curve25519_get_one_commit(struct sxt_ristretto255* one_commit, uint64_t n);

/* ref:568, 606  inner-product argument over the built-in generators [generators_offset,
 * generators_offset + np), np = 2^ceil(log2 n), Q = generator generators_offset + np.  l_vector /
 * r_vector: ceil(log2 n) compressed points; the 203-byte transcript is updated in place.  Proof
 * bytes and transcript state are identical to the reference's. */
void sxt_curve25519_prove_inner_product(struct sxt_ristretto255_compressed* l_vector,
                                        struct sxt_ristretto255_compressed* r_vector,
                                        struct sxt_curve25519_scalar* ap_value,
                                        struct sxt_transcript* transcript, uint64_t n,
                                        uint64_t generators_offset,
                                        const struct sxt_curve25519_scalar* a_vector,
                                        const struct sxt_curve25519_scalar* b_vector);

int sxt_curve25519_verify_inner_product(struct sxt_transcript* transcript, uint64_t n,
                                        uint64_t generators_offset,
                                        const struct sxt_curve25519_scalar* b_vector,
                                        const struct sxt_curve25519_scalar* product,
                                        const struct sxt_ristretto255* a_commit,
                                        const struct sxt_ristretto255_compressed* l_vector,
                                        const struct sxt_ristretto255_compressed* r_vector,
                                        const struct sxt_curve25519_scalar* ap_value);

/* ref:631  `generators`: n projective elements of the curve (sxt_ristretto255 /
 * sxt_bls12_381_g1_p2 / sxt_bn254_g1_p2 / sxt_grumpkin_p2) */
struct sxt_multiexp_handle* sxt_multiexp_handle_new(unsigned curve_id, const void* generators,
                                                    unsigned n);

/* ref:641, 649  partition-table file: u32 window width, then the table of compact elements */
struct sxt_multiexp_handle* sxt_multiexp_handle_new_from_file(unsigned curve_id,
                                                              const char* filename);

void sxt_multiexp_handle_write_to_file(const struct sxt_multiexp_handle* handle,
                                       const char* filename);

/* ref:655 */
void sxt_multiexp_handle_free(struct sxt_multiexp_handle* handle);

/* ref:685  res[k] = sum_j s_kj * g_j; row j of `scalars` holds the num_outputs scalars of
 * element_num_bytes each, back to back.  Results are projective elements of the curve. */
void sxt_fixed_multiexponentiation(void* res, const struct sxt_multiexp_handle* handle,
                                   unsigned element_num_bytes, unsigned num_outputs, unsigned n,
                                   const uint8_t* scalars);

/* ref:712  bit-packed rows: output k owns output_bit_table[k] bits of every row */
void sxt_fixed_packed_multiexponentiation(void* res, const struct sxt_multiexp_handle* handle,
                                          const unsigned* output_bit_table, unsigned num_outputs,
                                          unsigned n, const uint8_t* scalars);

/* ref:741  as packed, output k only uses rows < output_lengths[k] (ascending) */
void sxt_fixed_vlen_multiexponentiation(void* res, const struct sxt_multiexp_handle* handle,
                                        const unsigned* output_bit_table,
                                        const unsigned* output_lengths, unsigned num_outputs,
                                        const uint8_t* scalars);

/* ref:766  polynomials: (round_degree + 1) x num_variables FIELD elements, column-major;
 * evaluation_point: num_variables FIELD elements; transcript_callback:
 *   void (FIELD* r, void* context, const FIELD* polynomial, unsigned polynomial_length)
 * draws the round challenge.  round_degree <= 8 here (Limits). */
void sxt_prove_sumcheck(void* polynomials, void* evaluation_point, unsigned field_id,
                        const struct sumcheck_descriptor* descriptor, void* transcript_callback,
                        void* transcript_context);

#ifdef __cplusplus
} /* extern "C" */
#endif

#endif /* BLITZAR_AMD_BLITZAR_API_H */
