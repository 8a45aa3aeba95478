/* MI355X-native extensions next to the drop-in Blitzar ABI (include/blitzar_api.h).
 *
 * The reference API only accepts host buffers and re-uploads scalars and generators on every
 * call (sxt/multiexp/bucket_method/accumulation.h:68-71, bucket_method2/sum.h:103-107).  These
 * entry points are what SURVEY.md section 8(f) row 1 asks for: the same computation on operands that are
 * already resident in HBM, enqueued on a caller stream.  They are also what `bench.py` times
 * ("inputs already resident in HBM when the timed region starts").
 *
 * All pointers marked DEVICE must be valid on the current HIP device.  `stream` is a hipStream_t
 * passed as void* (NULL = the default stream).  Calls are asynchronous unless stated otherwise.
 *
 * Concurrency: the engine keeps one workspace per device.  Calls on one device are executed in the
 * order they were enqueued, whatever streams they arrive on (a call on another stream first waits
 * for everything enqueued so far on the stream of the previous call; do not destroy a stream the
 * engine may still have to wait for), and the host side of every call takes the device context's
 * lock, so several host threads may enqueue.  What can overlap is the tail of one call with the
 * next call (bzamd_pipeline_next below).  The blocking sxt_* entry points serialise on one
 * process-wide lock.
 */
#ifndef BLITZAR_AMD_BLITZAR_AMD_H
#define BLITZAR_AMD_BLITZAR_AMD_H

#include <stdint.h>

#include "blitzar_api.h"

#ifdef __cplusplus
extern "C" {
#endif

/* library / device introspection */
const char* bzamd_version(void);
int bzamd_device_count(void);
/* 0 = not initialised, SXT_CPU_BACKEND, SXT_GPU_BACKEND */
int bzamd_active_backend(void);
/* HIP devices the GPU backend drives from this process (0 before sxt_init / on the cpu backend).
 * sxt_init takes every visible device (the current one first), capped by the environment variable
 * BLITZAR_AMD_NUM_DEVICES; a blocking sxt_* call shards its columns / outputs (or the rows of a
 * single long column) over them, one host thread per device.  BLITZAR_AMD_FORCE_SHARDS=k makes k
 * logical devices out of the current physical one (testing the sharded paths on a one-GPU box). */
int bzamd_num_devices(void);
/* calls with fewer scalar bytes than this stay on one device (default 1 MiB) */
void bzamd_set_shard_min_bytes(uint64_t bytes);
/* A blocking call with host operands that stays on one device is cut into row chunks when that pays:
 * chunk k is committed to projective partials while chunk k + 1 crosses PCIe, one fold adds the
 * partials up (exact group addition: the same commitments).  0 = the cost model decides (default);
 * k = k chunks wherever the longest sequence has that many rows (tests). */
void bzamd_set_row_pipeline_chunks(uint32_t chunks);
/* rows of a sequence one pass of the engine takes (default 2^28, at most 2^31 - 1: the engine
 * indexes the rows of a pass with 31 bits).  A longer sequence -- the ABI's n is a uint64_t -- runs
 * in ceil(n / rows) passes over row ranges whose projective partial results are folded; the
 * canonical result is the same.  Tests lower it. */
void bzamd_set_max_rows_per_pass(uint64_t rows);
/* A fresh Merlin transcript with the application's domain-separation label: the 203 bytes a
 * caller hands to sxt_curve25519_prove_inner_product / _verify_inner_product (the reference leaves
 * their construction to the caller's Merlin implementation; Rust callers transmute
 * merlin::Transcript, prft::transcript{label} in C++). */
void bzamd_transcript_init(struct sxt_transcript* transcript, const char* label,
                           uint64_t label_len);
/* addition formula k_accumulate runs for curve25519 caller generators: 1 = Z = 1 addends (7 field
 * products, generators normalised per call by a batched inversion), 0 = projective (8 products) */
int bzamd_accumulate_form(void);
/* number of gfx950 kernel launches issued by this process so far (tests use it to prove that the
 * HIP path, not a host path, produced a result) */
uint64_t bzamd_kernel_launch_count(void);
/* the most blocking sxt_* calls that ever held devices of the GPU backend at the same time (calls
 * take per-device leases, not a process-wide lock: two host threads on a two-device backend run
 * side by side; tests use this to prove it) */
uint32_t bzamd_concurrent_calls_high_water(void);
/* 1 if the current device fetches code beyond its instruction cache more slowly than it executes it
 * (probed once per device; such devices run the bucket reduction of calls with few columns through
 * k_reduce_compact), 0 if not, -1 without an initialised GPU backend */
int bzamd_slow_instruction_fetch(void);
/* Issue rate of v_mad_u64_u32 -- the field products' one wide primitive, the binding bound of the
 * bucket accumulation -- on the current device, measured now: every SIMD holds 6 waves of 8
 * independent chains for about `target_ms` milliseconds, the fastest ~3 ms launch of the second half
 * of that load is what is reported.
 *   out[0] wave-instructions per second over the whole device
 *   out[1] effective shader clock in Hz (s_memtime ticks of the longest wave / wall time)
 *   out[2] shader cycles per wave-instruction and SIMD
 *   out[3] milliseconds of load the probe ran
 * Returns 0, or -1 without the GPU backend.  bench.py normalises `roofline.alu` with it. */
int bzamd_probe_mad_rate(double target_ms, double* out);
/* BLITZAR_AMD_GENERATOR_CACHE=1 (read by sxt_init; off by default): the blocking
 * sxt_*_compute_pedersen_commitments_with_generators entry points keep a caller's host generators on
 * the device across calls.  Key = (host pointer, count, curve) + a hash of a 1-in-256 sample of the
 * rows; the second call with the same key registers the set (one extra upload), later calls whose key
 * and sample still match upload scalars only (the reference re-uploads on every call:
 * sxt/multiexp/bucket_method/accumulation.h:68-71).  CAVEAT: rewriting rows the sample misses, in
 * place, is not noticed -- hence opt-in.  Counters: calls served from the cache / sets registered. */
void bzamd_generator_cache_stats(uint64_t* hits, uint64_t* builds);

/* drop the backend singleton so that sxt_init may be called again (reference:
 * cbn::reset_backend_for_testing, cbindings/backend.cc:111) */
void bzamd_reset_for_testing(void);

/* Engine knobs of the current device's MSM context (GPU backend; 0 keeps the current value):
 * cap on the window width c (2..16), and the batching limits for many-column jobs (tasks = column
 * windows per launch, device workspace bytes per batch).  Results never depend on them. */
void bzamd_set_tuning(uint32_t max_window_bits, uint64_t max_tasks_per_batch,
                      uint64_t max_workspace_bytes);
/* Every column takes window width `window_bits` (2..16) wherever its bit width allows, instead of
 * the width the cost model would choose from its length (0 restores the model).  For tests: the
 * c = 16 code paths at sizes a CPU reference finishes in seconds. */
void bzamd_set_window_bits(uint32_t window_bits);
/* Per-call window tables: a call of many columns over the same caller generators (the reference's
 * bucket_method2 regime, sxt/multiexp/bucket_method2/multiexponentiation.h:48-121) builds the
 * 2^(c w) multiples of its generators once, inside the call, and runs every column as ONE task with
 * one bucket set.  mode 0 = the cost model decides (default), 1 = never, 6..16 = a table of that
 * window width for every call with caller generators (tests, A/B runs), negative = change nothing.
 * Returns the number of tables built so far on the backend's contexts.  Results never depend on it.
 * Env: BLITZAR_AMD_CALL_TABLES=0, BLITZAR_AMD_CALL_TABLE_BITS=c, BLITZAR_AMD_CALL_TABLE_OVERLAP=0. */
uint64_t bzamd_set_call_tables(int mode);
/* Work per lane of the two bucket kernels, as log2 (0, the default, lets every launch choose from
 * its size): sorted entries per accumulation lane (2^3..2^10; 32 for a single column so that its
 * lanes fill the machine, up to 128 when hundreds of columns do -- every segment leaves one partial
 * sum to fold) and buckets per bucket-reduction lane (2^1..2^8; 8 for a single column: shortest
 * dependent chain, up to 64 for many: least total work).  Results never depend on them. */
void bzamd_set_segments(uint32_t log2_entries_per_accumulate_lane,
                        uint32_t log2_buckets_per_reduce_lane);

/* Per-stage device timing of the next `max_calls` MSM calls issued on the current device, measured
 * with HIP events on the launch stream.  `bzamd_stage_timing_collect` blocks until those calls
 * finished, writes the accumulated milliseconds of the six stages
 * {prepare_addends, recode, bucket_sort, accumulate, reduce, combine} to out_ms[6] and returns the
 * number of calls recorded.  Every recorded stage puts an event pair on the stream, i.e. two bubbles
 * of a few microseconds per call: `_masked` records only the stages whose bit is set in
 * `stage_mask` (bit 3 = accumulate), the others read 0. */
void bzamd_stage_timing_begin(uint64_t max_calls);
void bzamd_stage_timing_begin_masked(uint64_t max_calls, uint32_t stage_mask);
/* ... recording one MSM in `sample_every` only (an event pair costs the stream two bubbles per
 * recorded stage: a sample keeps a timed region honest); `max_calls` and the count
 * bzamd_stage_timing_collect returns are then of RECORDED calls */
void bzamd_stage_timing_begin_sampled(uint64_t max_calls, uint32_t stage_mask, uint32_t sample_every);
uint64_t bzamd_stage_timing_collect(double* out_ms);

/* Throughput mode for a sequence of device-resident MSM calls.  A call with few columns is four
 * stages with complementary bottlenecks: a front of short HBM-bound kernels (generator conversion,
 * recoding, bucket sort), the integer-issue-bound bucket accumulation, and two latency chains that
 * leave the machine nearly idle -- the bucket reduction at one wavefront per SIMD and the final
 * stage at ONE workgroup per column (~250 dependent doublings, the encoding's 250 squarings).
 * bzamd_pipeline_next() makes the NEXT MSM enqueued through a device entry point of this header
 * (bzamd_msm_device*, bzamd_fixed_packed_multiexponentiation_device) on the current device run its
 * two tail stages on internal streams of the engine, beside the front and the accumulation of call
 * k + 1, which stay on the caller's `stream` (the calls must have the same shape to overlap: the
 * engine otherwise simply waits).  The operands are consumed in stream order, as without the mode.
 * The commitments of such a call are complete on `stream` only once TWO later
 * such calls on the device have been enqueued on it, or after bzamd_pipeline_flush(stream): do not
 * read them earlier.  Calls with 64 or more columns ignore the request (their tails fill the
 * machine).  A pipelined sequence lives on ONE stream and one caller thread per device (the NULL
 * stream or a stream of the caller's own: the engine's internal streams are non-blocking streams
 * of the lowest priority, they synchronise with nobody implicitly).
 * (Measured on MI355X, 2^20 curve25519 rows: see DESIGN.md section 9.) */
void bzamd_pipeline_next(void);
void bzamd_pipeline_flush(void* stream);

/* Variable-base MSM on device-resident operands.
 *   commitments  DEVICE  num_sequences canonical encodings (32 / 48 / 72 / 72 bytes each)
 *   descriptors  HOST    array whose `data` members are DEVICE pointers
 *   generators   DEVICE  C-ABI layout of the curve (sxt_ristretto255[160] / bls 104-byte stride /
 *                        sxt_bn254_g1[72] / sxt_grumpkin[72]), max_i n_i entries
 * Same validation/abort behaviour as the sxt_*_compute_pedersen_commitments_with_generators calls. */
void bzamd_msm_device(unsigned curve_id, void* commitments, uint32_t num_sequences,
                      const struct sxt_sequence_descriptor* descriptors, const void* generators,
                      void* stream);

/* Multi-device MSM inside ONE process on device-resident operands (the
 * reference's gpu backend drives every visible device from one process too,
 * sxt/execution/device/for_each.cc:56-82, but bounces partial results through pinned host memory).
 * The backend drives D = bzamd_num_devices() devices; device slot d is HIP device bzamd_device_id(d)
 * (slot 0 = the device that was current at sxt_init).  The columns are cut into contiguous ranges
 * of per = bzamd_multi_device_columns_per_device(num_sequences) = ceil(num_sequences / D) columns:
 * column i belongs to slot i / per, and descriptors[i].data is a DEVICE pointer on THAT device;
 * generators[d] = the generator set (C-ABI layout) as a DEVICE pointer on slot d (may be NULL for a
 * slot that owns no column).  Every device commits its columns; ONE all-gather of the encodings --
 * ncclAllGather on the communicators the library creates with ncclCommInitAll on first use, RCCL
 * over the xGMI links; librccl is loaded on that first use only -- leaves ALL num_sequences
 * commitments on EVERY device, copied to commitments[d] (DEVICE pointer on slot d, NULL = not
 * wanted there).  Blocking.  Logical devices that share a physical one
 * (BLITZAR_AMD_FORCE_SHARDS; RCCL refuses duplicate devices within a communicator) exchange with
 * peer copies instead: bzamd_multi_device_exchange() says which ("rccl" / "peer-copies"). */
int bzamd_device_id(int slot);
uint32_t bzamd_multi_device_columns_per_device(uint32_t num_sequences);
const char* bzamd_multi_device_exchange(void);
void bzamd_msm_multi_device(unsigned curve_id, void* const* commitments, uint32_t num_sequences,
                            const struct sxt_sequence_descriptor* descriptors,
                            const void* const* generators);

/* Row-sharded MSM support (one column split by rows across GPUs):
 * the partial result of a shard as a raw projective element (sxt_ristretto255 160 B /
 * sxt_bls12_381_g1_p2 144 B / sxt_bn254_g1_p2, sxt_grumpkin_p2 96 B per column), and the fold
 *   commitments[k] = canonical encoding of  sum_r partials[r * num_outputs + k]
 * that every rank applies after the all-gather of the partials (RCCL has no user-defined
 * reduction; group addition is exact, so the canonical result equals the unsharded one). */
void bzamd_msm_device_projective(unsigned curve_id, void* res, uint32_t num_sequences,
                                 const struct sxt_sequence_descriptor* descriptors,
                                 const void* generators, void* stream);
/* HOST operands, either backend (blocking) */
void bzamd_msm_projective(unsigned curve_id, void* res, uint32_t num_sequences,
                          const struct sxt_sequence_descriptor* descriptors,
                          const void* generators);
/* HOST operands (no backend needed: G - 1 point additions and one encoding per output) */
void bzamd_fold_encode(unsigned curve_id, void* commitments, const void* partials,
                       uint32_t num_partials, uint32_t num_outputs);
/* DEVICE operands (async) */
void bzamd_fold_encode_device(unsigned curve_id, void* commitments, const void* partials,
                              uint32_t num_partials, uint32_t num_outputs, void* stream);

/* Resident generator set: generators converted once into the engine's addend layout and kept in
 * HBM across calls. */
struct bzamd_generators;
/* from DEVICE generators in C-ABI layout (blocking) */
struct bzamd_generators* bzamd_generators_new_device(unsigned curve_id, const void* generators,
                                                     uint64_t n, void* stream);
/* from HOST generators in C-ABI layout (blocking) */
struct bzamd_generators* bzamd_generators_new_host(unsigned curve_id, const void* generators,
                                                   uint64_t n);
void bzamd_generators_free(struct bzamd_generators* gens);
void bzamd_msm_device_resident(void* commitments, uint32_t num_sequences,
                               const struct sxt_sequence_descriptor* descriptors,
                               const struct bzamd_generators* gens, void* stream);

/* DEVICE built-in ristretto generators g_first .. g_first+n-1 as sxt_ristretto255 (async) */
void bzamd_ristretto255_generators_device(struct sxt_ristretto255* generators, uint64_t first,
                                          uint64_t n, void* stream);

/* DEVICE generators[i] = (i + 1) * base in the curve's C-ABI generator layout, base = one DEVICE
 * generator in the same layout (async).  Synthetic generator sets with known discrete logarithms
 * for benchmarks and full-size parity checks. */
void bzamd_generator_multiples_device(unsigned curve_id, void* generators, const void* base,
                                      uint64_t n, void* stream);

/* Fixed-base (handle) MSM with DEVICE scalars and DEVICE results; same packing rules as
 * sxt_fixed_packed_multiexponentiation / sxt_fixed_vlen_multiexponentiation
 * (output_lengths may be NULL = all rows). */
void bzamd_fixed_packed_multiexponentiation_device(void* res, const struct sxt_multiexp_handle* handle,
                                                   const unsigned* output_bit_table,
                                                   const unsigned* output_lengths,
                                                   unsigned num_outputs, unsigned n,
                                                   const uint8_t* scalars, void* stream);

#ifdef __cplusplus
} /* extern "C" */
#endif

#endif /* BLITZAR_AMD_BLITZAR_AMD_H */
