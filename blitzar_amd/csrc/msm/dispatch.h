// Type-erased entry points of the per-curve MSM translation units (msm_<curve>.hip).  Each curve
// is compiled in its own TU so the four gfx950 code objects build in parallel.
#pragma once

#include <cstdio>
#include <vector>

#include "blitzar_amd/csrc/base/device.h"
#include "blitzar_amd/csrc/msm/plan.h"

namespace bz {

struct msm_context;

struct curve_vtable {
  unsigned curve_id;
  size_t api_generator_size; // stride of caller generators in the C ABI
  size_t addend_size;        // resident addend
  size_t output_size;        // canonical commitment encoding
  size_t projective_size;    // raw projective element (fixed-base results, handle generators)
  // enqueue a variable-base MSM; exactly one of d_addends / d_api_generators is used
  void (*msm)(msm_context& ctx, u8* d_out, u32 out_stride, bool projective_out,
              const std::vector<host_column>& cols, const void* d_addends,
              const void* d_api_generators, hipStream_t stream);
  // C-ABI generators -> resident addends
  void (*prepare_addends)(void* d_addends, const void* d_api_generators, u64 n, hipStream_t stream);
  // projective elements (handle generators) -> resident addends (batch normalisation on device)
  void (*prepare_addends_projective)(void* d_addends, const void* d_projective, u64 n,
                                     hipStream_t stream);
  // SXT_CPU_BACKEND: all pointers are host pointers; generators in C-ABI layout, or in
  // projective layout when `generators_projective`
  void (*msm_host)(u8* out, u32 out_stride, bool projective_out,
                   const std::vector<host_column>& cols, const void* generators,
                   bool generators_projective, u64 num_generators);
  // canonical encodings of the sums of `num_partials` projective partial results per output
  void (*fold_encode_host)(u8* out, const void* partials, u32 num_partials, u32 num_outputs);
  void (*fold_encode_device)(u8* d_out, const void* d_partials, u32 num_partials, u32 num_outputs,
                             hipStream_t stream);
  // the same sums left as raw projective elements (`projective_size` apart)
  void (*fold_device)(u8* d_out, const void* d_partials, u32 num_partials, u32 num_outputs,
                      hipStream_t stream);
  // out[k] = sum_j 2^(shift_bits j) * pieces[first_k + j], first_k = piece_counts[0] + ... +
  // piece_counts[k - 1]: projective elements in and out, host memory (outputs of a fixed-base call
  // wider than one scalar)
  void (*fold_shifted_host)(u8* out, const void* pieces, const u32* piece_counts, u32 num_outputs,
                            u32 shift_bits);
  // d_out[i] = (i + 1) * base, C-ABI generator layout (synthetic generator sets)
  void (*generator_multiples)(void* d_out, const void* d_base_api, u64 n, hipStream_t stream);
  // resident generator sets (registered once, reused by many calls): their own addend layout
  // (curve25519: Z = 1, 128 bytes; the Weierstrass curves: the same affine addends)
  size_t resident_addend_size;
  // `tables`: d_addends is slice 0 of a window table (plan.h), or nullptr
  void (*msm_resident)(msm_context& ctx, u8* d_out, u32 out_stride, bool projective_out,
                       const std::vector<host_column>& cols, const void* d_addends,
                       hipStream_t stream, const window_table* tables);
  void (*prepare_resident)(void* d_addends, const void* d_api_generators, u64 n,
                           hipStream_t stream);
  void (*prepare_resident_projective)(void* d_addends, const void* d_projective, u64 n,
                                      hipStream_t stream);
  // window table of a resident set: d_table[w * stride + i] = addend of 2^(bits w) g_i for
  // w < windows, from C-ABI generators or projective elements on the device (blocking)
  void (*build_window_table)(void* d_table, const void* d_source, bool source_projective, u64 n,
                             u64 stride, u32 windows, u32 bits, hipStream_t stream);
  // Per-call window table (plan.h, choose_call_table): when the cost model wants one for `cols`
  // (only their shapes are read) over the caller generators `d_api_generators`, enqueue its build
  // and return its slice 0 -- resident-form addends for msm_resident with `*shape` -- else nullptr.
  // The table lives in the context until the next one is built there.
  const void* (*call_table)(msm_context& ctx, const std::vector<host_column>& cols,
                            const void* d_api_generators, window_table* shape, hipStream_t stream);
  // partition-table file interop of fixed-base handles (fixed/partition_table.h)
  size_t compact_size;
  bool (*write_partition_table)(std::FILE* f, unsigned window_width, const void* projective, u64 n);
  // the same file built on the current device (window widths up to 16)
  bool (*write_partition_table_device)(std::FILE* f, unsigned window_width, const void* projective,
                                       u64 n, hipStream_t stream);
  bool (*read_partition_generators)(std::FILE* f, unsigned& window_width,
                                    std::vector<u8>& projective_out, u64& n);
  // BLITZAR_DUMP_DIR recording (fixed/dump.h): compact generators + the reference's type names
  void (*write_compact_generators)(std::FILE* f, const void* projective, u64 n);
  const char* element_type_name;  // typeid(T).name() of the reference's projective element
  const char* accessor_type_name; // typeid(U).name() of its compact element
};

const curve_vtable& curve25519_vtable();
const curve_vtable& bls12_381_vtable();
const curve_vtable& bn254_vtable();
const curve_vtable& grumpkin_vtable();

// nullptr for an unknown id
const curve_vtable* curve_vtable_for(unsigned curve_id);

msm_context* msm_context_new();
// what the context's device answered to the instruction-fetch probe at creation (context.hip)
bool msm_context_slow_instruction_fetch(msm_context* ctx);
// measured issue rate of v_mad_u64_u32 on the current device under an all-SIMD load (context.hip)
bool msm_probe_mad_rate(double target_ms, double out[4]);
void msm_context_free(msm_context* ctx);
// engine knobs (0 keeps the current value): window width cap (2..16), tasks and workspace bytes
// per batch of columns
void msm_context_set_tuning(msm_context* ctx, u32 max_window_bits, size_t max_tasks_per_batch,
                            size_t max_workspace_bytes);
// tests: every column takes this window width where it can (0 = the cost model chooses)
void msm_context_set_window_bits(msm_context* ctx, u32 window_bits);
// per-call window tables (engine.h, msm_context): mode 0 = the cost model decides (default), 1 = never,
// 6..16 = a table of that width for every call with caller generators; returns the tables built so far
u64 msm_context_set_call_tables(msm_context* ctx, int mode);
// throughput mode (bzamd_msm_device_pipelined): the next MSM enqueued on this context leaves its last
// stages running on the context's own streams; `join_tail` makes `stream` wait for everything pending
void msm_context_defer_next_tail(msm_context* ctx);
void msm_context_join_tail(msm_context* ctx, hipStream_t stream);
// sorted entries per k_accumulate lane = 2^a (3..10), buckets per k_reduce lane = 2^r (1..8);
// 0 = chosen per launch from its entry / bucket counts (plan.h)
void msm_context_set_segments(msm_context* ctx, u32 log2_entries_per_accumulate_lane,
                              u32 log2_buckets_per_reduce_lane);
// per-stage HIP-event timing of the next `max_calls` MSM calls on this context
// (`stage_mask`: bit s set = record stage s; every recorded stage costs an event pair per call)
// `sample_every`: record one MSM batch in this many (the returned count is of RECORDED batches)
void msm_context_timing_begin(msm_context* ctx, size_t max_calls, unsigned stage_mask = 0x3f,
                              size_t sample_every = 1);
// accumulated ms per stage {prepare, recode, sort, accumulate, reduce, combine}; returns #calls
size_t msm_context_timing_collect(msm_context* ctx, double out_ms[6]);
} // namespace bz
