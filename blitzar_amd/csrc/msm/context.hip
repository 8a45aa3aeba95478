// msm_context lifetime + curve id dispatch.
#include <cstdlib>

#include "blitzar_amd/csrc/msm/dispatch.h"
#include "blitzar_amd/csrc/msm/engine.h"

namespace bz {
const curve_vtable* curve_vtable_for(unsigned curve_id) {
  switch (curve_id) {
  case 0:
    return &curve25519_vtable();
  case 1:
    return &bls12_381_vtable();
  case 2:
    return &bn254_vtable();
  case 3:
    return &grumpkin_vtable();
  default:
    return nullptr;
  }
}

namespace {
// Does this device fetch code that does not fit its instruction cache more slowly than it executes
// it?  One wavefront walks a dependent chain of 8-byte v_add_u32 laid out as straight-line code, once
// 16 KiB of it in a loop (stays in the 64 KiB cache), once 128 KiB: on one kind of MI355X box both
// run at 3.3-3.5 ns per instruction, on the other the large body takes 6.3.  ~1.5
// ms per device, at context creation.
#define BZ_PROBE_ADD(v) asm volatile("v_add_u32 %0, 0x12345679, %0" : "+v"(v));
#define BZ_R4(X) X X X X
#define BZ_R16(X) BZ_R4(BZ_R4(X))
#define BZ_R256(X) BZ_R16(BZ_R16(X))
#define BZ_R2048(X) BZ_R4(BZ_R256(X)) BZ_R4(BZ_R256(X))
template <int Blocks2048> __global__ void k_probe_code(u32* out, u32 seed, int rounds) {
  u32 a = seed + threadIdx.x;
  for (int r = 0; r < rounds; ++r) {
    BZ_R2048(BZ_PROBE_ADD(a))
    if constexpr (Blocks2048 >= 8) {
      BZ_R2048(BZ_PROBE_ADD(a)) BZ_R2048(BZ_PROBE_ADD(a)) BZ_R2048(BZ_PROBE_ADD(a))
      BZ_R2048(BZ_PROBE_ADD(a)) BZ_R2048(BZ_PROBE_ADD(a)) BZ_R2048(BZ_PROBE_ADD(a))
      BZ_R2048(BZ_PROBE_ADD(a))
    }
  }
  out[threadIdx.x] = a;
}

bool probe_slow_instruction_fetch() {
  u32* d_out = nullptr;
  hipStream_t stream = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (hipMalloc(reinterpret_cast<void**>(&d_out), sizeof(u32) * 64) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  BZ_HIP_CHECK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
  BZ_HIP_CHECK(hipEventCreate(&e0));
  BZ_HIP_CHECK(hipEventCreate(&e1));
  auto timed = [&](auto kernel, int rounds) {
    hipLaunchKernelGGL(kernel, dim3(1), dim3(64), 0, stream, d_out, 1u, 1); // load the code object, warm
    BZ_HIP_CHECK(hipEventRecord(e0, stream));
    hipLaunchKernelGGL(kernel, dim3(1), dim3(64), 0, stream, d_out, 2u, rounds);
    BZ_HIP_CHECK(hipEventRecord(e1, stream));
    BZ_HIP_CHECK(hipEventSynchronize(e1));
    float ms = 0;
    BZ_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    return static_cast<double>(ms);
  };
  // the same 2^17 instructions each: 64 rounds of 2048, 8 rounds of 16384
  (void)timed(k_probe_code<1>, 64); // clocks up
  const double small = timed(k_probe_code<1>, 64);
  const double large = timed(k_probe_code<8>, 8);
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  (void)hipStreamDestroy(stream);
  (void)hipFree(d_out);
  g_kernel_launches += 6;
  return small > 0 && large > 1.4 * small;
}
} // namespace

msm_context* msm_context_new() {
  auto* ctx = new msm_context();
  // development overrides of the sort geometry (plan.h), validated like bzamd_set_tuning: a window
  // width above 16 would overflow the int16 digit storage, one below 2 gives more windows than
  // k_horner's 256 lanes can fold
  if (const char* v = std::getenv("BLITZAR_AMD_GROUP_ENTRIES")) {
    const unsigned long e = std::strtoul(v, nullptr, 10);
    BZ_RELEASE_ASSERT(e >= 64 && e <= kLocalSortCapacity,
                      "BLITZAR_AMD_GROUP_ENTRIES must be in [64, 6144]");
    ctx->tuning.partition_group_entries = static_cast<u32>(e);
  }
  if (const char* v = std::getenv("BLITZAR_AMD_MAX_WINDOW_BITS")) {
    msm_context_set_tuning(ctx, static_cast<u32>(std::strtoul(v, nullptr, 10)), 0, 0);
  }
  if (const char* v = std::getenv("BLITZAR_AMD_BUCKET_COST")) {
    const double cost = std::strtod(v, nullptr);
    BZ_RELEASE_ASSERT(cost > 0 && cost < 1e6, "BLITZAR_AMD_BUCKET_COST must be positive");
    ctx->tuning.throughput_bucket_cost = cost;
  }
  if (const char* v = std::getenv("BLITZAR_AMD_REDUCE_SEGMENT_LOG2")) {
    msm_context_set_segments(ctx, ctx->tuning.force_segment_log2,
                             static_cast<u32>(std::strtoul(v, nullptr, 10)));
  }
  if (const char* v = std::getenv("BLITZAR_AMD_SEGMENT_LOG2")) {
    msm_context_set_segments(ctx, static_cast<u32>(std::strtoul(v, nullptr, 10)),
                             ctx->tuning.force_reduce_segment_log2);
  }
  if (const char* v = std::getenv("BLITZAR_AMD_FORCE_WINDOW_TABLES")) {
    ctx->tuning.force_window_tables = v[0] != '0';
  }
  // throughput mode (engine.h, msm_context): "0" switches a feature off, anything else on
  auto flag = [](const char* name, bool& out) {
    if (const char* v = std::getenv(name)) out = !(v[0] == '0' && v[1] == 0);
  };
  flag("BLITZAR_AMD_OVERLAP_TAILS", ctx->overlap_tails);
  if (const char* v = std::getenv("BLITZAR_AMD_COMPACT_REDUCE")) {
    const unsigned long m = std::strtoul(v, nullptr, 10);
    BZ_RELEASE_ASSERT(m <= 2, "BLITZAR_AMD_COMPACT_REDUCE must be 0 (never), 1 (always) or 2 (probe)");
    ctx->compact_reduce = static_cast<u32>(m);
  }
  if (ctx->compact_reduce == 2) ctx->slow_instruction_fetch = probe_slow_instruction_fetch();
  return ctx;
}
void msm_context_free(msm_context* ctx) { delete ctx; }
bool msm_context_slow_instruction_fetch(msm_context* ctx) { return ctx->slow_instruction_fetch; }
void msm_context_set_tuning(msm_context* ctx, u32 max_window_bits, size_t max_tasks_per_batch,
                            size_t max_workspace_bytes) {
  if (max_window_bits != 0) {
    BZ_RELEASE_ASSERT(max_window_bits >= 2 && max_window_bits <= 16, "window width cap must be 2..16");
    ctx->tuning.max_window_bits = max_window_bits;
  }
  if (max_tasks_per_batch != 0) {
    BZ_RELEASE_ASSERT(max_tasks_per_batch <= 65535, "tasks per batch are a launch-grid dimension");
    ctx->tuning.max_tasks_per_batch = max_tasks_per_batch;
  }
  if (max_workspace_bytes != 0) ctx->tuning.max_workspace_bytes = max_workspace_bytes;
}
void msm_context_set_segments(msm_context* ctx, u32 log2_entries_per_accumulate_lane,
                              u32 log2_buckets_per_reduce_lane) {
  BZ_RELEASE_ASSERT(log2_entries_per_accumulate_lane == 0 ||
                        (log2_entries_per_accumulate_lane >= 3 && log2_entries_per_accumulate_lane <= 10),
                    "entries per accumulate lane: 2^3 .. 2^10 (0 = automatic)");
  BZ_RELEASE_ASSERT(log2_buckets_per_reduce_lane <= 8,
                    "buckets per reduce lane: 2^1 .. 2^8 (0 = automatic)");
  ctx->tuning.force_segment_log2 = log2_entries_per_accumulate_lane;
  ctx->tuning.force_reduce_segment_log2 = log2_buckets_per_reduce_lane;
}
void msm_context_set_window_bits(msm_context* ctx, u32 window_bits) {
  BZ_RELEASE_ASSERT(window_bits == 0 || (window_bits >= 2 && window_bits <= 16),
                    "window width must be 2..16 (0 = automatic)");
  std::lock_guard<std::mutex> lock(ctx->mu);
  ctx->tuning.force_window_bits = window_bits;
}
void msm_context_defer_next_tail(msm_context* ctx) {
  std::lock_guard<std::mutex> lock(ctx->mu);
  ctx->defer_tail = true;
}
void msm_context_join_tail(msm_context* ctx, hipStream_t stream) {
  std::lock_guard<std::mutex> lock(ctx->mu);
  ctx->join_all(stream);
}
void msm_context_timing_begin(msm_context* ctx, size_t max_calls, unsigned stage_mask) {
  std::lock_guard<std::mutex> lock(ctx->mu);
  ctx->timer.begin(max_calls, stage_mask & 0x3f);
}
size_t msm_context_timing_collect(msm_context* ctx, double out_ms[6]) {
  std::lock_guard<std::mutex> lock(ctx->mu);
  return ctx->timer.collect(out_ms);
}
} // namespace bz
