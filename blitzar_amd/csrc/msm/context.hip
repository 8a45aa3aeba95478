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

// Issue rate of v_mad_u64_u32 on THIS device, now: every SIMD holds 6 waves, each runs 8 independent
// chains of the instruction (the field products' one wide primitive).  A launch lasts ~3 ms; launches
// repeat until `target_ms` of load have passed, and the best of the launches of the second half is
// reported (the part clocks to its power budget under this load, not to the nominal 2.4 GHz).
// (6, not the 8 a SIMD can hold: a grid that fills the machine EXACTLY runs in two rounds as soon as
// anything else holds a wave slot -- one box of round 5 reported half the rate that way.)
constexpr int kMadProbeIters = 8192, kMadProbeChains = 8;
__global__ void __launch_bounds__(256) k_probe_mad(u64* out, u64* ticks, u32 seed) {
  u64 a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 9, a5 = a0 * 11,
      a6 = a0 * 13, a7 = a0 * 15;
  const u32 x = seed * 2654435761u + threadIdx.x, y = x ^ 0x9e3779b9u;
#define BZ_PROBE_MAD(v) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(v) : "v"(x), "v"(y) : "vcc");
  const u64 t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < kMadProbeIters; ++it) {
    BZ_PROBE_MAD(a0) BZ_PROBE_MAD(a1) BZ_PROBE_MAD(a2) BZ_PROBE_MAD(a3)
    BZ_PROBE_MAD(a4) BZ_PROBE_MAD(a5) BZ_PROBE_MAD(a6) BZ_PROBE_MAD(a7)
  }
  const u64 t1 = __builtin_amdgcn_s_memtime();
#undef BZ_PROBE_MAD
  out[static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
  if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}
} // namespace

// out[0] wave-instructions per second over the whole device, out[1] effective shader clock (Hz:
// s_memtime ticks of the longest wave / wall time of its launch), out[2] shader cycles per
// wave-instruction and SIMD, out[3] milliseconds of load the probe ran
bool msm_probe_mad_rate(double target_ms, double out[4]) {
  int device = 0, cus = 0;
  BZ_HIP_CHECK(hipGetDevice(&device));
  BZ_HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device));
  constexpr u32 kWavesPerSimd = 6;
  const u32 blocks = static_cast<u32>(cus) * kWavesPerSimd; // 4 SIMDs per CU, 4 waves per block
  u64 *d_out = nullptr, *d_ticks = nullptr;
  if (hipMalloc(reinterpret_cast<void**>(&d_out), sizeof(u64) * 256 * blocks) != hipSuccess ||
      hipMalloc(reinterpret_cast<void**>(&d_ticks), sizeof(u64) * blocks) != hipSuccess) {
    (void)hipGetLastError();
    if (d_out != nullptr) (void)hipFree(d_out);
    return false;
  }
  hipStream_t stream = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr, first = nullptr;
  BZ_HIP_CHECK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
  BZ_HIP_CHECK(hipEventCreate(&e0));
  BZ_HIP_CHECK(hipEventCreate(&e1));
  BZ_HIP_CHECK(hipEventCreate(&first));
  BZ_HIP_CHECK(hipEventRecord(first, stream));
  float total_ms = 0, last_ms = 0, best_ms = 0;
  u32 launches = 0;
  u64 best_ticks = 0;
  std::vector<u64> ticks(blocks);
  do {
    BZ_HIP_CHECK(hipEventRecord(e0, stream));
    hipLaunchKernelGGL(k_probe_mad, dim3(blocks), dim3(256), 0, stream, d_out, d_ticks, 1u + launches);
    BZ_HIP_CHECK(hipEventRecord(e1, stream));
    BZ_HIP_CHECK(hipEventSynchronize(e1));
    BZ_HIP_CHECK(hipEventElapsedTime(&last_ms, e0, e1));
    BZ_HIP_CHECK(hipEventElapsedTime(&total_ms, first, e1));
    launches += 1;
    // past the clock ramp (the second half of the load): keep the fastest launch
    if (total_ms >= 0.5f * static_cast<float>(target_ms) && last_ms > 0 &&
        (best_ms == 0 || last_ms < best_ms)) {
      best_ms = last_ms;
      BZ_HIP_CHECK(hipMemcpy(ticks.data(), d_ticks, sizeof(u64) * blocks, hipMemcpyDeviceToHost));
      best_ticks = 0;
      for (u64 t : ticks) best_ticks = t > best_ticks ? t : best_ticks;
    }
  } while (total_ms < target_ms && launches < 4096);
  const double instructions_per_wave = static_cast<double>(kMadProbeIters) * kMadProbeChains;
  const double waves = static_cast<double>(blocks) * 4;
  out[0] = best_ms > 0 ? instructions_per_wave * waves / (static_cast<double>(best_ms) * 1e-3) : 0;
  out[1] = best_ms > 0 ? static_cast<double>(best_ticks) / (static_cast<double>(best_ms) * 1e-3) : 0;
  out[2] = static_cast<double>(best_ticks) / (instructions_per_wave * kWavesPerSimd);
  out[3] = total_ms;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  (void)hipEventDestroy(first);
  (void)hipStreamDestroy(stream);
  (void)hipFree(d_out);
  (void)hipFree(d_ticks);
  g_kernel_launches += launches;
  return best_ms > 0;
}

msm_context* msm_context_new() {
  auto* ctx = new msm_context();
  // development overrides of the launch geometry (plan.h) for A/B runs through the native drivers;
  // the same values as bzamd_set_tuning / bzamd_set_segments, validated there
  if (const char* v = std::getenv("BLITZAR_AMD_MAX_WINDOW_BITS")) {
    msm_context_set_tuning(ctx, static_cast<u32>(std::strtoul(v, nullptr, 10)), 0, 0);
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
  flag("BLITZAR_AMD_CALL_TABLES", ctx->call_tables);
  flag("BLITZAR_AMD_CALL_TABLE_OVERLAP", ctx->table_overlap);
  flag("BLITZAR_AMD_CALL_TABLE_WAVE_CHAIN", ctx->wave_chain);
  flag("BLITZAR_AMD_NORMALISE_CALLER", ctx->normalise_caller);
  flag("BLITZAR_AMD_MERGED_WAITS", ctx->merged_waits);
  if (const char* v = std::getenv("BLITZAR_AMD_ACC_LDS_PAD")) {
    const unsigned long pad = std::strtoul(v, nullptr, 10);
    BZ_RELEASE_ASSERT(pad <= 65536, "BLITZAR_AMD_ACC_LDS_PAD: at most 64 KiB of dynamic LDS");
    ctx->acc_lds_pad = static_cast<u32>(pad);
  }
  if (const char* v = std::getenv("BLITZAR_AMD_CALL_TABLE_BITS")) {
    const unsigned long b = std::strtoul(v, nullptr, 10);
    BZ_RELEASE_ASSERT(b == 0 || (b >= kCallTableMinBits && b <= kCallTableMaxBits),
                      "BLITZAR_AMD_CALL_TABLE_BITS must be 0 or in [6, 16]");
    ctx->force_call_table_bits = static_cast<u32>(b);
  }
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
u64 msm_context_set_call_tables(msm_context* ctx, int mode) {
  std::lock_guard<std::mutex> lock(ctx->mu);
  if (mode >= 0) {
    BZ_RELEASE_ASSERT(mode <= 1 || (mode >= static_cast<int>(kCallTableMinBits) &&
                                    mode <= static_cast<int>(kCallTableMaxBits)),
                      "call tables: 0 (model), 1 (never) or a width in [6, 16]");
    ctx->call_tables = mode != 1;
    ctx->force_call_table_bits = mode > 1 ? static_cast<u32>(mode) : 0;
  }
  return ctx->call_tables_built;
}
void msm_context_defer_next_tail(msm_context* ctx) {
  std::lock_guard<std::mutex> lock(ctx->mu);
  ctx->defer_tail = true;
}
void msm_context_join_tail(msm_context* ctx, hipStream_t stream) {
  std::lock_guard<std::mutex> lock(ctx->mu);
  ctx->join_all(stream);
}
void msm_context_timing_begin(msm_context* ctx, size_t max_calls, unsigned stage_mask,
                              size_t sample_every) {
  std::lock_guard<std::mutex> lock(ctx->mu);
  ctx->timer.begin(max_calls, stage_mask & 0x3f, sample_every);
}
size_t msm_context_timing_collect(msm_context* ctx, double out_ms[6]) {
  std::lock_guard<std::mutex> lock(ctx->mu);
  return ctx->timer.collect(out_ms);
}
} // namespace bz
