// Host orchestration of one variable-base MSM call on one device: plan -> arena carve-up ->
// kernel sequence on a single HIP stream.  Everything is asynchronous with respect to the host;
// callers synchronise the stream (the C ABI entry points do, they are blocking like the
// reference's).
#pragma once

#include <vector>

#include "blitzar_amd/csrc/base/device.h"
#include "blitzar_amd/csrc/msm/kernels.h"

namespace bz {

// Optional per-stage device timing (HIP events on the launch stream), used by bench.py to report
// the dominant kernel's measured duration next to its algorithmic bytes.
constexpr int kNumStages = 6; // prepare, recode, sort, accumulate, reduce, combine
struct stage_timer {
  bool enabled = false;
  std::vector<hipEvent_t> events; // (kNumStages + 1) per recorded call
  size_t calls = 0;
  size_t capacity_calls = 0;

  void begin(size_t max_calls) {
    release();
    events.resize(max_calls * (kNumStages + 1));
    for (auto& e : events) BZ_HIP_CHECK(hipEventCreate(&e));
    capacity_calls = max_calls;
    calls = 0;
    enabled = true;
  }
  hipEvent_t event(int stage) { return events[calls * (kNumStages + 1) + stage]; }
  bool recording() const { return enabled && calls < capacity_calls; }
  // accumulated milliseconds per stage over the recorded calls (blocks until they finished)
  size_t collect(double out_ms[kNumStages]) {
    for (int s = 0; s < kNumStages; ++s) out_ms[s] = 0;
    for (size_t c = 0; c < calls; ++c) {
      hipEvent_t* ev = &events[c * (kNumStages + 1)];
      BZ_HIP_CHECK(hipEventSynchronize(ev[kNumStages]));
      for (int s = 0; s < kNumStages; ++s) {
        float ms = 0;
        BZ_HIP_CHECK(hipEventElapsedTime(&ms, ev[s], ev[s + 1]));
        out_ms[s] += ms;
      }
    }
    const size_t n = calls;
    release();
    return n;
  }
  void release() {
    for (auto& e : events) (void)hipEventDestroy(e);
    events.clear();
    enabled = false;
    calls = 0;
    capacity_calls = 0;
  }
  ~stage_timer() { release(); }
};

struct msm_context {
  device_arena arena;
  msm_tuning tuning;
  stage_timer timer;
};

template <class C> struct msm_workspace_sizes {
  size_t total = 0;
};

// one-time kernel attributes: the sort kernels need up to 128 KiB of dynamic LDS
static void configure_sort_kernels() {
  static bool done = false;
  if (done) return;
  BZ_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_bucket_hist),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024));
  BZ_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_bucket_scatter),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024));
  done = true;
}

// device workspace of one batch of columns (everything carved from the arena)
template <class C>
size_t msm_workspace_bytes(const msm_plan& plan, bool needs_addends, u32 partial_stride) {
  using point = typename C::point;
  using addend = typename C::addend;
  const size_t num_tasks = plan.tasks.size(), num_cols = plan.columns.size();
  size_t need = 0;
  need += device_arena::padded(sizeof(column_desc) * num_cols);
  need += device_arena::padded(sizeof(task_desc) * (num_tasks + 1));
  if (needs_addends) need += device_arena::padded(sizeof(addend) * (plan.max_rows + 1));
  need += device_arena::padded(sizeof(i16) * (plan.total_entries + 8));
  need += device_arena::padded(sizeof(u32) * (plan.total_entries + 8));
  need += device_arena::padded(sizeof(u32) * (plan.total_hist + 2));
  need += device_arena::padded(sizeof(u32) * (plan.total_chunks + 1));
  need += device_arena::padded(sizeof(u32) * (plan.total_segments + 1));
  need += device_arena::padded(sizeof(u32) * (plan.total_buckets + 1));
  need += device_arena::padded(sizeof(point) * (plan.total_buckets + 1));
  need += device_arena::padded(sizeof(point) * (plan.total_segments + 1));
  need += device_arena::padded(sizeof(point) * (num_tasks * partial_stride + 1));
  return need;
}

inline u32 partial_stride_of(const msm_plan& plan) {
  return plan.max_task_buckets == 0 ? 1 : ceil_div_u32(plan.max_task_buckets, kReduceBlockBuckets);
}

template <class C>
void msm_enqueue_batch(msm_context& ctx, u8* d_out, u32 out_stride, bool projective_out,
                       const msm_plan& plan, const typename C::addend* d_addends,
                       const void* d_api_generators, hipStream_t stream);

// Enqueue the MSM.  `d_addends` covers rows [0, max n); `d_out` receives one encoding per column
// (`out_stride` bytes apart): canonical (`C::encode`) or raw projective when `projective_out`.
// Columns are processed in batches bounded by the launch grid (tasks per batch) and by
// `msm_tuning::max_workspace_bytes`; batches reuse the same arena back to back on the stream.
template <class C>
void msm_enqueue(msm_context& ctx, u8* d_out, u32 out_stride, bool projective_out,
                 const std::vector<host_column>& cols, const typename C::addend* d_addends,
                 const void* d_api_generators, hipStream_t stream) {
  if (cols.empty()) return;
  configure_sort_kernels();
  msm_tuning tune = ctx.tuning;
  bool any_signed = false;
  for (const auto& c : cols) any_signed = any_signed || c.is_signed;
  // Signed columns use |x| with all digits negated: cap c at 15 so that -D fits int16 either way.
  if (any_signed && tune.max_window_bits > 15) tune.max_window_bits = 15;

  std::vector<msm_plan> batches;
  std::vector<size_t> first_column;
  size_t need = 0;
  for (size_t begin = 0; begin < cols.size();) {
    size_t end = begin;
    msm_plan plan;
    // grow the batch column by column (re-planning a prefix is cheap: a few fields per task)
    while (end < cols.size()) {
      std::vector<host_column> trial(cols.begin() + begin, cols.begin() + end + 1);
      msm_plan p = make_msm_plan(trial, tune);
      const size_t bytes = msm_workspace_bytes<C>(p, d_addends == nullptr, partial_stride_of(p));
      if (end > begin && (p.tasks.size() > tune.max_tasks_per_batch || bytes > tune.max_workspace_bytes)) {
        break;
      }
      plan = std::move(p);
      ++end;
      // jump ahead in large uniform jobs: avoid quadratic re-planning
      if (end - begin >= 8) {
        const size_t step = end - begin;
        while (end + step <= cols.size()) {
          std::vector<host_column> t2(cols.begin() + begin, cols.begin() + end + step);
          msm_plan p2 = make_msm_plan(t2, tune);
          const size_t b2 = msm_workspace_bytes<C>(p2, d_addends == nullptr, partial_stride_of(p2));
          if (p2.tasks.size() > tune.max_tasks_per_batch || b2 > tune.max_workspace_bytes) break;
          plan = std::move(p2);
          end += step;
        }
      }
    }
    const size_t bytes = msm_workspace_bytes<C>(plan, d_addends == nullptr, partial_stride_of(plan));
    if (bytes > need) need = bytes;
    first_column.push_back(begin);
    batches.push_back(std::move(plan));
    begin = end;
  }
  // one allocation sized for the largest batch: later resets never reallocate (no mid-call sync)
  ctx.arena.reset(need, stream);
  for (size_t k = 0; k < batches.size(); ++k) {
    ctx.arena.reset(need, stream);
    msm_enqueue_batch<C>(ctx, d_out + first_column[k] * static_cast<size_t>(out_stride), out_stride,
                         projective_out, batches[k], d_addends, d_api_generators, stream);
  }
}

template <class C>
void msm_enqueue_batch(msm_context& ctx, u8* d_out, u32 out_stride, bool projective_out,
                       const msm_plan& plan, const typename C::addend* d_addends,
                       const void* d_api_generators, hipStream_t stream) {
  using point = typename C::point;
  using addend = typename C::addend;
  const u32 num_tasks = static_cast<u32>(plan.tasks.size());
  const u32 num_cols = static_cast<u32>(plan.columns.size());
  const u32 partial_stride = partial_stride_of(plan);

  column_desc* d_cols = ctx.arena.take<column_desc>(num_cols);
  task_desc* d_tasks = ctx.arena.take<task_desc>(num_tasks + 1);
  BZ_HIP_CHECK(hipMemcpyAsync(d_cols, plan.columns.data(), sizeof(column_desc) * num_cols,
                              hipMemcpyHostToDevice, stream));
  if (num_tasks > 0) {
    BZ_HIP_CHECK(hipMemcpyAsync(d_tasks, plan.tasks.data(), sizeof(task_desc) * num_tasks,
                                hipMemcpyHostToDevice, stream));
  }

  const bool timing = ctx.timer.recording() && num_tasks > 0;
  auto mark = [&](int stage) {
    if (timing) BZ_HIP_CHECK(hipEventRecord(ctx.timer.event(stage), stream));
  };
  if (num_tasks > 0) {
    mark(0);
    if (d_addends == nullptr) {
      addend* prepared = ctx.arena.take<addend>(plan.max_rows + 1);
      const u32 blocks = ceil_div_u32(plan.max_rows, 256);
      hipLaunchKernelGGL((k_prepare_addends<C>), dim3(blocks), dim3(256), 0, stream, prepared,
                         d_api_generators, plan.max_rows);
      d_addends = prepared;
    }
    i16* d_digits = ctx.arena.take<i16>(plan.total_entries + 8);
    u32* d_sorted = ctx.arena.take<u32>(plan.total_entries + 8);
    u32* d_hist = ctx.arena.take<u32>(plan.total_hist + 2);
    u32* d_chunk_totals = ctx.arena.take<u32>(plan.total_chunks + 1);
    u32* d_segment_bucket = ctx.arena.take<u32>(plan.total_segments + 1);
    u32* d_bucket_end = ctx.arena.take<u32>(plan.total_buckets + 1);
    point* d_bucket_sums = ctx.arena.take<point>(plan.total_buckets + 1);
    point* d_heads = ctx.arena.take<point>(plan.total_segments + 1);
    point* d_partials = ctx.arena.take<point>(static_cast<size_t>(num_tasks) * partial_stride + 1);

    mark(1);
    hipLaunchKernelGGL(k_recode, dim3(ceil_div_u32(plan.max_rows, 256), num_cols), dim3(256), 0,
                       stream, d_digits, d_cols, d_tasks);
    mark(2);

    // counting sort by bucket
    BZ_HIP_CHECK(hipMemsetAsync(d_chunk_totals, 0, sizeof(u32) * (plan.total_chunks + 1), stream));
    const size_t sort_lds = sizeof(u32) * plan.max_task_buckets;
    hipLaunchKernelGGL(k_bucket_hist, dim3(plan.max_task_slices, num_tasks), dim3(kSortThreads),
                       sort_lds, stream, d_hist, d_chunk_totals, d_digits, d_tasks);
    hipLaunchKernelGGL(k_bucket_offsets,
                       dim3(ceil_div_u32(plan.max_task_buckets, kOffsetChunkBuckets), num_tasks),
                       dim3(256), 0, stream, d_hist, d_bucket_end, d_chunk_totals, d_tasks);
    hipLaunchKernelGGL(k_bucket_scatter, dim3(plan.max_task_slices, num_tasks),
                       dim3(kSortThreads), sort_lds, stream, d_sorted, d_segment_bucket, d_hist,
                       d_digits, d_tasks);
    mark(3);

    const u32 seg_blocks =
        ceil_div_u32(plan.max_rows, static_cast<u64>(kSegmentEntries) * kAccumulateThreads);
    hipLaunchKernelGGL((k_accumulate<C>), dim3(seg_blocks, num_tasks), dim3(kAccumulateThreads), 0,
                       stream, d_bucket_sums, d_heads, d_bucket_end, d_segment_bucket, d_sorted,
                       d_addends, d_tasks);
    mark(4);

    hipLaunchKernelGGL((k_reduce<C>), dim3(partial_stride, num_tasks), dim3(kReduceThreads), 0,
                       stream, d_partials, partial_stride, d_bucket_sums, d_heads, d_bucket_end,
                       d_tasks);
    mark(5);

    hipLaunchKernelGGL((k_combine<C>), dim3(num_cols), dim3(kCombineThreads), 0, stream, d_out,
                       out_stride, projective_out ? 1 : 0, d_partials, partial_stride, d_cols);
    mark(6);
    if (timing) ctx.timer.calls += 1;
  } else {
    // every column is empty: identities only
    hipLaunchKernelGGL((k_combine<C>), dim3(num_cols), dim3(kCombineThreads), 0, stream, d_out,
                       out_stride, projective_out ? 1 : 0, nullptr, partial_stride, d_cols);
  }
  g_kernel_launches += num_tasks > 0 ? 7 : 1;
  BZ_HIP_CHECK(hipGetLastError());
}

} // namespace bz
