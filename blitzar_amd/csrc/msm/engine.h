// Host orchestration of one variable-base MSM call on one device: plan -> arena carve-up ->
// kernel sequence on a single HIP stream.  Everything is asynchronous with respect to the host;
// callers synchronise the stream (the C ABI entry points do, they are blocking like the
// reference's).
#pragma once

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include "blitzar_amd/csrc/base/device.h"
#include "blitzar_amd/csrc/msm/kernels.h"

namespace bz {

// Optional per-stage device timing (HIP event pairs on the launch stream), used by bench.py to
// report the dominant kernel's measured duration next to its algorithmic bytes.
constexpr int kNumStages = 6; // prepare, recode, sort, accumulate, reduce, combine
struct stage_timer {
  struct span {
    int stage;
    hipEvent_t begin, end;
  };
  bool enabled = false;
  std::vector<span> spans;
  size_t calls = 0;
  size_t capacity_calls = 0;

  unsigned stage_mask = 0x3f; // bit s: record stage s (an event pair costs two stream bubbles)

  void begin(size_t max_calls, unsigned mask = 0x3f) {
    release();
    capacity_calls = max_calls;
    calls = 0;
    stage_mask = mask;
    enabled = true;
  }
  bool recording() const { return enabled && calls < capacity_calls; }
  // bracket `launch()` with an event pair on `stream`
  template <class F> void timed(bool on, int stage, hipStream_t stream, F&& launch) {
    if (!on || ((stage_mask >> stage) & 1) == 0) {
      launch();
      return;
    }
    span sp{stage, nullptr, nullptr};
    BZ_HIP_CHECK(hipEventCreate(&sp.begin));
    BZ_HIP_CHECK(hipEventCreate(&sp.end));
    BZ_HIP_CHECK(hipEventRecord(sp.begin, stream));
    launch();
    BZ_HIP_CHECK(hipEventRecord(sp.end, stream));
    spans.push_back(sp);
  }
  // accumulated milliseconds per stage over the recorded calls (blocks until they finished)
  size_t collect(double out_ms[kNumStages]) {
    for (int s = 0; s < kNumStages; ++s) out_ms[s] = 0;
    for (auto& sp : spans) {
      BZ_HIP_CHECK(hipEventSynchronize(sp.end));
      float ms = 0;
      BZ_HIP_CHECK(hipEventElapsedTime(&ms, sp.begin, sp.end));
      out_ms[sp.stage] += ms;
    }
    const size_t n = calls;
    release();
    return n;
  }
  void release() {
    for (auto& sp : spans) {
      (void)hipEventDestroy(sp.begin);
      (void)hipEventDestroy(sp.end);
    }
    spans.clear();
    enabled = false;
    calls = 0;
    capacity_calls = 0;
  }
  ~stage_timer() { release(); }
};

// One context per device: the workspace arena, the pinned descriptor ring and the stage timer are
// shared by every MSM call enqueued on that device.  Calls are asynchronous on a caller stream, so
// two things keep them from trampling each other's workspace:
//   * `mu` serialises the host side (arena cursor, staging ring, timer) across caller threads;
//   * a call arriving on a DIFFERENT stream than the previous one first makes its stream wait for
//     everything enqueued on the previous stream (`order_after_previous`) -- calls on one device
//     therefore execute one after the other whatever streams they come in on; what does overlap
//     is the tail of a call with the call after it, on the context's own tail stream (below).
struct msm_context {
  device_arena arena;
  host_stage_ring descriptors; // pinned copies of the column / task descriptors in flight
  msm_tuning tuning;
  stage_timer timer;
  std::mutex mu;
  hipEvent_t last_done = nullptr;
  hipStream_t last_stream = nullptr;
  bool has_last = false;
  bool kernels_configured = false; // hipFuncSetAttribute applies to the device current at the call
  // side stream of a call: the conversion of caller generators (k_prepare_addends*: latency-bound
  // when it shares inversions) runs beside the recoding and sorting of the scalars, which do not
  // depend on it; joined before k_accumulate
  hipStream_t side = nullptr;
  hipEvent_t fork = nullptr, join = nullptr;
  // Device copies of the descriptors (columns | tasks | packed-recode ranges) in blocks of their
  // own -- two, one per tail set (below; plain calls use block 0) -- and `desc_shadow` = the bytes
  // each holds.  A batch whose descriptors are byte-identical to its block's -- the same shapes
  // over the same device pointers: a caller committing again and again from the same buffers --
  // skips the pinned staging, the H2D copy and the two stream bubbles around it (~15 us of a 1.3 ms
  // call).  Only msm_enqueue_batch writes a block, on the stream of the call and after the tail
  // that read it last has been joined; calls on one context are ordered (order_after_previous).
  // Tail stream.  The two stages after k_accumulate are latency chains that leave the machine
  // nearly idle: k_reduce runs one wavefront per SIMD, k_horner ONE workgroup per column (~250
  // dependent doublings and the encoding's 250 squarings); together 0.39 of a 1.24 ms call at
  // config 2.  In throughput mode they are enqueued on a stream of their own and run beside
  // whatever the caller's stream does next on this context: the generator conversion, recoding
  // and sorting of the next batch or call, and the start of its accumulation.  What the tail reads
  // and the next front writes -- bucket ends, bucket sums, head partials, partials, task totals --
  // then exists twice, and consecutive calls alternate (`tail_parity`); a call waits for the tail
  // that used its set two calls ago (`tail_done[parity]`) before its sort rewrites the bucket ends.
  // That only works while consecutive calls carve the arena identically (same shapes, same curve,
  // same mode: `tail_layout`); any other call first joins every pending tail (`join_tail`), and so
  // does a re-allocation of the arena, the end of a call that is not deferred, and
  // bzamd_pipeline_flush; a descriptor block is rewritten after the tail that read it was joined.  Used across calls issued through bzamd_pipeline_next
  // (whose results are complete on the caller's stream only after a later call or a flush); any
  // other call keeps everything on the caller's stream -- plain stream semantics, and forking
  // would only add stream bubbles (measured: 1.237 -> 1.266 ms at config 2).  The tail kernels
  // raise their wave priority (s_setprio): beside a k_accumulate that owns every SIMD they would
  // otherwise crawl (config 3: k_horner 1.3 -> 6.7 ms) and become the pipeline's bottleneck.
  // (Two tail streams: k_horner of call i on the second one, so that it also runs beside k_reduce of
  // call i + 1 -- the tails of a sequence then cost max(reduce, horner) per call instead of their
  // sum, which matters on the pool's slower kind of box, where they add up to more than the front
  // and the accumulation of a call; on the faster kind 1.037 -> 1.025 ms per call.)
  hipStream_t tail = nullptr, tail2 = nullptr;
  hipEvent_t tail_fork = nullptr, tail_mid = nullptr;
  hipEvent_t tail_done[2] = {nullptr, nullptr};
  bool tail_pending[2] = {false, false};
  u32 tail_parity = 0;
  u64 tail_layout = 0;        // layout tag of the calls whose tails are pending
  bool defer_tail = false;    // the next call leaves its tail pending (msm_context_defer_next_tail)
  bool overlap_tails = true;  // BLITZAR_AMD_OVERLAP_TAILS=0: never fork
  bool tail_includes_reduce = true; // BLITZAR_AMD_TAIL_REDUCE=0: only k_horner forks
  bool two_tail_streams = true;     // BLITZAR_AMD_TAIL_STREAMS=1: k_reduce and k_horner share one
  hipStream_t tail_stream() {
    if (tail == nullptr) {
      BZ_HIP_CHECK(hipStreamCreateWithFlags(&tail, hipStreamNonBlocking));
      BZ_HIP_CHECK(hipStreamCreateWithFlags(&tail2, hipStreamNonBlocking));
      BZ_HIP_CHECK(hipEventCreateWithFlags(&tail_fork, hipEventDisableTiming));
      BZ_HIP_CHECK(hipEventCreateWithFlags(&tail_mid, hipEventDisableTiming));
      for (auto& e : tail_done) BZ_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    return tail;
  }
  bool any_tail_pending() const { return tail_pending[0] || tail_pending[1]; }
  void join_tail(hipStream_t stream, u32 parity) {
    if (!tail_pending[parity]) return;
    BZ_HIP_CHECK(hipStreamWaitEvent(stream, tail_done[parity], 0));
    tail_pending[parity] = false;
  }
  void join_tail(hipStream_t stream) {
    join_tail(stream, 0);
    join_tail(stream, 1);
  }
  char* desc_dev[2] = {nullptr, nullptr};
  size_t desc_cap[2] = {0, 0};
  std::vector<char> desc_shadow[2], desc_image;
  char* descriptor_block(u32 which, size_t bytes) {
    if (bytes > desc_cap[which]) {
      // hipFree waits for the kernels reading the block
      if (desc_dev[which] != nullptr) BZ_HIP_CHECK(hipFree(desc_dev[which]));
      desc_cap[which] = bytes + bytes / 2 + 4096;
      BZ_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&desc_dev[which]), desc_cap[which]));
      desc_shadow[which].clear();
    }
    return desc_dev[which];
  }
  bool overlap_prepare = false; // BLITZAR_AMD_OVERLAP_PREPARE=1 (no gain for the HBM-bound per-point conversion)
  ~msm_context() {
    if (last_done != nullptr) (void)hipEventDestroy(last_done);
    if (fork != nullptr) (void)hipEventDestroy(fork);
    if (join != nullptr) (void)hipEventDestroy(join);
    if (side != nullptr) (void)hipStreamDestroy(side);
    for (auto& d : desc_dev) {
      if (d != nullptr) (void)hipFree(d);
    }
    if (tail_fork != nullptr) (void)hipEventDestroy(tail_fork);
    if (tail_mid != nullptr) (void)hipEventDestroy(tail_mid);
    if (tail2 != nullptr) (void)hipStreamDestroy(tail2);
    for (auto& e : tail_done) {
      if (e != nullptr) (void)hipEventDestroy(e);
    }
    if (tail != nullptr) (void)hipStreamDestroy(tail);
  }
  hipStream_t side_stream() {
    if (side == nullptr) {
      BZ_HIP_CHECK(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
      BZ_HIP_CHECK(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
      BZ_HIP_CHECK(hipEventCreateWithFlags(&join, hipEventDisableTiming));
    }
    return side;
  }
  // order `stream` behind the previous call on this context (no-op on the same stream).  The event
  // is recorded only now, on the previous call's stream -- behind that call and whatever the caller
  // enqueued there since: a superset -- so that a sequence of calls on ONE stream pays no event
  // record per call.
  void order_after_previous(hipStream_t stream) {
    if (!has_last || last_stream == stream) return;
    if (last_done == nullptr) {
      BZ_HIP_CHECK(hipEventCreateWithFlags(&last_done, hipEventDisableTiming));
    }
    if (hipEventRecord(last_done, last_stream) != hipSuccess) {
      (void)hipGetLastError(); // the caller destroyed that stream: its work drains on its own
      BZ_HIP_CHECK(hipDeviceSynchronize());
      return;
    }
    BZ_HIP_CHECK(hipStreamWaitEvent(stream, last_done, 0));
  }
  void mark_enqueued(hipStream_t stream) {
    last_stream = stream;
    has_last = true;
  }
};

// per-device kernel attributes: the partition kernels keep one counter per bucket group in dynamic
// LDS (a few KiB normally; up to 128 KiB for columns beyond 2^30 rows, plan.h).  Called with the
// context's mutex held and the context's device current.
static void configure_sort_kernels(msm_context& ctx) {
  if (ctx.kernels_configured) return;
  BZ_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_recode_packed),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024));
  BZ_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_group_hist),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024));
  BZ_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_group_scatter<true>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024));
  BZ_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_group_scatter<false>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024));
  ctx.kernels_configured = true;
}

// device workspace of one batch of columns (everything carved from the arena)
template <class C>
size_t msm_workspace_bytes(const msm_plan& plan, bool needs_addends, u32 partial_stride,
                           bool two_tail_sets = false) {
  using point = typename C::point;
  using addend = typename C::addend;
  const size_t num_tasks = plan.tasks.size(), num_cols = plan.columns.size();
  size_t need = 0;
  // (the descriptors live in a block of their own: msm_context::descriptor_block)
  if (needs_addends) need += device_arena::padded(sizeof(addend) * (plan.max_rows + 1));
  need += device_arena::padded(sizeof(i16) * (plan.total_entries + 8));
  need += 2 * device_arena::padded(sizeof(u32) * (plan.total_entries + 8));
  need += 3 * device_arena::padded(sizeof(u32) * (plan.total_groups + 1));
  need += device_arena::padded(sizeof(u32) * (num_tasks + 1));
  need += device_arena::padded(sizeof(u32) * 2 * (plan.total_buckets + 1));
  need += device_arena::padded(sizeof(u32) * (plan.total_segments + 1));
  // what the tail stages read (msm_context::tail): twice in throughput mode
  size_t tail = 0;
  tail += device_arena::padded(sizeof(u32) * (plan.total_buckets + 1));
  tail += device_arena::padded(sizeof(point) * (plan.total_buckets + 1));
  tail += device_arena::padded(sizeof(point) * (plan.total_segments + 1));
  tail += device_arena::padded(sizeof(point) * (num_tasks * partial_stride + 1));
  tail += device_arena::padded(sizeof(point) * (num_cols + 1));
  tail += device_arena::padded(sizeof(u32) * (num_tasks + 1));
  return need + (two_tail_sets ? 2 : 1) * tail;
}

static inline u32 partial_stride_of(const msm_plan& plan) {
  return plan.max_task_buckets == 0 ? 1 : ceil_div_u32(plan.max_task_buckets, plan.reduce_block_buckets());
}

template <class C>
void msm_enqueue_batch(msm_context& ctx, u8* d_out, u32 out_stride, bool projective_out,
                       const msm_plan& plan, const typename C::addend* d_addends,
                       const void* d_api_generators, hipStream_t stream, bool tail_on_side);

// Enqueue the MSM.  `d_addends` covers rows [0, max n); `d_out` receives one encoding per column
// (`out_stride` bytes apart): canonical (`C::encode`) or raw projective when `projective_out`.
// Columns are processed in batches bounded by the launch grid (tasks per batch) and by
// `msm_tuning::max_workspace_bytes`; batches reuse the same arena back to back on the stream.
// `tables`: `d_addends` is slice 0 of a window table (plan.h) whose further slices follow it.
template <class C>
void msm_enqueue(msm_context& ctx, u8* d_out, u32 out_stride, bool projective_out,
                 const std::vector<host_column>& cols, const typename C::addend* d_addends,
                 const void* d_api_generators, hipStream_t stream,
                 const window_table* tables = nullptr) {
  if (cols.empty()) return;
  std::lock_guard<std::mutex> lock(ctx.mu);
  configure_sort_kernels(ctx);
  ctx.order_after_previous(stream);
  // throughput mode is for latency-bound tails: with hundreds of columns k_reduce and k_horner
  // fill the machine themselves (plan.h, defer_max_columns), so such a call ignores the request
  // and completes on the caller's stream
  size_t nonempty_columns = 0;
  for (const auto& c : cols) nonempty_columns += c.n != 0 ? 1 : 0;
  const bool defer_tail = ctx.defer_tail && ctx.overlap_tails &&
                          nonempty_columns < ctx.tuning.defer_max_columns;
  ctx.defer_tail = false;
  msm_tuning tune = ctx.tuning;
  bool any_signed = false;
  for (const auto& c : cols) any_signed = any_signed || c.is_signed;
  // Signed columns use |x| with all digits negated: cap c at 15 so that -D fits int16 either way.
  if (any_signed && tune.max_window_bits > 15) tune.max_window_bits = 15;
  if (tables != nullptr) {
    const double table_bytes = static_cast<double>(sizeof(typename C::addend)) *
                               static_cast<double>(tables->stride) * tables->windows;
    tune.table_penalty = table_bytes > 200.0 * (1 << 20) ? 1.15 : 1.03;
  }

  // cut the columns into batches: the longest prefix of the remaining columns whose plan respects
  // the launch-grid and workspace limits (a single column is always accepted); found by bisection,
  // the usual case -- everything fits -- costs one planning pass
  std::vector<msm_plan> batches;
  std::vector<size_t> first_column;
  size_t need = 0;
  const bool needs_addends = d_addends == nullptr;
  bool two_tail_sets = false;
  auto plan_range = [&](size_t begin, size_t end, size_t& bytes) {
    msm_plan p = make_msm_plan(std::vector<host_column>(cols.begin() + begin, cols.begin() + end),
                               tune, tables);
    bytes = msm_workspace_bytes<C>(p, needs_addends, partial_stride_of(p), two_tail_sets);
    return p;
  };
  auto fits = [&](const msm_plan& p, size_t bytes) {
    // tasks and columns are launch-grid dimensions (grid.y <= 65535)
    return p.tasks.size() <= tune.max_tasks_per_batch && p.columns.size() <= 32768 &&
           bytes <= tune.max_workspace_bytes;
  };
  auto cut_batches = [&] {
    batches.clear();
    first_column.clear();
    need = 0;
    for (size_t begin = 0; begin < cols.size();) {
      size_t bytes = 0;
      size_t end = cols.size();
      msm_plan plan = plan_range(begin, end, bytes);
      if (!fits(plan, bytes) && end - begin > 1) {
        size_t lo = begin + 1, hi = end; // [begin, lo) is accepted, [begin, hi) is known not to fit
        plan = plan_range(begin, lo, bytes);
        while (hi - lo > 1) {
          const size_t mid = lo + (hi - lo) / 2;
          size_t mid_bytes = 0;
          msm_plan p = plan_range(begin, mid, mid_bytes);
          if (fits(p, mid_bytes)) {
            plan = std::move(p);
            bytes = mid_bytes;
            lo = mid;
          } else {
            hi = mid;
          }
        }
        end = lo;
      }
      if (bytes > need) need = bytes;
      first_column.push_back(begin);
      batches.push_back(std::move(plan));
      begin = end;
    }
  };
  // a deferred call runs its tail stages beside the front of the next call (msm_context::tail);
  // that mode keeps two sets of the tail's buffers
  two_tail_sets = defer_tail;
  cut_batches();
  const bool tail_on_side = two_tail_sets;
  // one allocation sized for the largest batch: later resets never reallocate (no mid-call sync)
  if (need > ctx.arena.capacity()) ctx.join_tail(stream); // the arena is about to be re-allocated
  ctx.arena.reset(need, stream);
  for (size_t k = 0; k < batches.size(); ++k) {
    ctx.arena.reset(need, stream);
    msm_enqueue_batch<C>(ctx, d_out + first_column[k] * static_cast<size_t>(out_stride), out_stride,
                         projective_out, batches[k], d_addends, d_api_generators, stream,
                         tail_on_side);
  }
  if (!defer_tail) ctx.join_tail(stream);
  ctx.mark_enqueued(stream);
}

// device arrays of one batch
template <class C> struct batch_buffers {
  column_desc* cols;
  task_desc* tasks;
  const typename C::addend* addends;
  i16* digits;
  u32* records;
  u32* sorted;
  u32* group_cursor;
  u32* group_start;
  u32* group_chunk;
  u32* big_tasks;    // [0] = count, then the tasks that have oversized groups
  u32* bucket_count; // [2][total_buckets + 1]: histograms and fill cursors of oversized groups
  u32* segment_bucket;
  u32* bucket_end;
  typename C::point* bucket_sums;
  typename C::point* heads;
  typename C::point* partials;
  typename C::point* horner_state;
  u32* task_total; // entries per task, written by k_reduce for k_horner
  u32 partial_stride;
};

template <class C>
void msm_enqueue_batch(msm_context& ctx, u8* d_out, u32 out_stride, bool projective_out,
                       const msm_plan& plan, const typename C::addend* d_addends,
                       const void* d_api_generators, hipStream_t stream, bool tail_on_side) {
  using point = typename C::point;
  using addend = typename C::addend;
  const u32 num_tasks = static_cast<u32>(plan.tasks.size());
  const u32 num_cols = static_cast<u32>(plan.columns.size());
  batch_buffers<C> b{};
  b.partial_stride = partial_stride_of(plan);
  // descriptors: one image (columns | tasks | ranges, each 16-byte aligned), copied to the
  // context's descriptor block through pinned staging -- `plan` does not outlive this call, the
  // copy does -- unless the block already holds exactly these bytes
  auto pad16 = [](size_t x) { return (x + 15) & ~size_t{15}; };
  const size_t col_bytes = pad16(sizeof(column_desc) * num_cols);
  const size_t task_bytes = pad16(sizeof(task_desc) * num_tasks);
  // packed fixed-base call? (columns = bit fields of the same wide rows, in row order): then
  // k_recode_packed reads the rows through LDS tiles, one range of columns per tile
  std::vector<recode_range> ranges = packed_recode_ranges(plan);
  const size_t range_bytes = pad16(sizeof(recode_range) * ranges.size());
  const size_t desc_bytes = col_bytes + task_bytes + range_bytes;
  std::vector<char>& image = ctx.desc_image;
  image.assign(desc_bytes, 0);
  std::memcpy(image.data(), plan.columns.data(), sizeof(column_desc) * num_cols);
  if (num_tasks != 0) {
    std::memcpy(image.data() + col_bytes, plan.tasks.data(), sizeof(task_desc) * num_tasks);
  }
  if (!ranges.empty()) {
    std::memcpy(image.data() + col_bytes + task_bytes, ranges.data(),
                sizeof(recode_range) * ranges.size());
  }
  // pending tails were carved from the arena exactly like this batch (every size that enters the
  // carving below, the curve and the mode), or they go first
  u64 layout = 0xcbf29ce484222325ull;
  for (u64 v : {static_cast<u64>(plan.total_entries), static_cast<u64>(plan.total_groups),
                static_cast<u64>(plan.total_buckets), static_cast<u64>(plan.total_segments),
                static_cast<u64>(plan.max_rows), static_cast<u64>(num_tasks),
                static_cast<u64>(num_cols), static_cast<u64>(b.partial_stride),
                static_cast<u64>(C::curve_id), static_cast<u64>(sizeof(addend)),
                static_cast<u64>(d_addends == nullptr), static_cast<u64>(tail_on_side)}) {
    layout = (layout ^ v) * 0x100000001b3ull;
  }
  if (ctx.any_tail_pending() && layout != ctx.tail_layout) ctx.join_tail(stream);
  // this batch's tail set and descriptor block (plain calls: set 0)
  const u32 parity = tail_on_side ? ctx.tail_parity : 0;
  char* desc = ctx.descriptor_block(parity, desc_bytes);
  if (image != ctx.desc_shadow[parity]) {
    ctx.join_tail(stream, parity); // the tail that read this block two batches ago
    char* staged = static_cast<char*>(ctx.descriptors.acquire(desc_bytes));
    std::memcpy(staged, image.data(), desc_bytes);
    BZ_HIP_CHECK(hipMemcpyAsync(desc, staged, desc_bytes, hipMemcpyHostToDevice, stream));
    ctx.descriptors.release(stream);
    ctx.desc_shadow[parity] = image;
  }
  b.cols = reinterpret_cast<column_desc*>(desc);
  b.tasks = reinterpret_cast<task_desc*>(desc + col_bytes);
  recode_range* d_ranges =
      ranges.empty() ? nullptr : reinterpret_cast<recode_range*>(desc + col_bytes + task_bytes);
  if (num_tasks == 0) {
    // every column is empty: identities only
    hipLaunchKernelGGL((k_horner<C>), dim3(num_cols), dim3(kCombineThreads), 0, stream, d_out,
                       out_stride, projective_out ? 1 : 0, static_cast<point*>(nullptr),
                       static_cast<const point*>(nullptr), 1u, b.cols,
                       static_cast<const task_desc*>(nullptr), static_cast<const u32*>(nullptr), 0u,
                       0u, 1, 1, plan.reduce_segment_log2);
    g_kernel_launches += 1;
    BZ_HIP_CHECK(hipGetLastError());
    return;
  }
  const bool timing = ctx.timer.recording();
  bool join_prepare = false;
  if (d_addends == nullptr) {
    addend* prepared = ctx.arena.take<addend>(plan.max_rows + 1);
    hipStream_t where = stream;
    if (ctx.overlap_prepare) {
      where = ctx.side_stream();
      BZ_HIP_CHECK(hipEventRecord(ctx.fork, stream));
      BZ_HIP_CHECK(hipStreamWaitEvent(where, ctx.fork, 0));
      join_prepare = true;
    }
    ctx.timer.timed(timing, 0, where, [&] {
      launch_prepare_addends<C>(prepared, d_api_generators, plan.max_rows, where);
    });
    if (join_prepare) BZ_HIP_CHECK(hipEventRecord(ctx.join, where));
    d_addends = prepared;
  }
  b.addends = d_addends;
  b.digits = ctx.arena.take<i16>(plan.total_entries + 8);
  b.records = ctx.arena.take<u32>(plan.total_entries + 8);
  b.sorted = ctx.arena.take<u32>(plan.total_entries + 8);
  b.group_cursor = ctx.arena.take<u32>(plan.total_groups + 1);
  b.group_start = ctx.arena.take<u32>(plan.total_groups + 1);
  b.group_chunk = ctx.arena.take<u32>(plan.total_groups + 1);
  b.big_tasks = ctx.arena.take<u32>(num_tasks + 1);
  b.bucket_count = ctx.arena.take<u32>(2 * (plan.total_buckets + 1));
  b.segment_bucket = ctx.arena.take<u32>(plan.total_segments + 1);
  // what the tail stages read: two sets in throughput mode, this batch uses set `parity`
  for (u32 set = 0; set < (tail_on_side ? 2u : 1u); ++set) {
    u32* bucket_end = ctx.arena.take<u32>(plan.total_buckets + 1);
    point* bucket_sums = ctx.arena.take<point>(plan.total_buckets + 1);
    point* heads = ctx.arena.take<point>(plan.total_segments + 1);
    point* partials = ctx.arena.take<point>(static_cast<size_t>(num_tasks) * b.partial_stride + 1);
    point* horner_state = ctx.arena.take<point>(num_cols);
    u32* task_total = ctx.arena.take<u32>(num_tasks + 1);
    if (set == parity) {
      b.bucket_end = bucket_end;
      b.bucket_sums = bucket_sums;
      b.heads = heads;
      b.partials = partials;
      b.horner_state = horner_state;
      b.task_total = task_total;
    }
  }
  const size_t part_lds = sizeof(u32) * plan.max_task_groups;
  const u32 seg_blocks =
      ceil_div_u32(plan.max_task_rows, static_cast<u64>(kAccumulateThreads) << plan.segment_log2);

  // One stream, stages in order.  (Running the sort of window group k+1 and the reduce / Horner of
  // group k-1 on side streams under the accumulation of group k was measured on MI355X and is
  // slower, 1.94 -> 2.0-2.3 ms at config 2: k_accumulate's waves hold 480 of a SIMD's 512 VGPRs,
  // so side kernels only get slots as accumulate waves retire and both sides lose.  A second
  // attempt -- two window groups, the high group's reduce + Horner on a highest-priority stream
  // beside the low group's accumulation, that launch held to two workgroups per CU with unused
  // dynamic LDS so registers stay free -- still queued the side kernels behind the accumulation:
  // 1.74 -> 1.87 ms.)
  const u64 zero_words = plan.total_groups + 1; // group cursors, cleared by the recode kernel
  ctx.timer.timed(timing, 1, stream, [&] {
    if (d_ranges != nullptr) {
      hipLaunchKernelGGL(k_recode_packed,
                         dim3(ceil_div_u32(plan.max_recode_rows, kPackedTileRows)),
                         dim3(kPackedRecodeThreads), kPackedTileBytes, stream, b.digits, b.cols,
                         b.tasks, d_ranges, static_cast<u32>(ranges.size()),
                         plan.columns[0].row_stride, plan.max_recode_rows, plan.max_rows,
                         b.group_cursor, zero_words);
      return;
    }
    const u32 chunks = ceil_div_u32(plan.max_recode_rows, 256);
    const u64 items = 8 * static_cast<u64>((chunks + 7) / 8) * num_cols;
    const u32 recode_blocks = static_cast<u32>(items < (u64{1} << 30) ? items : (u64{1} << 30));
    hipLaunchKernelGGL(k_recode, dim3(recode_blocks), dim3(256), 0, stream, b.digits, b.cols,
                       b.tasks, num_cols, chunks, b.group_cursor, zero_words);
  });
  ctx.join_tail(stream, parity); // (joined at the end of the previous batch already)
  ctx.timer.timed(timing, 2, stream, [&] {
    hipLaunchKernelGGL(k_group_hist, dim3(plan.max_task_slices, num_tasks), dim3(kSortThreads),
                       part_lds, stream, b.group_cursor, b.big_tasks, b.digits, b.tasks);
    u32* bucket_fill = b.bucket_count + plan.total_buckets + 1;
    hipLaunchKernelGGL(k_group_offsets, dim3(num_tasks), dim3(256), 0, stream, b.group_cursor,
                       b.group_start, b.group_chunk, b.bucket_count, bucket_fill, b.big_tasks,
                       b.tasks);
    // all tasks of a launch share one variant: staged unless some column needs the direct form
    if (plan.max_task_groups <= kMaxStagedGroups && plan.max_slice_rows <= kStagedSliceRows) {
      const size_t staged_lds = sizeof(u32) * (3 * plan.max_task_groups + 1 + kStagedSliceRows);
      hipLaunchKernelGGL(k_group_scatter<true>, dim3(plan.max_task_slices, num_tasks),
                         dim3(kSortThreads), staged_lds, stream, b.records, b.group_cursor,
                         b.digits, b.tasks);
    } else {
      hipLaunchKernelGGL(k_group_scatter<false>, dim3(plan.max_task_slices, num_tasks),
                         dim3(kSortThreads), part_lds, stream, b.records, b.group_cursor, b.digits,
                         b.tasks);
    }
    hipLaunchKernelGGL(k_group_sort, dim3(plan.max_task_groups, num_tasks), dim3(kGroupSortThreads),
                       0, stream, b.sorted, b.segment_bucket, b.bucket_end, b.records, b.group_start,
                       b.group_chunk, b.tasks);
    // oversized groups (skewed digits); both launches find nothing to do on uniform data
    hipLaunchKernelGGL(k_group_big_hist, dim3(kBigSortBlocks), dim3(kGroupSortThreads), 0, stream,
                       b.bucket_count, b.records, b.group_start, b.group_chunk, b.tasks,
                       b.big_tasks);
    hipLaunchKernelGGL(k_group_big_sort, dim3(kBigSortBlocks), dim3(kGroupSortThreads), 0, stream,
                       b.sorted, b.segment_bucket, b.bucket_end, b.bucket_count, bucket_fill,
                       b.records, b.group_start, b.group_chunk, b.tasks, b.big_tasks);
  });
  if (join_prepare) BZ_HIP_CHECK(hipStreamWaitEvent(stream, ctx.join, 0));
  ctx.timer.timed(timing, 3, stream, [&] {
    hipLaunchKernelGGL((k_accumulate<C>), dim3(seg_blocks, num_tasks), dim3(kAccumulateThreads), 0,
                       stream, b.bucket_sums, b.heads, b.bucket_end, b.segment_bucket, b.sorted,
                       b.addends, b.tasks);
  });
  // bucket reduction, then whole columns in one k_horner launch (the range covers every window,
  // first and last): on the caller's stream, or forked onto the tail stream (msm_context::tail)
  hipStream_t tail_stream = stream;
  auto fork = [&] {
    if (!tail_on_side) return;
    tail_stream = ctx.tail_stream();
    BZ_HIP_CHECK(hipEventRecord(ctx.tail_fork, stream));
    BZ_HIP_CHECK(hipStreamWaitEvent(tail_stream, ctx.tail_fork, 0));
  };
  if (ctx.tail_includes_reduce) fork();
  ctx.timer.timed(timing, 4, tail_stream, [&] {
    hipLaunchKernelGGL((k_reduce<C>), dim3(b.partial_stride, num_tasks), dim3(kReduceThreads), 0,
                       tail_stream, b.partials, b.partial_stride, b.task_total, b.bucket_sums,
                       b.heads, b.bucket_end, b.tasks, plan.reduce_segment_log2);
  });
  if (!ctx.tail_includes_reduce) fork();
  if (tail_on_side && ctx.tail_includes_reduce && ctx.two_tail_streams) {
    // k_horner on the second tail stream: beside the next call's k_reduce on the first
    BZ_HIP_CHECK(hipEventRecord(ctx.tail_mid, tail_stream));
    tail_stream = ctx.tail2;
    BZ_HIP_CHECK(hipStreamWaitEvent(tail_stream, ctx.tail_mid, 0));
  }
  ctx.timer.timed(timing, 5, tail_stream, [&] {
    hipLaunchKernelGGL((k_horner<C>), dim3(num_cols), dim3(kCombineThreads), 0, tail_stream, d_out,
                       out_stride, projective_out ? 1 : 0, b.horner_state, b.partials,
                       b.partial_stride, b.cols, b.tasks, b.task_total, 0u, 0xffffffffu, 1, 1,
                       plan.reduce_segment_log2);
  });
  if (tail_on_side) {
    BZ_HIP_CHECK(hipEventRecord(ctx.tail_done[parity], tail_stream));
    ctx.tail_pending[parity] = true;
    ctx.tail_layout = layout;
    ctx.tail_parity = parity ^ 1;
    // the tail before this one has had this batch's front and accumulation to finish beside: the
    // caller's stream picks it up here (so a deferred call's result is complete on the stream
    // once the next call has been enqueued)
    ctx.join_tail(stream, parity ^ 1);
  }
  if (timing) ctx.timer.calls += 1;
  g_kernel_launches += 10;
  BZ_HIP_CHECK(hipGetLastError());
}

} // namespace bz
