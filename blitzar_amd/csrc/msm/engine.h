// Host orchestration of one variable-base MSM call on one device: plan -> arena carve-up ->
// kernel sequence on a single HIP stream.  Everything is asynchronous with respect to the host;
// callers synchronise the stream (the C ABI entry points do, they are blocking like the
// reference's).
#pragma once

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <type_traits>
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
  // record one batch in `every` (a sample of the launches: every recorded stage is an event pair,
  // two stream bubbles); `calls` counts the batches recorded
  size_t every = 1, seen = 0;

  void begin(size_t max_calls, unsigned mask = 0x3f, size_t sample_every = 1) {
    release();
    capacity_calls = max_calls;
    calls = 0;
    stage_mask = mask;
    every = sample_every == 0 ? 1 : sample_every;
    seen = 0;
    enabled = true;
  }
  // asked once per batch: is this one recorded?
  bool recording() {
    if (!enabled || calls >= capacity_calls) return false;
    return seen++ % every == 0;
  }
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

// A stage's completion mark: an event and the stream it was last recorded on.  A wait from the same
// stream is implied by stream order and skipped (every record / wait is a packet in the queue).
struct stage_mark {
  hipEvent_t ev = nullptr;
  hipStream_t on = nullptr;
  bool set = false;
  void record(hipStream_t stream) {
    if (ev == nullptr) BZ_HIP_CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    BZ_HIP_CHECK(hipEventRecord(ev, stream));
    on = stream;
    set = true;
  }
  void wait(hipStream_t stream) const {
    if (!set || stream == on) return;
    BZ_HIP_CHECK(hipStreamWaitEvent(stream, ev, 0));
  }
  void destroy() {
    if (ev != nullptr) (void)hipEventDestroy(ev);
    ev = nullptr;
    set = false;
  }
};

// One context per device: the workspace arena, the pinned descriptor ring and the stage timer are
// shared by every MSM call enqueued on that device.  Calls are asynchronous on a caller stream, so
// two things keep them from trampling each other's workspace:
//   * `mu` serialises the host side (arena cursor, staging ring, timer) across caller threads;
//   * a call arriving on a DIFFERENT stream than the previous one first makes its stream wait for
//     everything enqueued on the previous stream (`order_after_previous`) -- calls on one device
//     therefore execute one after the other whatever streams they come in on; what does overlap
//     are the stages of consecutive calls of a pipelined sequence (throughput mode, below).
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
  // Device copies of the descriptors (columns | tasks | packed-recode ranges) in blocks of their
  // own -- two, one per buffer set of the throughput mode (plain calls use block 0) -- and
  // `desc_shadow` = the bytes each holds.  A batch whose descriptors are byte-identical to its
  // block's -- the same shapes over the same device pointers: a caller committing again and again
  // from the same buffers -- skips the pinned staging, the H2D copy and the two stream bubbles
  // around it (~15 us of a 1.3 ms call).  Only msm_enqueue_batch writes a block, after the last
  // stage that read it (k_horner, two batches ago) has finished.
  //
  // Throughput mode (bzamd_pipeline_next; batch = one launch sequence, a call of a few columns is
  // one batch).  A batch has four stages:
  //   front      conversion of caller generators, recoding, the two-pass bucket sort: short kernels
  //              (0.2-0.3 ms at 2^20 curve25519 rows, ~0.64 GB moved);
  //   accumulate the bucket additions: integer-issue bound, every SIMD full (0.65-0.7 ms);
  //   reduce     bucket reduction: a latency chain at one wavefront per SIMD (0.2-0.3 ms);
  //   horner     ONE workgroup per column, ~250 dependent doublings + the encoding (0.2-0.4 ms).
  // In a sequence of same-shaped calls the two tail stages run on streams of their own, beside the
  // front and the accumulation of the NEXT batch (they are tiny grids that slot in anywhere and
  // raise their wave priority, s_setprio): a step costs front + accumulate instead of the sum of
  // all four.  The buffers a stage writes while an earlier batch's later stage still reads exist
  // more than once, indexed by the batch's sequence number; completion marks (rings of 4 events)
  // order a stage behind the stage of an earlier batch whose buffers it reuses.
  //
  // (Front and accumulation on streams of their own as well -- plain streams, CU masks, queue
  // priorities -- were built and measured in rounds 3 and 4 and removed: the front is not HBM-bound,
  // there is nothing complementary to overlap, and what the arrangement gains or loses depends on how
  // a process's streams share the hardware queues.  DESIGN.md history table; logs
  // profiles/round3_ab_front_*.log, profiles/round4_front_arrangements.txt.)
  // A pipelined result is complete on the caller's stream once two further calls have
  // been enqueued, or after bzamd_pipeline_flush.  Only while consecutive batches carve the arena
  // identically (same shapes, curve, mode: `pipe_layout`); any other call first joins everything
  // pending, and so does a re-allocation of the arena, a call outside the mode and a flush.
  // BLITZAR_AMD_OVERLAP_TAILS=0 switches the mode off altogether.  A lone call never forks (every
  // fork / join pair costs ~25 us of stream bubbles).
  hipStream_t tail = nullptr, tail2 = nullptr; // k_reduce / k_horner of a pipelined batch
  stage_mark acc_done[4], reduce_done[4], horner_done[4];
  // `pre_horner[k]`: recorded on the k_horner stream behind its wait for reduce(k) and in front of
  // horner(k) -- that stream runs in order, so the mark fires when horner(k - 1) AND reduce(k) are
  // done.  The caller's stream waits for this ONE mark per batch (end of batch k: pre_horner[k - 1])
  // instead of for horner_done[k - 2] there and for reduce_done[k - 1] in front of the next batch's
  // front and again in front of its accumulation: every cross-stream wait is a barrier packet that
  // costs the queue 8-12 us (round 6 timeline: 24 us between k_accumulate and the next front, 13 us
  // in front of k_accumulate -- profiles/round6_timeline_sequence.txt).
  stage_mark pre_horner[4];
  u64 reduce_joined = 0; // `reduce_joined_on` has waited for the reduce of every batch below this
  hipStream_t reduce_joined_on = nullptr;
  // Per-call window tables (plan.h, choose_call_table; built by curve_tu.h, build_call_table): the
  // 2^(c w) multiples of a call's caller generators, rebuilt by every call that wants them into this
  // grow-only block.  Only k_accumulate reads the table, so the build -- a chain of W c dependent
  // doublings per generator on a handful of wavefronts, then one normalisation launch -- runs on a
  // side stream beside the recoding and the sort of the same call (`table_overlap`;
  // BLITZAR_AMD_CALL_TABLE_OVERLAP=0 keeps it on the caller's stream) and the accumulation waits for
  // `table_ready`.  BLITZAR_AMD_CALL_TABLES=0 switches the tables off, BLITZAR_AMD_CALL_TABLE_BITS=c
  // (6..16) forces a table of that width for every call with caller generators (tests, A/B runs).
  device_arena call_table;
  hipStream_t side = nullptr;
  stage_mark table_fork, table_ready;
  bool table_pending = false;
  bool call_tables = true;
  bool table_overlap = true;
  bool wave_chain = true; // BLITZAR_AMD_CALL_TABLE_WAVE_CHAIN=0: a lane per generator whatever the set's size
  // curve25519, BLITZAR_AMD_NORMALISE_CALLER=1: caller generators are normalised to Z = 1 in every
  // call (kernels.h, k_batch_*: three launches, no workgroup waits for an inversion) and the
  // accumulation runs the 7-product loop of resident sets.  OFF by default, measured on MI355X at
  // config 2 (profiles/round6_ab_normalise_caller.log): k_accumulate 0.618 -> 0.573 ms, but the
  // normalisation takes 0.200 ms against the 0.066 of the plain conversion -- a lone call 1.17 ->
  // 1.26 ms, a step in sequence 0.978 -> 1.054 -- and on the side stream beside recode + sort
  // (BLITZAR_AMD_CALL_TABLE_OVERLAP) every memory-bound kernel of the front takes 2.5-4x as long
  // (recode 0.018 -> 0.07, sort 0.11 -> 0.27; step 1.16).
  bool normalise_caller = false;
  u32 force_call_table_bits = 0;
  u64 call_tables_built = 0; // (tests: bzamd_set_call_tables returns it)
  const void* call_table_rows = nullptr; // slice 0 of the table built last
  void make_side_stream() {
    if (side != nullptr) return;
    BZ_HIP_CHECK(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
  }
  u64 seq = 0;            // pipelined batches enqueued so far on this context
  u64 joined = 0;         // the caller's stream `joined_on` has waited for every batch below this
  hipStream_t joined_on = nullptr;
  u64 pipe_layout = 0;        // layout tag of the pending batches
  bool defer_tail = false;    // the next call runs in throughput mode (msm_context_defer_next_tail)
  bool overlap_tails = true;  // BLITZAR_AMD_OVERLAP_TAILS=0: never fork
  // The tail streams are non-blocking streams of the LOWEST priority.  Non-blocking: callers on the
  // NULL stream must not meet a blocking stream (every NULL-stream launch takes a dependency on
  // every blocking stream of the process, even an idle one: 1.00 -> 1.19 ms per call).  Lowest
  // priority: the HIP runtime multiplexes the streams of a process over a few hardware queues PER
  // PRIORITY LEVEL (4 by default) and packets of one queue execute in order, so an internal stream
  // that lands on the queue of another one (PyTorch creates 32 normal-priority streams at its first
  // side stream) inherits that stream's waits and the stages serialise again (1.12 -> 1.48 ms per
  // step under torch); the lowest level is rarely used by anybody else.  And the tails have two
  // calls' time to finish: whatever the caller's stream has ready goes first (a sequence of
  // config-2 calls 1.009 -> 0.988 ms per call against tails on CU-masked queues of their own, the
  // round's first answer to the multiplexing; profiles/round3_ab_tail_priority.log).
  //
  // k_reduce_compact (the point addition at three places instead of ten: the code a wavefront walks
  // fits the instruction cache) instead of k_reduce.  Measured on both kinds of MI355X boxes
  // (profiles/round4_ab_compact_tails.log): where code beyond the instruction cache is fetched at half
  // speed (`slow_instruction_fetch`, probed once per device at context creation) a lone k_reduce goes
  // 0.317 -> 0.253 ms on curve25519 and 2.27 -> 1.24 ms on bls12-381 (a sequence of 2^22-row bls12-381
  // columns 13.2 -> 12.3 ms per call); on the other kind the inlined form is faster (0.185 against
  // 0.26), and a launch of many columns prefers it on both (its wavefronts share what they fetch).
  // BLITZAR_AMD_COMPACT_REDUCE: 0 never, 1 always, 2 (default) where the probe and the launch say so
  u32 compact_reduce = 2;
  bool slow_instruction_fetch = false;
  // compute units a launch on `stream` may use (a caller's stream may carry a CU mask).
  // k_group_sort_all's workers must all be resident at once.  Asked on every call -- a getter on the
  // stream object; a handle value can come back for a different stream, so nothing is cached per handle.
  u32 device_cus = 0;
  u32 stream_cus(hipStream_t stream) {
    if (device_cus == 0) {
      int device = 0, cus = 0;
      BZ_HIP_CHECK(hipGetDevice(&device));
      BZ_HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device));
      device_cus = cus > 0 ? static_cast<u32>(cus) : 1;
    }
    u32 usable = device_cus;
    if (stream != nullptr) {
      uint32_t mask[32] = {};
      if (hipExtStreamGetCUMask(stream, 32, mask) == hipSuccess) {
        u32 bits = 0;
        for (u32 w = 0; w < 32; ++w) bits += static_cast<u32>(__builtin_popcount(mask[w]));
        if (bits != 0 && bits < usable) usable = bits;
      } else {
        (void)hipGetLastError();
      }
    }
    return usable;
  }
  void make_pipe_streams() {
    if (tail != nullptr) return;
    int least = 0, greatest = 0;
    BZ_HIP_CHECK(hipDeviceGetStreamPriorityRange(&least, &greatest));
    BZ_HIP_CHECK(hipStreamCreateWithPriority(&tail, hipStreamNonBlocking, least));
    BZ_HIP_CHECK(hipStreamCreateWithPriority(&tail2, hipStreamNonBlocking, least));
  }
  bool any_pending() const { return joined < seq; }
  // End of pipelined batch k: the caller's stream picks up the batch TWO before it (whose buffers
  // batch k reused, so it is long done): a pipelined result is complete on the stream once two
  // further calls have been enqueued, or after a flush.  (Waiting for the previous batch here
  // would put its k_horner in front of whatever the caller enqueues next.)
  // BLITZAR_AMD_MERGED_WAITS=1: one wait per batch (pre_horner) instead of three.  OFF: measured on
  // one MI355X box, 4 x 300 steps each (profiles/round6_ab_merged_waits.log): the gaps shrink (24 ->
  // 17 us behind k_accumulate, 13 -> 6.5 us in front of it) and the step does not -- 0.9701 against
  // 0.9729 ms with the single wait: k_reduce and the sort stretch by what the gaps gave up.  The step
  // in throughput mode is bound by the work of its stages, not by the packets between them.
  bool merged_waits = false;
  // BLITZAR_AMD_ACC_LDS_PAD=<bytes>: dynamic LDS requested by k_accumulate (which uses none), i.e. a cap
  // on its workgroups per compute unit -- 55000 leaves two of the three wavefronts per SIMD its
  // registers allow, and the third slot's registers to whatever runs beside it (the tails of the
  // previous calls in throughput mode)
  u32 acc_lds_pad = 0;
  void join_two_back(hipStream_t stream, u64 k) {
    if (!merged_waits) {
      if (k >= 2 && (joined < k - 1 || stream != joined_on)) {
        horner_done[(k - 2) & 3].wait(stream);
        joined = k - 1;
        joined_on = stream;
      }
      return;
    }
    if (k >= 1 && (joined < k - 1 || reduce_joined < k || stream != joined_on ||
                   stream != reduce_joined_on)) {
      pre_horner[(k - 1) & 3].wait(stream); // horner(k - 2) and reduce(k - 1)
      joined = k - 1;
      joined_on = stream;
      reduce_joined = k;
      reduce_joined_on = stream;
    }
  }
  // has `stream` already waited for the reduce of batch `batch`?
  bool reduce_is_joined(hipStream_t stream, u64 batch) const {
    return reduce_joined > batch && stream == reduce_joined_on;
  }
  // make `stream` wait for every pipelined batch enqueued so far (k_horner runs on ONE stream, in
  // order, and is the last stage of a batch: the last batch's mark covers everything)
  void join_all(hipStream_t stream) {
    if (seq == 0 || (joined == seq && stream == joined_on)) return;
    horner_done[(seq - 1) & 3].wait(stream);
    joined = seq;
    joined_on = stream;
    reduce_joined = seq; // (horner(k) runs behind reduce(k))
    reduce_joined_on = stream;
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
  ~msm_context() {
    if (last_done != nullptr) (void)hipEventDestroy(last_done);
    for (auto& d : desc_dev) {
      if (d != nullptr) (void)hipFree(d);
    }
    for (int i = 0; i < 4; ++i) {
      acc_done[i].destroy();
      reduce_done[i].destroy();
      horner_done[i].destroy();
      pre_horner[i].destroy();
    }
    table_fork.destroy();
    table_ready.destroy();
    for (hipStream_t s : {side, tail2, tail}) {
      if (s != nullptr) (void)hipStreamDestroy(s);
    }
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
  auto allow_lds = [](auto kernel) {
    BZ_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024));
  };
  allow_lds(k_recode_packed<i16>);
  allow_lds(k_recode_packed<i32>);
  allow_lds(k_group_hist<i16>);
  allow_lds(k_group_hist<i32>);
  allow_lds(k_group_scatter<true, i16>);
  allow_lds(k_group_scatter<true, i32>);
  allow_lds(k_group_scatter<false, i16>);
  allow_lds(k_group_scatter<false, i32>);
  ctx.kernels_configured = true;
}

// buffer sets of a batch (msm_context: throughput mode)
struct pipe_mode {
  bool piped = false; // reduce + horner on the tail streams
  u32 end_sets() const { return piped ? 2 : 1; }
  u32 tail_sets() const { return piped ? 2 : 1; }
};

// The block of words the recode kernel clears for the sort of the same batch, one allocation:
//   group cursors [total_groups + 1] | arrival tickets of k_group_hist [num_tasks] | barrier of the
//   oversized-group workers [1] | big_tasks: count [1] + task list [num_tasks]
// (`zero_words` = everything up to and including the count)
static inline size_t zeroed_block_words(u64 total_groups, size_t num_tasks) {
  return static_cast<size_t>(total_groups) + 1 + num_tasks + 1 + 1 + num_tasks;
}

// device workspace of one batch of columns (everything carved from the arena)
template <class C>
size_t msm_workspace_bytes(const msm_plan& plan, bool needs_addends, u32 partial_stride,
                           pipe_mode mode = {}) {
  using point = typename C::point;
  using addend = typename C::addend;
  const size_t num_tasks = plan.tasks.size(), num_cols = plan.columns.size();
  // (the descriptors live in a block of their own: msm_context::descriptor_block)
  // what the front writes and the accumulation reads
  size_t front = 0;
  if (needs_addends) {
    front += device_arena::padded(sizeof(addend) * (plan.max_rows + 1));
    if constexpr (C::has_batched_prepare) {
      front += device_arena::padded(sizeof(typename C::batch_fe) *
                                    batch_prepare_scratch_elements(plan.max_rows));
    }
  }
  front += device_arena::padded((plan.wide_digits ? sizeof(i32) : sizeof(i16)) * (plan.total_entries + 8));
  front += 2 * device_arena::padded(sizeof(u32) * (plan.total_entries + 8));
  front += 2 * device_arena::padded(sizeof(u32) * (plan.total_groups + 1));
  front += device_arena::padded(sizeof(u32) * zeroed_block_words(plan.total_groups, num_tasks));
  front += device_arena::padded(sizeof(u32) * 2 * (plan.total_buckets + 1));
  front += device_arena::padded(sizeof(u32) * (plan.total_segments + 1));
  const size_t ends = device_arena::padded(sizeof(u32) * (plan.total_buckets + 1));
  // what accumulate, reduce and horner hand on
  size_t tail = 0;
  tail += device_arena::padded(sizeof(point) * (plan.total_buckets + 1));
  tail += device_arena::padded(sizeof(point) * (plan.total_segments + 1));
  tail += device_arena::padded(sizeof(point) * (num_tasks * partial_stride + 1));
  tail += device_arena::padded(sizeof(point) * (num_cols + 1));
  tail += device_arena::padded(sizeof(u32) * (num_tasks + 1));
  return front + mode.end_sets() * ends + mode.tail_sets() * tail;
}

static inline u32 partial_stride_of(const msm_plan& plan) {
  return plan.max_task_buckets == 0 ? 1 : ceil_div_u32(plan.max_task_buckets, plan.reduce_block_buckets());
}

template <class C>
void msm_enqueue_batch(msm_context& ctx, u8* d_out, u32 out_stride, bool projective_out,
                       const msm_plan& plan, const typename C::addend* d_addends,
                       const void* d_api_generators, hipStream_t stream, pipe_mode mode);

// Enqueue the MSM.  `d_addends` covers rows [0, max n); `d_out` receives one encoding per column
// (`out_stride` bytes apart): canonical (`C::encode`) or raw projective when `projective_out`.
// Columns are processed in batches bounded by the launch grid (tasks per batch) and by
// `msm_tuning::max_workspace_bytes`; batches reuse the same arena back to back on the stream.
// `tables`: `d_addends` is slice 0 of a window table (plan.h) whose further slices follow it.
// (`msm_enqueue_locked`: the caller holds ctx.mu, has configured the kernels and ordered `stream`
// behind the previous call -- curve_tu.h builds a per-call window table in between;
// `force_tables`: merge every column the table allows, whatever the cost model says)
template <class C>
void msm_enqueue_locked(msm_context& ctx, u8* d_out, u32 out_stride, bool projective_out,
                        const std::vector<host_column>& cols, const typename C::addend* d_addends,
                        const void* d_api_generators, hipStream_t stream,
                        const window_table* tables = nullptr, bool force_tables = false);

template <class C>
void msm_enqueue(msm_context& ctx, u8* d_out, u32 out_stride, bool projective_out,
                 const std::vector<host_column>& cols, const typename C::addend* d_addends,
                 const void* d_api_generators, hipStream_t stream,
                 const window_table* tables = nullptr) {
  if (cols.empty()) return;
  std::lock_guard<std::mutex> lock(ctx.mu);
  configure_sort_kernels(ctx);
  ctx.order_after_previous(stream);
  msm_enqueue_locked<C>(ctx, d_out, out_stride, projective_out, cols, d_addends, d_api_generators,
                        stream, tables);
}

template <class C>
void msm_enqueue_locked(msm_context& ctx, u8* d_out, u32 out_stride, bool projective_out,
                        const std::vector<host_column>& cols, const typename C::addend* d_addends,
                        const void* d_api_generators, hipStream_t stream,
                        const window_table* tables, bool force_tables) {
  // throughput mode is for latency-bound tails: with hundreds of columns k_reduce and k_horner
  // fill the machine themselves (plan.h, defer_max_columns), so such a call ignores the request
  // and completes on the caller's stream
  size_t nonempty_columns = 0;
  for (const auto& c : cols) nonempty_columns += c.n != 0 ? 1 : 0;
  pipe_mode mode;
  mode.piped = ctx.defer_tail && ctx.overlap_tails && nonempty_columns < ctx.tuning.defer_max_columns;
  ctx.defer_tail = false;
  msm_tuning tune = ctx.tuning;
  tune.in_sequence = mode.piped;
  if (force_tables) tune.force_window_tables = true;
  tune.accumulate_wave_slots =
      static_cast<u32>(C::accumulate_waves_per_simd) * 4 * ctx.stream_cus(stream);
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
  auto plan_range = [&](size_t begin, size_t end, size_t& bytes) {
    msm_plan p = make_msm_plan(std::vector<host_column>(cols.begin() + begin, cols.begin() + end),
                               tune, tables);
    bytes = msm_workspace_bytes<C>(p, needs_addends, partial_stride_of(p), mode);
    return p;
  };
  auto fits = [&](const msm_plan& p, size_t bytes) {
    // tasks and columns are launch-grid dimensions (grid.y <= 65535)
    return p.tasks.size() <= tune.max_tasks_per_batch && p.columns.size() <= 32768 &&
           bytes <= tune.max_workspace_bytes;
  };
  for (size_t begin = 0; begin < cols.size();) {
    size_t bytes = 0;
    size_t end = cols.size();
    msm_plan plan = plan_range(begin, end, bytes);
    if (!fits(plan, bytes) && end - begin > 1) {
      size_t lo = begin + 1, hi = end; // [begin, lo) is accepted, [begin, hi) is known not to fit
      // Columns of one call mostly share a shape: try the prefix the limits allow in proportion
      // first (3 % short of it) and take it if it fits -- a planning pass over a thousand columns
      // costs ~0.5 ms of host time, and ten of them per call made launches of 1024 short columns
      // host-bound (7-11 ms of enqueueing for 4 ms of device work).
      {
        const double by_tasks = static_cast<double>(tune.max_tasks_per_batch) /
                                static_cast<double>(plan.tasks.size() ? plan.tasks.size() : 1);
        const double by_bytes = static_cast<double>(tune.max_workspace_bytes) /
                                static_cast<double>(bytes ? bytes : 1);
        const double share = 0.97 * (by_tasks < by_bytes ? by_tasks : by_bytes);
        const size_t guess = begin + static_cast<size_t>(static_cast<double>(end - begin) * share);
        if (guess > begin + 1 && guess < end) {
          size_t guess_bytes = 0;
          msm_plan p = plan_range(begin, guess, guess_bytes);
          if (fits(p, guess_bytes)) {
            if (guess_bytes > need) need = guess_bytes;
            first_column.push_back(begin);
            batches.push_back(std::move(p));
            begin = guess;
            continue;
          }
          hi = guess;
        }
      }
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
  if (mode.piped) ctx.make_pipe_streams();
  // a call outside the mode uses buffer set 0 on the caller's stream: everything pending first
  if (!mode.piped) ctx.join_all(stream);
  // one allocation sized for the largest batch: later resets never reallocate (no mid-call sync)
  if (need > ctx.arena.capacity()) ctx.join_all(stream); // the arena is about to be re-allocated
  ctx.arena.reset(need, stream);
  for (size_t k = 0; k < batches.size(); ++k) {
    ctx.arena.reset(need, stream);
    msm_enqueue_batch<C>(ctx, d_out + first_column[k] * static_cast<size_t>(out_stride), out_stride,
                         projective_out, batches[k], d_addends, d_api_generators, stream, mode);
  }
  ctx.mark_enqueued(stream);
}

// device arrays of one batch
template <class C> struct batch_buffers {
  column_desc* cols;
  task_desc* tasks;
  const typename C::addend* addends;
  void* digits; // i16 [total_entries], or i32 when plan.wide_digits
  u32* records;
  u32* sorted;
  u32* group_cursor;
  u32* arrivals;    // tickets of k_group_hist's workgroups, per task
  u32* big_barrier; // arrival counter of the oversized-group workers
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
                       const void* d_api_generators, hipStream_t stream, pipe_mode mode) {
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
  // pending batches were carved from the arena exactly like this one (every size that enters the
  // carving below, the curve and the mode), or they go first
  u64 layout = 0xcbf29ce484222325ull;
  for (u64 v : {static_cast<u64>(plan.total_entries), static_cast<u64>(plan.total_groups),
                static_cast<u64>(plan.total_buckets), static_cast<u64>(plan.total_segments),
                static_cast<u64>(plan.max_rows), static_cast<u64>(num_tasks),
                static_cast<u64>(num_cols), static_cast<u64>(b.partial_stride),
                static_cast<u64>(C::curve_id), static_cast<u64>(sizeof(addend)),
                static_cast<u64>(d_addends == nullptr),
                static_cast<u64>(mode.piped) | static_cast<u64>(plan.wide_digits) << 1 |
                    static_cast<u64>(C::has_batched_prepare) << 2}) {
    layout = (layout ^ v) * 0x100000001b3ull;
  }
  if (ctx.any_pending() && layout != ctx.pipe_layout) ctx.join_all(stream);
  // this batch's place in the sequence, its buffer sets, its streams
  const u64 k = ctx.seq;
  const u32 parity = mode.piped ? static_cast<u32>(k & 1) : 0;
  const u32 end_set = mode.piped ? static_cast<u32>(k % mode.end_sets()) : 0;
  hipStream_t fs = stream, as = stream; // front and accumulation: the caller's stream
  hipStream_t rs = mode.piped ? ctx.tail : stream;
  hipStream_t hs = mode.piped ? ctx.tail2 : stream;
  // completion marks of earlier batches (none outside the mode: join_all came first)
  auto earlier = [&](stage_mark* ring, u64 back) -> const stage_mark* {
    return mode.piped && k >= back ? &ring[(k - back) & 3] : nullptr;
  };
  auto wait_for = [](const stage_mark* m, hipStream_t s) {
    if (m != nullptr) m->wait(s);
  };
  char* desc = ctx.descriptor_block(parity, desc_bytes);
  if (image != ctx.desc_shadow[parity]) {
    wait_for(earlier(ctx.horner_done, 2), fs); // the last reader of this block
    char* staged = static_cast<char*>(ctx.descriptors.acquire(desc_bytes));
    std::memcpy(staged, image.data(), desc_bytes);
    BZ_HIP_CHECK(hipMemcpyAsync(desc, staged, desc_bytes, hipMemcpyHostToDevice, fs));
    ctx.descriptors.release(fs);
    ctx.desc_shadow[parity] = image;
  }
  b.cols = reinterpret_cast<column_desc*>(desc);
  b.tasks = reinterpret_cast<task_desc*>(desc + col_bytes);
  recode_range* d_ranges =
      ranges.empty() ? nullptr : reinterpret_cast<recode_range*>(desc + col_bytes + task_bytes);
  if (num_tasks == 0) {
    // every column is empty: identities only, on the stream that carries every k_horner (in order
    // behind the previous batch's), behind the descriptor upload
    if (mode.piped) {
      ctx.acc_done[k & 3].record(fs);
      ctx.acc_done[k & 3].wait(hs);
    }
    hipLaunchKernelGGL((k_horner<C>), dim3(num_cols), dim3(kCombineThreads), 0, hs, d_out,
                       out_stride, projective_out ? 1 : 0, static_cast<point*>(nullptr),
                       static_cast<const point*>(nullptr), 1u, b.cols,
                       static_cast<const task_desc*>(nullptr), static_cast<const u32*>(nullptr), 0u,
                       0u, 1, 1, plan.reduce_block_log2());
    g_kernel_launches += 1;
    BZ_HIP_CHECK(hipGetLastError());
    if (mode.piped) {
      // keeps its place in the sequence: every mark of the slot points behind this launch
      ctx.acc_done[k & 3].record(hs);
      ctx.reduce_done[k & 3].record(hs);
      if (ctx.merged_waits) ctx.pre_horner[k & 3].record(hs);
      ctx.horner_done[k & 3].record(hs);
      ctx.pipe_layout = layout;
      ctx.seq = k + 1;
      ctx.join_two_back(stream, k);
    }
    return;
  }
  const bool timing = ctx.timer.recording();
  [[maybe_unused]] void* prepare_scratch = nullptr; // k_batch_* (curves that normalise per call)
  // carve the arena: the same walk for every batch of a layout, this batch's sets picked out
  {
    addend* prepared = d_addends == nullptr ? ctx.arena.take<addend>(plan.max_rows + 1) : nullptr;
    if constexpr (C::has_batched_prepare) {
      if (d_addends == nullptr) {
        prepare_scratch = ctx.arena.take<typename C::batch_fe>(
            batch_prepare_scratch_elements(plan.max_rows));
      }
    }
    void* digits = plan.wide_digits ? static_cast<void*>(ctx.arena.take<i32>(plan.total_entries + 8))
                                    : static_cast<void*>(ctx.arena.take<i16>(plan.total_entries + 8));
    u32* records = ctx.arena.take<u32>(plan.total_entries + 8);
    u32* sorted = ctx.arena.take<u32>(plan.total_entries + 8);
    u32* group_cursor = ctx.arena.take<u32>(zeroed_block_words(plan.total_groups, num_tasks));
    u32* group_start = ctx.arena.take<u32>(plan.total_groups + 1);
    u32* group_chunk = ctx.arena.take<u32>(plan.total_groups + 1);
    u32* arrivals = group_cursor + plan.total_groups + 1;
    u32* big_barrier = arrivals + num_tasks;
    u32* big_tasks = big_barrier + 1;
    u32* bucket_count = ctx.arena.take<u32>(2 * (plan.total_buckets + 1));
    u32* segment_bucket = ctx.arena.take<u32>(plan.total_segments + 1);
    b.addends = d_addends == nullptr ? prepared : d_addends;
    b.digits = digits;
    b.records = records;
    b.sorted = sorted;
    b.group_cursor = group_cursor;
    b.arrivals = arrivals;
    b.big_barrier = big_barrier;
    b.group_start = group_start;
    b.group_chunk = group_chunk;
    b.big_tasks = big_tasks;
    b.bucket_count = bucket_count;
    b.segment_bucket = segment_bucket;
  }
  for (u32 set = 0; set < mode.end_sets(); ++set) {
    u32* bucket_end = ctx.arena.take<u32>(plan.total_buckets + 1);
    if (set == end_set) b.bucket_end = bucket_end;
  }
  for (u32 set = 0; set < mode.tail_sets(); ++set) {
    point* bucket_sums = ctx.arena.take<point>(plan.total_buckets + 1);
    point* heads = ctx.arena.take<point>(plan.total_segments + 1);
    point* partials = ctx.arena.take<point>(static_cast<size_t>(num_tasks) * b.partial_stride + 1);
    point* horner_state = ctx.arena.take<point>(num_cols);
    u32* task_total = ctx.arena.take<u32>(num_tasks + 1);
    if (set == parity) {
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

  // ---- front: on the caller's stream, behind the previous batch's accumulation.  The bucket ends
  // it rewrites were last read by the reduce `end_sets` batches ago.
  // (Overlapping stages of ONE call was measured in round 1 and is slower: k_accumulate's waves
  // hold 480 of a SIMD's 512 VGPRs, side kernels only get slots as they retire, and everything a
  // call runs feeds its next stage.)
  if (!(mode.piped && k >= mode.end_sets() && ctx.reduce_is_joined(fs, k - mode.end_sets()))) {
    wait_for(earlier(ctx.reduce_done, mode.end_sets()), fs);
  }
  // caller generators -> addends.  (Dealing this kernel's workgroups into the group sort's launch --
  // the one HBM-saturating kernel of the front inside the LDS-bound one -- was built and measured in
  // round 3: sort + conversion 0.171 -> 0.183 ms, profiles/round3_ab_front_fusion.log; removed.)
  if (d_addends == nullptr) {
    if constexpr (C::has_batched_prepare) {
      // Z = 1 normalisation (kernels.h, k_batch_*): only k_accumulate reads the addends, so the three
      // launches -- the middle one a latency chain on ONE compute unit -- run on the side stream
      // beside the recoding and the sort, like the build of a per-call window table
      hipStream_t ps = fs;
      if (ctx.table_overlap) {
        ctx.make_side_stream();
        ctx.table_fork.record(fs);
        ctx.table_fork.wait(ctx.side);
        ps = ctx.side;
      }
      ctx.timer.timed(timing, 0, ps, [&] {
        launch_prepare_addends_split<C>(const_cast<addend*>(b.addends), d_api_generators,
                                        plan.max_rows,
                                        static_cast<typename C::batch_fe*>(prepare_scratch), ps);
      });
      g_kernel_launches += 2;
      if (ctx.table_overlap) {
        ctx.table_ready.record(ps);
        ctx.table_pending = true;
      }
    } else {
      ctx.timer.timed(timing, 0, fs, [&] {
        launch_prepare_addends<C>(const_cast<addend*>(b.addends), d_api_generators, plan.max_rows, fs);
      });
    }
  }
  // group cursors, arrival tickets, the workers' barrier, big_tasks[0]: cleared by the recode kernel
  const u64 zero_words = plan.total_groups + 1 + num_tasks + 2;
  // the common shape has a recode kernel of its own: byte-aligned unsigned 32-byte scalars,
  // 16-bit windows, one task per window (grid.y = columns)
  bool rows32_c16 = num_cols <= 65535;
  for (const column_desc& c : plan.columns) {
    rows32_c16 = rows32_c16 &&
                 (c.n == 0 || (c.bit_offset == 0 && c.bit_width == 256 && c.row_stride == 32 &&
                               c.is_signed == 0 && c.merged_stride == 0 && c.window_bits == 16 &&
                               c.num_windows == 17 &&
                               (reinterpret_cast<uintptr_t>(c.data) & 15) == 0));
  }
  // A bucket group of more records than pass 2 stages in LDS is streamed by ONE workgroup (up to 16 x
  // the capacity) or cut into chunks for the workers of the chunked path.  Streaming wins where
  // there are many such groups (2^20 rows of 252-bit scalars: 16 of them per column, sort 0.128 against
  // 0.157 ms chunked, profiles/round4_ab_sort_stream_limit.log); in a short launch there is ONE --
  // the top window's 2^16 records in a handful of buckets -- and eleven serial rounds of a single
  // workgroup are most of the sort (2^16 rows: 0.090 ms): chunks over the workers instead.
  // (The chunked path's workers spin at a barrier inside the launch, so all of them must become
  // resident: `workers` below is bounded by the stream's compute units.  Work of OTHER streams that
  // holds wave slots delays them until its workgroups retire -- it cannot hang them, nothing it runs
  // waits for this launch: tests/test_round6.py commits 2^16-row columns beside 1.5 s of a kernel
  // that fills every SIMD.)
  const u32 stream_limit =
      plan.total_entries <= (u64{1} << 22) ? kLocalSortCapacity : kStreamedSortRecords;
  // every task is ONE bucket group of at most an LDS stage's worth of rows (hundreds of short columns):
  // k_task_sort does the whole sort of a task in one workgroup, one launch instead of three
  const bool small_tasks = plan.max_task_groups == 1 && plan.max_task_rows <= kLocalSortCapacity;
  // the kernels that write / read the stored digits, for the launch's digit type
  auto recode_and_partition = [&](auto digit_tag) {
    using D = decltype(digit_tag);
    D* digits = static_cast<D*>(b.digits);
    ctx.timer.timed(timing, 1, fs, [&] {
      if (d_ranges != nullptr) {
        hipLaunchKernelGGL((k_recode_packed<D>),
                           dim3(ceil_div_u32(plan.max_recode_rows, kPackedTileRows)),
                           dim3(kPackedRecodeThreads), kPackedTileBytes, fs, digits, b.cols,
                           b.tasks, d_ranges, static_cast<u32>(ranges.size()),
                           plan.columns[0].row_stride, plan.max_recode_rows, plan.max_rows,
                           b.group_cursor, zero_words);
        return;
      }
      const u32 chunks = ceil_div_u32(plan.max_recode_rows, 256);
      if constexpr (sizeof(D) == 2) {
        if (rows32_c16) {
          hipLaunchKernelGGL(k_recode_rows32_c16, dim3(chunks, num_cols), dim3(256), 0, fs, digits,
                             b.cols, b.tasks, b.group_cursor, zero_words);
          return;
        }
      }
      const u64 items = 8 * static_cast<u64>((chunks + 7) / 8) * num_cols;
      const u32 recode_blocks = static_cast<u32>(items < (u64{1} << 30) ? items : (u64{1} << 30));
      hipLaunchKernelGGL((k_recode<D>), dim3(recode_blocks), dim3(256), 0, fs, digits, b.cols,
                         b.tasks, num_cols, chunks, b.group_cursor, zero_words);
    });
    ctx.timer.timed(timing, 2, fs, [&] {
      if (small_tasks) {
        hipLaunchKernelGGL((k_task_sort<D>), dim3(num_tasks), dim3(kGroupSortThreads), 0, fs, b.sorted,
                           b.segment_bucket, b.bucket_end, digits, b.tasks);
        return;
      }
      u32* bucket_fill = b.bucket_count + plan.total_buckets + 1;
      // pass 1a (+ 1b in the last workgroup of every task)
      hipLaunchKernelGGL((k_group_hist<D>), dim3(plan.max_task_slices, num_tasks), dim3(kSortThreads),
                         part_lds, fs, b.group_cursor, b.big_tasks, digits, b.tasks, b.arrivals,
                         b.group_start, b.group_chunk, b.bucket_count, bucket_fill, stream_limit);
      // pass 1c; all tasks of a launch share one variant: staged unless some column needs the direct form
      if (plan.max_task_groups <= kMaxStagedGroups && plan.max_slice_rows <= kStagedSliceRows) {
        const size_t staged_lds = sizeof(u32) * (3 * plan.max_task_groups + 1 + kStagedSliceRows);
        hipLaunchKernelGGL((k_group_scatter<true, D>), dim3(plan.max_task_slices, num_tasks),
                           dim3(kSortThreads), staged_lds, fs, b.records, b.group_cursor, digits,
                           b.tasks);
      } else {
        hipLaunchKernelGGL((k_group_scatter<false, D>), dim3(plan.max_task_slices, num_tasks),
                           dim3(kSortThreads), part_lds, fs, b.records, b.group_cursor, digits,
                           b.tasks);
      }
    });
  };
  if (plan.wide_digits) {
    recode_and_partition(i32{});
  } else {
    recode_and_partition(i16{});
  }
  const u32 sort_launches = small_tasks ? 1 : 3;
  if (!small_tasks) {
    ctx.timer.timed(timing, 2, fs, [&] {
      u32* bucket_fill = b.bucket_count + plan.total_buckets + 1;
      // pass 2, the oversized groups (skewed digits) in one more row of the same grid: their workers
      // meet at a barrier, so there may only be as many as the stream's compute units hold at once
      const u32 cus = ctx.stream_cus(fs);
      const u32 workers = 2 * cus < kBigSortBlocks ? 2 * cus : kBigSortBlocks;
      hipLaunchKernelGGL(k_group_sort_all, dim3(plan.max_task_groups, num_tasks + 1),
                         dim3(kGroupSortThreads), 0, fs, b.sorted, b.segment_bucket, b.bucket_end,
                         b.records, b.group_start, b.group_chunk, b.tasks, num_tasks, b.bucket_count,
                         bucket_fill, b.big_tasks, b.big_barrier, workers);
    });
  }

  // ---- accumulate: rewrites the bucket sums / head partials the reduce two batches ago read
  if (!(mode.piped && k >= 2 && ctx.reduce_is_joined(as, k - 2))) {
    wait_for(earlier(ctx.reduce_done, 2), as);
  }
  if (ctx.table_pending) {
    // the call's window table, built on the side stream beside this front
    ctx.table_ready.wait(as);
    ctx.table_pending = false;
  }
  ctx.timer.timed(timing, 3, as, [&] {
    hipLaunchKernelGGL((k_accumulate<C>), dim3(seg_blocks, num_tasks), dim3(kAccumulateThreads),
                       ctx.acc_lds_pad, as, b.bucket_sums, b.heads, b.bucket_end, b.segment_bucket, b.sorted,
                       b.addends, b.tasks);
  });
  if (mode.piped) ctx.acc_done[k & 3].record(as);

  // ---- reduce: rewrites the partials / entry counts the horner two batches ago read
  if (mode.piped) ctx.acc_done[k & 3].wait(rs);
  wait_for(earlier(ctx.horner_done, 2), rs);
  ctx.timer.timed(timing, 4, rs, [&] {
    // (the compact form has 256-lane blocks only: launches of many small tasks keep the inlined one)
    const bool compact = plan.reduce_threads == kReduceThreads &&
                         (ctx.compact_reduce == 1 ||
                          (ctx.compact_reduce == 2 && ctx.slow_instruction_fetch &&
                           num_cols < ctx.tuning.throughput_columns));
    if (compact) {
      hipLaunchKernelGGL((k_reduce_compact<C>), dim3(b.partial_stride, num_tasks),
                         dim3(kReduceThreads), 0, rs, b.partials, b.partial_stride, b.task_total,
                         b.bucket_sums, b.heads, b.bucket_end, b.tasks, plan.reduce_segment_log2);
    } else {
      auto launch = [&](auto scan) {
        if (plan.reduce_threads == 64) {
          hipLaunchKernelGGL((k_reduce<C, decltype(scan)::value, 64>), dim3(b.partial_stride, num_tasks),
                             dim3(64), 0, rs, b.partials, b.partial_stride, b.task_total,
                             b.bucket_sums, b.heads, b.bucket_end, b.tasks, plan.reduce_segment_log2);
          return;
        }
        hipLaunchKernelGGL((k_reduce<C, decltype(scan)::value>), dim3(b.partial_stride, num_tasks),
                           dim3(kReduceThreads), 0, rs, b.partials, b.partial_stride, b.task_total,
                           b.bucket_sums, b.heads, b.bucket_end, b.tasks, plan.reduce_segment_log2);
      };
      if constexpr (!C::has_wave_add_multiple) {
        launch(std::false_type{});
      } else if constexpr (!C::reduce_scan_few_columns_only) {
        launch(std::true_type{});
      } else if (num_cols < ctx.tuning.throughput_columns) {
        launch(std::true_type{});
      } else {
        launch(std::false_type{});
      }
    }
  });
  if (mode.piped) ctx.reduce_done[k & 3].record(rs);

  // ---- horner: whole columns in one launch (the range covers every window, first and last)
  if (mode.piped) {
    ctx.reduce_done[k & 3].wait(hs);
    if (ctx.merged_waits) ctx.pre_horner[k & 3].record(hs);
  }
  ctx.timer.timed(timing, 5, hs, [&] {
    // (hundreds of columns: one-wavefront blocks, kernels.h)
    if (num_cols >= 64 && plan.max_windows <= 64) {
      hipLaunchKernelGGL((k_horner<C, 64>), dim3(num_cols), dim3(64), 0, hs, d_out, out_stride,
                         projective_out ? 1 : 0, b.horner_state, b.partials, b.partial_stride, b.cols,
                         b.tasks, b.task_total, 0u, 0xffffffffu, 1, 1, plan.reduce_block_log2());
      return;
    }
    hipLaunchKernelGGL((k_horner<C>), dim3(num_cols), dim3(kCombineThreads), 0, hs, d_out,
                       out_stride, projective_out ? 1 : 0, b.horner_state, b.partials,
                       b.partial_stride, b.cols, b.tasks, b.task_total, 0u, 0xffffffffu, 1, 1,
                       plan.reduce_block_log2());
  });
  if (mode.piped) {
    ctx.horner_done[k & 3].record(hs);
    ctx.pipe_layout = layout;
    ctx.seq = k + 1;
    ctx.join_two_back(stream, k);
  }
  if (timing) ctx.timer.calls += 1;
  g_kernel_launches += 5 + sort_launches; // (prepare), recode, the sort, accumulate, reduce, horner
  BZ_HIP_CHECK(hipGetLastError());
}

} // namespace bz
