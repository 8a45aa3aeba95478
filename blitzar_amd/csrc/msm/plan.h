// Host-side planning of a variable-base multi-scalar multiplication.
//
// The unit of device work is a *task* = (column, window):
//   * a column is one `sxt_sequence_descriptor` / `mtxb::exponent_sequence`
//     (cbindings/blitzar_api.h:115-131, sxt/multiexp/base/exponent_sequence.h:25-42),
//   * a window is one signed radix-2^c digit position of the column's scalars.
// Every task owns 2^(c-1) buckets (bucket id = |digit| - 1).  The column result is
//   sum_w 2^(c w) * sum_b (b + 1) * bucket[w][b].
// The digits of a task are sorted by bucket in two passes (kernels.h): the buckets are cut into
// *groups* of 2^s consecutive buckets, the rows into *slices* (one workgroup each); pass 1
// partitions every slice's digits by group, pass 2 sorts one group per workgroup inside LDS.  The
// sorted entry list is then cut into *segments* (32..128 entries, one lane each) for the
// bucket accumulation, so the parallelism of every stage is proportional to the number of rows
// and independent of how the digits are distributed over the buckets.
//
// This replaces, as one mechanism, the reference's three dispatch tiers
// (sxt/multiexp/curve/multiexponentiation.h:147-200: bucket_method2 / bucket_method / generic
// bit-plane multiproduct), which all use unsigned 8-bit windows or single bit planes.
#pragma once

#include <cstdint>
#include <vector>

#include "blitzar_amd/csrc/base/macros.h"

namespace bz {

// sorted entries per accumulation lane = 2^s (msm_plan::segment_log2): every segment that begins
// inside a bucket leaves a head partial for k_reduce to add (one projective addition per segment),
// so a launch with entries to spare (hundreds of columns) takes longer segments; a single column
// keeps 32 so that its lanes fill the machine
constexpr u32 kSegmentLog2 = 5;
constexpr u32 kSegmentLog2Max = 7;
// lanes k_accumulate should still have (measured on MI355X: one 2^22-row bls12-381 column, 71 M
// entries, is 3 % faster with 64 or 128 entries per lane than with 32; a 2^20-row curve25519 column,
// 18 M entries, is not)
constexpr u64 kSegmentFillLanes = u64{1} << 19;
constexpr u32 kSegmentEntries = 1u << kSegmentLog2;
constexpr u32 kSegmentLog2Min = 3;
// `task_rows` (of a typical entry's task): a task's lanes are rows / 2^s of ONE workgroup row of k_accumulate's grid, so short
// tasks must keep s small enough for whole wavefronts: 1024 columns x 4096 rows (the reference's
// bucket_method2 regime) chose 128 entries per lane from its 1.2e8 entries and ran every task on HALF
// a wavefront -- k_accumulate 10.8 ms against 5.7 ms at 32 entries per lane
// (profiles/round5_ab_wide_tables_and_short_columns.log).  At least 128 lanes per task where the rows allow it.
// A SHORT launch (a lone column of 2^12 .. 2^17 rows) cannot fill the machine at 32 entries per lane:
// 2^14 rows x 26 windows are 13 K lanes -- a fifth of a wavefront per SIMD, each walking 32 dependent
// additions -- so k_accumulate is a latency chain (0.10 ms at 2^14 rows, the price of 3 M additions,
// for 0.4 M).  While the launch has less than one wavefront per SIMD (kSegmentLatencyLanes), shorter
// segments, down to 8 entries: the chain shortens in proportion and the extra head partials are
// additions k_reduce has idle lanes for.
constexpr u64 kSegmentLatencyLanes = u64{1} << 16;
inline u32 choose_segment_log2(u64 total_entries, u64 task_rows = ~u64{0}) {
  u32 s = kSegmentLog2;
  while (s < kSegmentLog2Max && (total_entries >> (s + 1)) >= kSegmentFillLanes) ++s;
  while (s > kSegmentLog2Min && (task_rows >> s) < 128) --s;
  while (s > kSegmentLog2Min && (total_entries >> s) < kSegmentLatencyLanes) --s;
  return s;
}
constexpr u32 kStagedSliceRows = 1u << 14; // rows per partition workgroup: staged in LDS / direct
constexpr u32 kDirectSliceRows = 1u << 16;
constexpr u32 kMaxStagedGroups = 1024;     // the staged partition keeps 3 counters per group in LDS
constexpr u32 kGroupTargetEntries = 4096; // entries a bucket group should hold (uniform digits)
constexpr u32 kLocalSortCapacity = 6144;  // entries the per-group sort stages in LDS
constexpr u32 kMaxGroupBits = 10;         // 2^s counters of the per-group sort live in LDS
constexpr u32 kMaxGroupsLog2 = 10;        // pass 1 writes one stream per group: keep them few

// buckets a reduce block covers (threads per block x buckets per thread), shared with kernels.h
#ifndef BZ_REDUCE_THREADS
#define BZ_REDUCE_THREADS 256
#endif
constexpr u32 kReduceThreads = BZ_REDUCE_THREADS;
#ifndef BZ_REDUCE_SEGMENT_LOG2
#define BZ_REDUCE_SEGMENT_LOG2 3
#endif
// buckets per k_reduce lane = 2^s, chosen per launch (msm_plan::reduce_segment_log2): a lane pays
// 2 additions per bucket plus ~30 per lane (its weight and the tree), so few buckets per lane keep
// the dependent chain of a latency-bound launch short (one column: s = 3), many buckets per lane
// keep the total work of a throughput-bound launch small (hundreds of columns: s = 6)
constexpr u32 kReduceSegmentLog2 = BZ_REDUCE_SEGMENT_LOG2;
#ifndef BZ_REDUCE_SEGMENT_LOG2_MAX
#define BZ_REDUCE_SEGMENT_LOG2_MAX 6
#endif
constexpr u32 kReduceSegmentLog2Max = BZ_REDUCE_SEGMENT_LOG2_MAX;
// lanes that fill the machine twice over (1024 SIMDs x 64 lanes x 2 waves)
constexpr u64 kReduceFillLanes = 131072;
// `in_sequence`: a call in throughput mode (msm_context).  Its k_reduce runs beside the next call
// and has a whole step to finish, so what counts is not the length of its dependent chain but the
// work it takes out of a machine that runs at its package power limit: twice the
// buckets per lane, ~20 % fewer additions (per call in sequence, profiles/round3_ab_reduce_in_sequence.log:
// config 2 0.987 -> 0.978 ms, on resident generators 0.892 -> 0.871, one bls12-381 column of 2^22
// rows 12.6 -> 11.8; a lone k_reduce of this geometry takes 0.38 instead of 0.20 ms, which is why
// lone calls keep the shorter chain).
// `lone_latency`: a launch of few columns outside the throughput mode whose buckets fit the machine
// at 2 per lane (<= 2^18 buckets: 512 workgroups): k_reduce is then the longer of a lone call's two
// tail chains and 2 buckets per lane halve it -- 2^16 curve25519 rows: k_reduce 0.166 -> 0.090 ms, the
// lone call 0.55 -> 0.47; in a sequence the same geometry costs 3 % (more workgroups beside the next
// call), and at 2^20 rows twice the workgroups no longer fit (0.19 -> 0.31 ms): both keep theirs.
// `throughput`: a launch of many columns.  Its small tasks (256 buckets at 9-bit windows: thousands of
// short columns) keep at least ONE wavefront's worth of lanes with several buckets each -- k_reduce
// runs its scan and tree over the lanes in use only, so 64 lanes x 4 buckets cost a third of the
// additions of 256 lanes x 1 bucket; a lone narrow column keeps the shortest chain (below).
inline u32 choose_reduce_segment_log2(u64 total_buckets, u32 max_task_buckets,
                                      bool in_sequence = false, bool lone_latency = false,
                                      bool throughput = false) {
  u32 s = kReduceSegmentLog2 + (in_sequence ? 1 : 0);
  if (lone_latency && total_buckets <= (u64{1} << 18)) s = 1;
  const u64 lanes_min = throughput ? 64 : kReduceThreads;
  // narrow columns (bytes, booleans: 128..1024 buckets per task) have one block per task anyway:
  // fewer buckets per lane, down to one, shorten its chain (a 1-byte column of 2^20 rows: k_reduce
  // 0.25 -> 0.1 ms of a 0.6 ms call)
  while (s > 0 && (lanes_min << s) > max_task_buckets) --s;
  // ... as long as the largest task still fills a block's 256 lanes (idle waves of a block hold
  // registers the other blocks of the CU could use)
  while (s < kReduceSegmentLog2Max && (total_buckets >> (s + 1)) >= kReduceFillLanes &&
         (static_cast<u64>(kReduceThreads) << (s + 1)) <= max_task_buckets) {
    ++s;
  }
  return s;
}

// device-visible task descriptor
struct task_desc {
  u32 column;
  u32 window;
  u64 rows;         // rows of the column
  u32 num_buckets;  // 2^(c-1)
  u32 num_slices;   // ceil(rows / slice_rows)
  u32 slice_rows;   // rows per partition workgroup (power of two, multiple of 8)
  u32 group_bits;   // s: a group is 2^s consecutive buckets
  u32 num_groups;   // num_buckets >> s
  u32 segment_log2; // sorted entries per k_accumulate lane = 2^this (one value per launch)
  u64 bucket_base;  // first bucket of this task in the flat bucket arrays
  u64 entry_base;   // first entry of this task in the flat digit / record / sorted-index arrays
  u64 group_base;   // first entry of this task's group tables (num_groups + 1 entries)
  u64 segment_base; // first segment of this task in the flat per-segment arrays
};

// device-visible column descriptor
struct column_desc {
  const u8* data;  // device pointer to row 0
  u64 n;
  u64 row_stride;  // bytes between consecutive rows
  u32 bit_offset;  // first bit of the scalar inside its row (LSB-first within bytes)
  u32 bit_width;   // 1..256; a byte-aligned column has bit_offset 0, width 8 * nbytes
  u32 is_signed;   // two's complement over bit_width bits
  u32 window_bits; // c
  u32 num_windows; // W: digits per scalar
  u32 first_task;  // task index of window 0; task = first + window
  // window tables (resident generator sets with 2^(c w) g_i precomputed, see window_table below):
  // all W windows of the column share ONE task whose "rows" are (window, row) pairs,
  // virtual row = window * merged_stride + row; 0 = one task per window
  u32 num_tasks;     // tasks of the column: W, or 1 when merged
  u32 merged_stride; // rows of one table slice, or 0
};

// Precomputed multiples of a resident generator set: slice w holds 2^(bits w) g_i, `stride` rows
// per slice (>= the number of generators, multiple of 8), slices back to back -- so a sorted
// entry's virtual row IS its row in the table and k_accumulate needs no change.  With every window
// accumulating into the same 2^(bits-1) buckets, a column costs one bucket reduction instead of W
// and its Horner chain over windows disappears.
// `bits` may exceed 16 (kMaxTableWindowBits): a merged column's digits are then stored as 32-bit
// words (msm_plan::wide_digits) -- one bucket set per column makes 2^17 buckets affordable where W
// separate sets were not: 15 windows instead of 17 for 256-bit scalars at bits = 18.  The 32-bit
// partition record (sign | bucket mod 2^s | row) bounds what is practical: a merged task of W x
// stride virtual rows needs s >= bits - 11 record bits for at most 1024 bucket groups, i.e.
// log2(W stride) + bits <= 42 (2^20 generators: bits = 18; 2^18: bits = 20); beyond that pass 1 falls
// back to its direct form (one 4-byte store per record) -- correct, slower.
constexpr u32 kMaxTableWindowBits = 20;
struct window_table {
  u64 stride = 0;
  u32 windows = 0; // slices available
  u32 bits = 16;   // window width the slices were built for
};

struct msm_plan {
  std::vector<column_desc> columns;
  std::vector<task_desc> tasks;
  u64 total_buckets = 0;
  u64 total_entries = 0;
  u64 total_groups = 0;    // entries of all group tables (group_start, group_cursor)
  u64 total_segments = 0;
  u64 max_rows = 0;        // longest column
  u64 max_task_rows = 0;   // most (virtual) rows of a task: k_accumulate's grid
  u64 max_recode_rows = 0; // rows k_recode visits per column (merged columns: the table stride)
  u32 max_task_buckets = 0;
  u32 max_task_slices = 0;
  u32 max_task_groups = 0;
  u32 max_slice_rows = 0;
  u32 max_windows = 0;
  bool wide_digits = false; // some column has c > 16: the digits of the launch are 32-bit words
  u32 segment_log2 = kSegmentLog2;               // sorted entries per k_accumulate lane
  u32 reduce_segment_log2 = kReduceSegmentLog2; // buckets per k_reduce lane
  u32 reduce_threads = kReduceThreads;          // lanes of a k_reduce block: 256, or 64 (small tasks)
  u32 reduce_block_buckets() const { return reduce_threads << reduce_segment_log2; }
  // log2 of the buckets a k_reduce block covers (what k_horner counts a task's partials by)
  static_assert(kReduceThreads == 256, "reduce_block_log2 counts 256-lane blocks as 2^8 lanes");
  u32 reduce_block_log2() const { return (reduce_threads == 64 ? 6 : 8) + reduce_segment_log2; }
};

#ifndef BZ_THROUGHPUT_BUCKET_COST
#define BZ_THROUGHPUT_BUCKET_COST 3.5
#endif
struct msm_tuning {
  u32 max_window_bits = 16; // separate windows store their digits as int16 (merged tasks of wide tables: int32)
  // batching of many-column jobs: tasks per launch (grid.y) and device workspace per batch
  size_t max_tasks_per_batch = 65000; // (tasks + 1 is a launch-grid dimension: at most 65535)
  size_t max_workspace_bytes = size_t{64} << 30;
  // two-pass sort geometry (choose_partition below): entries per bucket group
  u32 partition_group_entries = kGroupTargetEntries;
  // window-width cost model: from this many columns on, buckets cost `throughput_bucket_cost`
  size_t throughput_columns = 4;
  double throughput_bucket_cost = BZ_THROUGHPUT_BUCKET_COST;
  u32 force_window_bits = 0;         // tests (bzamd_set_window_bits): this width wherever a column allows it
  bool in_sequence = false; // set per call by msm_enqueue: the call runs in throughput mode
  // wavefronts of k_accumulate the device holds at once (waves per SIMD of the curve's loop x SIMDs;
  // set per call by msm_enqueue, 0 = unknown): make_msm_plan shortens the segments of a launch whose
  // wavefronts would fill the device a few times only and leave its last round mostly idle
  u32 accumulate_wave_slots = 0;
  u32 force_reduce_segment_log2 = 0; // development override (BLITZAR_AMD_REDUCE_SEGMENT_LOG2), 0 = choose
  u32 force_segment_log2 = 0;        // development override (BLITZAR_AMD_SEGMENT_LOG2), 0 = choose
  // throughput mode (engine.h, msm_context): calls with this many columns or more ignore
  // bzamd_pipeline_next.  Measured on MI355X, curve25519, k columns x 2^20
  // rows, ms per call lone / in sequence (tools/multi_column_bench.py): 2: 2.41 / 2.03, 4: 4.19 / 3.73,
  // 8: 7.60 / 7.13, 16: 14.46 / 13.93, 32: 27.73 / 27.22; with 256 columns (bn254) the fork and the
  // second set of tail buffers cost more than the overlap buys (352.5 against 348.0).
  size_t defer_max_columns = 64;
  // window tables: gathers from a table beyond the 256 MiB Infinity Cache cost this much more per
  // addition (measured on MI355X: 51 against 43 ps with a 2 GiB curve25519 table); a table that
  // fits costs `table_penalty_cached`.  `force_window_tables` (tests) merges whenever possible.
  double table_penalty = 1.0;
  bool force_window_tables = false;
};

inline u32 ceil_div_u32(u64 a, u64 b) { return static_cast<u32>((a + b - 1) / b); }

// choose the window width for a column of n rows and B significant bits: minimise
//   W * (n + bucket_cost * 2^(c-1)),  W = ceil((B + 1) / c)
// where bucket_cost is the price of reducing one bucket in units of one bucket addition.  A
// single column is latency-bound in k_reduce (few waves, the dependent chain is what matters): its
// buckets are nearly free (2.5).  Many columns fill the machine with reduce work, so buckets are
// priced at their throughput cost: with 64 buckets per reduce lane (choose_reduce_segment_log2)
// that is two projective additions per bucket plus its share of the lane's ~30, measured on
// MI355X (bn254, 256 x 2^20, c = 15 against 16: 16.5 ms for 71 M more buckets = 0.23 ns per bucket
// against 0.07 ns per accumulated entry) as 3.5.  (Round 1, with 8 buckets per lane whatever the
// launch: 15 additions per bucket, which had pushed such jobs to c = 13-14.)
// `task_cost` (launches of many columns): what a task costs whatever its size, in bucket additions --
// its k_reduce block, its share of the sort and of k_accumulate's grid.  Fitted on 1024 columns of 256 /
// 1024 / 4096 rows (profiles/round5_ab_many_short_columns_final.log: a task's k_reduce is 28 ns + 0.165
// ns per bucket beside 0.048 ns per entry): without it the model took one bit too few below 4096 rows
// (1024 rows: c = 7, 3.73 ms; c = 8, 3.50; 256 rows: c = 6, 3.02 ms; c = 7, 2.58).
constexpr double kThroughputTaskCost = 800.0;
inline u32 choose_window_bits(u64 n, u32 bits, const msm_tuning& tune, double bucket_cost = 2.5,
                              double task_cost = 0.0) {
  u32 best_c = 1;
  double best_cost = 1e300;
  const u32 cmax = tune.max_window_bits < bits + 1 ? tune.max_window_bits : bits + 1;
  if (tune.force_window_bits != 0) {
    const u32 c = tune.force_window_bits < cmax ? tune.force_window_bits : cmax;
    return c < 2 ? 2 : c;
  }
  for (u32 c = 2; c <= cmax; ++c) {
    const u32 w = ceil_div_u32(bits + 1, c);
    const double cost =
        static_cast<double>(w) *
        (static_cast<double>(n) + bucket_cost * static_cast<double>(1u << (c - 1)) + task_cost);
    if (cost < best_cost) {
      best_cost = cost;
      best_c = c;
    }
  }
  return best_c;
}

struct host_column {
  const u8* data; // pointer to row 0 (device pointer for the GPU engine)
  u64 n;
  u64 row_stride;
  u32 bit_offset;
  u32 bit_width;
  bool is_signed;
};

// one `sxt_sequence_descriptor`: n little-endian integers of nbytes bytes, back to back
inline host_column byte_column(const u8* data, u64 n, u32 nbytes, bool is_signed) {
  return host_column{data, n, nbytes, 0, 8 * nbytes, is_signed};
}

// Partition geometry of a task with n rows and 2^(c-1) buckets.
//   s (group_bits): groups of ~kGroupTargetEntries entries so that pass 2 sorts a group inside
//     LDS; bounded by the 32-bit partition record, which packs sign | bucket-in-group (s bits) |
//     row (31 - s bits), by the LDS counters of pass 2, and from below so that pass 1 keeps at
//     most 2^kMaxGroupsLog2 output streams whenever the record has room for it.  For very long
//     columns s shrinks to 0 and the scheme degenerates into a plain one-pass bucket scatter.
//   slice_rows: what pass 1 can stage in LDS (so that it writes whole runs, not single records);
//     with more than kMaxStagedGroups groups it writes records directly, from larger slices.
struct partition_geometry {
  u32 group_bits, num_groups, slice_rows, num_slices;
};
inline partition_geometry choose_partition(u64 n, u32 window_bits,
                                            u32 target_entries = kGroupTargetEntries) {
  const u32 bucket_bits = window_bits - 1;
  u32 row_bits = 0;
  while (row_bits < 31 && (u64{1} << row_bits) < n) ++row_bits;
  // desired: largest s with n * 2^s / 2^bucket_bits <= kGroupTargetEntries
  u32 desired = 0;
  while (desired < bucket_bits &&
         ((n << (desired + 1)) >> bucket_bits) <= target_entries)
    ++desired;
  const u32 few_streams = bucket_bits > kMaxGroupsLog2 ? bucket_bits - kMaxGroupsLog2 : 0;
  u32 s = desired > few_streams ? desired : few_streams;
  if (s > bucket_bits) s = bucket_bits;
  if (s > kMaxGroupBits) s = kMaxGroupBits;
  if (s > 31 - row_bits) s = 31 - row_bits;
  partition_geometry g;
  g.group_bits = s;
  g.num_groups = 1u << (bucket_bits - s);
  const u64 slice = g.num_groups <= kMaxStagedGroups ? kStagedSliceRows : kDirectSliceRows;
  g.slice_rows = static_cast<u32>(slice);
  g.num_slices = static_cast<u32>((n + slice - 1) / slice);
  return g;
}

// a column uses the window tables when it fills at least half a slice (the rest of the slice is
// recoded as zero digits), has more than one window and is unsigned (signed columns store -D and
// stay at c <= 15)
inline bool use_window_table(const host_column& hc, const window_table* tables,
                             const msm_tuning& tune, double bucket_cost, double task_cost = 0.0) {
  if (tables == nullptr || tables->windows == 0 || hc.is_signed || hc.n == 0) return false;
  if (hc.n > tables->stride || 2 * hc.n < tables->stride) return false;
  const u32 w = ceil_div_u32(hc.bit_width + 1, tables->bits);
  if (w <= 1 || w > tables->windows) return false;
  if (tune.force_window_tables) return true;
  // same cost model as choose_window_bits: one bucket set for all windows, dearer gathers
  const u32 c = choose_window_bits(hc.n, hc.bit_width, tune, bucket_cost, task_cost);
  const double separate =
      static_cast<double>(ceil_div_u32(hc.bit_width + 1, c)) *
      (static_cast<double>(hc.n) + bucket_cost * static_cast<double>(1u << (c - 1)) + task_cost);
  const double merged = static_cast<double>(w) * static_cast<double>(hc.n) * tune.table_penalty +
                        bucket_cost * static_cast<double>(1u << (tables->bits - 1)) + task_cost;
  return merged < separate;
}

// Per-call window tables (the reference's bucket_method2 regime: many columns over the SAME caller
// generators, sxt/multiexp/bucket_method2/multiexponentiation.h:48-121).  A launch of hundreds of short
// columns pays W separate windows per column -- 29 tasks of 256 buckets each at 4096 rows -- because
// every window owns a bucket set; with the 2^(c w) multiples of the call's generators built once IN
// the call (W c doublings per generator, shared by every column) a column is ONE task with one
// bucket set, and c can grow: 1024 x 4096 x 256-bit: 29 x (4096 + 3.5 x 256 + 800) = 168 K units per
// column on separate 9-bit windows against 22 x 4096 + 3.5 x 2048 + 800 = 98 K merged at c = 12.
// The build is priced in the same unit (one bucket addition at the machine's throughput): a doubling
// ~0.8, the normalisation of a slice row ~0.5, plus what the chain of W c dependent doublings costs
// in idle machine when nothing hides it (`kCallTableLatencyUnits`, ~0.1 ms).
struct call_table_choice {
  window_table shape; // shape.windows == 0: no table for this call
  double separate_cost = 0, merged_cost = 0, build_cost = 0;
};
constexpr u32 kCallTableMinBits = 6;
constexpr u32 kCallTableMaxBits = 16;
constexpr size_t kCallTableMinColumns = 8;
constexpr double kCallTableLatencyUnits = 2.5e6;
constexpr double kCallTableMaxBytes = 4.0 * (u64{1} << 30);
// `entry_cost`: an accumulated entry against the table's addend form relative to the per-call form
// (curve25519: Z = 1 addends, 7 field products instead of 8)
inline call_table_choice choose_call_table(const std::vector<host_column>& cols,
                                           const msm_tuning& tune, size_t addend_size,
                                           double entry_cost = 1.0, u32 force_bits = 0) {
  call_table_choice out;
  u64 max_n = 0;
  size_t nonempty = 0;
  for (const auto& c : cols) {
    if (c.n == 0) continue;
    ++nonempty;
    if (c.n > max_n) max_n = c.n;
  }
  if (nonempty < (force_bits != 0 ? 1 : kCallTableMinColumns) || max_n == 0) return out;
  if (nonempty < tune.throughput_columns && force_bits == 0) return out;
  const u64 stride = (max_n + 7) & ~u64{7};
  // distinct column shapes (a call's columns mostly share one)
  struct shape_count {
    u64 n;
    u32 bits;
    bool is_signed;
    size_t count;
  };
  std::vector<shape_count> shapes;
  for (const auto& c : cols) {
    if (c.n == 0) continue;
    bool found = false;
    for (auto& s : shapes) {
      if (s.n == c.n && s.bits == c.bit_width && s.is_signed == c.is_signed) {
        ++s.count;
        found = true;
        break;
      }
    }
    if (!found) {
      if (shapes.size() >= 64) return out; // a ragged call: not the regime
      shapes.push_back({c.n, c.bit_width, c.is_signed, 1});
    }
  }
  const double bucket_cost = tune.throughput_bucket_cost, task_cost = kThroughputTaskCost;
  auto separate_of = [&](const shape_count& s) {
    msm_tuning t = tune;
    if (s.is_signed && t.max_window_bits > 15) t.max_window_bits = 15;
    const u32 c = choose_window_bits(s.n, s.bits, t, bucket_cost, task_cost);
    return static_cast<double>(ceil_div_u32(s.bits + 1, c)) *
           (static_cast<double>(s.n) + bucket_cost * static_cast<double>(1u << (c - 1)) + task_cost);
  };
  double separate_total = 0;
  for (const auto& s : shapes) separate_total += separate_of(s) * static_cast<double>(s.count);
  out.separate_cost = separate_total;
  double best = separate_total;
  const u32 lo = force_bits != 0 ? force_bits : kCallTableMinBits;
  const u32 hi = force_bits != 0 ? force_bits : kCallTableMaxBits;
  u32 widest = 0;
  for (const auto& s : shapes) widest = s.bits > widest ? s.bits : widest;
  for (u32 c = lo; c <= hi; ++c) {
    u32 windows = 0;
    double total = 0;
    // gathers from a table beyond the Infinity Cache are dearer (msm_tuning::table_penalty; the
    // same figures msm_enqueue prices a resident table with)
    const double table_bytes = static_cast<double>(addend_size) * static_cast<double>(stride) *
                               ceil_div_u32(widest + 1, c);
    const double penalty = table_bytes > 200.0 * (1 << 20) ? 1.15 : 1.03;
    for (const auto& s : shapes) {
      const u32 w = ceil_div_u32(s.bits + 1, c);
      const bool can = !s.is_signed && w > 1 && 2 * s.n >= stride;
      const double merged = static_cast<double>(w) * static_cast<double>(s.n) * entry_cost * penalty +
                            bucket_cost * static_cast<double>(1u << (c - 1)) + task_cost;
      const double sep = separate_of(s) * entry_cost; // (plain columns gather slice 0: the same form)
      if (can && (merged < sep || force_bits != 0)) {
        total += merged * static_cast<double>(s.count);
        if (w > windows) windows = w;
      } else {
        total += sep * static_cast<double>(s.count);
      }
    }
    if (windows < 2) continue;
    if (static_cast<double>(addend_size) * static_cast<double>(stride) * windows > kCallTableMaxBytes) {
      continue;
    }
    const double build = static_cast<double>(max_n) * (0.8 * c * (windows - 1) + 0.5 * windows) +
                         kCallTableLatencyUnits;
    if (total + build < best || (force_bits != 0 && out.shape.windows == 0)) {
      best = total + build;
      out.shape.stride = stride;
      out.shape.windows = windows;
      out.shape.bits = c;
      out.merged_cost = total;
      out.build_cost = build;
    }
  }
  return out;
}

inline msm_plan make_msm_plan(const std::vector<host_column>& cols, const msm_tuning& tune = {},
                              const window_table* tables = nullptr) {
  msm_plan plan;
  plan.columns.reserve(cols.size());
  size_t nonempty = 0;
  for (const auto& c : cols) nonempty += c.n != 0 ? 1 : 0;
  const double bucket_cost = nonempty >= tune.throughput_columns ? tune.throughput_bucket_cost : 2.5;
  const double task_cost = nonempty >= tune.throughput_columns ? kThroughputTaskCost : 0.0;
  // window width, window count and (virtual) rows per task of a column
  struct column_shape {
    bool merged;
    u32 c, w;
    u64 task_rows;
  };
  auto shape_of = [&](const host_column& hc) {
    column_shape sh{};
    sh.merged = use_window_table(hc, tables, tune, bucket_cost, task_cost);
    sh.c = sh.merged ? tables->bits
                     : choose_window_bits(hc.n, hc.bit_width, tune, bucket_cost, task_cost);
    sh.w = ceil_div_u32(hc.bit_width + 1, sh.c);
    // rows of a task: the column's, or every (window, row) pair up to the last window's rows
    sh.task_rows = sh.merged ? static_cast<u64>(sh.w - 1) * tables->stride + hc.n : hc.n;
    return sh;
  };
  // entries per accumulation lane: one value for the launch, from its total number of entries
  // ... and from the rows of the task a typical ENTRY lives in (the entry-weighted mean of the tasks'
  // rows: one long column among a thousand short ones keeps its own geometry)
  u64 launch_entries = 0;
  double rows_squared = 0;
  struct task_run {
    u64 rows, tasks;
  };
  std::vector<task_run> runs; // (rows, count) of the launch's tasks, equal neighbours folded
  for (const auto& hc : cols) {
    if (hc.n == 0) continue;
    const column_shape sh = shape_of(hc);
    const u64 tasks = sh.merged ? 1 : sh.w;
    launch_entries += sh.task_rows * tasks;
    rows_squared += static_cast<double>(sh.task_rows) * static_cast<double>(sh.task_rows) *
                    static_cast<double>(tasks);
    if (!runs.empty() && runs.back().rows == sh.task_rows) {
      runs.back().tasks += tasks;
    } else {
      runs.push_back({sh.task_rows, tasks});
    }
  }
  const u64 typical_task_rows =
      launch_entries != 0 ? static_cast<u64>(rows_squared / static_cast<double>(launch_entries)) : 0;
  plan.segment_log2 = tune.force_segment_log2 != 0
                          ? tune.force_segment_log2
                          : choose_segment_log2(launch_entries, typical_task_rows);
  // Rounds.  A wavefront of k_accumulate works through 64 segments of equal length, so the launch's
  // wavefronts all take about the same time and the device runs them in rounds of
  // `accumulate_wave_slots`: 1024 merged tasks of 20 x 4096 rows at 128 entries per lane are 10240
  // wavefronts = 3.33 rounds of 3072 and take the time of FOUR (measured: k_accumulate 3.43 ms at 84 M
  // entries and 3.39 ms at 100 M, which are 3.97 rounds).  While the last round is less than 90 % full
  // and there are fewer than 12 rounds, halve the segments (down to 32 entries: twice the head
  // partials for k_reduce, which is the cheaper side).
  if (tune.force_segment_log2 == 0 && tune.accumulate_wave_slots != 0) {
    auto rounds_at = [&](u32 s) {
      u64 waves = 0;
      for (const auto& r : runs) waves += r.tasks * ((r.rows + (u64{64} << s) - 1) / (u64{64} << s));
      return static_cast<double>(waves) / static_cast<double>(tune.accumulate_wave_slots);
    };
    while (plan.segment_log2 > kSegmentLog2) {
      const double rounds = rounds_at(plan.segment_log2);
      const double whole = static_cast<double>(static_cast<u64>(rounds));
      const double up = whole < rounds ? whole + 1 : whole;
      if (rounds >= 12 || rounds >= 0.9 * up) break;
      --plan.segment_log2;
    }
  }
  const u64 seg_entries = u64{1} << plan.segment_log2;
  for (size_t ci = 0; ci < cols.size(); ++ci) {
    const host_column& hc = cols[ci];
    column_desc cd{};
    cd.data = hc.data;
    cd.n = hc.n;
    cd.row_stride = hc.row_stride;
    cd.bit_offset = hc.bit_offset;
    cd.bit_width = hc.bit_width;
    cd.is_signed = hc.is_signed ? 1 : 0;
    cd.first_task = static_cast<u32>(plan.tasks.size());
    if (hc.n == 0) {
      cd.window_bits = 1;
      cd.num_windows = 0;
      plan.columns.push_back(cd);
      continue;
    }
    const column_shape sh = shape_of(hc);
    const bool merged = sh.merged;
    const u32 c = sh.c, w = sh.w;
    const u32 buckets = 1u << (c - 1);
    const u64 task_rows = sh.task_rows;
    // A merged task's top slice is not uniform: scalars below 2^252 in 256-bit fields leave a few bits
    // (or only the carry) to the top window, so ALL of that slice's rows land in the lowest bucket
    // group, on top of the group's share of the other slices.  With short slices (per-call tables of a
    // few thousand generators) size the groups so that share + slice still fits the LDS stage of pass 2
    // -- otherwise every task has one oversized group, streamed by a single workgroup (1024 x 4096
    // rows, c = 13: sort 1.06 ms against 0.55 at c = 11, whose top slice holds carries only).
    u32 group_entries = tune.partition_group_entries;
    if (merged && tables->stride < kLocalSortCapacity - 1024) {
      const u32 room = kLocalSortCapacity - static_cast<u32>(tables->stride);
      if (room < group_entries) group_entries = room;
    }
    const partition_geometry geo = choose_partition(task_rows, c, group_entries);
    const u32 slices = geo.num_slices;
    cd.window_bits = c;
    cd.num_windows = w;
    cd.num_tasks = merged ? 1 : w;
    cd.merged_stride = merged ? static_cast<u32>(tables->stride) : 0;
    for (u32 wi = 0; wi < cd.num_tasks; ++wi) {
      task_desc t{};
      t.column = static_cast<u32>(ci);
      t.window = wi;
      t.rows = task_rows;
      t.num_buckets = buckets;
      t.num_slices = slices;
      t.slice_rows = geo.slice_rows;
      t.group_bits = geo.group_bits;
      t.segment_log2 = plan.segment_log2;
      t.num_groups = geo.num_groups;
      t.bucket_base = plan.total_buckets;
      t.entry_base = plan.total_entries;
      t.group_base = plan.total_groups;
      t.segment_base = plan.total_segments;
      plan.total_buckets += buckets;
      // keep every task's entry range 16-byte aligned for both the i16 and the u32 views
      plan.total_entries += (task_rows + 7) & ~7ull;
      plan.total_groups += geo.num_groups + 1;
      plan.total_segments += (task_rows + seg_entries - 1) / seg_entries;
      plan.tasks.push_back(t);
    }
    if (task_rows > plan.max_task_rows) plan.max_task_rows = task_rows;
    const u64 recode_rows = merged ? tables->stride : hc.n;
    if (recode_rows > plan.max_recode_rows) plan.max_recode_rows = recode_rows;
    if (buckets > plan.max_task_buckets) plan.max_task_buckets = buckets;
    if (slices > plan.max_task_slices) plan.max_task_slices = slices;
    if (geo.num_groups > plan.max_task_groups) plan.max_task_groups = geo.num_groups;
    if (geo.slice_rows > plan.max_slice_rows) plan.max_slice_rows = geo.slice_rows;
    if (w > plan.max_windows) plan.max_windows = w;
    if (c > 16) plan.wide_digits = true;
    if (hc.n > plan.max_rows) plan.max_rows = hc.n;
    plan.columns.push_back(cd);
  }
  plan.reduce_segment_log2 = tune.force_reduce_segment_log2 != 0
                                 ? tune.force_reduce_segment_log2
                                 : choose_reduce_segment_log2(
                                       plan.total_buckets, plan.max_task_buckets,
                                       // (short columns are all tail even in a sequence: 2^16 rows
                                       // 0.285 -> 0.351 ms per call with the longer chain)
                                       tune.in_sequence && plan.total_entries >= (u64{1} << 23),
                                       !tune.in_sequence && nonempty < tune.throughput_columns,
                                       nonempty >= tune.throughput_columns);
  // many small tasks: one wavefront per block (kernels.h, k_reduce<.., 64>)
  if (tune.force_reduce_segment_log2 == 0 && nonempty >= tune.throughput_columns &&
      (u64{64} << plan.reduce_segment_log2) >= plan.max_task_buckets) {
    plan.reduce_threads = 64;
  }
  return plan;
}

// k_recode_packed (kernels.h): tile geometry and the column ranges one tile covers
constexpr u32 kPackedTileRows = 64;
constexpr u32 kPackedTileSpan = 1984; // bytes of a row one tile covers
constexpr u32 kPackedTilePitch = kPackedTileSpan + 48; // + alignment slack, 16-byte multiple
constexpr u32 kPackedRecodeThreads = 1024;
// + the read-ahead of the last field
constexpr u32 kPackedTileBytes = kPackedTileRows * kPackedTilePitch + 64;
struct recode_range {
  const u8* base; // lowest column base pointer of the range (row 0)
  u32 first_column, num_columns;
  u32 span;       // bytes of a row the range needs, from `base`
  u32 pad;
};


// Column ranges for k_recode_packed, or none when the batch is not a packed fixed-base call:
// every column must be a field of the same rows (equal strides, base pointers ascending and all
// inside the first row).
inline std::vector<recode_range> packed_recode_ranges(const msm_plan& plan) {
  std::vector<recode_range> ranges;
  const auto& cols = plan.columns;
  if (cols.size() < 2) return ranges;
  const u64 stride = cols[0].row_stride;
  // bytes of its row a column's recoder reads, from its base pointer (msm/recode.h)
  auto bytes_of = [](const column_desc& c) { return ((c.bit_offset & 7) + c.bit_width + 7) / 8; };
  for (size_t i = 0; i < cols.size(); ++i) {
    if (cols[i].row_stride != stride || cols[i].data == nullptr) return {};
    if (i > 0 && cols[i].data < cols[i - 1].data) return {};
    if (static_cast<u64>(cols[i].data - cols[0].data) + bytes_of(cols[i]) > stride) return {};
  }
  for (size_t i = 0; i < cols.size();) {
    recode_range r{cols[i].data, static_cast<u32>(i), 0, 0, 0};
    size_t j = i;
    u64 span = 0;
    while (j < cols.size()) {
      const u64 end = static_cast<u64>(cols[j].data - r.base) + bytes_of(cols[j]);
      if (end > kPackedTileSpan) break;
      if (end > span) span = end;
      ++j;
    }
    r.num_columns = static_cast<u32>(j - i);
    r.span = static_cast<u32>(span);
    ranges.push_back(r);
    i = j;
  }
  return ranges;
}

} // namespace bz
