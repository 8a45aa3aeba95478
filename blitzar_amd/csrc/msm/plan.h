// Host-side planning of a variable-base multi-scalar multiplication.
//
// The unit of device work is a *task* = (column, window):
//   * a column is one `sxt_sequence_descriptor` / `mtxb::exponent_sequence`
//     (cbindings/blitzar_api.h:115-131, sxt/multiexp/base/exponent_sequence.h:25-42),
//   * a window is one signed radix-2^c digit position of the column's scalars.
// Every task owns 2^(c-1) buckets (bucket id = |digit| - 1).  The column result is
//   sum_w 2^(c w) * sum_b (b + 1) * bucket[w][b].
// Inside a task the rows are cut into *slices* (kSliceRows rows, one workgroup each) for the
// counting sort, and the sorted entry list is cut into *segments* (kSegmentEntries entries, one
// lane each) for the bucket accumulation, so the parallelism of every stage is proportional to
// the number of rows and independent of how the digits are distributed over the buckets.
//
// This replaces, as one mechanism, the reference's three dispatch tiers
// (sxt/multiexp/curve/multiexponentiation.h:147-200: bucket_method2 / bucket_method / generic
// bit-plane multiproduct), which all use unsigned 8-bit windows or single bit planes.
#pragma once

#include <cstdint>
#include <vector>

#include "blitzar_amd/csrc/base/macros.h"

namespace bz {

constexpr u32 kSliceRows = 1u << 16;    // rows per counting-sort workgroup
constexpr u32 kSegmentEntries = 32;     // sorted entries per accumulation lane
constexpr u32 kOffsetChunkBuckets = 512; // buckets per k_bucket_offsets workgroup

// buckets a reduce block covers (threads per block x buckets per thread), shared with kernels.h
constexpr u32 kReduceThreads = 256;
constexpr u32 kReduceSegment = 8;
constexpr u32 kReduceBlockBuckets = kReduceThreads * kReduceSegment;

// device-visible task descriptor
struct task_desc {
  u32 column;
  u32 window;
  u64 rows;         // rows of the column
  u32 num_buckets;  // 2^(c-1)
  u32 num_slices;   // ceil(rows / kSliceRows)
  u64 bucket_base;  // first bucket of this task in the flat bucket arrays
  u64 entry_base;   // first entry of this task in the flat digit / sorted-index arrays
  u64 hist_base;    // first counter of this task's [slice][bucket] histogram
  u64 segment_base; // first segment of this task in the flat per-segment arrays
  u32 chunk_base;   // first k_bucket_offsets chunk total of this task
  u32 pad;
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
  u32 num_windows; // W
  u32 first_task;  // task index of window 0; task = first + window
};

struct msm_plan {
  std::vector<column_desc> columns;
  std::vector<task_desc> tasks;
  u64 total_buckets = 0;
  u64 total_entries = 0;
  u64 total_hist = 0;
  u64 total_segments = 0;
  u32 total_chunks = 0;
  u64 max_rows = 0;        // longest column
  u32 max_task_buckets = 0;
  u32 max_task_slices = 0;
  u32 max_windows = 0;
};

struct msm_tuning {
  u32 max_window_bits = 16; // the LDS histogram holds 2^(c-1) 32-bit counters (128 KiB at c = 16)
  // batching of many-column jobs: tasks per launch (grid.y) and device workspace per batch
  size_t max_tasks_per_batch = 32768;
  size_t max_workspace_bytes = size_t{64} << 30;
  // window-width cost model: from this many columns on, buckets cost `throughput_bucket_cost`
  size_t throughput_columns = 4;
  double throughput_bucket_cost = 12.0;
  // k_bucket_scatter: bytes of sorted entries one (task, bucket range) unit covers
  size_t scatter_range_bytes = size_t{1} << 19;
};

inline u32 ceil_div_u32(u64 a, u64 b) { return static_cast<u32>((a + b - 1) / b); }

// choose the window width for a column of n rows and B significant bits: minimise
//   W * (n + bucket_cost * 2^(c-1)),  W = ceil((B + 1) / c)
// where bucket_cost is the price of reducing one bucket in units of one bucket addition.  A
// single column is latency-bound in k_reduce (few waves, the dependent chain is what matters): its
// buckets are nearly free (2.5).  Many columns fill the machine with reduce work (measured on
// MI355X, bn254, 32 x 2^20: 2.5 ns per bucket against 0.16 ns per addition), so buckets are priced
// at their throughput cost and narrower windows win.
inline u32 choose_window_bits(u64 n, u32 bits, const msm_tuning& tune, double bucket_cost = 2.5) {
  u32 best_c = 1;
  double best_cost = 1e300;
  const u32 cmax = tune.max_window_bits < bits + 1 ? tune.max_window_bits : bits + 1;
  for (u32 c = 2; c <= cmax; ++c) {
    const u32 w = ceil_div_u32(bits + 1, c);
    const double cost =
        static_cast<double>(w) *
        (static_cast<double>(n) + bucket_cost * static_cast<double>(1u << (c - 1)));
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

inline msm_plan make_msm_plan(const std::vector<host_column>& cols, const msm_tuning& tune = {}) {
  msm_plan plan;
  plan.columns.reserve(cols.size());
  size_t nonempty = 0;
  for (const auto& c : cols) nonempty += c.n != 0 ? 1 : 0;
  const double bucket_cost = nonempty >= tune.throughput_columns ? tune.throughput_bucket_cost : 2.5;
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
    const u32 bits = hc.bit_width;
    const u32 c = choose_window_bits(hc.n, bits, tune, bucket_cost);
    const u32 w = ceil_div_u32(bits + 1, c);
    const u32 buckets = 1u << (c - 1);
    const u32 slices = ceil_div_u32(hc.n, kSliceRows);
    cd.window_bits = c;
    cd.num_windows = w;
    for (u32 wi = 0; wi < w; ++wi) {
      task_desc t{};
      t.column = static_cast<u32>(ci);
      t.window = wi;
      t.rows = hc.n;
      t.num_buckets = buckets;
      t.num_slices = slices;
      t.bucket_base = plan.total_buckets;
      t.entry_base = plan.total_entries;
      t.hist_base = plan.total_hist;
      t.segment_base = plan.total_segments;
      t.chunk_base = plan.total_chunks;
      plan.total_buckets += buckets;
      // keep every task's entry range 16-byte aligned for both the i16 and the u32 views
      plan.total_entries += (hc.n + 7) & ~7ull;
      plan.total_hist += static_cast<u64>(slices) * buckets;
      plan.total_segments += (hc.n + kSegmentEntries - 1) / kSegmentEntries;
      plan.total_chunks += ceil_div_u32(buckets, kOffsetChunkBuckets);
      plan.tasks.push_back(t);
    }
    if (buckets > plan.max_task_buckets) plan.max_task_buckets = buckets;
    if (slices > plan.max_task_slices) plan.max_task_slices = slices;
    if (w > plan.max_windows) plan.max_windows = w;
    if (hc.n > plan.max_rows) plan.max_rows = hc.n;
    plan.columns.push_back(cd);
  }
  return plan;
}
} // namespace bz
