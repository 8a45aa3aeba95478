// Host-side planning of a variable-base multi-scalar multiplication.
//
// The unit of device work is a *task* = (column, window, row group):
//   * a column is one `sxt_sequence_descriptor` / `mtxb::exponent_sequence`
//     (cbindings/blitzar_api.h:115-131, sxt/multiexp/base/exponent_sequence.h:25-42),
//   * a window is one signed radix-2^c digit position of the column's scalars,
//   * a row group is a contiguous slice of the column's rows (groups exist so that narrow columns,
//     which have few windows and few buckets, still expose enough parallel buckets).
// Every task owns 2^(c-1) buckets (bucket id = |digit| - 1).  The column result is
//   sum_w 2^(c w) * sum_g  sum_b (b + 1) * bucket[w][g][b].
//
// This replaces, as one mechanism, the reference's three dispatch tiers
// (sxt/multiexp/curve/multiexponentiation.h:147-200: bucket_method2 / bucket_method / generic
// bit-plane multiproduct), which all use unsigned 8-bit windows or single bit planes.
#pragma once

#include <cstdint>
#include <vector>

#include "blitzar_amd/csrc/base/macros.h"

namespace bz {

// device-visible task descriptor
struct task_desc {
  u32 column;
  u32 window;
  u64 row_begin;   // first row of the column covered by this task
  u32 row_count;   // rows covered
  u32 num_buckets; // 2^(c-1)
  u64 bucket_base; // first bucket of this task in the flat bucket arrays
  u64 entry_base;  // first entry of this task in the flat digit / sorted-index arrays
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
  u32 num_groups;  // G
  u32 rows_per_group;
  u32 first_task;  // task index of (window 0, group 0); task = first + window * G + group
  u32 pad;
};

struct msm_plan {
  std::vector<column_desc> columns;
  std::vector<task_desc> tasks;
  u64 total_buckets = 0;
  u64 total_entries = 0;
  u64 max_rows = 0;        // longest column
  u32 max_task_buckets = 0;
  u32 max_task_rows = 0;
  u32 max_windows = 0;
  u32 max_partials_per_window = 0; // groups * reduce blocks, per column maximum
};

struct msm_tuning {
  u32 max_window_bits = 16;     // LDS histogram holds 2^(c-1) 32-bit counters (128 KiB at c = 16)
  u32 target_bucket_threads = 1u << 17; // ~ 256 CUs x 4 SIMDs x 64 lanes x 2 waves
  u32 min_group_rows = 1024;
  u32 max_group_rows = 1u << 20; // entries per task are indexed with 31 bits + sign
};

inline u32 ceil_div_u32(u64 a, u64 b) { return static_cast<u32>((a + b - 1) / b); }

// buckets a reduce block covers (threads per block x buckets per thread), shared with kernels.h
constexpr u32 kReduceThreads = 256;
constexpr u32 kReduceSegment = 8;
constexpr u32 kReduceBlockBuckets = kReduceThreads * kReduceSegment;

// choose the window width for a column of n rows and B significant bits: minimise
// point additions = W * (n + 2 * 2^(c-1)) with W = ceil((B + 1) / c)
inline u32 choose_window_bits(u64 n, u32 bits, const msm_tuning& tune) {
  u32 best_c = 1;
  double best_cost = 1e300;
  const u32 cmax = tune.max_window_bits < bits + 1 ? tune.max_window_bits : bits + 1;
  for (u32 c = 2; c <= cmax; ++c) {
    const u32 w = ceil_div_u32(bits + 1, c);
    const double cost =
        static_cast<double>(w) * (static_cast<double>(n) + 2.5 * static_cast<double>(1u << (c - 1)));
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
      cd.num_groups = 0;
      cd.rows_per_group = 0;
      plan.columns.push_back(cd);
      continue;
    }
    const u32 bits = hc.bit_width;
    const u32 c = choose_window_bits(hc.n, bits, tune);
    const u32 w = ceil_div_u32(bits + 1, c);
    const u32 buckets = 1u << (c - 1);
    // row groups: enough (window, group, bucket) threads to fill the chip, but keep every
    // group large enough that its buckets see several points
    u64 want_groups = tune.target_bucket_threads / (static_cast<u64>(w) * buckets);
    if (want_groups < 1) want_groups = 1;
    u64 max_groups_by_rows = hc.n / tune.min_group_rows;
    if (max_groups_by_rows < 1) max_groups_by_rows = 1;
    // a group should also average >= 8 points per bucket
    u64 max_groups_by_load = hc.n / (8ull * buckets);
    if (max_groups_by_load < 1) max_groups_by_load = 1;
    u64 groups = want_groups;
    if (groups > max_groups_by_rows) groups = max_groups_by_rows;
    if (groups > max_groups_by_load) groups = max_groups_by_load;
    const u64 min_groups = (hc.n + tune.max_group_rows - 1) / tune.max_group_rows;
    if (groups < min_groups) groups = min_groups;
    u64 rows_per_group = (hc.n + groups - 1) / groups;
    groups = (hc.n + rows_per_group - 1) / rows_per_group;

    cd.window_bits = c;
    cd.num_windows = w;
    cd.num_groups = static_cast<u32>(groups);
    cd.rows_per_group = static_cast<u32>(rows_per_group);
    for (u32 wi = 0; wi < w; ++wi) {
      for (u32 g = 0; g < groups; ++g) {
        task_desc t{};
        t.column = static_cast<u32>(ci);
        t.window = wi;
        t.row_begin = static_cast<u64>(g) * rows_per_group;
        const u64 end = t.row_begin + rows_per_group < hc.n ? t.row_begin + rows_per_group : hc.n;
        t.row_count = static_cast<u32>(end - t.row_begin);
        t.num_buckets = buckets;
        t.bucket_base = plan.total_buckets;
        t.entry_base = plan.total_entries;
        plan.total_buckets += buckets;
        // keep every task's entry range 16-byte aligned for both the i16 and the u32 views
        plan.total_entries += (static_cast<u64>(t.row_count) + 7) & ~7ull;
        plan.tasks.push_back(t);
        if (t.row_count > plan.max_task_rows) plan.max_task_rows = t.row_count;
      }
    }
    if (buckets > plan.max_task_buckets) plan.max_task_buckets = buckets;
    if (w > plan.max_windows) plan.max_windows = w;
    const u32 blocks = ceil_div_u32(buckets, kReduceBlockBuckets);
    const u32 partials = static_cast<u32>(groups) * blocks;
    if (partials > plan.max_partials_per_window) plan.max_partials_per_window = partials;
    if (hc.n > plan.max_rows) plan.max_rows = hc.n;
    plan.columns.push_back(cd);
  }
  return plan;
}
} // namespace bz
