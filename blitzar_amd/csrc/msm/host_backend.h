// `SXT_CPU_BACKEND`: a deliberately plain host implementation of the same MSM,
// selected only when the caller asks for it through `sxt_config.backend` / BLITZAR_BACKEND=cpu
// (reference: cpu_backend::compute_commitments, sxt/cbindings/backend/cpu_backend.cc:117-152).
// It is never used as a fallback: the GPU backend aborts when no device is present.
//
// Algorithm: signed radix-2^c bucket method per column (same recoding as the device path,
// recode.h), buckets folded with the running-sum trick, Horner over windows from the top.  The
// windows of a column are independent until the Horner step: they are dealt out to host threads
// (BLITZAR_AMD_HOST_THREADS, default: the hardware's, at most one per window; 1 = the serial
// loop the reference's cpu backend is).
#pragma once

#include <atomic>
#include <cstdlib>
#include <thread>
#include <vector>

#include "blitzar_amd/csrc/msm/curve_traits.h"
#include "blitzar_amd/csrc/msm/plan.h"
#include "blitzar_amd/csrc/msm/recode.h"

namespace bz {

// host threads for a column of `windows` windows
inline u32 host_threads(u32 windows) {
  static const u32 configured = [] {
    if (const char* v = std::getenv("BLITZAR_AMD_HOST_THREADS")) {
      const unsigned long t = std::strtoul(v, nullptr, 10);
      if (t >= 1) return static_cast<u32>(t);
    }
    const unsigned hw = std::thread::hardware_concurrency();
    return static_cast<u32>(hw == 0 ? 1 : hw);
  }();
  return configured < windows ? configured : windows;
}

// addends[i] for i < n; cols[].data are HOST pointers here
template <class C>
void msm_host(u8* out, u32 out_stride, bool projective_out, const std::vector<host_column>& cols,
              const typename C::addend* addends) {
  using point = typename C::point;
  msm_tuning tune;
  tune.max_window_bits = 13;
  for (size_t ci = 0; ci < cols.size(); ++ci) {
    const host_column& col = cols[ci];
    point acc = C::identity();
    if (col.n > 0) {
      const u32 bits = col.bit_width;
      const u32 c = choose_window_bits(col.n, bits, tune);
      const u32 W = ceil_div_u32(bits + 1, c);
      const u32 nb = 1u << (c - 1);
      std::vector<int16_t> digits(static_cast<size_t>(col.n) * W);
      for (u64 r = 0; r < col.n; ++r) {
        digit_recoder rec;
        rec.init(col.data + r * col.row_stride, col.bit_offset, col.bit_width, col.is_signed, c);
        for (u32 w = 0; w < W; ++w) digits[r * W + w] = static_cast<int16_t>(rec.next());
      }
      // window sums, in parallel; identity for a window without digits
      std::vector<point> window_sum(W, C::identity());
      auto do_window = [&](u32 w, std::vector<point>& buckets) {
        for (auto& b : buckets) b = C::identity();
        bool any = false;
        for (u64 r = 0; r < col.n; ++r) {
          const int d = digits[r * W + w];
          if (d == 0) continue;
          any = true;
          C::accumulate(buckets[(d < 0 ? -d : d) - 1], addends[r], d < 0);
        }
        if (!any) return;
        point run = C::identity();
        point sum = C::identity();
        for (u32 b = nb; b-- > 0;) {
          run = C::add(run, buckets[b]);
          sum = C::add(sum, run);
        }
        window_sum[w] = sum;
      };
      const u32 threads = host_threads(W);
      if (threads <= 1) {
        std::vector<point> buckets(nb);
        for (u32 w = 0; w < W; ++w) do_window(w, buckets);
      } else {
        std::atomic<u32> next{0};
        std::vector<std::thread> pool;
        for (u32 t = 0; t < threads; ++t) {
          pool.emplace_back([&] {
            std::vector<point> buckets(nb);
            for (u32 w = next.fetch_add(1); w < W; w = next.fetch_add(1)) do_window(w, buckets);
          });
        }
        for (auto& th : pool) th.join();
      }
      for (u32 w = W; w-- > 0;) {
        acc = C::dbl_n(acc, static_cast<int>(c));
        acc = C::add(acc, window_sum[w]);
      }
    }
    u8* dst = out + static_cast<size_t>(ci) * out_stride;
    if (projective_out) {
      C::store_projective(dst, acc);
    } else {
      C::encode(dst, acc);
    }
  }
}
} // namespace bz
