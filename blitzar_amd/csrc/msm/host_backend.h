// `SXT_CPU_BACKEND`: a deliberately plain, single-threaded host implementation of the same MSM,
// selected only when the caller asks for it through `sxt_config.backend` / BLITZAR_BACKEND=cpu
// (reference: cpu_backend::compute_commitments, sxt/cbindings/backend/cpu_backend.cc:117-152).
// It is never used as a fallback: the GPU backend aborts when no device is present.
//
// Algorithm: signed radix-2^c bucket method per column (same recoding as the device path,
// recode.h), buckets folded with the running-sum trick, Horner over windows from the top.
#pragma once

#include <vector>

#include "blitzar_amd/csrc/msm/curve_traits.h"
#include "blitzar_amd/csrc/msm/plan.h"
#include "blitzar_amd/csrc/msm/recode.h"

namespace bz {

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
      std::vector<point> buckets(nb);
      for (u32 w = W; w-- > 0;) {
        acc = C::dbl_n(acc, static_cast<int>(c));
        for (auto& b : buckets) b = C::identity();
        bool any = false;
        for (u64 r = 0; r < col.n; ++r) {
          const int d = digits[r * W + w];
          if (d == 0) continue;
          any = true;
          C::accumulate(buckets[(d < 0 ? -d : d) - 1], addends[r], d < 0);
        }
        if (!any) continue;
        point run = C::identity();
        point sum = C::identity();
        for (u32 b = nb; b-- > 0;) {
          run = C::add(run, buckets[b]);
          sum = C::add(sum, run);
        }
        acc = C::add(acc, sum);
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
