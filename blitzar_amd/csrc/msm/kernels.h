// gfx950 kernels of the variable-base MSM (Pippenger with signed windows, counting-sorted
// buckets).  Stage by stage (reference counterparts in SURVEY.md section 2.3):
//
//   k_prepare_addends  caller generators (C-ABI layout) -> resident addend array
//   k_recode           scalars -> signed radix-2^c digits, transposed to [task][row] int16
//                      (reference: mtxb digit extraction, sxt/multiexp/base/digit_utility.cc:27-98,
//                       and the scalar transpose sxt/multiexp/base/scalar_array.cc:34-104)
//   k_bucket_sort      per task: LDS histogram -> LDS exclusive scan -> LDS-cursor scatter of row
//                      indices, i.e. a counting sort by bucket (reference K1/K2:
//                      bucket_method2/multiproduct_table_kernel.h:32-93, multiproduct_table.cc:74-82)
//   k_accumulate       one lane per bucket: gather addends by sorted index, mixed-add
//                      (reference K3 bucket_method2/sum.h:41-72, K5 bucket_method/
//                       accumulation_kernel.h:37-75)
//   k_reduce           per task: sum_b (b + 1) * bucket[b]  (reference K4 bucket_method2/reduce.h:50-78,
//                      K8 + host combine_buckets bucket_method/combination.h:27-63)
//   k_combine          per column: group sums, Horner over windows, canonical encoding
//                      (reference: host rsto::batch_compress / batch_to_element_affine,
//                       sxt/cbindings/backend/cpu_backend.cc:117-152)
//
// No MFMA anywhere: the arithmetic is carry-propagating multi-limb integer math.
#pragma once

#include "blitzar_amd/csrc/msm/curve_traits.h"
#include "blitzar_amd/csrc/msm/plan.h"
#include "blitzar_amd/csrc/msm/recode.h"

namespace bz {

using i16 = int16_t;

constexpr u32 kSortThreads = 1024;
constexpr u32 kAccumulateThreads = 256;
constexpr u32 kCombineThreads = 256;

//--------------------------------------------------------------------------------------------------
// k_prepare_addends
//--------------------------------------------------------------------------------------------------
template <class C>
__global__ void __launch_bounds__(256)
    k_prepare_addends(typename C::addend* __restrict__ addends, const void* __restrict__ api_generators,
                      u64 n) {
  const u64 i = static_cast<u64>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  addends[i] = C::make_addend(api_generators, i);
}

//--------------------------------------------------------------------------------------------------
// k_recode
//--------------------------------------------------------------------------------------------------
// One lane per row.  Loads the little-endian scalar (1..32 bytes; two's complement when the column
// is signed, in which case |x| is recoded and every digit is negated), produces W signed digits
// D_w in [-2^(c-1), 2^(c-1)] with  x = sum_w D_w 2^(c w), and stores E = -D as int16 at
// digits[task.entry_base + row_in_group].  (Storing -D keeps c = 16 inside int16: D in
// [-32767, 32768].  Signed columns use c <= 15, enforced by the planner.)
static __global__ void __launch_bounds__(256)
    k_recode(i16* __restrict__ digits, const column_desc* __restrict__ columns,
             const task_desc* __restrict__ tasks) {
  const column_desc col = columns[blockIdx.y];
  const u64 row = static_cast<u64>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (row >= col.n) return;
  digit_recoder rec;
  rec.init(col.data + row * col.row_stride, col.bit_offset, col.bit_width, col.is_signed != 0,
           col.window_bits);
  const u32 group = static_cast<u32>(row / col.rows_per_group);
  const u32 r = static_cast<u32>(row - static_cast<u64>(group) * col.rows_per_group);
  for (u32 wi = 0; wi < col.num_windows; ++wi) {
    const int d = rec.next();
    const task_desc& task = tasks[col.first_task + wi * col.num_groups + group];
    digits[task.entry_base + r] = static_cast<i16>(-d);
  }
}

//--------------------------------------------------------------------------------------------------
// k_bucket_sort
//--------------------------------------------------------------------------------------------------
// One 1024-lane workgroup per task.  The task's 2^(c-1) bucket counters live in LDS (128 KiB at
// c = 16): pass 1 histograms the digits with LDS atomics, an in-LDS exclusive scan turns counts
// into cursors, pass 2 re-reads the digits (L2-resident) and scatters `row | sign << 31` through
// the LDS cursors.  Output: sorted[entry_base + ...] grouped by bucket, bucket_end[bucket_base + b]
// = end offset of bucket b (start = end of b - 1).
static __global__ void __launch_bounds__(kSortThreads)
    k_bucket_sort(u32* __restrict__ sorted, u32* __restrict__ bucket_end,
                  const i16* __restrict__ digits, const task_desc* __restrict__ tasks) {
  extern __shared__ __attribute__((aligned(16))) u32 lds[];
  __shared__ u32 wave_sums[kSortThreads / 64];
  const task_desc task = tasks[blockIdx.x];
  const u32 nb = task.num_buckets;
  const u32 tid = threadIdx.x;
  for (u32 b = tid; b < nb; b += kSortThreads) lds[b] = 0;
  __syncthreads();

  const i16* dig = digits + task.entry_base;
  const u32 rows = task.row_count;
  // entry ranges are padded to multiples of 8 entries -> 16-byte vector loads are in bounds
  const u32 nvec = (rows + 7) / 8;
  const uint4* dig4 = reinterpret_cast<const uint4*>(dig);
  for (u32 v = tid; v < nvec; v += kSortThreads) {
    const uint4 pack = dig4[v];
    const u32 words[4] = {pack.x, pack.y, pack.z, pack.w};
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const u32 r = v * 8 + k;
      const int e = static_cast<i16>(words[k >> 1] >> (16 * (k & 1)));
      if (r < rows && e != 0) {
        const u32 mag = e < 0 ? static_cast<u32>(-e) : static_cast<u32>(e);
        atomicAdd(&lds[mag - 1], 1u);
      }
    }
  }
  __syncthreads();

  // exclusive scan of nb counters by 1024 lanes, nb / 1024 consecutive counters per lane
  const u32 per = (nb + kSortThreads - 1) / kSortThreads;
  const u32 first = tid * per;
  u32 local = 0;
  for (u32 k = 0; k < per; ++k) {
    const u32 b = first + k;
    if (b < nb) local += lds[b];
  }
  // wave-level inclusive scan
  u32 incl = local;
  const u32 lane = tid & 63;
#pragma unroll
  for (u32 off = 1; off < 64; off <<= 1) {
    const u32 up = __shfl_up(incl, off, 64);
    if (lane >= off) incl += up;
  }
  if (lane == 63) wave_sums[tid >> 6] = incl;
  __syncthreads();
  if (tid < 64) {
    const u32 nw = kSortThreads / 64;
    u32 s = tid < nw ? wave_sums[tid] : 0;
    u32 si = s;
#pragma unroll
    for (u32 off = 1; off < 64; off <<= 1) {
      const u32 up = __shfl_up(si, off, 64);
      if (tid >= off) si += up;
    }
    if (tid < nw) wave_sums[tid] = si - s; // exclusive
  }
  __syncthreads();
  u32 run = wave_sums[tid >> 6] + incl - local;
  for (u32 k = 0; k < per; ++k) {
    const u32 b = first + k;
    if (b < nb) {
      const u32 cnt = lds[b];
      lds[b] = run; // cursor = start offset
      run += cnt;
      bucket_end[task.bucket_base + b] = run;
    }
  }
  __syncthreads();

  u32* out = sorted + task.entry_base;
  for (u32 v = tid; v < nvec; v += kSortThreads) {
    const uint4 pack = dig4[v];
    const u32 words[4] = {pack.x, pack.y, pack.z, pack.w};
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const u32 r = v * 8 + k;
      const int e = static_cast<i16>(words[k >> 1] >> (16 * (k & 1)));
      if (r < rows && e != 0) {
        // E = -D: positive E means the digit is negative -> subtract the generator
        const u32 mag = e < 0 ? static_cast<u32>(-e) : static_cast<u32>(e);
        const u32 pos = atomicAdd(&lds[mag - 1], 1u);
        out[pos] = r | (e > 0 ? 0x80000000u : 0u);
      }
    }
  }
}

//--------------------------------------------------------------------------------------------------
// k_accumulate
//--------------------------------------------------------------------------------------------------
// One lane per bucket: walk the bucket's slice of the sorted index array, gather each addend from
// the resident generator array and mixed-add it into a register-resident accumulator.
template <class C>
__global__ void __launch_bounds__(kAccumulateThreads)
    k_accumulate(typename C::point* __restrict__ bucket_sums, const u32* __restrict__ bucket_end,
                 const u32* __restrict__ sorted, const typename C::addend* __restrict__ addends,
                 const task_desc* __restrict__ tasks) {
  const task_desc task = tasks[blockIdx.y];
  const u32 b = blockIdx.x * kAccumulateThreads + threadIdx.x;
  if (b >= task.num_buckets) return;
  const u32* ends = bucket_end + task.bucket_base;
  const u32 begin = b == 0 ? 0 : ends[b - 1];
  const u32 end = ends[b];
  const u32* idx = sorted + task.entry_base;
  const typename C::addend* gens = addends + task.row_begin;
  typename C::point acc = C::identity();
  for (u32 i = begin; i < end; ++i) {
    const u32 e = idx[i];
    const typename C::addend q = gens[e & 0x7fffffffu];
    C::accumulate(acc, q, (e >> 31) != 0);
  }
  bucket_sums[task.bucket_base + b] = acc;
}

//--------------------------------------------------------------------------------------------------
// k_reduce
//--------------------------------------------------------------------------------------------------
// partial[task][block] = sum over the block's buckets of (b + 1) * bucket[b].
// Each lane owns kReduceSegment consecutive buckets: running sums give S = sum B_j and
// R = sum (j + 1) B_j; the lane's contribution is R + (first bucket index) * S, the small multiple
// by double-and-add; a workgroup LDS tree folds the 256 contributions.
template <class C>
__global__ void __launch_bounds__(kReduceThreads)
    k_reduce(typename C::point* __restrict__ partials, u32 partial_stride,
             const typename C::point* __restrict__ bucket_sums, const u32* __restrict__ bucket_end,
             const task_desc* __restrict__ tasks) {
  using point = typename C::point;
  __shared__ point tree[kReduceThreads];
  const task_desc task = tasks[blockIdx.y];
  const u32 nb = task.num_buckets;
  const u32 block_first = blockIdx.x * kReduceBlockBuckets;
  if (block_first >= nb) return;
  const u32 tid = threadIdx.x;
  const u32* ends = bucket_end + task.bucket_base;
  const u32 total = ends[nb - 1];
  point* dst = partials + static_cast<u64>(blockIdx.y) * partial_stride + blockIdx.x;
  if (total == 0) {
    if (tid == 0) *dst = C::identity();
    return;
  }
  const u32 seg_first = block_first + tid * kReduceSegment;
  point contrib = C::identity();
  if (seg_first < nb) {
    const u32 seg_last = seg_first + kReduceSegment < nb ? seg_first + kReduceSegment : nb;
    const u32 lo = seg_first == 0 ? 0 : ends[seg_first - 1];
    const u32 hi = ends[seg_last - 1];
    if (hi != lo) {
      const point* bs = bucket_sums + task.bucket_base;
      point s = C::identity();
      point r = C::identity();
      for (u32 b = seg_last; b-- > seg_first;) {
        s = C::add(s, bs[b]);
        r = C::add(r, s);
      }
      // r = sum (b - seg_first + 1) B_b ; add seg_first * s
      contrib = r;
      if (seg_first != 0) {
        point m = C::identity();
        bool started = false;
        for (int bit = 31 - __builtin_clz(seg_first); bit >= 0; --bit) {
          if (started) m = C::dbl_n(m, 1);
          if ((seg_first >> bit) & 1) {
            m = started ? C::add(m, s) : s;
            started = true;
          }
        }
        contrib = C::add(contrib, m);
      }
    }
  }
  tree[tid] = contrib;
  __syncthreads();
  for (u32 stride = kReduceThreads / 2; stride > 0; stride >>= 1) {
    if (tid < stride) tree[tid] = C::add(tree[tid], tree[tid + stride]);
    __syncthreads();
  }
  if (tid == 0) *dst = tree[0];
}

//--------------------------------------------------------------------------------------------------
// k_combine
//--------------------------------------------------------------------------------------------------
// One workgroup per column: fold the per-(window, group, block) partials into one sum per window
// (all windows concurrently, a power-of-two team of lanes per window), then lane 0 runs the Horner
// recurrence  acc = 2^c acc + window[w]  from the top window down and writes the canonical
// encoding (or the raw projective point when `projective_out`).
template <class C>
__global__ void __launch_bounds__(kCombineThreads)
    k_combine(u8* __restrict__ out, u32 out_stride, int projective_out,
              const typename C::point* __restrict__ partials, u32 partial_stride,
              const column_desc* __restrict__ columns, const task_desc* __restrict__ tasks) {
  using point = typename C::point;
  __shared__ point tree[kCombineThreads];
  const column_desc col = columns[blockIdx.x];
  const u32 tid = threadIdx.x;
  u8* dst = out + static_cast<u64>(blockIdx.x) * out_stride;
  const u32 W = col.num_windows;
  if (W == 0) {
    if (tid == 0) {
      if (projective_out) {
        C::store_projective(dst, C::identity());
      } else {
        C::encode(dst, C::identity());
      }
    }
    return;
  }
  // lanes per window: largest power of two with W * team <= 256
  u32 team = 1;
  while (team * 2 * W <= kCombineThreads) team *= 2;
  const u32 w = tid / team;
  const u32 lane = tid % team;
  const u32 nb = 1u << (col.window_bits - 1);
  const u32 blocks = (nb + kReduceBlockBuckets - 1) / kReduceBlockBuckets;
  const u32 P = col.num_groups * blocks;
  point sum = C::identity();
  if (w < W) {
    for (u32 e = lane; e < P; e += team) {
      const u32 g = e / blocks, blk = e % blocks;
      const u32 t = col.first_task + w * col.num_groups + g;
      sum = C::add(sum, partials[static_cast<u64>(t) * partial_stride + blk]);
    }
  }
  tree[tid] = sum;
  __syncthreads();
  for (u32 stride = team / 2; stride > 0; stride >>= 1) {
    if (w < W && lane < stride) tree[tid] = C::add(tree[tid], tree[tid + stride]);
    __syncthreads();
  }
  if (tid == 0) {
    point acc = tree[(W - 1) * team];
    for (u32 wi = W - 1; wi-- > 0;) {
      acc = C::dbl_n(acc, static_cast<int>(col.window_bits));
      acc = C::add(acc, tree[wi * team]);
    }
    if (projective_out) {
      C::store_projective(dst, acc);
    } else {
      C::encode(dst, acc);
    }
  }
}
} // namespace bz
