// gfx950 kernels of the variable-base MSM (Pippenger with signed windows, counting-sorted
// buckets, load-balanced segment accumulation).  Stage by stage (reference counterparts in
// SURVEY.md section 2.3):
//
//   k_prepare_addends  caller generators (C-ABI layout) -> resident addend array
//   k_recode           scalars -> signed radix-2^c digits, transposed to [task][row] int16
//                      (reference: mtxb digit extraction, sxt/multiexp/base/digit_utility.cc:27-98,
//                       and the scalar transpose sxt/multiexp/base/scalar_array.cc:34-104)
//   k_group_hist       per (task, slice of rows): LDS histogram of the digits by bucket *group*; the
//                      last workgroup of a task scans the group totals -> start of every group
//   k_group_scatter    per (task, slice): partition the digits into per-group runs of records
//   k_group_sort_all   per (task, group; one more grid row: the chunks of oversized groups): counting
//                      sort by bucket inside LDS -> `row | sign << 31` list, bucket end offsets,
//                      segment -> bucket map
//                      (together a two-pass radix sort by bucket; reference K1/K2:
//                       bucket_method2/multiproduct_table_kernel.h:32-93, multiproduct_table.cc:74-82)
//   k_accumulate       one lane per 32 consecutive *sorted entries* (not per bucket): gather
//                      addends, mixed-add, flush at bucket boundaries; a bucket that straddles
//                      lanes leaves "head" partials that k_reduce folds in
//                      (reference K3 bucket_method2/sum.h:41-72, K5 bucket_method/
//                       accumulation_kernel.h:37-75: one thread per bucket)
//   k_reduce           per task: sum_b (b + 1) * bucket[b]  (reference K4 bucket_method2/reduce.h:50-78,
//                      K8 + host combine_buckets bucket_method/combination.h:27-63)
//   k_horner           per column (and window range): Horner over windows, canonical encoding
//                      (reference: host rsto::batch_compress / batch_to_element_affine,
//                       sxt/cbindings/backend/cpu_backend.cc:117-152)
//
// No MFMA anywhere: the arithmetic is carry-propagating multi-limb integer math.
#pragma once

#include <utility>

#include "blitzar_amd/csrc/msm/curve_traits.h"
#include "blitzar_amd/csrc/msm/plan.h"
#include "blitzar_amd/csrc/msm/recode.h"

namespace bz {

using i16 = int16_t;
using i32 = int32_t;
// Stored digits E = -D: int16 for window widths up to 16 (every launch without a window table), int32
// for the merged tasks of wider window tables (msm_plan::wide_digits; class D below).
template <class D> inline constexpr u32 kDigitsPerVector = 16 / sizeof(D); // per 16-byte load

constexpr u32 kSortThreads = 1024;
// k_reduce folds buckets with more than kReduceHeavyHeads head partials cooperatively, up to
// kReduceMaxHeavy of them per workgroup (any further ones stay with their lane)
constexpr u32 kReduceHeavyHeads = 16;
constexpr u32 kReduceMaxHeavy = 8;
constexpr u32 kAccumulateThreads = 256;
constexpr u32 kCombineThreads = 256;
#ifndef BZ_FRONT_PRIO
#define BZ_FRONT_PRIO 0
#endif
__device__ __forceinline__ void front_priority() {
#if BZ_FRONT_PRIO != 0
  __builtin_amdgcn_s_setprio(BZ_FRONT_PRIO);
#endif
}

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

// The same through LDS.  A lane reading its own generator touches ten (curve25519: 160-byte
// elements) different 128-byte lines per load instruction of the wavefront, and its addend leaves
// the same way: 288 MB moved at 4.2 TB/s.  Here every wavefront copies its 64 generators -- one
// contiguous 10 KiB piece -- into LDS with 16-byte loads that are consecutive across the lanes,
// converts from LDS, puts the addends back into the same LDS region and writes them out as one
// contiguous piece.  Needs 16-byte aligned arrays (the launcher checks); no workgroup barrier: a
// wavefront only reads what it wrote itself.
__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
template <class C>
__global__ void __launch_bounds__(256)
    k_prepare_addends_staged(typename C::addend* __restrict__ addends,
                             const void* __restrict__ api_generators, u64 n) {
  front_priority();
  using addend = typename C::addend;
  constexpr u32 G = static_cast<u32>(C::api_generator_size);
  constexpr u32 A = static_cast<u32>(sizeof(addend));
  static_assert(G % 8 == 0 && A % 16 == 0 && A <= G && (64 * G) % 16 == 0);
  __shared__ __attribute__((aligned(16))) u8 stage[4 * 64 * G];
  const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const u64 first = (static_cast<u64>(blockIdx.x) * 4 + wave) * 64;
  if (first >= n) return;
  const u32 count = static_cast<u32>(n - first < 64 ? n - first : 64);
  u8* region = stage + wave * (64 * G);
  const u8* src = static_cast<const u8*>(api_generators) + first * G;
  const u32 in_bytes = count * G;
  for (u32 off = lane * 16; off + 16 <= in_bytes; off += 64 * 16) {
    *reinterpret_cast<uint4*>(region + off) = *reinterpret_cast<const uint4*>(src + off);
  }
  if ((in_bytes & 8) != 0 && lane == 0) { // G = 8 (mod 16) and an odd count
    *reinterpret_cast<u64*>(region + in_bytes - 8) = *reinterpret_cast<const u64*>(src + in_bytes - 8);
  }
  wave_lds_fence();
  addend a;
  if (lane < count) a = C::make_addend(region, lane);
  wave_lds_fence(); // every lane has read its generator: the region is free
  if (lane < count) *reinterpret_cast<addend*>(region + lane * A) = a;
  wave_lds_fence();
  u8* dst = reinterpret_cast<u8*>(addends + first);
  const u32 out_bytes = count * A;
  for (u32 off = lane * 16; off < out_bytes; off += 64 * 16) {
    *reinterpret_cast<uint4*>(dst + off) = *reinterpret_cast<const uint4*>(region + off);
  }
}

// curve25519: caller generators arrive as projective element_p3 (any Z).  Normalising them to
// Z = 1 lets k_accumulate run the 7-product addition on 128-byte raw-limb rows (ed29_niels: nothing
// to unpack) instead of the 8-product one -- 16 additions per generator saved one product each --
// but costs an inversion per generator unless the inversions are shared: a workgroup takes
// 256 x kBatchPreparePoints generators, every lane multiplies its Z's up (Montgomery's trick), the
// 256 lane products meet in a binary product tree in LDS, the root is inverted ONCE by the first
// wavefront with the element spread over its lanes (ed16w::pow22523: ~250 row-parallel squarings,
// ~50 us -- every workgroup of the launch is in this phase at the same time, so it is paid once),
// and the inverse travels back down the tree and through the lanes' prefix products: 3 products per
// generator for the inversion instead of ~265, then x = X / Z, y = Y / Z, 2d x y.
// (C::batch_* hooks: curve_traits.h.  T is recomputed from x and y; a caller's T is only ever
// XY / Z, cbindings/blitzar_api.h:66-71.)
constexpr u32 kBatchPrepareThreads = 256;
constexpr u32 kBatchPreparePoints = 4;
// f(0), f(1), ..., f(N - 1) with compile-time indices (per-lane arrays stay in registers)
template <u32 N, class F> __device__ __forceinline__ void static_for(F&& f) {
  [&]<u32... J>(std::integer_sequence<u32, J...>) {
    (f(std::integral_constant<u32, J>{}), ...);
  }(std::make_integer_sequence<u32, N>{});
}
#ifndef BZ_BATCH_PREPARE_WAVES
#define BZ_BATCH_PREPARE_WAVES 1
#endif
// Montgomery's trick across a workgroup of T lanes (T a power of two): `tree` has 2 T entries in
// heap order (root 1, leaf T + lane).  tree_products: leaves -> products of every subtree, root in
// tree[1].  tree_inverses: with tree[1] replaced by the inverse of the root, every node becomes
// the inverse of its subtree's product (children get inv * sibling); the leaves end up holding the
// inverse of each lane's own product.
template <class C, u32 T>
__device__ __forceinline__ void tree_products(typename C::batch_fe* tree, u32 tid) {
  __syncthreads();
  for (u32 s = T / 2; s >= 1; s >>= 1) {
    if (tid < s) tree[s + tid] = C::batch_mul(tree[2 * (s + tid)], tree[2 * (s + tid) + 1]);
    __syncthreads();
  }
}
template <class C, u32 T>
__device__ __forceinline__ void tree_inverses(typename C::batch_fe* tree, u32 tid) {
  __syncthreads();
  for (u32 s = 1; s < T; s <<= 1) {
    if (tid < s) {
      const u32 i = s + tid;
      const typename C::batch_fe inv = tree[i], left = tree[2 * i], right = tree[2 * i + 1];
      tree[2 * i] = C::batch_mul(inv, right);
      tree[2 * i + 1] = C::batch_mul(inv, left);
    }
    __syncthreads();
  }
}

template <class C>
__global__ void __launch_bounds__(kBatchPrepareThreads, BZ_BATCH_PREPARE_WAVES)
    k_prepare_addends_batched(typename C::addend* __restrict__ addends,
                              const void* __restrict__ api_generators, u64 n) {
  using fe = typename C::batch_fe;
  __shared__ fe tree[2 * kBatchPrepareThreads];
  const u32 tid = threadIdx.x;
  const u64 base = static_cast<u64>(blockIdx.x) * kBatchPrepareThreads * kBatchPreparePoints;
  fe z[kBatchPreparePoints], prefix[kBatchPreparePoints];
  static_for<kBatchPreparePoints>([&](auto jc) {
    constexpr u32 j = decltype(jc)::value;
    const u64 i = base + static_cast<u64>(j) * kBatchPrepareThreads + tid;
    z[j] = i < n ? C::batch_load_z(api_generators, i) : C::batch_one();
    if constexpr (j == 0) {
      prefix[0] = z[0];
    } else {
      prefix[j] = C::batch_mul(prefix[j - 1], z[j]);
    }
  });
  tree[kBatchPrepareThreads + tid] = prefix[kBatchPreparePoints - 1];
  tree_products<C, kBatchPrepareThreads>(tree, tid);
  if (tid < 64) {
    const fe inv = C::batch_wave_invert(tree[1]); // all 64 lanes cooperate, all get the result
    if (tid == 0) tree[1] = inv;
  }
  tree_inverses<C, kBatchPrepareThreads>(tree, tid);
  fe inv = tree[kBatchPrepareThreads + tid];
  static_for<kBatchPreparePoints>([&](auto jc) {
    constexpr u32 j = kBatchPreparePoints - 1 - decltype(jc)::value;
    fe zinv = inv;
    if constexpr (j != 0) {
      zinv = C::batch_mul(inv, prefix[j - 1]);
      inv = C::batch_mul(inv, z[j]);
    }
    const u64 i = base + static_cast<u64>(j) * kBatchPrepareThreads + tid;
    if (i < n) addends[i] = C::batch_make_addend(api_generators, i, zinv);
  });
}

// The same normalisation without a workgroup waiting for its own inversion: k_prepare_addends_batched
// has every workgroup of the launch sit through a ~50 us exponentiation at the same time (all but one
// of its wavefronts idle), which is what made per-call normalisation lose in round 2.  Split in
// three launches, the machine-wide part has no latency chain in it and the chain runs ONCE, on one
// compute unit, beside whatever else the call has to do (recoding, the sort):
//   k_batch_products   every workgroup multiplies its 1024 Z's up (lane prefixes + the LDS tree) and
//                      writes ONE field element, the product of them all;
//   k_batch_invert     one workgroup per 1024 of those products: the same trick one level up, one
//                      wave-cooperative inversion per workgroup (2^20 generators: ONE workgroup);
//   k_batch_finish     every workgroup rebuilds its prefixes and tree (3 + 1 products per generator:
//                      cheaper than keeping 40 KiB of them per workgroup between the launches), takes
//                      the inverse of its product from k_batch_invert, walks it down the tree and
//                      the prefixes, and writes the Z = 1 addends.
template <class C>
__device__ __forceinline__ void batch_prefixes(typename C::batch_fe (&z)[kBatchPreparePoints],
                                               typename C::batch_fe (&prefix)[kBatchPreparePoints],
                                               const void* __restrict__ api_generators, u64 base,
                                               u64 n, u32 tid) {
  static_for<kBatchPreparePoints>([&](auto jc) {
    constexpr u32 j = decltype(jc)::value;
    const u64 i = base + static_cast<u64>(j) * kBatchPrepareThreads + tid;
    z[j] = i < n ? C::batch_load_z(api_generators, i) : C::batch_one();
    if constexpr (j == 0) {
      prefix[0] = z[0];
    } else {
      prefix[j] = C::batch_mul(prefix[j - 1], z[j]);
    }
  });
}

template <class C>
__global__ void __launch_bounds__(kBatchPrepareThreads)
    k_batch_products(typename C::batch_fe* __restrict__ block_products,
                     const void* __restrict__ api_generators, u64 n) {
  using fe = typename C::batch_fe;
  front_priority();
  __shared__ fe tree[2 * kBatchPrepareThreads];
  const u32 tid = threadIdx.x;
  const u64 base = static_cast<u64>(blockIdx.x) * kBatchPrepareThreads * kBatchPreparePoints;
  fe z[kBatchPreparePoints], prefix[kBatchPreparePoints];
  batch_prefixes<C>(z, prefix, api_generators, base, n, tid);
  tree[kBatchPrepareThreads + tid] = prefix[kBatchPreparePoints - 1];
  tree_products<C, kBatchPrepareThreads>(tree, tid);
  if (tid == 0) block_products[blockIdx.x] = tree[1];
}

// block_inverses[b] = 1 / block_products[b], 1024 per workgroup with one shared inversion
template <class C>
__global__ void __launch_bounds__(kBatchPrepareThreads)
    k_batch_invert(typename C::batch_fe* __restrict__ block_inverses,
                   const typename C::batch_fe* __restrict__ block_products, u32 num_blocks) {
  using fe = typename C::batch_fe;
  __builtin_amdgcn_s_setprio(3); // a latency chain on one compute unit, beside the call's front
  __shared__ fe tree[2 * kBatchPrepareThreads];
  const u32 tid = threadIdx.x;
  const u32 base = blockIdx.x * kBatchPrepareThreads * kBatchPreparePoints;
  fe z[kBatchPreparePoints], prefix[kBatchPreparePoints];
  static_for<kBatchPreparePoints>([&](auto jc) {
    constexpr u32 j = decltype(jc)::value;
    const u32 i = base + j * kBatchPrepareThreads + tid;
    z[j] = i < num_blocks ? block_products[i] : C::batch_one();
    if constexpr (j == 0) {
      prefix[0] = z[0];
    } else {
      prefix[j] = C::batch_mul(prefix[j - 1], z[j]);
    }
  });
  tree[kBatchPrepareThreads + tid] = prefix[kBatchPreparePoints - 1];
  tree_products<C, kBatchPrepareThreads>(tree, tid);
  if (tid < 64) {
    const fe inv = C::batch_wave_invert(tree[1]);
    if (tid == 0) tree[1] = inv;
  }
  tree_inverses<C, kBatchPrepareThreads>(tree, tid);
  fe inv = tree[kBatchPrepareThreads + tid];
  static_for<kBatchPreparePoints>([&](auto jc) {
    constexpr u32 j = kBatchPreparePoints - 1 - decltype(jc)::value;
    fe zinv = inv;
    if constexpr (j != 0) {
      zinv = C::batch_mul(inv, prefix[j - 1]);
      inv = C::batch_mul(inv, z[j]);
    }
    const u32 i = base + j * kBatchPrepareThreads + tid;
    if (i < num_blocks) block_inverses[i] = zinv;
  });
}

template <class C>
__global__ void __launch_bounds__(kBatchPrepareThreads)
    k_batch_finish(typename C::addend* __restrict__ addends,
                   const typename C::batch_fe* __restrict__ block_inverses,
                   const void* __restrict__ api_generators, u64 n) {
  using fe = typename C::batch_fe;
  front_priority();
  __shared__ fe tree[2 * kBatchPrepareThreads];
  const u32 tid = threadIdx.x;
  const u64 base = static_cast<u64>(blockIdx.x) * kBatchPrepareThreads * kBatchPreparePoints;
  fe z[kBatchPreparePoints], prefix[kBatchPreparePoints];
  batch_prefixes<C>(z, prefix, api_generators, base, n, tid);
  tree[kBatchPrepareThreads + tid] = prefix[kBatchPreparePoints - 1];
  tree_products<C, kBatchPrepareThreads>(tree, tid);
  if (tid == 0) tree[1] = block_inverses[blockIdx.x];
  tree_inverses<C, kBatchPrepareThreads>(tree, tid);
  fe inv = tree[kBatchPrepareThreads + tid];
  static_for<kBatchPreparePoints>([&](auto jc) {
    constexpr u32 j = kBatchPreparePoints - 1 - decltype(jc)::value;
    fe zinv = inv;
    if constexpr (j != 0) {
      zinv = C::batch_mul(inv, prefix[j - 1]);
      inv = C::batch_mul(inv, z[j]);
    }
    const u64 i = base + static_cast<u64>(j) * kBatchPrepareThreads + tid;
    if (i < n) addends[i] = C::batch_make_addend(api_generators, i, zinv);
  });
}

// field elements of scratch the three launches need for n generators (products | inverses)
inline size_t batch_prepare_scratch_elements(u64 n) {
  const u64 per_block = static_cast<u64>(kBatchPrepareThreads) * kBatchPreparePoints;
  return 2 * static_cast<size_t>((n + per_block - 1) / per_block) + 2;
}
template <class C>
void launch_prepare_addends_split(typename C::addend* d_addends, const void* d_api_generators, u64 n,
                                  typename C::batch_fe* d_scratch, hipStream_t stream) {
  if (n == 0) return;
  const u64 per_block = static_cast<u64>(kBatchPrepareThreads) * kBatchPreparePoints;
  const u32 blocks = ceil_div_u32(n, per_block);
  typename C::batch_fe* products = d_scratch;
  typename C::batch_fe* inverses = d_scratch + blocks + 1;
  hipLaunchKernelGGL((k_batch_products<C>), dim3(blocks), dim3(kBatchPrepareThreads), 0, stream,
                     products, d_api_generators, n);
  hipLaunchKernelGGL((k_batch_invert<C>), dim3(ceil_div_u32(blocks, per_block)),
                     dim3(kBatchPrepareThreads), 0, stream, inverses, products, blocks);
  hipLaunchKernelGGL((k_batch_finish<C>), dim3(blocks), dim3(kBatchPrepareThreads), 0, stream,
                     d_addends, inverses, d_api_generators, n);
}

#ifndef BZ_PREPARE_STAGED
#define BZ_PREPARE_STAGED 1
#endif
// C-ABI generators -> addends, by the curve's cheapest route
template <class C>
void launch_prepare_addends(typename C::addend* d_addends, const void* d_api_generators, u64 n,
                            hipStream_t stream) {
  if (n == 0) return;
  if constexpr (C::has_batched_prepare) {
    const u64 per_block = static_cast<u64>(kBatchPrepareThreads) * kBatchPreparePoints;
    hipLaunchKernelGGL((k_prepare_addends_batched<C>), dim3(ceil_div_u32(n, per_block)),
                       dim3(kBatchPrepareThreads), 0, stream, d_addends, d_api_generators, n);
  } else if ((reinterpret_cast<uintptr_t>(d_api_generators) & 15) == 0 &&
             (reinterpret_cast<uintptr_t>(d_addends) & 15) == 0 && BZ_PREPARE_STAGED != 0) {
    hipLaunchKernelGGL((k_prepare_addends_staged<C>), dim3(ceil_div_u32(n, 256)), dim3(256), 0,
                       stream, d_addends, d_api_generators, n);
  } else {
    hipLaunchKernelGGL((k_prepare_addends<C>), dim3(ceil_div_u32(n, 256)), dim3(256), 0, stream,
                       d_addends, d_api_generators, n);
  }
}

//--------------------------------------------------------------------------------------------------
// k_recode
//--------------------------------------------------------------------------------------------------
// One lane per row.  Loads the little-endian scalar (1..32 bytes; two's complement when the column
// is signed, in which case |x| is recoded and every digit is negated), produces W signed digits
// D_w in [-2^(c-1), 2^(c-1)] with  x = sum_w D_w 2^(c w), and stores E = -D as int16 at
// digits[task.entry_base + row].  (Storing -D keeps c = 16 inside int16: D in [-32767, 32768].
// Signed columns use c <= 15, enforced by the planner.)
// Work item = (column, chunk of 256 rows), numbered so that the workgroups one XCD receives in
// a row (ids = xcd mod 8) walk through the COLUMNS of one row chunk: the columns of a packed
// fixed-base call are bit fields of the same rows (`row_stride` ~12 KiB at config 5), so every
// lane touches its own cache line and the neighbouring columns find it in that XCD's L2.
template <class D>
__global__ void __launch_bounds__(256)
    k_recode(D* __restrict__ digits, const column_desc* __restrict__ columns,
             const task_desc* __restrict__ tasks, u32 num_columns, u32 num_chunks,
             u32* __restrict__ zero, u64 zero_words) {
  front_priority();
  // the group cursors of the sort start at zero: cleared here, by the first kernel of the call,
  // instead of by a memset (two fill kernels and two stream bubbles per call)
  for (u64 i = static_cast<u64>(blockIdx.x) * blockDim.x + threadIdx.x; i < zero_words;
       i += static_cast<u64>(gridDim.x) * blockDim.x) {
    zero[i] = 0;
  }
  const u64 chunk_groups = (num_chunks + 7) / 8;
  const u64 total = 8 * chunk_groups * num_columns;
  for (u64 id = blockIdx.x; id < total; id += gridDim.x) {
    const u32 xcd = static_cast<u32>(id & 7);
    const u64 rest = id >> 3;
    const column_desc col = columns[rest % num_columns];
    const u64 chunk = (rest / num_columns) * 8 + xcd;
    const u64 row = chunk * blockDim.x + threadIdx.x;
    if (col.merged_stride != 0) {
      // window tables: the column's windows are slices of one task, virtual row = window *
      // stride + row; rows between the column's end and the slice's are zero digits
      if (row >= col.merged_stride) continue;
      D* dst = digits + tasks[col.first_task].entry_base + row;
      digit_recoder rec;
      if (row < col.n) {
        rec.init(col.data + row * col.row_stride, col.bit_offset, col.bit_width, false,
                 col.window_bits);
      }
      for (u32 wi = 0; wi < col.num_windows; ++wi) {
        const int d = row < col.n ? rec.next() : 0;
        if (wi + 1 < col.num_windows || row < col.n) {
          dst[static_cast<u64>(wi) * col.merged_stride] = static_cast<D>(-d);
        }
      }
      continue;
    }
    if (row >= col.n) continue;
    digit_recoder rec;
    rec.init(col.data + row * col.row_stride, col.bit_offset, col.bit_width, col.is_signed != 0,
             col.window_bits);
    for (u32 wi = 0; wi < col.num_windows; ++wi) {
      const int d = rec.next();
      digits[tasks[col.first_task + wi].entry_base + row] = static_cast<D>(-d);
    }
  }
}

// k_recode for the common shape: byte-aligned unsigned 32-byte scalars, 16-bit windows, one task
// per window (the Pedersen entry points at 2^17 rows and more).  The generic kernel walks a bit
// cursor through a register array with run-time indices, which hipcc parks in LDS (14 KiB per
// workgroup, a dozen LDS round trips per digit: 35-50 us for 2^20 rows); here the sixteen 16-bit
// halves of the eight words ARE the raw digits, the carry chain is unrolled, and the tasks of a
// column lie `(n + 7) & ~7` entries apart, so nothing is looked up per window.  The stored value
// E = -D is the low half of -(raw + carry): D = t for t <= 2^15, t - 2^16 above.
static __global__ void __launch_bounds__(256)
    k_recode_rows32_c16(i16* __restrict__ digits, const column_desc* __restrict__ columns,
                        const task_desc* __restrict__ tasks, u32* __restrict__ zero, u64 zero_words) {
  front_priority();
  const u64 threads = static_cast<u64>(gridDim.x) * gridDim.y * blockDim.x;
  for (u64 i = (static_cast<u64>(blockIdx.y) * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x;
       i < zero_words; i += threads) {
    zero[i] = 0; // the sort's group cursors (see k_recode)
  }
  const column_desc col = columns[blockIdx.y];
  const u64 row = static_cast<u64>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (row >= col.n) return;
  const uint4* src = reinterpret_cast<const uint4*>(col.data + row * 32);
  const uint4 lo = src[0], hi = src[1];
  const u32 words[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
  const u64 stride = (col.n + 7) & ~u64{7};
  i16* dst = digits + tasks[col.first_task].entry_base + row;
  u32 carry = 0;
#pragma unroll
  for (int w = 0; w < 16; ++w) {
    const u32 t = ((words[w >> 1] >> (16 * (w & 1))) & 0xffffu) + carry;
    carry = t > 32768u ? 1u : 0u;
    dst[w * stride] = static_cast<i16>(0u - t);
  }
  dst[16 * stride] = static_cast<i16>(0u - carry);
}

// k_recode for the packed fixed-base calls (cbindings/blitzar_api.h:688-712): the columns are bit
// fields of the same rows, `row_stride` (~12 KiB at BASELINE configs[4]) apart, so a lane per row
// touches one cache line per row and column and uses a few bytes of it.  Here a workgroup takes 64
// rows: for every range of columns whose bytes span at most kPackedTileSpan it copies the rows'
// span into LDS with aligned 16-byte loads (coalesced along the row), then every wavefront takes
// columns of the range, lane = row, and recodes from LDS; a task's 64 digits leave as one 128-byte
// line.
template <class D>
__global__ void __launch_bounds__(kPackedRecodeThreads)
    k_recode_packed(D* __restrict__ digits, const column_desc* __restrict__ columns,
                    const task_desc* __restrict__ tasks, const recode_range* __restrict__ ranges,
                    u32 num_ranges, u64 row_stride, u64 max_rows, u64 data_rows,
                    u32* __restrict__ zero, u64 zero_words) {
  for (u64 i = static_cast<u64>(blockIdx.x) * blockDim.x + threadIdx.x; i < zero_words;
       i += static_cast<u64>(gridDim.x) * blockDim.x) {
    zero[i] = 0; // the sort's group cursors (see k_recode)
  }
  extern __shared__ __attribute__((aligned(16))) u8 tile[];
  __shared__ u32 row_shift[kPackedTileRows];
  const u64 row0 = static_cast<u64>(blockIdx.x) * kPackedTileRows;
  const u32 rows =
      static_cast<u32>(max_rows - row0 < kPackedTileRows ? max_rows - row0 : kPackedTileRows);
  const u32 tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (u32 ri = 0; ri < num_ranges; ++ri) {
    const recode_range range = ranges[ri];
    // rows [row0, row0 + rows) x bytes [base, base + span): 16-byte chunks from the aligned-down
    // start of every row; row_shift = what the alignment skipped
    const u32 chunks = (range.span + 15 + 15) / 16;
    for (u32 idx = tid; idx < rows * chunks; idx += kPackedRecodeThreads) {
      const u32 r = idx / chunks, k = idx % chunks;
      // rows past the caller's data (zero-digit filler of window-table slices) are never read
      if (row0 + r >= data_rows) continue;
      const uintptr_t start = reinterpret_cast<uintptr_t>(range.base) + (row0 + r) * row_stride;
      const uintptr_t aligned = start & ~static_cast<uintptr_t>(15);
      const uintptr_t lo = aligned + 16 * k; // this chunk covers [lo, lo + 16)
      u8* dst = tile + r * kPackedTilePitch + 16 * k;
      if (lo >= start && lo + 16 <= start + range.span) {
        *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(lo);
      } else {
        // first / last chunk of the row: never touch a byte outside the caller's buffer
        for (u32 i = 0; i < 16; ++i) {
          const uintptr_t at = lo + i;
          dst[i] = at >= start && at < start + range.span ? *reinterpret_cast<const u8*>(at) : 0;
        }
      }
      if (k == 0) row_shift[r] = static_cast<u32>(start & 15);
    }
    __syncthreads();
    for (u32 c = wave; c < range.num_columns; c += kPackedRecodeThreads / 64) {
      const column_desc col = columns[range.first_column + c];
      const u64 row = row0 + lane;
      if (col.merged_stride != 0 && lane < rows && row >= col.n && row < col.merged_stride) {
        // window tables: zero digits between the column's end and the slice's
        D* dst = digits + tasks[col.first_task].entry_base + row;
        for (u32 wi = 0; wi + 1 < col.num_windows; ++wi) {
          dst[static_cast<u64>(wi) * col.merged_stride] = 0;
        }
      }
      if (lane < rows && row < col.n) {
        // ten aligned words from the tile cover the <= 33 bytes of the field
        const u32 at = lane * kPackedTilePitch + row_shift[lane] +
                       static_cast<u32>(col.data - range.base);
        const u32* src = reinterpret_cast<const u32*>(tile + (at & ~3u));
        u32 words[10];
#pragma unroll
        for (int i = 0; i < 10; ++i) words[i] = src[i];
        digit_recoder rec;
        rec.init_words32(words, 8 * (at & 3) + col.bit_offset, col.bit_width, col.is_signed != 0,
                         col.window_bits);
        for (u32 wi = 0; wi < col.num_windows; ++wi) {
          const int d = rec.next();
          const u64 at = col.merged_stride != 0
                             ? tasks[col.first_task].entry_base +
                                   static_cast<u64>(wi) * col.merged_stride + row
                             : tasks[col.first_task + wi].entry_base + row;
          digits[at] = static_cast<D>(-d);
        }
      }
    }
    __syncthreads();
  }
}

//--------------------------------------------------------------------------------------------------
// sort by bucket: k_group_hist (+ the scan of the group totals) -> k_group_scatter (partition by
// bucket group) -> k_group_sort_all (counting sort of one group inside LDS)
//--------------------------------------------------------------------------------------------------
// Both partition sweeps visit the digits of a (task, slice) in the same vectorised order.  `fn(r, e)`
// is called for every non-zero stored digit e = -D of row r (relative to the slice).
// the stored digits E = -D of one 16-byte vector: eight int16 or four int32
template <class D>
__device__ __forceinline__ void unpack_digits(const uint4& pack, int e[kDigitsPerVector<D>]) {
  const u32 words[4] = {pack.x, pack.y, pack.z, pack.w};
  if constexpr (sizeof(D) == 2) {
#pragma unroll
    for (int k = 0; k < 8; ++k) e[k] = static_cast<i16>(words[k >> 1] >> (16 * (k & 1)));
  } else {
#pragma unroll
    for (int k = 0; k < 4; ++k) e[k] = static_cast<int>(words[k]);
  }
}

template <class D, class F>
__device__ __forceinline__ void for_each_slice_digit(const D* __restrict__ dig, u32 rows, F&& fn) {
  // slices start at multiples of 8 rows and entry ranges are padded to multiples of 8
  // entries, so 16-byte vector loads are aligned and in bounds
  constexpr u32 V = kDigitsPerVector<D>;
  const u32 nvec = (rows + V - 1) / V;
  const uint4* dig4 = reinterpret_cast<const uint4*>(dig);
  for (u32 v = threadIdx.x; v < nvec; v += kSortThreads) {
    int e[V];
    unpack_digits<D>(dig4[v], e);
#pragma unroll
    for (u32 k = 0; k < V; ++k) {
      const u32 r = v * V + k;
      if (r < rows && e[k] != 0) fn(r, e[k]);
    }
  }
}

// chunks an oversized group of `total` records is cut into in pass 2 (0: one workgroup sorts it)
// `stream_limit` = records up to which ONE workgroup streams a group through LDS in rounds of
// kLocalSortCapacity (16 x the capacity: sending such groups to the chunked path earlier was measured
// twice and is slower, profiles/round4_ab_sort_stream_limit.log); beyond it the group goes to the
// chunked path, whose workers share its chunks
constexpr u32 kStreamedSortRecords = 16 * kLocalSortCapacity;
__device__ __forceinline__ u32 big_chunks_of(u32 total, u32 stream_limit) {
  return total <= stream_limit ? 0 : (total + kLocalSortCapacity - 1) / kLocalSortCapacity;
}

// Pass 1b, one workgroup of T threads per task: exclusive scans over the groups.
//   group_start[task.group_base + g] = first record of group g, entry [G] = records of the task;
//   group_cursor (the totals, in place) = the same offsets, bumped by k_group_scatter;
//   group_chunk[task.group_base + g] = chunks of the oversized groups before g, entry [G] = their
//   number (all zero on uniform digits); the bucket counters of oversized groups are cleared and
//   the task is appended to big_tasks (big_tasks[0] = their count, zeroed by the recode kernel).
// The totals are read with agent-scope loads: they were written by atomics of other workgroups of
// the SAME launch (k_group_hist below).
template <u32 T>
__device__ __forceinline__ void
group_offsets_block(const task_desc& task, u32 task_index, u32* __restrict__ group_cursor,
                    u32* __restrict__ group_start, u32* __restrict__ group_chunk,
                    u32* __restrict__ bucket_count, u32* __restrict__ bucket_fill,
                    u32* __restrict__ big_tasks, u32* wave_sums, u32* wave_chunks,
                    u32 stream_limit) {
  constexpr u32 kWaves = T / 64;
  const u32 groups = task.num_groups;
  const u32 tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  u32* cur = group_cursor + task.group_base;
  u32* gs = group_start + task.group_base;
  u32* gc = group_chunk + task.group_base;
  u32 carry = 0, chunk_carry = 0;
  for (u32 g0 = 0; g0 < groups; g0 += T) {
    const u32 g = g0 + tid;
    const u32 total =
        g < groups ? __hip_atomic_load(&cur[g], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
    const u32 chunks = g < groups ? big_chunks_of(total, stream_limit) : 0;
    u32 incl = total, chunk_incl = chunks;
#pragma unroll
    for (u32 off = 1; off < 64; off <<= 1) {
      const u32 up = __shfl_up(incl, off, 64);
      const u32 chunk_up = __shfl_up(chunk_incl, off, 64);
      if (lane >= off) {
        incl += up;
        chunk_incl += chunk_up;
      }
    }
    __syncthreads(); // the sums of the previous round have been read
    if (lane == 63) {
      wave_sums[wave] = incl;
      wave_chunks[wave] = chunk_incl;
    }
    __syncthreads();
    u32 base = carry, chunk_base = chunk_carry;
    u32 round_sum = 0, round_chunks = 0;
    for (u32 w = 0; w < kWaves; ++w) {
      if (w < wave) {
        base += wave_sums[w];
        chunk_base += wave_chunks[w];
      }
      round_sum += wave_sums[w];
      round_chunks += wave_chunks[w];
    }
    if (g < groups) {
      gs[g] = base + incl - total;
      cur[g] = base + incl - total;
      gc[g] = chunk_base + chunk_incl - chunks;
      if (chunks != 0) {
        // an oversized group: its chunks meet in these counters (big_hist_body / big_sort_body)
        const u64 first = task.bucket_base + (static_cast<u64>(g) << task.group_bits);
        for (u32 b = 0; b < (1u << task.group_bits); ++b) {
          bucket_count[first + b] = 0;
          bucket_fill[first + b] = 0;
        }
      }
    }
    carry += round_sum;
    chunk_carry += round_chunks;
  }
  if (tid == 0) {
    gs[groups] = carry;
    gc[groups] = chunk_carry;
    // big_tasks[0] = number of tasks with oversized groups, then their indices
    if (chunk_carry != 0) big_tasks[1 + atomicAdd(&big_tasks[0], 1u)] = task_index;
  }
}

// Pass 1a.  group_total[task.group_base + g] += digits of the slice whose bucket lies in group g
// (2^s consecutive buckets): LDS histogram, one global atomic per populated group.
// The workgroup that finishes LAST on a task (a ticket per task, zeroed with the group cursors by
// the recode kernel) goes on to run pass 1b for it: no 17-workgroup launch of its own (~5 us).
template <class D>
__global__ void __launch_bounds__(kSortThreads, 8) // <= 64 VGPRs: two workgroups per CU
    k_group_hist(u32* __restrict__ group_total, u32* __restrict__ big_tasks,
                 const D* __restrict__ digits, const task_desc* __restrict__ tasks,
                 u32* __restrict__ arrivals, u32* __restrict__ group_start,
                 u32* __restrict__ group_chunk, u32* __restrict__ bucket_count,
                 u32* __restrict__ bucket_fill, u32 stream_limit) {
  front_priority();
  extern __shared__ __attribute__((aligned(16))) u32 lds[];
  __shared__ u32 wave_sums[kSortThreads / 64];
  __shared__ u32 wave_chunks[kSortThreads / 64];
  __shared__ u32 last_flag;
  const task_desc task = tasks[blockIdx.y];
  const u32 slice = blockIdx.x;
  if (slice >= task.num_slices) return;
  const u32 groups = task.num_groups, s = task.group_bits;
  const u32 tid = threadIdx.x;
  for (u32 g = tid; g < groups; g += kSortThreads) lds[g] = 0;
  __syncthreads();
  const u64 row0 = static_cast<u64>(slice) * task.slice_rows;
  const u32 rows =
      static_cast<u32>(task.rows - row0 < task.slice_rows ? task.rows - row0 : task.slice_rows);
  for_each_slice_digit<D>(digits + task.entry_base + row0, rows, [&](u32, int e) {
    const u32 mag = e < 0 ? static_cast<u32>(-e) : static_cast<u32>(e);
    atomicAdd(&lds[(mag - 1) >> s], 1u);
  });
  __syncthreads();
  u32* out = group_total + task.group_base;
  for (u32 g = tid; g < groups; g += kSortThreads) {
    if (lds[g] != 0) atomicAdd(&out[g], lds[g]);
  }
  // No fences: the group totals are only ever touched by agent-scope atomics (performed at the
  // device's coherence point, not in an XCD's L2) and read back with agent-scope atomic loads.
  // Hardware assumption, stated: on gfx950 an atomic without return counts in vmcnt until the
  // coherence point has acknowledged it, so every wavefront drains vmcnt explicitly before the
  // barrier behind which the ticket is taken (the HIP memory model alone would ask for a release /
  // acquire pair here; a __threadfence() per lane is an L2 write-back + invalidate per wavefront on
  // a part whose eight L2s are not coherent with each other: measured, it took this kernel from 11
  // to 600 us).
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    const u32 ticket = __hip_atomic_fetch_add(&arrivals[blockIdx.y], 1u, __ATOMIC_RELAXED,
                                              __HIP_MEMORY_SCOPE_AGENT);
    last_flag = ticket + 1 == task.num_slices ? 1u : 0u;
  }
  __syncthreads();
  if (last_flag == 0) return;
  group_offsets_block<kSortThreads>(task, blockIdx.y, group_total, group_start, group_chunk,
                                    bucket_count, bucket_fill, big_tasks, wave_sums, wave_chunks,
                                    stream_limit);
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains the wave's global
// memory counter (its release fence), which parks every wave until its fire-and-forget stores and
// prefetched loads have come back; the sort kernels exchange data through LDS alone.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// Pass 1c.  Every non-zero digit of the slice becomes one 32-bit record in its group's piece of
// the record list:
//   record = (digit negative) << 31 | (bucket mod 2^s) << (31 - s) | row        (row < 2^(31-s))
// The workgroup histograms its slice by group and claims one contiguous run per group with a
// global atomic on the group's cursor (so the order of the runs inside a group is arbitrary;
// bucket contents are order-free).  Scattered 4-byte stores cost one L2 transaction each (16.8 M
// of them at config 2, ~80 us), so the Staged variant first assembles the slice's records grouped
// in LDS and then copies every run out with whole-line stores, one wavefront per run; the direct
// variant (more than kMaxStagedGroups groups: columns beyond ~2^25 rows) stores records one by
// one.  Per vector the eight cursor bumps are issued before the eight dependent stores.
//   dynamic LDS: Staged ? 3 * groups + 1 + kStagedSliceRows : groups   words
// Staged slices are short enough for every digit vector to stay in registers, so the counting pass
// keeps what its LDS atomic returns -- the digit's rank inside its group -- and the second pass places
// the record at local_start[group] + rank with a plain LDS read instead of a second atomic.
template <bool Staged, class D>
__global__ void __launch_bounds__(kSortThreads, 8) // <= 64 VGPRs: two workgroups per CU
    k_group_scatter(u32* __restrict__ records, u32* __restrict__ group_cursor,
                    const D* __restrict__ digits, const task_desc* __restrict__ tasks) {
  front_priority();
  constexpr u32 V = kDigitsPerVector<D>;                   // digits per 16-byte vector: 8 or 4
  constexpr u32 H = kStagedSliceRows / (V * kSortThreads); // vectors a thread holds: 2 or 4
  static_assert(H * V * kSortThreads == kStagedSliceRows);
  extern __shared__ __attribute__((aligned(16))) u32 lds[];
  __shared__ u32 wave_sums[kSortThreads / 64];
  const task_desc task = tasks[blockIdx.y];
  const u32 slice = blockIdx.x;
  if (slice >= task.num_slices) return;
  const u32 groups = task.num_groups, s = task.group_bits;
  const u32 tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  u32* cursor = lds;                  // [groups]     count, then write cursor of every group
  u32* local_start = lds + groups;    // [groups + 1] Staged: first staged record of the group
  u32* run_base = lds + 2 * groups + 1; // [groups]   Staged: where the run goes in `records`
  u32* staging = lds + 3 * groups + 1;  // [kStagedSliceRows]
  for (u32 g = tid; g < groups; g += kSortThreads) cursor[g] = 0;
  lds_barrier();
  const u64 row0 = static_cast<u64>(slice) * task.slice_rows;
  const u32 rows =
      static_cast<u32>(task.rows - row0 < task.slice_rows ? task.rows - row0 : task.slice_rows);
  const u32 nvec = (rows + V - 1) / V;
  const uint4* dig4 = reinterpret_cast<const uint4*>(digits + task.entry_base + row0);
  // the first H vectors of every thread stay in registers (all of them at 16384-row slices)
  uint4 held[H];
#pragma unroll
  for (u32 j = 0; j < H; ++j) {
    held[j] = make_uint4(0, 0, 0, 0);
    if (tid + j * kSortThreads < nvec) held[j] = dig4[tid + j * kSortThreads];
  }
  u32 rank[H][V];
  if constexpr (Staged) {
#pragma unroll
    for (u32 j = 0; j < H; ++j) {
      const u32 v = tid + j * kSortThreads;
      int e[V];
      unpack_digits<D>(held[j], e);
#pragma unroll
      for (u32 k = 0; k < V; ++k) {
        const u32 mag = e[k] < 0 ? static_cast<u32>(-e[k]) : static_cast<u32>(e[k]);
        rank[j][k] = 0;
        if (v * V + k < rows && e[k] != 0) rank[j][k] = atomicAdd(&cursor[(mag - 1) >> s], 1u);
      }
    }
  } else {
    // (held[] is only ever indexed with compile-time constants: a run-time index would park it in
    // scratch)
    auto count_vector = [&](const uint4& vec, u32 v) {
      int e[V];
      unpack_digits<D>(vec, e);
#pragma unroll
      for (u32 k = 0; k < V; ++k) {
        const u32 mag = e[k] < 0 ? static_cast<u32>(-e[k]) : static_cast<u32>(e[k]);
        if (v * V + k < rows && e[k] != 0) atomicAdd(&cursor[(mag - 1) >> s], 1u);
      }
    };
#pragma unroll
    for (u32 j = 0; j < H; ++j) {
      if (tid + j * kSortThreads < nvec) count_vector(held[j], tid + j * kSortThreads);
    }
    for (u32 v = tid + H * kSortThreads; v < nvec; v += kSortThreads) count_vector(dig4[v], v);
  }
  lds_barrier();
  u32* cur = group_cursor + task.group_base;
  if constexpr (Staged) {
    // groups <= kSortThreads: one group per thread; exclusive scan of the counts
    const u32 count = tid < groups ? cursor[tid] : 0;
    u32 incl = count;
#pragma unroll
    for (u32 off = 1; off < 64; off <<= 1) {
      const u32 up = __shfl_up(incl, off, 64);
      if (lane >= off) incl += up;
    }
    if (lane == 63) wave_sums[wave] = incl;
    lds_barrier();
    u32 start = incl - count;
    for (u32 w = 0; w < wave; ++w) start += wave_sums[w];
    if (tid < groups) {
      local_start[tid] = start;
      run_base[tid] = count != 0 ? atomicAdd(&cur[tid], count) : 0;
      if (tid + 1 == groups) local_start[groups] = start + count;
    }
  } else {
    for (u32 g = tid; g < groups; g += kSortThreads) {
      const u32 count = cursor[g];
      cursor[g] = count != 0 ? atomicAdd(&cur[g], count) : 0;
    }
  }
  lds_barrier();
  u32* out = records + task.entry_base;
  const u32 in_group = (1u << s) - 1, shift = 31 - s;
  if constexpr (Staged) {
#pragma unroll
    for (u32 j = 0; j < H; ++j) {
      const u32 v = tid + j * kSortThreads;
      int e[V];
      unpack_digits<D>(held[j], e);
      u32 pos[V], rec[V];
      bool take[V];
#pragma unroll
      for (u32 k = 0; k < V; ++k) {
        // E = -D: positive E means the digit is negative -> subtract the generator
        const u32 bucket = (e[k] < 0 ? static_cast<u32>(-e[k]) : static_cast<u32>(e[k])) - 1;
        take[k] = v * V + k < rows && e[k] != 0;
        rec[k] = (e[k] > 0 ? 0x80000000u : 0u) | ((bucket & in_group) << shift) |
                 (static_cast<u32>(row0) + v * V + k);
        pos[k] = take[k] ? local_start[bucket >> s] + rank[j][k] : 0;
      }
#pragma unroll
      for (u32 k = 0; k < V; ++k) {
        if (take[k]) staging[pos[k]] = rec[k];
      }
    }
    lds_barrier();
    for (u32 g = wave; g < groups; g += kSortThreads / 64) {
      const u32 from = local_start[g], count = local_start[g + 1] - from;
      u32* dst = out + run_base[g];
      for (u32 i = lane; i < count; i += 64) dst[i] = staging[from + i];
    }
  } else {
    auto place_vector = [&](const uint4& vec, u32 v) {
      int e[V];
      unpack_digits<D>(vec, e);
      u32 pos[V], rec[V];
      bool take[V];
#pragma unroll
      for (u32 k = 0; k < V; ++k) {
        const u32 bucket = (e[k] < 0 ? static_cast<u32>(-e[k]) : static_cast<u32>(e[k])) - 1;
        take[k] = v * V + k < rows && e[k] != 0;
        rec[k] = (e[k] > 0 ? 0x80000000u : 0u) | ((bucket & in_group) << shift) |
                 (static_cast<u32>(row0) + v * V + k);
        pos[k] = 0;
        if (take[k]) pos[k] = atomicAdd(&cursor[bucket >> s], 1u);
      }
#pragma unroll
      for (u32 k = 0; k < V; ++k) {
        if (take[k]) out[pos[k]] = rec[k];
      }
    };
#pragma unroll
    for (u32 j = 0; j < H; ++j) {
      if (tid + j * kSortThreads < nvec) place_vector(held[j], tid + j * kSortThreads);
    }
    for (u32 v = tid + H * kSortThreads; v < nvec; v += kSortThreads) place_vector(dig4[v], v);
  }
}

// Pass 2: counting sort of every group's records by bucket.
//   sorted[task.entry_base + pos] = row | (digit negative) << 31, grouped by bucket;
//   bucket_end[task.bucket_base + b] = end offset of bucket b in the task's sorted list;
//   segment_bucket[task.segment_base + pos / 32] = bucket of the entry that starts a segment.
// One workgroup per (task, group): a group of at most kLocalSortCapacity records
// (nearly every group, on uniform digits) is held in registers, count -> scan -> rank run in LDS, the
// group's piece of the sorted list is assembled in LDS and written out in order (coalesced).
// Up to 16 times that (a window whose digits use few of its buckets, like the top one) the
// workgroup streams its records twice and writes its piece directly.
// A larger group still (skewed digits: constants, booleans; very long columns) is cut into chunks of
// kLocalSortCapacity records that cooperate through global memory; the workers of one more grid row
// loop over the chunks in two phases (nothing to do on uniform digits):
//   big_hist_body  adds every chunk's bucket histogram into bucket_count,
//   big_sort_body  scans the group's bucket counts, claims a run per (chunk, bucket) with an
//                  atomic on bucket_fill, sorts the chunk in LDS and copies the runs out.
// Equal keys of a wavefront are combined before they touch an LDS counter (wave_aggregated_add):
// on skewed data all 64 lanes hit the same counter, which would serialise them.
constexpr u32 kGroupSortThreads = 512;
constexpr u32 kLocalSortPerThread = kLocalSortCapacity / kGroupSortThreads;
constexpr u32 kBigSortBlocks = 128; // at most this many workgroups share the chunks of the
                                    // oversized groups (they loop)
static_assert(kLocalSortPerThread * kGroupSortThreads == kLocalSortCapacity);
static_assert(2 * kGroupSortThreads >= (1u << kMaxGroupBits));

// LDS of the pass-2 kernels.  `run_base` / `local_start` are used by the oversized-group path only.
struct sort_lds {
  u32 cursor[1u << kMaxGroupBits];
  u32 staging[kLocalSortCapacity];
  u32 wave_sums[kGroupSortThreads / 64];
};
struct big_sort_lds {
  u32 run_base[1u << kMaxGroupBits];    // where this chunk's run of the bucket goes (group-relative)
  u32 local_start[1u << kMaxGroupBits]; // first staged entry of the bucket
};

// The counting pass keeps the rank its atomic returns, the placing pass adds the bucket's start with
// a plain LDS read (one LDS atomic per record instead of two).
__device__ __forceinline__ void
group_sort_block(u32 g, const task_desc& task, u32* __restrict__ sorted,
                 u32* __restrict__ segment_bucket, u32* __restrict__ bucket_end,
                 const u32* __restrict__ records, const u32* __restrict__ group_start,
                 const u32* __restrict__ group_chunk, sort_lds& lds) {
  u32* cursor = lds.cursor;
  u32* staging = lds.staging;
  u32* wave_sums = lds.wave_sums;
  if (g >= task.num_groups) return;
  const u32* gc = group_chunk + task.group_base;
  if (gc[g + 1] != gc[g]) return; // oversized: the chunked path
  const u32 s = task.group_bits, buckets = 1u << s;
  const u32 tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const u32* gs = group_start + task.group_base;
  const u32 begin = gs[g], total = gs[g + 1] - begin;
  for (u32 b = tid; b < buckets; b += kGroupSortThreads) cursor[b] = 0;
  lds_barrier();
  const u32* rec = records + task.entry_base + begin;
  const u32 in_group = buckets - 1, shift = 31 - s, row_mask = (1u << shift) - 1;
  const bool staged = total <= kLocalSortCapacity; // uniform over the workgroup
  u32 mine[kLocalSortPerThread];
  // ranks inside the bucket (< kLocalSortCapacity < 2^16), two per register
  u32 rank2[(kLocalSortPerThread + 1) / 2];
  if (staged) {
#pragma unroll
    for (u32 k = 0; k < kLocalSortPerThread; ++k) {
      const u32 i = tid + k * kGroupSortThreads;
      mine[k] = i < total ? rec[i] : 0;
    }
#pragma unroll
    for (u32 k = 0; k < kLocalSortPerThread; ++k) {
      u32 r = 0;
      if (tid + k * kGroupSortThreads < total) {
        r = atomicAdd(&cursor[(mine[k] >> shift) & in_group], 1u);
      }
      if ((k & 1) == 0) {
        rank2[k / 2] = r;
      } else {
        rank2[k / 2] |= r << 16;
      }
    }
  } else {
    // up to kStreamedSortRecords records (a window whose digits use few of its buckets): streamed
    // twice, kLocalSortCapacity at a time
    for (u32 base = 0; base < total; base += kLocalSortCapacity) {
#pragma unroll
      for (u32 k = 0; k < kLocalSortPerThread; ++k) {
        const u32 i = base + tid + k * kGroupSortThreads;
        mine[k] = i < total ? rec[i] : 0;
      }
#pragma unroll
      for (u32 k = 0; k < kLocalSortPerThread; ++k) {
        if (base + tid + k * kGroupSortThreads < total) {
          atomicAdd(&cursor[(mine[k] >> shift) & in_group], 1u);
        }
      }
    }
  }
  lds_barrier();
  // exclusive scan of the <= 2^kMaxGroupBits bucket counts, two adjacent buckets per lane
  const u32 b0 = 2 * tid;
  const u32 c0 = b0 < buckets ? cursor[b0] : 0, c1 = b0 + 1 < buckets ? cursor[b0 + 1] : 0;
  const u32 local = c0 + c1;
  u32 incl = local;
#pragma unroll
  for (u32 off = 1; off < 64; off <<= 1) {
    const u32 up = __shfl_up(incl, off, 64);
    if (lane >= off) incl += up;
  }
  if (lane == 63) wave_sums[wave] = incl;
  lds_barrier();
  u32 start0 = incl - local;
  for (u32 w = 0; w < wave; ++w) start0 += wave_sums[w];
  const u32 start1 = start0 + c0;
  u32* ends = bucket_end + task.bucket_base + (static_cast<u64>(g) << s);
  if (b0 < buckets) {
    cursor[b0] = start0;
    ends[b0] = begin + start0 + c0;
  }
  if (b0 + 1 < buckets) {
    cursor[b0 + 1] = start1;
    ends[b0 + 1] = begin + start1 + c1;
  }
  lds_barrier();
  u32* out = sorted + task.entry_base + begin;
  u32* seg = segment_bucket + task.segment_base;
  const u32 seg_log2 = task.segment_log2, seg_mask = (1u << seg_log2) - 1;
  if (staged) {
    u32 at[kLocalSortPerThread];
#pragma unroll
    for (u32 k = 0; k < kLocalSortPerThread; ++k) {
      at[k] = 0;
      if (tid + k * kGroupSortThreads < total) {
        const u32 b = (mine[k] >> shift) & in_group;
        at[k] = cursor[b] + ((rank2[k / 2] >> (16 * (k & 1))) & 0xffffu);
      }
    }
#pragma unroll
    for (u32 k = 0; k < kLocalSortPerThread; ++k) {
      if (tid + k * kGroupSortThreads < total) {
        staging[at[k]] = (mine[k] & 0x80000000u) | (mine[k] & row_mask);
        if (((begin + at[k]) & seg_mask) == 0) {
          seg[(begin + at[k]) >> seg_log2] = (g << s) + ((mine[k] >> shift) & in_group);
        }
      }
    }
    lds_barrier();
    for (u32 i = tid; i < total; i += kGroupSortThreads) out[i] = staging[i];
  } else {
    for (u32 base = 0; base < total; base += kLocalSortCapacity) {
      u32 pos[kLocalSortPerThread];
#pragma unroll
      for (u32 k = 0; k < kLocalSortPerThread; ++k) {
        const u32 i = base + tid + k * kGroupSortThreads;
        mine[k] = i < total ? rec[i] : 0;
      }
#pragma unroll
      for (u32 k = 0; k < kLocalSortPerThread; ++k) {
        pos[k] = 0;
        if (base + tid + k * kGroupSortThreads < total) {
          pos[k] = atomicAdd(&cursor[(mine[k] >> shift) & in_group], 1u);
        }
      }
#pragma unroll
      for (u32 k = 0; k < kLocalSortPerThread; ++k) {
        if (base + tid + k * kGroupSortThreads < total) {
          out[pos[k]] = (mine[k] & 0x80000000u) | (mine[k] & row_mask);
          if (((begin + pos[k]) & seg_mask) == 0) {
            seg[(begin + pos[k]) >> seg_log2] = (g << s) + ((mine[k] >> shift) & in_group);
          }
        }
      }
    }
  }
}

// The oversized group that chunk `index` (counted over all oversized groups of the task) belongs
// to: the largest g with chunk_first[g] <= index; groups that fit one workgroup repeat the value
// of their successor, so the search lands on an oversized one.
__device__ __forceinline__ u32 locate_big_group(const u32* __restrict__ chunk_first, u32 groups,
                                                u32 index) {
  u32 lo = 0, hi = groups;
  while (hi - lo > 1) {
    const u32 mid = lo + (hi - lo) / 2;
    if (chunk_first[mid] <= index) {
      lo = mid;
    } else {
      hi = mid;
    }
  }
  return lo;
}

// counters[key] += 1 for every active lane, returning what the lane's own atomicAdd(..., 1) would
// have: keys shared by many lanes of the wavefront (the heavy hitters of skewed data, which would
// serialise 64 lanes on one LDS counter) are combined into one atomic per key, up to three keys;
// the remaining lanes -- all of them when the keys are spread -- use their own atomic.
__device__ __forceinline__ u32 wave_aggregated_add(u32* counters, u32 key, bool active) {
  const u32 lane = threadIdx.x & 63;
  u32 result = 0;
  unsigned long long todo = __ballot(active);
  for (int round = 0; round < 3 && todo != 0; ++round) {
    const int leader = __ffsll(static_cast<long long>(todo)) - 1;
    const u32 leader_key = __shfl(key, leader, 64);
    const unsigned long long same = __ballot(active && key == leader_key) & todo;
    if (__popcll(same) < 8) break;
    u32 base = 0;
    if (static_cast<int>(lane) == leader) {
      base = atomicAdd(&counters[leader_key], static_cast<u32>(__popcll(same)));
    }
    base = __shfl(base, leader, 64);
    if ((same >> lane) & 1) {
      result = base + static_cast<u32>(__popcll(same & ((1ull << lane) - 1)));
    }
    todo &= ~same;
  }
  if ((todo >> lane) & 1) result = atomicAdd(&counters[key], 1u);
  return result;
}

// Oversized groups, phase 1: worker `worker` of `workers` adds the bucket histograms of its chunks
// (of every listed task) into bucket_count.
__device__ __forceinline__ void
big_hist_body(u32 worker, u32 workers, u32* __restrict__ bucket_count,
              const u32* __restrict__ records, const u32* __restrict__ group_start,
              const u32* __restrict__ group_chunk, const task_desc* __restrict__ tasks,
              const u32* __restrict__ big_tasks, u32 num_big_tasks, u32* cursor) {
  for (u32 t = 0; t < num_big_tasks; ++t) {
    const task_desc task = tasks[big_tasks[1 + t]];
    const u32* gc = group_chunk + task.group_base;
    const u32 big_chunks = gc[task.num_groups];
    const u32 s = task.group_bits, buckets = 1u << s;
    const u32 tid = threadIdx.x;
    const u32 in_group = buckets - 1, shift = 31 - s;
    const u32* gs = group_start + task.group_base;
    // every worker takes chunks of every listed task, starting at a different one
    for (u32 index = (worker + 5 * t) % workers; index < big_chunks; index += workers) {
      const u32 g = locate_big_group(gc, task.num_groups, index);
      const u32 begin = gs[g] + (index - gc[g]) * kLocalSortCapacity;
      const u32 left = gs[g + 1] - begin;
      const u32 total = left < kLocalSortCapacity ? left : kLocalSortCapacity;
      for (u32 b = tid; b < buckets; b += kGroupSortThreads) cursor[b] = 0;
      lds_barrier();
      const u32* rec = records + task.entry_base + begin;
      u32 mine[kLocalSortPerThread];
#pragma unroll
      for (u32 k = 0; k < kLocalSortPerThread; ++k) {
        const u32 i = tid + k * kGroupSortThreads;
        mine[k] = i < total ? rec[i] : 0;
      }
#pragma unroll
      for (u32 k = 0; k < kLocalSortPerThread; ++k) {
        wave_aggregated_add(cursor, (mine[k] >> shift) & in_group,
                            tid + k * kGroupSortThreads < total);
      }
      lds_barrier();
      u32* counts = bucket_count + task.bucket_base + (static_cast<u64>(g) << s);
      for (u32 b = tid; b < buckets; b += kGroupSortThreads) {
        if (cursor[b] != 0) atomicAdd(&counts[b], cursor[b]);
      }
      lds_barrier(); // cursor is reused by the next chunk
    }
  }
}

// Oversized groups, phase 2 (all of phase 1 is complete and visible): scan the group's bucket
// counts, claim a run per (chunk, bucket) with an atomic on bucket_fill, sort the chunk in LDS and
// copy the runs out.  The staged records keep their bucket bits until they are copied out.
__device__ __forceinline__ void
big_sort_body(u32 worker, u32 workers, u32* __restrict__ sorted, u32* __restrict__ segment_bucket,
              u32* __restrict__ bucket_end, const u32* __restrict__ bucket_count,
              u32* __restrict__ bucket_fill, const u32* __restrict__ records,
              const u32* __restrict__ group_start, const u32* __restrict__ group_chunk,
              const task_desc* __restrict__ tasks, const u32* __restrict__ big_tasks,
              u32 num_big_tasks, sort_lds& lds, big_sort_lds& big) {
  u32* cursor = lds.cursor;
  u32* staging = lds.staging;
  u32* wave_sums = lds.wave_sums;
  u32* run_base = big.run_base;
  u32* local_start = big.local_start;
  for (u32 t = 0; t < num_big_tasks; ++t) {
    const task_desc task = tasks[big_tasks[1 + t]];
    const u32* gc = group_chunk + task.group_base;
    const u32 big_chunks = gc[task.num_groups];
    const u32 s = task.group_bits, buckets = 1u << s;
    const u32 tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const u32* gs = group_start + task.group_base;
    const u32 in_group = buckets - 1, shift = 31 - s, row_mask = (1u << shift) - 1;
    // block-wide exclusive scan of two adjacent values per lane (b0 = 2 tid)
    const u32 b0 = 2 * tid;
    auto scan_pairs = [&](u32 v0, u32 v1, u32& start0, u32& start1) {
      const u32 local = v0 + v1;
      u32 incl = local;
#pragma unroll
      for (u32 off = 1; off < 64; off <<= 1) {
        const u32 up = __shfl_up(incl, off, 64);
        if (lane >= off) incl += up;
      }
      lds_barrier(); // wave_sums of an earlier scan have been read
      if (lane == 63) wave_sums[wave] = incl;
      lds_barrier();
      start0 = incl - local;
      for (u32 w = 0; w < wave; ++w) start0 += wave_sums[w];
      start1 = start0 + v0;
    };
    for (u32 index = (worker + 5 * t) % workers; index < big_chunks; index += workers) {
      const u32 g = locate_big_group(gc, task.num_groups, index);
      const u32 chunk = index - gc[g];
      const u32 group_begin = gs[g], group_end = gs[g + 1];
      const u32 begin = group_begin + chunk * kLocalSortCapacity;
      const u32 left = group_end - begin;
      const u32 total = left < kLocalSortCapacity ? left : kLocalSortCapacity;
      const u32* rec = records + task.entry_base + begin;
      u32 mine[kLocalSortPerThread];
#pragma unroll
      for (u32 k = 0; k < kLocalSortPerThread; ++k) {
        const u32 i = tid + k * kGroupSortThreads;
        mine[k] = i < total ? rec[i] : 0;
      }
      // the group's bucket starts from the global histogram (written by atomics of other
      // workgroups, possibly of this very launch: agent-scope loads)
      const u32* counts = bucket_count + task.bucket_base + (static_cast<u64>(g) << s);
      const u32 g0 = b0 < buckets ? __hip_atomic_load(&counts[b0], __ATOMIC_RELAXED,
                                                      __HIP_MEMORY_SCOPE_AGENT)
                                  : 0;
      const u32 g1 = b0 + 1 < buckets ? __hip_atomic_load(&counts[b0 + 1], __ATOMIC_RELAXED,
                                                          __HIP_MEMORY_SCOPE_AGENT)
                                      : 0;
      u32 gstart0, gstart1;
      scan_pairs(g0, g1, gstart0, gstart1);
      if (chunk == 0) {
        u32* ends = bucket_end + task.bucket_base + (static_cast<u64>(g) << s);
        if (b0 < buckets) ends[b0] = group_begin + gstart0 + g0;
        if (b0 + 1 < buckets) ends[b0 + 1] = group_begin + gstart1 + g1;
      }
      // the chunk's own histogram
      for (u32 b = tid; b < buckets; b += kGroupSortThreads) cursor[b] = 0;
      lds_barrier();
#pragma unroll
      for (u32 k = 0; k < kLocalSortPerThread; ++k) {
        wave_aggregated_add(cursor, (mine[k] >> shift) & in_group,
                            tid + k * kGroupSortThreads < total);
      }
      lds_barrier();
      const u32 c0 = b0 < buckets ? cursor[b0] : 0, c1 = b0 + 1 < buckets ? cursor[b0 + 1] : 0;
      u32 lstart0, lstart1;
      scan_pairs(c0, c1, lstart0, lstart1);
      u32* fill = bucket_fill + task.bucket_base + (static_cast<u64>(g) << s);
      if (b0 < buckets) {
        local_start[b0] = lstart0;
        cursor[b0] = lstart0;
        run_base[b0] = gstart0 + (c0 != 0 ? atomicAdd(&fill[b0], c0) : 0);
      }
      if (b0 + 1 < buckets) {
        local_start[b0 + 1] = lstart1;
        cursor[b0 + 1] = lstart1;
        run_base[b0 + 1] = gstart1 + (c1 != 0 ? atomicAdd(&fill[b0 + 1], c1) : 0);
      }
      lds_barrier();
#pragma unroll
      for (u32 k = 0; k < kLocalSortPerThread; ++k) {
        const bool active = tid + k * kGroupSortThreads < total;
        const u32 b = (mine[k] >> shift) & in_group;
        const u32 pos = wave_aggregated_add(cursor, b, active);
        if (active) staging[pos] = mine[k];
      }
      lds_barrier();
      u32* out = sorted + task.entry_base + group_begin;
      u32* seg = segment_bucket + task.segment_base;
      const u32 seg_log2 = task.segment_log2, seg_mask = (1u << seg_log2) - 1;
      for (u32 i = tid; i < total; i += kGroupSortThreads) {
        const u32 r = staging[i];
        const u32 b = (r >> shift) & in_group;
        const u32 at = run_base[b] + (i - local_start[b]); // group-relative position
        out[at] = (r & 0x80000000u) | (r & row_mask);
        if (((group_begin + at) & seg_mask) == 0) {
          seg[(group_begin + at) >> seg_log2] = (g << s) + b;
        }
      }
      lds_barrier(); // the LDS arrays are reused by the next chunk
    }
  }
}

// The whole sort of a SMALL task in one workgroup (launches whose every task has ONE bucket group and
// at most kLocalSortCapacity rows: hundreds of short columns, the reference's bucket_method2 regime).
// For such a task pass 1 only turned digits into records of the one group; here the workgroup reads
// the task's digits, counting-sorts them by bucket in LDS and writes the sorted entries, the bucket
// ends and the segment -> bucket map -- one launch instead of three, two passes over the digits less
// (1024 columns x 4096 rows: the sort 1.5 ms of a 9.8 ms call).
template <class D>
__global__ void __launch_bounds__(kGroupSortThreads, 8)
    k_task_sort(u32* __restrict__ sorted, u32* __restrict__ segment_bucket,
                u32* __restrict__ bucket_end, const D* __restrict__ digits,
                const task_desc* __restrict__ tasks) {
  front_priority();
  __shared__ sort_lds lds;
  u32* cursor = lds.cursor;
  u32* staging = lds.staging;
  u32* wave_sums = lds.wave_sums;
  const task_desc task = tasks[blockIdx.x];
  const u32 buckets = task.num_buckets; // <= 2^kMaxGroupBits: the task is one group
  const u32 rows = static_cast<u32>(task.rows); // <= kLocalSortCapacity
  const u32 tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (u32 b = tid; b < buckets; b += kGroupSortThreads) cursor[b] = 0;
  lds_barrier();
  const D* dig = digits + task.entry_base;
  // sign << 31 | bucket << 13 | row (row < 6144 < 2^13, bucket < 2^10); 0 = a zero digit
  u32 mine[kLocalSortPerThread];
  u32 rank2[(kLocalSortPerThread + 1) / 2];
#pragma unroll
  for (u32 k = 0; k < kLocalSortPerThread; ++k) {
    const u32 i = tid + k * kGroupSortThreads;
    const int e = i < rows ? static_cast<int>(dig[i]) : 0;
    const u32 mag = e < 0 ? static_cast<u32>(-e) : static_cast<u32>(e);
    // E = -D: positive E means the digit is negative -> subtract the generator
    mine[k] = e == 0 ? 0u : ((e > 0 ? 0x80000000u : 0u) | ((mag - 1) << 13) | i | 0x40000000u);
    u32 r = 0;
    if (e != 0) r = atomicAdd(&cursor[mag - 1], 1u);
    if ((k & 1) == 0) {
      rank2[k / 2] = r;
    } else {
      rank2[k / 2] |= r << 16;
    }
  }
  lds_barrier();
  // exclusive scan of the bucket counts, two adjacent buckets per lane (as group_sort_block)
  const u32 b0 = 2 * tid;
  const u32 c0 = b0 < buckets ? cursor[b0] : 0, c1 = b0 + 1 < buckets ? cursor[b0 + 1] : 0;
  const u32 local = c0 + c1;
  u32 incl = local;
#pragma unroll
  for (u32 off = 1; off < 64; off <<= 1) {
    const u32 up = __shfl_up(incl, off, 64);
    if (lane >= off) incl += up;
  }
  if (lane == 63) wave_sums[wave] = incl;
  lds_barrier();
  u32 start0 = incl - local;
  u32 total = 0;
  for (u32 w = 0; w < kGroupSortThreads / 64; ++w) {
    if (w < wave) start0 += wave_sums[w];
    total += wave_sums[w];
  }
  const u32 start1 = start0 + c0;
  u32* ends = bucket_end + task.bucket_base;
  if (b0 < buckets) {
    cursor[b0] = start0;
    ends[b0] = start0 + c0;
  }
  if (b0 + 1 < buckets) {
    cursor[b0 + 1] = start1;
    ends[b0 + 1] = start1 + c1;
  }
  lds_barrier();
  u32* seg = segment_bucket + task.segment_base;
  const u32 seg_log2 = task.segment_log2, seg_mask = (1u << seg_log2) - 1;
#pragma unroll
  for (u32 k = 0; k < kLocalSortPerThread; ++k) {
    if (mine[k] != 0) {
      const u32 b = (mine[k] >> 13) & 0x3ffu;
      const u32 at = cursor[b] + ((rank2[k / 2] >> (16 * (k & 1))) & 0xffffu);
      staging[at] = (mine[k] & 0x80000000u) | (mine[k] & 0x1fffu);
      if ((at & seg_mask) == 0) seg[at >> seg_log2] = b;
    }
  }
  lds_barrier();
  u32* out = sorted + task.entry_base;
  for (u32 i = tid; i < total; i += kGroupSortThreads) out[i] = staging[i];
}

// Pass 2 with the oversized groups inside the same launch: one more row of the grid
// (blockIdx.y == num_tasks), whose first `workers` workgroups run the chunked path.  They return at
// once when no task has an oversized group (uniform digits: big_tasks[0] was settled by pass 1b).
// The two phases of that path are separated by a barrier among the workers (a counter in global
// memory, zeroed by the recode kernel), so ALL `workers` workgroups must be resident at once: the host
// sizes `workers` from the compute units the launch stream may use (engine.h: at most kBigSortBlocks,
// two per available CU -- 512 threads and 36 KiB of LDS each, four fit a CU), and every other
// workgroup of the launch terminates on its own, so the workers do get dispatched.  Phase 2 needs
// ~100 VGPRs; inside this kernel's budget of 64 (four workgroups per CU for pass 2 proper) it spills
// a little, on the skewed path only.
static __global__ void __launch_bounds__(kGroupSortThreads, 8) // <= 64 VGPRs: four workgroups per CU
    k_group_sort_all(u32* __restrict__ sorted, u32* __restrict__ segment_bucket,
                     u32* __restrict__ bucket_end, const u32* __restrict__ records,
                     const u32* __restrict__ group_start, const u32* __restrict__ group_chunk,
                     const task_desc* __restrict__ tasks, u32 num_tasks,
                     u32* __restrict__ bucket_count, u32* __restrict__ bucket_fill,
                     const u32* __restrict__ big_tasks, u32* __restrict__ big_barrier,
                     u32 max_workers) {
  front_priority();
  __shared__ sort_lds lds;
  if (blockIdx.y < num_tasks) {
    const task_desc task = tasks[blockIdx.y];
    group_sort_block(blockIdx.x, task, sorted, segment_bucket, bucket_end, records, group_start,
                     group_chunk, lds);
    return;
  }
  const u32 workers = gridDim.x < max_workers ? gridDim.x : max_workers;
  if (blockIdx.x >= workers) return;
  const u32 num_big_tasks = big_tasks[0];
  if (num_big_tasks == 0) return;
  big_hist_body(blockIdx.x, workers, bucket_count, records, group_start, group_chunk, tasks,
                big_tasks, num_big_tasks, lds.cursor);
  __shared__ big_sort_lds big;
  // The histogram counters are touched by agent-scope atomics only and read back with agent-scope
  // loads.  Same hardware assumption as k_group_hist's ticket: an atomic counts in vmcnt until the
  // coherence point has acknowledged it, so every wavefront drains vmcnt before the workgroup takes
  // its place at the barrier; no L2 write-back / invalidate per wavefront.
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(big_barrier, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (__hip_atomic_load(big_barrier, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < workers) {
      __builtin_amdgcn_s_sleep(16);
    }
  }
  __syncthreads();
  big_sort_body(blockIdx.x, workers, sorted, segment_bucket, bucket_end, bucket_count, bucket_fill,
                records, group_start, group_chunk, tasks, big_tasks, num_big_tasks, lds, big);
}

//--------------------------------------------------------------------------------------------------
// k_accumulate
//--------------------------------------------------------------------------------------------------
// One lane per segment of 2^task.segment_log2 (32..128, one value per launch: plan.h) consecutive
// sorted entries.  The lane walks its entries,
// gathers each addend from the resident generator array and adds it into a register-resident
// accumulator; at a bucket boundary the accumulator is flushed.  The lane in whose segment a
// bucket *starts* owns bucket_sums[bucket]; a lane that begins in the middle of a bucket writes
// that first partial to heads[segment] instead (k_reduce adds the heads of a bucket to its sum).
// Every lane therefore does at most one segment's worth of additions, whatever the digit
// distribution.
//
// Skewed data (many equal scalars) makes one bucket span many whole segments.  Consecutive lanes
// of a wavefront whose segments lie entirely inside the same bucket fold their partials with a
// segmented shuffle reduction (taken only when such a run exists: wave-uniform branch), and only
// the first lane of the run writes a head: `load_bucket` below applies the same geometric rule,
// so a bucket of m entries costs its consumer m / (64 segments) additions instead of m / segment.
template <class P> __device__ __forceinline__ P wave_shfl_down(const P& v, u32 delta) {
  static_assert(sizeof(P) % 4 == 0);
  P r;
  const u32* src = reinterpret_cast<const u32*>(&v);
  u32* dst = reinterpret_cast<u32*>(&r);
#pragma unroll
  for (u32 i = 0; i < sizeof(P) / 4; ++i) dst[i] = __shfl_down(src[i], delta, 64);
  return r;
}

// wave priorities (s_setprio, 0..3) of the two tail kernels, see k_reduce; BZ_FRONT_PRIO: of the
// front kernels (conversion, recoding, the sort), 0 = the hardware default, no instruction
#ifndef BZ_REDUCE_PRIO
#define BZ_REDUCE_PRIO 3
#endif
#ifndef BZ_HORNER_PRIO
#define BZ_HORNER_PRIO 3
#endif
// waves per SIMD the accumulation loop is compiled for: the curve's constant, unless the translation
// unit that INSTANTIATES the kernel overrides it (blitzar_amd/build.py: the curve25519 loop under a
// bound of two waves keeps its 140 VGPRs but loses a third of its s_nop padding).  The override only
// changes an attribute of the kernel's definition; every class constant is the same in every unit.
#ifdef BZ_ACCUMULATE_WAVES_OVERRIDE
#define BZ_ACCUMULATE_WAVES(C) (BZ_ACCUMULATE_WAVES_OVERRIDE)
#else
#define BZ_ACCUMULATE_WAVES(C) (C::accumulate_waves_per_simd)
#endif
// BZ_ACCUMULATE_NT (A/B): bit 0 -- the flushed sums / head partials leave through nontemporal stores
// (written once here, read once by k_reduce: they need not displace the addend table from the caches);
// bit 1 -- the sorted entries arrive through nontemporal loads (read once)
#ifndef BZ_ACCUMULATE_NT
#define BZ_ACCUMULATE_NT 0
#endif
template <class P> __device__ __forceinline__ void store_flushed(P* dst, const P& v) {
  if constexpr ((BZ_ACCUMULATE_NT & 1) != 0) {
    typedef u32 vec4 __attribute__((ext_vector_type(4)));
    const u32* w = reinterpret_cast<const u32*>(&v);
    u32* d = reinterpret_cast<u32*>(dst);
    constexpr u32 words = sizeof(P) / 4;
#pragma unroll
    for (u32 k = 0; k + 4 <= words; k += 4) {
      const vec4 piece = {w[k], w[k + 1], w[k + 2], w[k + 3]};
      __builtin_nontemporal_store(piece, reinterpret_cast<vec4*>(d + k));
    }
#pragma unroll
    for (u32 k = words & ~3u; k < words; ++k) __builtin_nontemporal_store(w[k], d + k);
  } else {
    *dst = v;
  }
}
__device__ __forceinline__ u32 load_entry(const u32* p) {
  if constexpr ((BZ_ACCUMULATE_NT & 2) != 0) {
    return __builtin_nontemporal_load(p);
  } else {
    return *p;
  }
}
template <class C>
__global__ void __launch_bounds__(kAccumulateThreads, BZ_ACCUMULATE_WAVES(C))
    k_accumulate(typename C::point* __restrict__ bucket_sums, typename C::point* __restrict__ heads,
                 const u32* __restrict__ bucket_end, const u32* __restrict__ segment_bucket,
                 const u32* __restrict__ sorted, const typename C::addend* __restrict__ addends,
                 const task_desc* __restrict__ tasks) {
  const task_desc task = tasks[blockIdx.y];
  const u32* ends = bucket_end + task.bucket_base;
  const u32 total = ends[task.num_buckets - 1];
  const u32 seg = blockIdx.x * kAccumulateThreads + threadIdx.x;
  const u32 seg_entries = 1u << task.segment_log2; // 32, or more for throughput-bound launches
  const u32 lo = seg << task.segment_log2;
  if (lo >= total) return;
  const u32 hi = lo + seg_entries < total ? lo + seg_entries : total;
  u32 b = segment_bucket[task.segment_base + seg];
  u32 b_end = ends[b];
  const u32 b_start = b == 0 ? 0 : ends[b - 1];
  bool owned = b_start == lo;
  // the whole segment lies inside a bucket that started in an earlier segment
  const bool whole = !owned && b_end >= lo + seg_entries;
  const u32* idx = sorted + task.entry_base;
  typename C::point* sums = bucket_sums + task.bucket_base;
  typename C::point acc = C::identity();
  // Software pipeline, one entry ahead, with no register copies: at the top of an iteration the
  // staged row (the packed words the previous iteration's gather delivered) is turned into the
  // operand the addition consumes -- pinned there, so the staging registers are dead before the
  // next gather is issued into them -- and that gather then has the whole addition (~1500 VALU
  // instructions) to land.  (The first version carried the packed row through the addition and
  // copied next -> current at the loop end: 33 v_mov_b64 + ~100 v_mov_b32 per iteration.)
  //
  // Bucket boundaries: with 64 lanes and ~1 boundary per 32 entries, SOME lane of the wavefront
  // flushes in ~87 % of the iterations, so the whole wavefront walks the flush block almost every
  // time.  It is therefore one store sequence to a selected destination (the bucket's sum if the
  // bucket started in this segment, else this segment's head partial), and the end of the next
  // bucket is fetched an iteration ahead instead of being waited for inside the block (fetched by
  // every iteration, outside the divergent block: issued inside it, hipcc waits for it on the spot
  // to merge it into the lanes that did not flush).
  const u32 last_bucket = task.num_buckets - 1;
  u32 next_end = ends[b < last_bucket ? b + 1 : last_bucket];
  typename C::point* flush_to = owned ? sums + b : heads + task.segment_base + seg;
  u32 e_cur = load_entry(idx + lo);
  u32 e_next = lo + 1 < hi ? load_entry(idx + lo + 1) : 0;
  // (curves with C::has_signed_gather fetch the row in the order the digit's sign asks for)
  // BZ_ACCUMULATE_ROW_MASK (timing experiments only, WRONG results): every gather lands in the first
  // mask + 1 rows of the table -- what the loop costs when its rows come from the nearest cache
#ifndef BZ_ACCUMULATE_ROW_MASK
#define BZ_ACCUMULATE_ROW_MASK 0x7fffffffu
#endif
  auto gather = [&](u32 entry) {
    if constexpr (C::has_signed_gather) {
      return C::gather(addends, entry & BZ_ACCUMULATE_ROW_MASK, (entry >> 31) != 0);
    } else {
      return addends[entry & BZ_ACCUMULATE_ROW_MASK];
    }
  };
  // BZ_ACCUMULATE_DIRECT=1 (A/B): no row in flight across the addition -- 32 registers fewer, for a
  // fourth wavefront per SIMD to hide the gather instead
#ifndef BZ_ACCUMULATE_DIRECT
#define BZ_ACCUMULATE_DIRECT 0
#endif
  // BZ_ACCUMULATE_DIRECT=2 (curves with C::has_split_add): the next row is requested between the two
  // halves of the addition, where neither the accumulator nor the operand is live
  constexpr bool kSplit = BZ_ACCUMULATE_DIRECT == 2 && C::has_split_add;
  constexpr bool kDirect = BZ_ACCUMULATE_DIRECT == 1 || (BZ_ACCUMULATE_DIRECT == 2 && !kSplit);
  typename C::addend staged = gather(e_cur);
  // The first entry of a segment meets the identity in every lane of the wavefront (and never a
  // bucket boundary: `b` is the bucket that holds entry `lo`), so it is loaded, not added:
  // curve25519 one field product instead of eight, the Weierstrass curves none (C::first).
  // BZ_ACCUMULATE_PEEL=0 keeps the uniform loop (A/B).
#ifndef BZ_ACCUMULATE_PEEL
#define BZ_ACCUMULATE_PEEL 1
#endif
  u32 first = lo;
  if constexpr (BZ_ACCUMULATE_PEEL != 0) {
    const typename C::operand q = C::stage(staged);
    const bool negate = (e_cur >> 31) != 0;
    const u32 next_entry = lo + 1 < hi ? e_next : e_cur;
    e_cur = e_next;
    if constexpr (!kDirect) staged = gather(next_entry);
    if (lo + 2 < hi) e_next = load_entry(idx + lo + 2);
    if constexpr (C::has_signed_gather) {
      acc = C::first_gathered(q, negate);
    } else {
      acc = C::first(q, negate);
    }
    first = lo + 1;
    // (materialised here: left to itself hipcc no longer updates the loop's accumulator in place
    // and copies it at the end of every iteration -- curve25519 36 v_mov_b32, 140 -> 179 VGPRs
    // -- C::first_pinned; the Weierstrass loops are better off without)
    if constexpr (C::first_pinned) {
      u32* w = reinterpret_cast<u32*>(&acc);
#pragma unroll
      for (u32 k = 0; k < sizeof(acc) / 4; ++k) asm volatile("" : "+v"(w[k]));
    }
  }
  for (u32 i = first; i < hi; ++i) {
    // the staged row first: the waits hipcc puts in front of its registers count every memory
    // operation of the wavefront in order, so behind the flush block they would also wait for the
    // block's nine stores to complete (curve25519 k_accumulate 0.629 -> 0.625 ms alone, 0.655 ->
    // 0.648 in a sequence; the Weierstrass kernels unchanged)
    if constexpr (kDirect) staged = gather(e_cur);
    const typename C::operand q = C::stage(staged);
    const bool negate = (e_cur >> 31) != 0;
    if (i == b_end) {
      store_flushed(flush_to, acc);
      ++b;
      b_end = next_end;
      while (b_end == i) { // empty buckets
        ++b;
        b_end = ends[b];
      }
      flush_to = sums + b;
      owned = true;
      acc = C::identity();
    }
    next_end = ends[b < last_bucket ? b + 1 : last_bucket];
    // the gather is unconditional (the last iteration re-reads its own row, a cache hit): a load
    // under `if (i + 1 < hi)` writes its registers in some lanes only, and hipcc then keeps two
    // copies of the row and moves it back and forth (bn254: 16 v_mov_b64 per iteration)
    const u32 next_entry = i + 1 < hi ? e_next : e_cur;
    e_cur = e_next;
    if constexpr (kSplit) {
      const typename C::completed mid = C::add_front(acc, q, negate);
      asm volatile("" ::: "memory");
      staged = gather(next_entry);
      if (i + 2 < hi) e_next = load_entry(idx + i + 2);
      asm volatile("" ::: "memory");
      acc = C::add_back(mid);
    } else {
      if constexpr (!kDirect) staged = gather(next_entry);
      if (i + 2 < hi) e_next = load_entry(idx + i + 2);
      if constexpr (C::has_signed_gather) {
        C::accumulate_gathered(acc, q, negate);
      } else {
        C::accumulate(acc, q, negate);
      }
    }
  }
  // runs of `whole` lanes (necessarily of one bucket) -> one head per run and wavefront
  const unsigned long long whole_mask = __ballot(whole);
  const u32 lane = threadIdx.x & 63;
  bool write_head = !owned;
  if ((whole_mask & (whole_mask >> 1)) != 0) {
    const u32 run = whole ? static_cast<u32>(__ffsll(static_cast<long long>(~(whole_mask >> lane)))) - 1 : 0;
    for (u32 d = 1; d < 64; d <<= 1) {
      const typename C::point other = wave_shfl_down(acc, d);
      if (whole && d < run) acc = C::add(acc, other);
    }
    if (whole && lane != 0 && ((whole_mask >> (lane - 1)) & 1) != 0) write_head = false;
  }
  if (owned || write_head) store_flushed(flush_to, acc);
}

// k_accumulate<C> is instantiated in a translation unit of its own per curve
// (msm_<curve>_accumulate.hip), which is compiled with the instruction-scheduling strategy that
// suits this one loop (blitzar_amd/build.py, TU_FLAGS); the TU that launches it declares the
// instantiation with this macro
#define BZ_ACCUMULATE_INSTANCE(KEYWORD, C)                                                         \
  KEYWORD template __global__ void k_accumulate<C>(                                                \
      typename C::point* __restrict__, typename C::point* __restrict__, const u32* __restrict__,   \
      const u32* __restrict__, const u32* __restrict__, const typename C::addend* __restrict__,    \
      const task_desc* __restrict__)

// complete sum of bucket b of a task: the owner's partial plus the heads of the following
// segments the bucket extends into.  Whole segments of one wavefront (64 consecutive segments)
// were folded into the first of them by k_accumulate.
template <class C>
__device__ __forceinline__ typename C::point
load_bucket(const typename C::point* __restrict__ sums, const typename C::point* __restrict__ heads,
            u32 begin, u32 end, u32 b, u32 seg_log2) {
  typename C::point v = sums[b];
  const u32 first = begin >> seg_log2, last = (end - 1) >> seg_log2;
  const u32 after_whole = end >> seg_log2; // first segment not entirely below `end`
  u32 s = first + 1;
  while (s <= last) {
    v = C::add(v, heads[s]);
    if (s < after_whole) {
      const u32 next_wave = (s / 64 + 1) * 64;
      s = next_wave < after_whole ? next_wave : after_whole;
    } else {
      ++s;
    }
  }
  return v;
}

// The head partials load_bucket visits for a bucket covering sorted entries [begin, end), in
// closed form: segment first + 1, every multiple of 64 strictly between that and the last whole
// segment boundary, and the trailing partial segment.
struct bucket_heads {
  u32 s0, k_lo, middle, tail_index, count; // indices: s0 | (k_lo + j) * 64, j < middle | tail_index
  __device__ __forceinline__ u32 index(u32 j) const {
    return j == 0 ? s0 : (j <= middle ? (k_lo + j - 1) * 64 : tail_index);
  }
};
__device__ __forceinline__ bucket_heads heads_of(u32 begin, u32 end, u32 seg_log2) {
  const u32 first = begin >> seg_log2, last = (end - 1) >> seg_log2;
  const u32 after_whole = end >> seg_log2;
  bucket_heads h;
  h.s0 = first + 1;
  h.k_lo = h.s0 / 64 + 1;
  const u32 k_hi_plus_1 = (after_whole + 63) / 64; // multiples 64 k with s0 < 64 k < after_whole
  h.middle = k_hi_plus_1 > h.k_lo ? k_hi_plus_1 - h.k_lo : 0;
  h.tail_index = after_whole;
  const u32 tail = after_whole <= last && after_whole > h.s0 ? 1 : 0;
  h.count = h.s0 > last ? 0 : 1 + h.middle + tail;
  return h;
}

//--------------------------------------------------------------------------------------------------
// k_reduce
//--------------------------------------------------------------------------------------------------
// partial[task][block] = sum over the block's buckets of (b + 1) * bucket[b]; a block covers
// 256 * 2^lane_log2 consecutive buckets, lane t the 2^lane_log2 buckets from block_first + t 2^lane_log2
// on (8 per lane for a latency-bound launch, up to 64 for a throughput-bound one: plan.h).
// Running sums over the lane's buckets give s_t = sum B_j and r_t = sum (j + 1) B_j.  The kernel
// is bound by its total number of point additions (one wavefront per SIMD already keeps the
// 64-bit multiplier busy), so what matters is how the weights of the lanes are applied:
//   * generic: every lane forms (first bucket index) * s_t by double-and-add (~22 operations per
//     lane), then an LDS tree folds the 256 contributions;
//   * curves with C::wave_add_multiple (curve25519): the lane weights come from an inclusive
//     suffix scan over the lanes,  sum_t t s_t = sum_{u >= 1} suffix_u  (8 additions), the tree
//     folds v_t = r_t + 8 suffix_t, and the one remaining multiple, block_first * (sum of the s_t),
//     is formed by the first wavefront with the point spread over its lanes (curve/ed16_wave.h,
//     ~10x less latency per dependent operation): ~28 % fewer additions per block.
// (The tree below costs 37-43 KiB of LDS per workgroup.  Round 3 replaced it by wave shuffles --
// 3 KiB, two barriers per level fewer, a lone k_reduce 0.200 -> 0.197 ms at config 2 -- hoping
// the front of the next call would run faster beside it: it did not, and the 9-limb Weierstrass
// curves went from 209 to 284 registers, one wavefront per SIMD instead of two, config 4's
// k_reduce 33.0 -> 42.6 ms.  Reverted: profiles/round3_ab_reduce_lds.log.)
// Scan: the lane weights as a suffix scan over the lanes + one multiple per workgroup
// (C::wave_add_multiple) instead of a double-and-add per lane.  curve25519: always.  Weierstrass
// curves: for launches of few columns, where it is the shorter chain (a lone bls12-381 k_reduce
// 0.82 -> 0.73 ms); a launch of many columns is bound by issue slots and barriers, not by the
// length of a lane's chain, and runs 30-60 % LONGER under the scan (config 4: 32.7 -> 42.6 ms,
// config 5: 13.7 -> 22.0, profiles/round4_ab_weierstrass_tails.txt)
// T: lanes per workgroup.  256, or 64 for launches of many SMALL tasks (256 buckets per task: thousands
// of short columns): a 256-lane workgroup would run such a task on one wavefront while three idle ones
// hold a CU's registers -- 1024 columns x 4096 rows: k_reduce 4.5 ms at two working wavefronts per CU.
template <class C, bool Scan, u32 T = kReduceThreads>
__global__ void __launch_bounds__(T)
    k_reduce(typename C::point* __restrict__ partials, u32 partial_stride,
             u32* __restrict__ task_total, const typename C::point* __restrict__ bucket_sums,
             const typename C::point* __restrict__ heads, const u32* __restrict__ bucket_end,
             const task_desc* __restrict__ tasks, u32 lane_log2) {
  using point = typename C::point;
  __shared__ point tree[T];
  // a latency chain at one wavefront per SIMD: when it runs beside another batch's k_accumulate
  // (msm_context: throughput mode) its instructions go first, the accumulation fills the slots it leaves
  __builtin_amdgcn_s_setprio(BZ_REDUCE_PRIO);
  const task_desc task = tasks[blockIdx.y];
  const u32 nb = task.num_buckets;
  const u32 seg_log2 = task.segment_log2; // k_accumulate's segments: where the head partials are
  const u32 lane_buckets = 1u << lane_log2;
  const u32 block_first = blockIdx.x * (T << lane_log2);
  if (block_first >= nb) return;
  const u32 tid = threadIdx.x;
  const u32* ends = bucket_end + task.bucket_base;
  const u32 total = ends[nb - 1];
  // entries of the task, for k_horner (which then reads nothing a following sort overwrites)
  if (blockIdx.x == 0 && tid == 0) task_total[blockIdx.y] = total;
  point* dst = partials + static_cast<u64>(blockIdx.y) * partial_stride + blockIdx.x;
  // a block none of whose buckets holds an entry (the upper blocks of a top window whose digits
  // use few of its buckets, the carry window): the identity, without the scan and the tree -- at
  // 252-bit scalars in 16-bit windows that is 30 of a column's 272 blocks, which otherwise share
  // compute units with blocks that have work
  const u32 block_end = block_first + (T << lane_log2) < nb
                            ? block_first + (T << lane_log2)
                            : nb;
  if (total == 0 || ends[block_end - 1] == (block_first == 0 ? 0 : ends[block_first - 1])) {
    if (tid == 0) *dst = C::identity();
    return;
  }
  const u32 seg_first = block_first + tid * lane_buckets;
  // lanes that own buckets, rounded up to a power of two (at least one wavefront): a task of few
  // buckets (many short columns: 256 buckets per task at 9-bit windows) fills a quarter of the
  // workgroup, and the scan and the tree below only involve these lanes -- the other wavefronts hold
  // identities and skip their additions
  u32 active = 64;
  while (active < T && active * lane_buckets < block_end - block_first) active <<= 1;
  const point* bs = bucket_sums + task.bucket_base;
  const point* hd = heads + task.segment_base;
  // Heavy buckets first.  A bucket that holds a large share of a skewed column (constants,
  // booleans) spans thousands of segments and leaves one head partial per 64 of them: its owner
  // lane would add them one after the other (2^20 equal scalars: 512 dependent additions, 2 ms).
  // The workgroup folds such buckets cooperatively -- the heads dealt out over the 256 lanes, an
  // LDS tree on top -- and phase 1 picks the finished sums up from LDS.
  __shared__ u32 heavy_bucket[kReduceMaxHeavy];
  __shared__ u32 heavy_count;
  __shared__ point heavy_sum[kReduceMaxHeavy];
  if (tid == 0) heavy_count = 0;
  __syncthreads();
  if (seg_first < nb) {
    const u32 seg_last = seg_first + lane_buckets < nb ? seg_first + lane_buckets : nb;
    u32 begin = seg_first == 0 ? 0 : ends[seg_first - 1];
    for (u32 b = seg_first; b < seg_last; ++b) {
      const u32 end = ends[b];
      if (end != begin && heads_of(begin, end, seg_log2).count > kReduceHeavyHeads) {
        const u32 slot = atomicAdd(&heavy_count, 1u);
        if (slot < kReduceMaxHeavy) heavy_bucket[slot] = b;
      }
      begin = end;
    }
  }
  __syncthreads();
  const u32 num_heavy = heavy_count < kReduceMaxHeavy ? heavy_count : kReduceMaxHeavy;
  for (u32 h = 0; h < num_heavy; ++h) {
    const u32 b = heavy_bucket[h];
    const bucket_heads list = heads_of(b == 0 ? 0 : ends[b - 1], ends[b], seg_log2);
    point part = C::identity();
    bool any = false;
    for (u32 j = tid; j < list.count; j += T) {
      part = any ? C::add(part, hd[list.index(j)]) : hd[list.index(j)];
      any = true;
    }
    tree[tid] = part;
    __syncthreads();
    for (u32 stride = T / 2; stride > 0; stride >>= 1) {
      if (tid < stride && tid + stride < list.count) tree[tid] = C::add(tree[tid], tree[tid + stride]);
      __syncthreads();
    }
    if (tid == 0) heavy_sum[h] = C::add(tree[0], bs[b]);
    __syncthreads();
  }
  point s = C::identity();
  point r = C::identity();
  bool populated = false;
  if (seg_first < nb) {
    const u32 seg_last = seg_first + lane_buckets < nb ? seg_first + lane_buckets : nb;
    const u32 lo = seg_first == 0 ? 0 : ends[seg_first - 1];
    const u32 hi = ends[seg_last - 1];
    if (hi != lo) {
      populated = true;
      u32 end = hi;
      for (u32 b = seg_last; b-- > seg_first;) {
        const u32 begin = b == 0 ? 0 : ends[b - 1];
        if (begin != end) {
          u32 folded = kReduceMaxHeavy;
          if (num_heavy != 0 && heads_of(begin, end, seg_log2).count > kReduceHeavyHeads) {
            for (u32 h = 0; h < num_heavy; ++h) folded = heavy_bucket[h] == b ? h : folded;
          }
          s = C::add(s, folded < kReduceMaxHeavy ? heavy_sum[folded]
                                                 : load_bucket<C>(bs, hd, begin, end, b, seg_log2));
        }
        r = C::add(r, s);
        end = begin;
      }
    }
  }
  if constexpr (Scan) {
    // inclusive suffix scan of s over the active lanes
    point x = s;
    for (u32 d = 1; d < active; d <<= 1) {
      tree[tid] = x;
      __syncthreads();
      if (tid + d < active) x = C::add(x, tree[tid + d]);
      __syncthreads();
    }
    // v_t = r_t + 2^lane_log2 * suffix_t (t >= 1), folded by the tree
    if (tid != 0 && tid < active) r = C::add(r, C::dbl_n(x, static_cast<int>(lane_log2)));
    tree[tid] = r;
    __syncthreads();
    for (u32 stride = active / 2; stride > 0; stride >>= 1) {
      if (tid < stride) tree[tid] = C::add(tree[tid], tree[tid + stride]);
      __syncthreads();
    }
    // block offset: lane 0 holds suffix_0 = the sum of the block's buckets
    if (block_first == 0) {
      if (tid == 0) *dst = tree[0];
      return;
    }
    if (tid == 0) tree[1] = x;
    __syncthreads();
    if (tid < 64) {
      const point sum = C::wave_add_multiple(tree[0], tree[1], block_first);
      if (tid == 0) *dst = sum;
    }
  } else {
    // r = sum (b - seg_first + 1) B_b ; add seg_first * s
    point contrib = r;
    if (populated && seg_first != 0) {
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
    tree[tid] = contrib;
    __syncthreads();
    for (u32 stride = active / 2; stride > 0; stride >>= 1) {
      if (tid < stride) tree[tid] = C::add(tree[tid], tree[tid + stride]);
      __syncthreads();
    }
    if (tid == 0) *dst = tree[0];
  }
}

// k_reduce with few copies of the point addition (round 4).  The form above inlines C::add at ten
// places (115 KB of code for curve25519, 226-445 KB for the Weierstrass curves), a wavefront walks
// nearly all of it once, and the instruction cache holds 64 KB.  On one of the two kinds of MI355X
// boxes in the pool that is harmless -- a wavefront fetches straight-line code as fast as it executes
// it -- on the other kind code beyond the cache arrives at half the speed (6.3 against 3.5 ns per
// instruction, tools/ubench/tail_latency.hip, profiles/round4_tail_latency_*), and this kernel takes
// 0.32 instead of 0.18 ms.  Here the same arithmetic runs through TWO addition sites that loops
// return to: one for the bucket walk (head partials, s += B, r += s take turns in it) and one for
// everything that exchanges points through LDS (suffix scan or lane weights, the tree, the folding
// of heavy buckets).  The operands are moved into place around each site (~100 register moves per
// 1250-instruction addition); the code a wavefront executes fits the cache.
template <class C>
__global__ void __launch_bounds__(kReduceThreads)
    k_reduce_compact(typename C::point* __restrict__ partials, u32 partial_stride,
                     u32* __restrict__ task_total, const typename C::point* __restrict__ bucket_sums,
                     const typename C::point* __restrict__ heads, const u32* __restrict__ bucket_end,
                     const task_desc* __restrict__ tasks, u32 lane_log2) {
  using point = typename C::point;
  __shared__ point tree[kReduceThreads];
  __builtin_amdgcn_s_setprio(BZ_REDUCE_PRIO);
  const task_desc task = tasks[blockIdx.y];
  const u32 nb = task.num_buckets;
  const u32 seg_log2 = task.segment_log2;
  const u32 lane_buckets = 1u << lane_log2;
  const u32 block_first = blockIdx.x * (kReduceThreads << lane_log2);
  if (block_first >= nb) return;
  const u32 tid = threadIdx.x;
  const u32* ends = bucket_end + task.bucket_base;
  const u32 total = ends[nb - 1];
  if (blockIdx.x == 0 && tid == 0) task_total[blockIdx.y] = total;
  point* dst = partials + static_cast<u64>(blockIdx.y) * partial_stride + blockIdx.x;
  const u32 block_end = block_first + (kReduceThreads << lane_log2) < nb
                            ? block_first + (kReduceThreads << lane_log2)
                            : nb;
  if (total == 0 || ends[block_end - 1] == (block_first == 0 ? 0 : ends[block_first - 1])) {
    if (tid == 0) *dst = C::identity();
    return;
  }
  const u32 seg_first = block_first + tid * lane_buckets;
  const point* bs = bucket_sums + task.bucket_base;
  const point* hd = heads + task.segment_base;
  __shared__ u32 heavy_bucket[kReduceMaxHeavy];
  __shared__ u32 heavy_count;
  __shared__ point heavy_sum[kReduceMaxHeavy];
  if (tid == 0) heavy_count = 0;
  __syncthreads();
  const u32 seg_last = seg_first < nb ? (seg_first + lane_buckets < nb ? seg_first + lane_buckets : nb)
                                      : seg_first;
  if (seg_first < nb) {
    u32 begin = seg_first == 0 ? 0 : ends[seg_first - 1];
    for (u32 b = seg_first; b < seg_last; ++b) {
      const u32 end = ends[b];
      if (end != begin && heads_of(begin, end, seg_log2).count > kReduceHeavyHeads) {
        const u32 slot = atomicAdd(&heavy_count, 1u);
        if (slot < kReduceMaxHeavy) heavy_bucket[slot] = b;
      }
      begin = end;
    }
  }
  __syncthreads();
  const u32 num_heavy = heavy_count < kReduceMaxHeavy ? heavy_count : kReduceMaxHeavy;

  point s = C::identity();
  point r = C::identity();
  bool populated = false;

  // ---- heavy buckets: one addition site, three kinds of step ----------------------------------------
  // every step: each lane publishes a point in its LDS slot, the active lanes add the point of slot
  // `src` to their partial.  Gather steps publish a head partial for the lane itself, tree steps the
  // partial for a lane `stride` below, the last step the bucket's own partial for lane 0.
  for (u32 h = 0; h < num_heavy; ++h) {
    const u32 b = heavy_bucket[h];
    const bucket_heads list = heads_of(b == 0 ? 0 : ends[b - 1], ends[b], seg_log2);
    point part = C::identity();
    const u32 gathers = (list.count + kReduceThreads - 1) / kReduceThreads;
    for (u32 step = 0; step < gathers + 9; ++step) {
      bool act;
      u32 src = tid;
      point publish = part;
      if (step < gathers) {
        const u32 j = tid + step * kReduceThreads;
        act = j < list.count;
        if (act) publish = hd[list.index(j)];
      } else if (step < gathers + 8) {
        const u32 stride = (kReduceThreads / 2) >> (step - gathers);
        act = tid < stride && tid + stride < list.count;
        if (act) src = tid + stride;
      } else {
        act = tid == 0;
        if (act) publish = bs[b];
      }
      tree[tid] = publish;
      __syncthreads();
      if (__ballot(act) != 0) { // (a wavefront with no active lane skips the addition, not the barriers)
        const point other = tree[src];
        const point out = C::add(part, other);
        if (act) part = out;
      }
      __syncthreads();
    }
    if (tid == 0) heavy_sum[h] = part;
    __syncthreads();
  }

  // ---- the bucket walk: one addition site, three roles ---------------------------------------------
  // per bucket (from the lane's last one down): v = the bucket's partials, s += v, r += s.  The
  // r += s of a bucket runs at the top of the NEXT iteration, while the loads of that iteration's
  // bucket are in flight (one extra iteration at the end finishes the last bucket).
  {
    const u32 lo = seg_first < nb ? (seg_first == 0 ? 0 : ends[seg_first - 1]) : 0;
    const u32 hi = seg_first < nb ? ends[seg_last - 1] : 0;
    populated = seg_first < nb && hi != lo;
    u32 end = hi;
    bool prev_valid = false;
    for (u32 it = 0; it <= lane_buckets; ++it) {
      const u32 b = seg_first + (lane_buckets - 1 - it); // (wraps in the extra iteration: unused)
      const bool valid = it != lane_buckets && populated && b < seg_last;
      if (__ballot(valid || prev_valid) == 0) continue; // nothing for this wavefront here
      const u32 begin = valid ? (b == 0 ? 0 : ends[b - 1]) : 0;
      const bool nonempty = valid && begin != end;
      point v = C::identity();
      u32 hs = 1, hlast = 0, after_whole = 0; // the head partials load_bucket would visit
      if (nonempty) {
        u32 folded = kReduceMaxHeavy;
        if (num_heavy != 0 && heads_of(begin, end, seg_log2).count > kReduceHeavyHeads) {
          for (u32 h = 0; h < num_heavy; ++h) folded = heavy_bucket[h] == b ? h : folded;
        }
        if (folded < kReduceMaxHeavy) {
          v = heavy_sum[folded];
        } else {
          v = bs[b];
          hs = (begin >> seg_log2) + 1;
          hlast = (end - 1) >> seg_log2;
          after_whole = end >> seg_log2;
        }
      }
      // phases, uniform over the wavefront: 2 (r += s of the previous bucket), then 0 (v += a head
      // partial, as long as some lane has one), then 1 (s += v)
      u32 phase = 2;
      for (;;) {
        bool act;
        point pa, pb;
        if (phase == 2) {
          act = prev_valid;
          pa = r;
          pb = s;
        } else if (phase == 0) {
          act = nonempty && hs <= hlast;
          if (__ballot(act) == 0) {
            phase = 1;
            continue;
          }
          pa = v;
          pb = act ? hd[hs] : v;
        } else {
          act = nonempty;
          pa = s;
          pb = v;
        }
        if (__ballot(act) != 0) {
          const point out = C::add(pa, pb);
          if (act) {
            if (phase == 2) {
              r = out;
            } else if (phase == 0) {
              v = out;
              if (hs < after_whole) {
                const u32 next_wave = (hs / 64 + 1) * 64;
                hs = next_wave < after_whole ? next_wave : after_whole;
              } else {
                ++hs;
              }
            } else {
              s = out;
            }
          }
        }
        if (phase == 2) {
          phase = 0;
        } else if (phase == 1) {
          break;
        }
      }
      prev_valid = valid;
      if (valid) end = begin;
    }
  }

  // ---- lane weights and the tree: one addition site --------------------------------------------------
  // every step as above: publish a point, add the point of slot `src` where active
  if constexpr (C::has_wave_add_multiple) {
    // steps 0..7: inclusive suffix scan of s over the lanes; step 8: v_t = r_t + 2^lane_log2 suffix_t
    // (t >= 1; the multiple is published for the lane itself); steps 9..16: the tree over v_t
    point val = s;
    point x = s;
    for (u32 step = 0; step < 17; ++step) {
      u32 src = tid;
      bool act;
      point publish = val;
      if (step < 8) {
        act = tid + (1u << step) < kReduceThreads;
        if (act) src = tid + (1u << step);
      } else if (step == 8) {
        x = val; // this lane's inclusive suffix sum (lane 0: the sum of the block's buckets)
        publish = C::dbl_n(val, static_cast<int>(lane_log2));
        val = r;
        act = tid != 0;
      } else {
        const u32 stride = (kReduceThreads / 2) >> (step - 9);
        act = tid < stride;
        if (act) src = tid + stride;
      }
      tree[tid] = publish;
      __syncthreads();
      if (__ballot(act) != 0) {
        const point other = tree[src];
        const point out = C::add(val, other);
        if (act) val = out;
      }
      __syncthreads();
    }
    if (block_first == 0) {
      if (tid == 0) *dst = val;
      return;
    }
    if (tid == 0) {
      tree[0] = val;
      tree[1] = x;
    }
    __syncthreads();
    if (tid < 64) {
      const point sum = C::wave_add_multiple(tree[0], tree[1], block_first);
      if (tid == 0) *dst = sum;
    }
  } else {
    // contrib_t = r_t + seg_first * s_t: seg_first by double-and-add from the top bit of the block's
    // last bucket index down (complete formulas: doubling or adding the identity is harmless), then
    // the tree.  Steps: `bits` x (m = 2 m; m += s where the bit is set) | contrib = r + m | 8 tree steps
    const u32 bits = block_end > 1 ? 32 - __builtin_clz(block_end - 1) : 0;
    point val = C::identity(); // m, then contrib
    for (u32 step = 0; step < bits + 1 + 8; ++step) {
      u32 src = tid;
      bool act;
      point publish = val;
      if (step < bits) {
        val = C::dbl_n(val, 1);
        publish = s;
        act = populated && ((seg_first >> (bits - 1 - step)) & 1u) != 0;
      } else if (step == bits) {
        // publish m, continue with r
        val = r;
        act = populated && seg_first != 0;
      } else {
        const u32 stride = (kReduceThreads / 2) >> (step - bits - 1);
        act = tid < stride;
        if (act) src = tid + stride;
      }
      tree[tid] = publish;
      __syncthreads();
      if (__ballot(act) != 0) {
        const point other = tree[src];
        const point out = C::add(val, other);
        if (act) val = out;
      }
      __syncthreads();
    }
    if (tid == 0) *dst = val;
  }
}

//--------------------------------------------------------------------------------------------------
// k_horner
//--------------------------------------------------------------------------------------------------
// One workgroup per column, windows [w_lo, w_hi) of it: fold the per-(window, block) partials into
// one sum per window (all windows concurrently, a power-of-two team of lanes per window), then the
// Horner recurrence  acc = 2^c acc + window[w]  from the top window of the range down -- on one
// wavefront (all lanes redundantly, or lane-cooperatively when the curve offers C::wave_horner).
// `first`: the range contains the column's top window (no incoming state); otherwise the chain
// continues from state[column].  `last`: the range ends at window 0: write the canonical encoding
// (or the raw projective point when `projective_out`); otherwise leave the chain value in
// state[column].  The whole column in one launch = first && last with the full range.
// T: lanes per workgroup.  256 lanes fold a column's windows fastest (a lone call's last stage); a
// launch of hundreds of columns takes 64 -- only the first wavefront walks the chain, and the other
// three of a 256-lane block would hold a CU's LDS and registers for the length of it (1024 columns:
// 0.67 ms at two resident blocks per CU).  W <= T windows.
template <class C, u32 T = kCombineThreads>
__global__ void __launch_bounds__(T)
    k_horner(u8* __restrict__ out, u32 out_stride, int projective_out,
             typename C::point* __restrict__ state, const typename C::point* __restrict__ partials,
             u32 partial_stride, const column_desc* __restrict__ columns,
             const task_desc* __restrict__ tasks, const u32* __restrict__ task_total, u32 w_lo_arg,
             u32 w_hi_arg, int first, int last, u32 reduce_block_log2) {
  using point = typename C::point;
  __shared__ point tree[T];
  __builtin_amdgcn_s_setprio(BZ_HORNER_PRIO); // one workgroup per column, possibly beside k_accumulate
  const column_desc col = columns[blockIdx.x];
  const u32 tid = threadIdx.x;
  u8* dst = out + static_cast<u64>(blockIdx.x) * out_stride;
  // windows = the column's tasks (ONE for a column on window tables: no chain at all)
  u32 w_hi = w_hi_arg < col.num_tasks ? w_hi_arg : col.num_tasks;
  const u32 w_lo = w_lo_arg < w_hi ? w_lo_arg : w_hi;
  if (first) {
    // windows above the highest populated one contribute nothing: start the chain below them
    // (a 64-bit value in a 32-byte column keeps 4 of its 17 windows: 48 doublings instead of 256)
    while (w_hi > w_lo) {
      if (task_total[col.first_task + w_hi - 1] != 0) break;
      --w_hi;
    }
  }
  const u32 W = w_hi - w_lo;
  if (W == 0) {
    // nothing to add in this range (empty column, or a range above the column's top window)
    if (tid == 0) {
      const point acc = first ? C::identity() : state[blockIdx.x];
      if (!last) {
        state[blockIdx.x] = acc;
      } else if (projective_out) {
        C::store_projective(dst, acc);
      } else {
        C::encode(dst, acc);
      }
    }
    return;
  }
  // lanes per window: largest power of two with W * team <= 256
  u32 team = 1;
  while (team * 2 * W <= T) team *= 2;
  const u32 w = tid / team;
  const u32 lane = tid % team;
  const u32 nb = 1u << (col.window_bits - 1);
  const u32 reduce_block = 1u << reduce_block_log2; // buckets per k_reduce block (lanes x 2^s)
  const u32 blocks = (nb + reduce_block - 1) / reduce_block;
  point sum = C::identity();
  if (w < W && lane < blocks) {
    const point* p = partials + static_cast<u64>(col.first_task + w_lo + w) * partial_stride;
    sum = p[lane];
    for (u32 blk = lane + team; blk < blocks; blk += team) sum = C::add(sum, p[blk]);
  }
  tree[tid] = sum;
  __syncthreads();
  for (u32 stride = team / 2; stride > 0; stride >>= 1) {
    if constexpr (C::has_coop_add) {
      // a level with at most one addition per quad of the workgroup: the four lanes of a DPP quad
      // share each addition (the latency of ~4 field products instead of 12; on the Weierstrass
      // curves the fold is a third of a lone k_horner)
      if (W * stride * 4 <= T) {
        const u32 a = tid >> 2;
        if (a < W * stride) {
          const u32 e = (a / stride) * team + a % stride;
          const point sum2 = C::add_coop4(tree[e], tree[e + stride], tid & 3);
          if ((tid & 3) == 0) tree[e] = sum2;
        }
        __syncthreads();
        continue;
      }
    }
    if (w < W && lane < stride) tree[tid] = C::add(tree[tid], tree[tid + stride]);
    __syncthreads();
  }
  // all lanes of the first wavefront run the chain: keeping the data in vector registers stops
  // hipcc from moving the multi-limb chain onto the scalar unit (it did: s_mul_hi_u32 chains with
  // hundreds of SGPR spills, ~3x slower than the VALU form)
  if (tid < 64) {
    point acc;
    if constexpr (C::has_wave_horner) {
      acc = C::wave_horner(first ? C::identity() : state[blockIdx.x], first == 0, tree, team, W,
                           col.window_bits);
    } else {
      u32 i = W;
      if (first) {
        acc = tree[(W - 1) * team];
        i = W - 1;
      } else {
        acc = state[blockIdx.x];
      }
      while (i-- > 0) {
        acc = C::dbl_n(acc, static_cast<int>(col.window_bits));
        acc = C::add(acc, tree[i * team]);
      }
    }
    // every lane of the wavefront holds the chain value
    if (!last) {
      if (tid == 0) state[blockIdx.x] = acc;
    } else if (projective_out) {
      if (tid == 0) C::store_projective(dst, acc);
    } else if constexpr (C::has_wave_encode) {
      C::wave_encode(dst, acc);
    } else {
      if (tid == 0) C::encode(dst, acc);
    }
  }
}
} // namespace bz
