// GPU construction of the reference's fixed-base partition table for
// sxt_multiexp_handle_write_to_file (reference: mtxpp2::compute_partition_table, a serial host
// loop, sxt/multiexp/pippenger2/partition_table.h:36-98).
//
// A window of w generators has 2^w subset sums.  Split the mask m = (hi, lo) into two halves of
// at most 8 bits: the two half tables (<= 256 sums each) are built per window by popcount levels,
// and every entry of the slice is ONE complete addition hi_table[hi] + lo_table[lo] -- 2^w
// independent additions instead of a 2^w-long dependency chain.  Entries are stored affine, which
// needs an inversion each; they are shared two levels deep with Montgomery's trick: a workgroup
// multiplies the Z's of its 2048 entries up a product tree (k_table_sums), ONE small launch
// inverts the roots of up to 2048 workgroups the same way with a single field inversion
// (k_table_invert_roots), and k_table_emit walks back down to every entry's 1 / Z and writes the
// compact element.  A 2^30-entry table (2^14 generators, w = 16) costs 256 field inversions in all;
// the host loop it replaces did 2^30 of them.
//
// Byte-exactness: Weierstrass compact entries are canonical Montgomery residues.  curve25519
// entries are radix-2^51 limbs; the limbs the reference's field products leave are the canonical
// digits of the value whatever the operation order (checked against the reference's own tables in
// tests/), so canonical digits are what is written -- except for the value 0, which also fits
// below 2^255 as p: the identity-valued entries of padded windows are patched on the host
// (patch_identity_submasks).
#pragma once

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <vector>

#include "blitzar_amd/csrc/base/device.h"
#include "blitzar_amd/csrc/fixed/partition_table.h"
#include "blitzar_amd/csrc/msm/kernels.h"

namespace bz {

constexpr u32 kTableThreads = 256;
constexpr u32 kTablePoints = 8;                              // entries per lane
constexpr u32 kTableChunk = kTableThreads * kTablePoints;    // entries per workgroup
constexpr u32 kTableHalfBits = 8;                            // 2^8 sums per half table
constexpr u64 kTableBatchEntries = u64{kTableChunk} * kTableChunk; // roots of one batch fit one chunk

// halves[window][0 .. 255] = sums over the low min(w, 8) generators of the window,
// halves[window][256 .. 511] = sums over the remaining w - 8 (one workgroup per window)
template <class C>
__global__ void __launch_bounds__(kTableThreads)
    k_table_halves(typename C::point* __restrict__ halves,
                   const typename C::api_projective* __restrict__ generators, u64 n,
                   u64 first_window, u32 w) {
  using point = typename C::point;
  __shared__ point g[2 * kTableHalfBits];
  const u32 tid = threadIdx.x;
  if (tid < w) {
    const u64 idx = (first_window + blockIdx.x) * w + tid;
    g[tid] = idx < n ? C::point_from_api_projective(generators, idx) : C::identity();
  }
  __syncthreads();
  const u32 w_lo = w < kTableHalfBits ? w : kTableHalfBits, w_hi = w - w_lo;
  point* lo = halves + static_cast<u64>(blockIdx.x) * 2 * kTableThreads;
  point* hi = lo + kTableThreads;
  if (tid == 0) {
    lo[0] = C::identity();
    hi[0] = C::identity();
  }
  // popcount levels: entry m = entry[m without its lowest set bit] + generator[lowest set bit]
  for (u32 level = 1; level <= kTableHalfBits; ++level) {
    if (tid != 0 && static_cast<u32>(__popc(tid)) == level) {
      const u32 low = static_cast<u32>(__ffs(static_cast<int>(tid))) - 1, rest = tid & (tid - 1);
      if (tid < (1u << w_lo)) lo[tid] = level == 1 ? g[low] : C::add(lo[rest], g[low]);
      if (tid < (1u << w_hi)) hi[tid] = level == 1 ? g[w_lo + low] : C::add(hi[rest], g[w_lo + low]);
    }
    __syncthreads(); // also orders the workgroup's global writes before the next level's reads
  }
}

// this lane's entries of chunk `chunk`: e = chunk * kTableChunk + j * kTableThreads + tid
template <class C, class F> __device__ __forceinline__ void for_each_table_entry(u32 chunk, F&& f) {
  static_for<kTablePoints>([&](auto jc) {
    constexpr u32 j = decltype(jc)::value;
    f(jc, static_cast<u64>(chunk) * kTableChunk + j * kTableThreads + threadIdx.x);
  });
}

// sums[e] = hi_table[m >> 8] + lo_table[m & 255] for entry e = (window, m) of the batch;
// roots[chunk] = product of the chunk's Z coordinates (identity entries count as 1)
template <class C>
__global__ void __launch_bounds__(kTableThreads)
    k_table_sums(typename C::point* __restrict__ sums, typename C::batch_fe* __restrict__ roots,
                 const typename C::point* __restrict__ halves, u64 entries, u32 w) {
  using fe = typename C::batch_fe;
  __shared__ fe tree[2 * kTableThreads];
  const u32 w_lo = w < kTableHalfBits ? w : kTableHalfBits;
  fe product = C::batch_one();
  for_each_table_entry<C>(blockIdx.x, [&](auto, u64 e) {
    if (e >= entries) return;
    const u64 window = e >> w;
    const u32 m = static_cast<u32>(e & ((u64{1} << w) - 1));
    const typename C::point* lo = halves + window * 2 * kTableThreads;
    const typename C::point p = C::add(lo[kTableThreads + (m >> w_lo)], lo[m & ((1u << w_lo) - 1)]);
    sums[e] = p;
    bool is_identity;
    product = C::batch_mul(product, C::batch_z(p, is_identity));
  });
  tree[kTableThreads + threadIdx.x] = product;
  tree_products<C, kTableThreads>(tree, threadIdx.x);
  if (threadIdx.x == 0) roots[blockIdx.x] = tree[1];
}

// roots[i] <- 1 / roots[i] for i < count <= kTableChunk: one workgroup, one field inversion
template <class C>
__global__ void __launch_bounds__(kTableThreads)
    k_table_invert_roots(typename C::batch_fe* __restrict__ roots, u32 count) {
  using fe = typename C::batch_fe;
  __shared__ fe tree[2 * kTableThreads];
  const u32 tid = threadIdx.x;
  fe z[kTablePoints], prefix[kTablePoints];
  static_for<kTablePoints>([&](auto jc) {
    constexpr u32 j = decltype(jc)::value;
    const u32 i = j * kTableThreads + tid;
    z[j] = i < count ? roots[i] : C::batch_one();
    if constexpr (j == 0) {
      prefix[0] = z[0];
    } else {
      prefix[j] = C::batch_mul(prefix[j - 1], z[j]);
    }
  });
  tree[kTableThreads + tid] = prefix[kTablePoints - 1];
  tree_products<C, kTableThreads>(tree, tid);
  if (tid < 64) {
    const fe inv = C::batch_wave_invert(tree[1]);
    if (tid == 0) tree[1] = inv;
  }
  tree_inverses<C, kTableThreads>(tree, tid);
  fe inv = tree[kTableThreads + tid];
  static_for<kTablePoints>([&](auto jc) {
    constexpr u32 j = kTablePoints - 1 - decltype(jc)::value;
    fe zinv = inv;
    if constexpr (j != 0) {
      zinv = C::batch_mul(inv, prefix[j - 1]);
      inv = C::batch_mul(inv, z[j]);
    }
    const u32 i = j * kTableThreads + tid;
    if (i < count) roots[i] = zinv;
  });
}

// out[e] = compact(sums[e]): the chunk's product tree is rebuilt from the stored Z's, its root's
// inverse comes from k_table_invert_roots, and the inverses travel down to the entries
template <class C>
__global__ void __launch_bounds__(kTableThreads)
    k_table_emit(u8* __restrict__ out, const typename C::point* __restrict__ sums,
                 const typename C::batch_fe* __restrict__ inverse_roots, u64 entries) {
  using fe = typename C::batch_fe;
  __shared__ fe tree[2 * kTableThreads];
  const u32 tid = threadIdx.x;
  fe z[kTablePoints], prefix[kTablePoints];
  bool is_identity[kTablePoints];
  for_each_table_entry<C>(blockIdx.x, [&](auto jc, u64 e) {
    constexpr u32 j = decltype(jc)::value;
    is_identity[j] = false;
    z[j] = e < entries ? C::batch_z(sums[e], is_identity[j]) : C::batch_one();
    if constexpr (j == 0) {
      prefix[0] = z[0];
    } else {
      prefix[j] = C::batch_mul(prefix[j - 1], z[j]);
    }
  });
  tree[kTableThreads + tid] = prefix[kTablePoints - 1];
  tree_products<C, kTableThreads>(tree, tid);
  if (tid == 0) tree[1] = inverse_roots[blockIdx.x];
  tree_inverses<C, kTableThreads>(tree, tid);
  fe inv = tree[kTableThreads + tid];
  static_for<kTablePoints>([&](auto jc) {
    constexpr u32 j = kTablePoints - 1 - decltype(jc)::value;
    fe zinv = inv;
    if constexpr (j != 0) {
      zinv = C::batch_mul(inv, prefix[j - 1]);
      inv = C::batch_mul(inv, z[j]);
    }
    const u64 e = static_cast<u64>(blockIdx.x) * kTableChunk + j * kTableThreads + tid;
    if (e < entries) C::store_compact(out + e * C::compact_size, sums[e], zinv, is_identity[j]);
  });
}

// Entries whose mask only selects identity generators (the padding of the last window): their
// value is the identity, whose x = 0 and t = 0 have TWO limb patterns below 2^255 (0 and p), and
// which one the reference's recurrence leaves depends on its operation order
// (partition_table.h:52-66 through add_inplace).  These few entries are recomputed on the host
// with that exact recurrence; submasks of the identity set only depend on each other.
template <class C>
void patch_identity_submasks(u8* slice, unsigned w, const typename compact_ops<C>::point* gens) {
  using ops = compact_ops<C>;
  using compact = typename ops::compact;
  const compact ident = ops::shrink(ops::identity());
  u32 identity_set = 0;
  for (unsigned i = 0; i < w; ++i) {
    const compact c = ops::shrink(gens[i]);
    if (std::memcmp(&c, &ident, sizeof(c)) == 0) identity_set |= 1u << i;
  }
  if ((identity_set & (identity_set - 1)) == 0) return; // fewer than two identity generators
  compact* entries = reinterpret_cast<compact*>(slice);
  // ascending submasks of identity_set: every dependency (rest, lowest bit) comes first
  for (u32 m = 1; m < (1u << w); ++m) {
    if ((m & ~identity_set) != 0) continue;
    const u32 rest = m & (m - 1);
    if (rest == 0) continue; // single generators are direct conversions
    entries[m] = ops::shrink(ops::add(ops::expand(entries[rest]), ops::expand(entries[m ^ rest])));
  }
}

// The whole table of `n` projective generators (host, ABI layout) at window width w <= 16, written
// to `f` after the 4-byte width; the device is the current one.  Returns false on a short write.
template <class C>
bool write_partition_table_device(std::FILE* f, unsigned w, const void* projective_generators,
                                  u64 n, hipStream_t stream) {
  using point = typename C::point;
  using fe = typename C::batch_fe;
  const u32 w32 = w;
  if (std::fwrite(&w32, sizeof(w32), 1, f) != 1) return false;
  const u64 windows = (n + w - 1) / w;
  if (windows == 0) return true;
  const u64 per_window = u64{1} << w;
  const u64 batch_windows = std::max<u64>(1, std::min<u64>(windows, kTableBatchEntries >> w));
  const u64 batch_entries = batch_windows * per_window;
  const u32 batch_chunks = ceil_div_u32(batch_entries, kTableChunk);
  typename C::api_projective* d_gens = nullptr;
  point* d_halves = nullptr;
  point* d_sums = nullptr;
  fe* d_roots = nullptr;
  u8* d_out = nullptr;
  BZ_HIP_CHECK(hipMalloc(&d_gens, sizeof(typename C::api_projective) * n));
  BZ_HIP_CHECK(hipMalloc(&d_halves, sizeof(point) * batch_windows * 2 * kTableThreads));
  BZ_HIP_CHECK(hipMalloc(&d_sums, sizeof(point) * batch_entries));
  BZ_HIP_CHECK(hipMalloc(&d_roots, sizeof(fe) * batch_chunks));
  BZ_HIP_CHECK(hipMalloc(&d_out, C::compact_size * batch_entries));
  BZ_HIP_CHECK(hipMemcpyAsync(d_gens, projective_generators,
                              sizeof(typename C::api_projective) * n, hipMemcpyHostToDevice,
                              stream));
  std::vector<u8> host(C::compact_size * batch_entries);
  bool ok = true;
  for (u64 first = 0; first < windows && ok; first += batch_windows) {
    const u64 count = std::min<u64>(batch_windows, windows - first);
    const u64 entries = count * per_window;
    const u32 chunks = ceil_div_u32(entries, kTableChunk);
    hipLaunchKernelGGL((k_table_halves<C>), dim3(static_cast<u32>(count)), dim3(kTableThreads), 0,
                       stream, d_halves, d_gens, n, first, w);
    hipLaunchKernelGGL((k_table_sums<C>), dim3(chunks), dim3(kTableThreads), 0, stream, d_sums,
                       d_roots, d_halves, entries, w);
    hipLaunchKernelGGL((k_table_invert_roots<C>), dim3(1), dim3(kTableThreads), 0, stream, d_roots,
                       chunks);
    hipLaunchKernelGGL((k_table_emit<C>), dim3(chunks), dim3(kTableThreads), 0, stream, d_out,
                       d_sums, d_roots, entries);
    BZ_HIP_CHECK(hipGetLastError());
    g_kernel_launches += 4;
    BZ_HIP_CHECK(hipMemcpyAsync(host.data(), d_out, C::compact_size * entries,
                                hipMemcpyDeviceToHost, stream));
    BZ_HIP_CHECK(hipStreamSynchronize(stream));
    for (u64 k = 0; k < count; ++k) {
      std::vector<typename compact_ops<C>::point> slice(w, compact_ops<C>::identity());
      const auto* g = static_cast<const typename compact_ops<C>::point*>(projective_generators);
      for (unsigned i = 0; i < w; ++i) {
        const u64 idx = (first + k) * w + i;
        if (idx < n) slice[i] = g[idx];
      }
      patch_identity_submasks<C>(host.data() + k * per_window * C::compact_size, w, slice.data());
    }
    ok = std::fwrite(host.data(), C::compact_size, entries, f) == entries;
  }
  BZ_HIP_CHECK(hipFree(d_out));
  BZ_HIP_CHECK(hipFree(d_roots));
  BZ_HIP_CHECK(hipFree(d_sums));
  BZ_HIP_CHECK(hipFree(d_halves));
  BZ_HIP_CHECK(hipFree(d_gens));
  return ok;
}
} // namespace bz
