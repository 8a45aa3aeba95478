// Body shared by the four msm_<curve>.hip translation units.
#pragma once

#include "blitzar_amd/csrc/msm/dispatch.h"
#include "blitzar_amd/csrc/msm/engine.h"
#include "blitzar_amd/csrc/fixed/partition_table.h"
#include "blitzar_amd/csrc/fixed/partition_table_device.h"
#include "blitzar_amd/csrc/msm/host_backend.h"

namespace bz {

// projective handle generators -> addends
template <class C>
__global__ void __launch_bounds__(256)
    k_prepare_addends_projective(typename C::addend* __restrict__ addends,
                                 const typename C::api_projective* __restrict__ points, u64 n) {
  const u64 i = static_cast<u64>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  addends[i] = C::addend_from_api_projective(points, i);
}

// commitments[k] = canonical encoding of sum_r partials[r * num_outputs + k]: the fold of the
// per-rank partial results of a row-sharded MSM (SURVEY 8(e) way 2); one lane per output
template <class C, bool Encode = true>
__global__ void __launch_bounds__(64)
    k_fold_encode(u8* __restrict__ out, const typename C::api_projective* __restrict__ partials,
                  u32 num_partials, u32 num_outputs) {
  const u32 k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= num_outputs) return;
  typename C::point acc = C::identity();
  for (u32 r = 0; r < num_partials; ++r) {
    acc = C::add(acc, C::point_from_api_projective(partials, static_cast<u64>(r) * num_outputs + k));
  }
  if constexpr (Encode) {
    C::encode(out + static_cast<u64>(k) * C::output_size, acc);
  } else {
    C::store_projective(out + static_cast<u64>(k) * C::projective_size, acc);
  }
}

// out[i] = (i + 1) * base in the curve's C-ABI generator layout: large synthetic generator sets
// with known discrete logarithms (bench / full-size parity checks; the reference's
// generate_random_element costs a 255-bit scalar multiplication per point on the host)
template <class C>
__global__ void __launch_bounds__(64)
    k_generator_multiples(u8* __restrict__ out, const void* __restrict__ base_api, u64 n) {
  const u64 i = static_cast<u64>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const typename C::addend g = C::make_addend(base_api, 0);
  typename C::point acc = C::identity();
  const u64 k = i + 1;
  for (int bit = 63 - __builtin_clzll(k); bit >= 0; --bit) {
    acc = C::dbl_n(acc, 1);
    if ((k >> bit) & 1) C::accumulate(acc, g, false);
  }
  C::store_api_generator(out + i * C::api_generator_size, acc);
}

//--------------------------------------------------------------------------------------------------
// window tables of resident generator sets (plan.h: window_table): slice w = 2^(bits w) g_i as
// addends, built by a chain of `bits` doublings per slice on engine points and one shared-inversion
// normalisation per slice
//--------------------------------------------------------------------------------------------------
template <class R>
__global__ void __launch_bounds__(256)
    k_load_points(typename R::point* __restrict__ points, const void* __restrict__ source, u64 n,
                  int projective) {
  const u64 i = static_cast<u64>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  points[i] = projective ? R::point_from_api_projective(source, i)
                         : R::point_from_api_generator(source, i);
}

template <class R>
__global__ void __launch_bounds__(256)
    k_double_points(typename R::point* __restrict__ points, u64 n, int doublings) {
  const u64 i = static_cast<u64>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  points[i] = R::dbl_n(points[i], doublings);
}

// Per-call window tables (plan.h, choose_call_table): points[w * stride + i] = 2^(bits w) g_i for
// every slice in ONE launch -- a lane walks its generator's whole chain of doublings (the chain is
// what the build costs: one-wavefront blocks spread the few lanes of a short generator set over as
// many compute units as there are) -- rows between the set's end and the slice's are identities.
// rows up to which the chain runs with a wavefront per generator.  Measured on MI355X, 1024 columns x
// 256 / 1024 / 4096 rows, a lane per generator -> a wavefront per generator, ms per call
// (profiles/round6_ab_wave_chain.log): curve25519 1.04 -> 0.77 / 1.77 -> 1.50 / 4.89 -> 5.00, bn254
// 2.31 -> 1.69 / 3.92 -> 3.38 / 8.99 -> 9.33, bls12-381 4.79 -> 3.08 / 9.36 -> 7.60 / 21.5 -> 20.2: at 4096
// rows the lane form hides behind recode + sort anyway and 4096 chain wavefronts get in their way
// (recode 0.12 -> 0.24 ms), except on bls12-381, whose doubling is three times as long.
template <class R> constexpr u64 wave_chain_max_rows() {
  return sizeof(typename R::point) > 160 ? 4096 : 2048; // (bls12-381: 3 x 14 limbs = 168 bytes)
}
template <class R>
__global__ void __launch_bounds__(64)
    k_chain_points(typename R::point* __restrict__ points, const void* __restrict__ api_generators,
                   u64 n, u64 stride, u32 windows, int bits) {
  const u64 i = static_cast<u64>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= stride) return;
  typename R::point p = i < n ? R::point_from_api_generator(api_generators, i) : R::identity();
  for (u32 w = 0; w < windows; ++w) {
    points[static_cast<u64>(w) * stride + i] = p;
    if (w + 1 < windows && i < n) p = R::dbl_n(p, bits);
  }
}

// The same chain with ONE WAVEFRONT per generator (R::wave_chain: the point spread over the lanes,
// curve/ed16_wave.h, curve/sw_wave.h): for the short generator sets the regime is about -- a few
// thousand generators -- a lane per generator leaves the machine to 64 wavefronts that each walk
// ~256 dependent doublings alone (0.35 ms at 256 curve25519 generators, a third of the whole call).
template <class R>
__global__ void __launch_bounds__(64)
    k_chain_points_wave(typename R::point* __restrict__ points,
                        const void* __restrict__ api_generators, u64 n, u64 stride, u32 windows,
                        u32 bits) {
  const u64 i = blockIdx.x;
  if (i >= n) {
    if (threadIdx.x < windows) points[static_cast<u64>(threadIdx.x) * stride + i] = R::identity();
    for (u32 w = 64 + threadIdx.x; w < windows; w += 64) {
      points[static_cast<u64>(w) * stride + i] = R::identity();
    }
    return;
  }
  R::wave_chain(points + i, stride, R::point_from_api_generator(api_generators, i), windows, bits);
}

// addends[i] = the affine (Z = 1) addend of points[i]; a workgroup's points share one inversion
// (tree_products / tree_inverses, msm/kernels.h)
template <class R>
__global__ void __launch_bounds__(256)
    k_points_to_addends(typename R::addend* __restrict__ addends,
                        const typename R::point* __restrict__ points, u64 n) {
  using fe = typename R::batch_fe;
  constexpr u32 K = R::batch_points_per_lane;
  __shared__ fe tree[512];
  const u32 tid = threadIdx.x;
  const u64 base = static_cast<u64>(blockIdx.x) * 256 * K;
  fe z[K], prefix[K];
  bool is_identity[K];
  static_for<K>([&](auto jc) {
    constexpr u32 j = decltype(jc)::value;
    const u64 i = base + static_cast<u64>(j) * 256 + tid;
    is_identity[j] = false;
    z[j] = i < n ? R::batch_z(points[i], is_identity[j]) : R::batch_one();
    if constexpr (j == 0) {
      prefix[0] = z[0];
    } else {
      prefix[j] = R::batch_mul(prefix[j - 1], z[j]);
    }
  });
  tree[256 + tid] = prefix[K - 1];
  tree_products<R, 256>(tree, tid);
  if (tid < 64) {
    const fe inv = R::batch_wave_invert(tree[1]);
    if (tid == 0) tree[1] = inv;
  }
  tree_inverses<R, 256>(tree, tid);
  fe inv = tree[256 + tid];
  static_for<K>([&](auto jc) {
    constexpr u32 j = K - 1 - decltype(jc)::value;
    fe zinv = inv;
    if constexpr (j != 0) {
      zinv = R::batch_mul(inv, prefix[j - 1]);
      inv = R::batch_mul(inv, z[j]);
    }
    const u64 i = base + static_cast<u64>(j) * 256 + tid;
    if (i < n) addends[i] = R::addend_from_point(points[i], zinv, is_identity[j]);
  });
}

// R = the trait used against resident generator sets, H = the trait of the host backend (both C
// itself, except curve25519: the host backend keeps the projective cached addends -- one inversion
// per generator is only worth it where the inversions are batched on the device)
template <class C, class R = C, class H = C> struct curve_tu {
  static void msm_resident(msm_context& ctx, u8* d_out, u32 out_stride, bool projective_out,
                           const std::vector<host_column>& cols, const void* d_addends,
                           hipStream_t stream, const window_table* tables) {
    if (cols.empty()) return;
    std::lock_guard<std::mutex> lock(ctx.mu);
    configure_sort_kernels(ctx);
    ctx.order_after_previous(stream);
    // (a forced per-call table -- tests, A/B runs -- merges whatever the cost model says)
    const bool force = ctx.force_call_table_bits != 0 && tables != nullptr &&
                       d_addends == ctx.call_table_rows;
    msm_enqueue_locked<R>(ctx, d_out, out_stride, projective_out, cols,
                          static_cast<const typename R::addend*>(d_addends), nullptr, stream, tables,
                          force);
  }
  // slices 0 .. windows-1 of a resident set: d_table[w * stride + i] = addend of 2^(bits w) g_i
  // (blocking: scratch for the chain of points is allocated and freed here)
  static void build_window_table(void* d_table, const void* d_source, bool source_projective,
                                 u64 n, u64 stride, u32 windows, u32 bits, hipStream_t stream) {
    if (n == 0) return;
    using point = typename R::point;
    point* d_points = nullptr;
    BZ_HIP_CHECK(hipMalloc(&d_points, sizeof(point) * n));
    const u32 blocks = ceil_div_u32(n, 256);
    hipLaunchKernelGGL((k_load_points<R>), dim3(blocks), dim3(256), 0, stream, d_points, d_source,
                       n, source_projective ? 1 : 0);
    auto* table = static_cast<typename R::addend*>(d_table);
    for (u32 w = 0; w < windows; ++w) {
      if (w != 0) {
        hipLaunchKernelGGL((k_double_points<R>), dim3(blocks), dim3(256), 0, stream, d_points, n,
                           static_cast<int>(bits));
      }
      hipLaunchKernelGGL((k_points_to_addends<R>),
                         dim3(ceil_div_u32(n, 256ull * R::batch_points_per_lane)), dim3(256), 0,
                         stream, table + static_cast<u64>(w) * stride, d_points, n);
    }
    BZ_HIP_CHECK(hipGetLastError());
    g_kernel_launches += 1 + 2 * windows;
    BZ_HIP_CHECK(hipStreamSynchronize(stream));
    BZ_HIP_CHECK(hipFree(d_points));
  }
  static void prepare_resident(void* d_addends, const void* d_api_generators, u64 n,
                               hipStream_t stream) {
    launch_prepare_addends<R>(static_cast<typename R::addend*>(d_addends), d_api_generators, n,
                              stream);
    BZ_HIP_CHECK(hipGetLastError());
  }
  static void prepare_resident_projective(void* d_addends, const void* d_projective, u64 n,
                                          hipStream_t stream) {
    if (n == 0) return;
    hipLaunchKernelGGL((k_prepare_addends_projective<R>), dim3(ceil_div_u32(n, 256)), dim3(256), 0,
                       stream, static_cast<typename R::addend*>(d_addends),
                       static_cast<const typename R::api_projective*>(d_projective), n);
    BZ_HIP_CHECK(hipGetLastError());
  }
  // the 2^(bits w) multiples of a call's caller generators in the resident addend form, into the
  // context's table block; asynchronous (side stream beside the call's front, or the caller's)
  static const typename R::addend* build_call_table(msm_context& ctx, const void* d_api_generators,
                                                    u64 n, const window_table& t,
                                                    hipStream_t stream) {
    using point = typename R::point;
    using addend = typename R::addend;
    const u64 rows = t.stride * t.windows;
    ctx.call_table.reset(device_arena::padded(sizeof(addend) * (rows + 1)) +
                             device_arena::padded(sizeof(point) * (rows + 1)),
                         stream);
    addend* table = ctx.call_table.take<addend>(rows + 1);
    point* points = ctx.call_table.take<point>(rows + 1);
    hipStream_t bs = stream;
    if (ctx.table_overlap) {
      ctx.make_side_stream();
      ctx.table_fork.record(stream);
      ctx.table_fork.wait(ctx.side);
      bs = ctx.side;
    }
    if (t.stride <= wave_chain_max_rows<R>() && ctx.wave_chain) {
      hipLaunchKernelGGL((k_chain_points_wave<R>), dim3(static_cast<u32>(t.stride)), dim3(64), 0, bs,
                         points, d_api_generators, n, t.stride, t.windows, t.bits);
    } else {
      hipLaunchKernelGGL((k_chain_points<R>), dim3(ceil_div_u32(t.stride, 64)), dim3(64), 0, bs,
                         points, d_api_generators, n, t.stride, t.windows, static_cast<int>(t.bits));
    }
    hipLaunchKernelGGL((k_points_to_addends<R>),
                       dim3(ceil_div_u32(rows, 256ull * R::batch_points_per_lane)), dim3(256), 0, bs,
                       table, points, rows);
    BZ_HIP_CHECK(hipGetLastError());
    g_kernel_launches += 2;
    ctx.call_tables_built += 1;
    ctx.call_table_rows = table;
    if (ctx.table_overlap) {
      ctx.table_ready.record(bs);
      ctx.table_pending = true;
    }
    return table;
  }
  // the model's choice for `cols` and, if it wants a table, the enqueued build (ctx.mu held)
  static const typename R::addend* call_table_locked(msm_context& ctx,
                                                     const std::vector<host_column>& cols,
                                                     const void* d_api_generators,
                                                     window_table& shape, hipStream_t stream) {
    shape = window_table{};
    shape.windows = 0;
    if (!ctx.call_tables || d_api_generators == nullptr) return nullptr;
    const call_table_choice ch =
        choose_call_table(cols, ctx.tuning, sizeof(typename R::addend), R::call_table_entry_cost,
                          ctx.force_call_table_bits);
    if (ch.shape.windows == 0) return nullptr;
    u64 n = 0;
    for (const auto& c : cols) n = c.n > n ? c.n : n;
    shape = ch.shape;
    return build_call_table(ctx, d_api_generators, n, shape, stream);
  }
  static const void* call_table(msm_context& ctx, const std::vector<host_column>& cols,
                                const void* d_api_generators, window_table* shape,
                                hipStream_t stream) {
    std::lock_guard<std::mutex> lock(ctx.mu);
    ctx.order_after_previous(stream);
    const void* table = call_table_locked(ctx, cols, d_api_generators, *shape, stream);
    if (table != nullptr) ctx.mark_enqueued(stream);
    return table;
  }
  static void msm(msm_context& ctx, u8* d_out, u32 out_stride, bool projective_out,
                  const std::vector<host_column>& cols, const void* d_addends,
                  const void* d_api_generators, hipStream_t stream) {
    if (cols.empty()) return;
    std::lock_guard<std::mutex> lock(ctx.mu);
    configure_sort_kernels(ctx);
    ctx.order_after_previous(stream);
    // many columns over the same caller generators (the reference's bucket_method2 regime): the
    // window table of the generators is built in the call and every column becomes one task
    if (d_addends == nullptr) {
      window_table shape;
      const typename R::addend* table = call_table_locked(ctx, cols, d_api_generators, shape, stream);
      if (table != nullptr) {
        msm_enqueue_locked<R>(ctx, d_out, out_stride, projective_out, cols, table, nullptr, stream,
                              &shape, ctx.force_call_table_bits != 0);
        return;
      }
    }
    if constexpr (!std::is_same_v<C, R> && R::has_batched_prepare) {
      if (d_addends == nullptr && ctx.normalise_caller) {
        msm_enqueue_locked<R>(ctx, d_out, out_stride, projective_out, cols, nullptr,
                              d_api_generators, stream);
        return;
      }
    }
    msm_enqueue_locked<C>(ctx, d_out, out_stride, projective_out, cols,
                          static_cast<const typename C::addend*>(d_addends), d_api_generators, stream);
  }
  static void prepare_addends(void* d_addends, const void* d_api_generators, u64 n,
                              hipStream_t stream) {
    launch_prepare_addends<C>(static_cast<typename C::addend*>(d_addends), d_api_generators, n,
                              stream);
    BZ_HIP_CHECK(hipGetLastError());
  }
  static void prepare_addends_projective(void* d_addends, const void* d_projective, u64 n,
                                         hipStream_t stream) {
    if (n == 0) return;
    hipLaunchKernelGGL((k_prepare_addends_projective<C>), dim3(ceil_div_u32(n, 256)), dim3(256), 0,
                       stream, static_cast<typename C::addend*>(d_addends),
                       static_cast<const typename C::api_projective*>(d_projective), n);
    BZ_HIP_CHECK(hipGetLastError());
  }
  static void msm_host_entry(u8* out, u32 out_stride, bool projective_out,
                             const std::vector<host_column>& cols, const void* generators,
                             bool generators_projective, u64 num_generators) {
    std::vector<typename H::addend> addends(num_generators);
    for (u64 i = 0; i < num_generators; ++i) {
      addends[i] = generators_projective
                       ? H::addend_from_api_projective(generators, i)
                       : H::make_addend(generators, i);
    }
    msm_host<H>(out, out_stride, projective_out, cols, addends.data());
  }
  static void fold_encode_host(u8* out, const void* partials, u32 num_partials, u32 num_outputs) {
    for (u32 k = 0; k < num_outputs; ++k) {
      typename C::point acc = C::identity();
      for (u32 r = 0; r < num_partials; ++r) {
        acc = C::add(acc,
                     C::point_from_api_projective(partials, static_cast<u64>(r) * num_outputs + k));
      }
      C::encode(out + static_cast<u64>(k) * C::output_size, acc);
    }
  }
  // out[k] = sum_j 2^(shift_bits j) * pieces[first_k + j] (projective elements in, projective
  // out): an output of a fixed-base call that is wider than a scalar was computed in pieces of
  // `shift_bits` bits (api/capi.hip); Horner from the top piece down
  static void fold_shifted_host(u8* out, const void* pieces, const u32* piece_counts,
                                u32 num_outputs, u32 shift_bits) {
    u64 first = 0;
    for (u32 k = 0; k < num_outputs; ++k) {
      typename C::point acc = C::identity();
      for (u32 j = piece_counts[k]; j-- > 0;) {
        if (j + 1 != piece_counts[k]) acc = C::dbl_n(acc, static_cast<int>(shift_bits));
        acc = C::add(acc, C::point_from_api_projective(pieces, first + j));
      }
      C::store_projective(out + static_cast<u64>(k) * C::projective_size, acc);
      first += piece_counts[k];
    }
  }
  static void fold_encode_device(u8* d_out, const void* d_partials, u32 num_partials,
                                 u32 num_outputs, hipStream_t stream) {
    if (num_outputs == 0) return;
    hipLaunchKernelGGL((k_fold_encode<C>), dim3(ceil_div_u32(num_outputs, 64)), dim3(64), 0, stream,
                       d_out, static_cast<const typename C::api_projective*>(d_partials),
                       num_partials,
                       num_outputs);
    BZ_HIP_CHECK(hipGetLastError());
  }
  static void fold_device(u8* d_out, const void* d_partials, u32 num_partials, u32 num_outputs,
                          hipStream_t stream) {
    if (num_outputs == 0) return;
    hipLaunchKernelGGL((k_fold_encode<C, false>), dim3(ceil_div_u32(num_outputs, 64)), dim3(64), 0,
                       stream, d_out,
                       static_cast<const typename C::api_projective*>(d_partials), num_partials,
                       num_outputs);
    BZ_HIP_CHECK(hipGetLastError());
  }
  static void generator_multiples(void* d_out, const void* d_base_api, u64 n, hipStream_t stream) {
    if (n == 0) return;
    hipLaunchKernelGGL((k_generator_multiples<C>), dim3(ceil_div_u32(n, 64)), dim3(64), 0, stream,
                       static_cast<u8*>(d_out), d_base_api, n);
    BZ_HIP_CHECK(hipGetLastError());
  }
  static const curve_vtable& vtable() {
    static const curve_vtable vt{C::curve_id,
                                 C::api_generator_size,
                                 sizeof(typename C::addend),
                                 C::output_size,
                                 C::projective_size,
                                 &curve_tu::msm,
                                 &curve_tu::prepare_addends,
                                 &curve_tu::prepare_addends_projective,
                                 &curve_tu::msm_host_entry,
                                 &curve_tu::fold_encode_host,
                                 &curve_tu::fold_encode_device,
                                 &curve_tu::fold_device,
                                 &curve_tu::fold_shifted_host,
                                 &curve_tu::generator_multiples,
                                 sizeof(typename R::addend),
                                 &curve_tu::msm_resident,
                                 &curve_tu::prepare_resident,
                                 &curve_tu::prepare_resident_projective,
                                 &curve_tu::build_window_table,
                                 &curve_tu::call_table,
                                 sizeof(typename compact_ops<H>::compact),
                                 &write_partition_table<H>,
                                 &write_partition_table_device<H>,
                                 &read_partition_generators<H>,
                                 &write_compact_generators<H>,
                                 C::reference_element_name,
                                 C::reference_compact_name};
    return vt;
  }
};
} // namespace bz
