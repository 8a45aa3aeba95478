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

// R = the trait used against resident generator sets, H = the trait of the host backend (both C
// itself, except curve25519: the host backend keeps the projective cached addends -- one inversion
// per generator is only worth it where the inversions are batched on the device)
template <class C, class R = C, class H = C> struct curve_tu {
  static void msm_resident(msm_context& ctx, u8* d_out, u32 out_stride, bool projective_out,
                           const std::vector<host_column>& cols, const void* d_addends,
                           hipStream_t stream) {
    msm_enqueue<R>(ctx, d_out, out_stride, projective_out, cols,
                   static_cast<const typename R::addend*>(d_addends), nullptr, stream);
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
  static void msm(msm_context& ctx, u8* d_out, u32 out_stride, bool projective_out,
                  const std::vector<host_column>& cols, const void* d_addends,
                  const void* d_api_generators, hipStream_t stream) {
    msm_enqueue<C>(ctx, d_out, out_stride, projective_out, cols,
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
                                 &curve_tu::generator_multiples,
                                 sizeof(typename R::addend),
                                 &curve_tu::msm_resident,
                                 &curve_tu::prepare_resident,
                                 &curve_tu::prepare_resident_projective,
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
