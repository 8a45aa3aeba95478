// Body shared by the four msm_<curve>.hip translation units.
#pragma once

#include "blitzar_amd/csrc/msm/dispatch.h"
#include "blitzar_amd/csrc/msm/engine.h"
#include "blitzar_amd/csrc/fixed/partition_table.h"
#include "blitzar_amd/csrc/msm/host_backend.h"

namespace bz {

// projective handle generators -> addends
template <class C>
__global__ void __launch_bounds__(256)
    k_prepare_addends_projective(typename C::addend* __restrict__ addends,
                                 const typename C::point* __restrict__ points, u64 n) {
  const u64 i = static_cast<u64>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  addends[i] = C::addend_from_point(points[i]);
}

template <class C> struct curve_tu {
  static void msm(msm_context& ctx, u8* d_out, u32 out_stride, bool projective_out,
                  const std::vector<host_column>& cols, const void* d_addends,
                  const void* d_api_generators, hipStream_t stream) {
    msm_enqueue<C>(ctx, d_out, out_stride, projective_out, cols,
                   static_cast<const typename C::addend*>(d_addends), d_api_generators, stream);
  }
  static void prepare_addends(void* d_addends, const void* d_api_generators, u64 n,
                              hipStream_t stream) {
    if (n == 0) return;
    hipLaunchKernelGGL((k_prepare_addends<C>), dim3(ceil_div_u32(n, 256)), dim3(256), 0, stream,
                       static_cast<typename C::addend*>(d_addends), d_api_generators, n);
    BZ_HIP_CHECK(hipGetLastError());
  }
  static void prepare_addends_projective(void* d_addends, const void* d_projective, u64 n,
                                         hipStream_t stream) {
    if (n == 0) return;
    hipLaunchKernelGGL((k_prepare_addends_projective<C>), dim3(ceil_div_u32(n, 256)), dim3(256), 0,
                       stream, static_cast<typename C::addend*>(d_addends),
                       static_cast<const typename C::point*>(d_projective), n);
    BZ_HIP_CHECK(hipGetLastError());
  }
  static void msm_host_entry(u8* out, u32 out_stride, bool projective_out,
                             const std::vector<host_column>& cols, const void* generators,
                             bool generators_projective, u64 num_generators) {
    std::vector<typename C::addend> addends(num_generators);
    for (u64 i = 0; i < num_generators; ++i) {
      addends[i] = generators_projective
                       ? C::addend_from_point(static_cast<const typename C::point*>(generators)[i])
                       : C::make_addend(generators, i);
    }
    msm_host<C>(out, out_stride, projective_out, cols, addends.data());
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
                                 sizeof(typename compact_ops<C>::compact),
                                 &write_partition_table<C>,
                                 &read_partition_generators<C>};
    return vt;
  }
};
} // namespace bz
