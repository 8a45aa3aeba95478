#include "blitzar_amd/csrc/generators/builtin.h"

namespace bz {
namespace {
__global__ void __launch_bounds__(64) k_base_elements(ed_point* __restrict__ out, u64 first, u64 n) {
  const u64 i = static_cast<u64>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  out[i] = ed::base_element(first + i);
}

} // namespace

void builtin_generators_enqueue(ed_point* d_out, u64 first, u64 n, hipStream_t stream) {
  if (n == 0) return;
  hipLaunchKernelGGL(k_base_elements, dim3(static_cast<unsigned>((n + 63) / 64)), dim3(64), 0,
                     stream, d_out, first, n);
  BZ_HIP_CHECK(hipGetLastError());
}

} // namespace bz
