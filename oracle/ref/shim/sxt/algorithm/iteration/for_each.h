// Oracle-build-only stand-in (TEST INFRASTRUCTURE) for the one reference header of the fixed-base
// closure that is CUDA syntax (`__global__`, `<<<>>>`: sxt/algorithm/iteration/for_each.h:32-55).
// The include path lists oracle/ref/shim before /root/reference, so the reference's pippenger2
// headers pick this file up instead.  Same interface (launch_for_each_kernel, for_each); the index
// functor simply runs for every index on the calling host thread, over the host memory the
// cuda_runtime.h stand-in hands out.  The reference's HOST entry points (mtxpp2::multiexponentiate,
// partition_product, reduce_products) never call it; it only has to parse.
#pragma once

#include <utility>

#include "sxt/algorithm/base/index_functor.h"
#include "sxt/base/device/stream.h"
#include "sxt/base/type/raw_stream.h"
#include "sxt/execution/async/future.h"
#include "sxt/execution/device/synchronization.h"

namespace sxt::algi {
template <algb::index_functor F>
void launch_for_each_kernel(bast::raw_stream_t /*stream*/, F f, unsigned n) noexcept {
  for (unsigned index = 0; index < n; ++index) f(n, index);
}

template <algb::index_functor F>
xena::future<> for_each(basdv::stream&& stream, F f, unsigned n) noexcept {
  launch_for_each_kernel(stream, f, n);
  return xendv::await_and_own_stream(std::move(stream));
}

template <algb::index_functor F> xena::future<> for_each(F f, unsigned n) noexcept {
  return for_each(basdv::stream{}, f, n);
}
} // namespace sxt::algi
