// TEST INFRASTRUCTURE ONLY -- never linked into or called from the product (blitzar_amd/).
//
// extern "C" driver over the reference's OWN fixed-base ("pippenger2") host path, compiled from the
// sources where they lie under /root/reference:
//   mtxpp2::make_in_memory_partition_table_accessor   in_memory_partition_table_accessor_utility.h:41-93
//   mtxpp2::in_memory_partition_table_accessor<U>      in_memory_partition_table_accessor.h:38-111
//   mtxpp2::multiexponentiate (host overloads)          multiexponentiation.h:239-288,
//                                                        variable_length_multiexponentiation.h:197-232
//   -> partition_product (partition_product.h:167-194), reduce_products (reduce.h:120-146)
// The calls below are the ones cpu_backend makes (sxt/cbindings/backend/cpu_backend.cc:196-262:
// make_partition_table_accessor, the three fixed_multiexponentiation overloads, and
// make_scalars_span of computational_backend_utility.cc:28-52, whose TU is linked as it stands);
// nothing of the algorithm is restated here.  Those headers also hold the CUDA variants of the same
// functions, so two stand-ins make them parse on the host (oracle/ref/shim): cuda_runtime.h (host
// memory, synchronous streams) and sxt/algorithm/iteration/for_each.h (the one `<<<>>>` header).
//
// Built by oracle/ref/build_ref.py into oracle/_ref/libblitzar_ref.so.
#include <cstdint>
#include <memory>
#include <numeric>
#include <string>

#include "sxt/base/container/span.h"
#include "sxt/base/memory/alloc.h"
#include "sxt/base/num/divide_up.h"
#include "sxt/cbindings/backend/computational_backend_utility.h"
#include "sxt/execution/schedule/scheduler.h"
#include "sxt/curve21/operation/add.h"
#include "sxt/curve21/operation/double.h"
#include "sxt/curve21/operation/neg.h"
#include "sxt/curve21/type/compact_element.h"
#include "sxt/curve21/type/element_p3.h"
#include "sxt/curve_bng1/operation/add.h"
#include "sxt/curve_bng1/operation/double.h"
#include "sxt/curve_bng1/operation/neg.h"
#include "sxt/curve_bng1/type/compact_element.h"
#include "sxt/curve_bng1/type/element_p2.h"
#include "sxt/curve_g1/operation/add.h"
#include "sxt/curve_g1/operation/double.h"
#include "sxt/curve_g1/operation/neg.h"
#include "sxt/curve_g1/type/compact_element.h"
#include "sxt/curve_g1/type/element_p2.h"
#include "sxt/curve_gk/operation/add.h"
#include "sxt/curve_gk/operation/double.h"
#include "sxt/curve_gk/operation/neg.h"
#include "sxt/curve_gk/type/compact_element.h"
#include "sxt/curve_gk/type/element_p2.h"
#include "sxt/multiexp/pippenger2/in_memory_partition_table_accessor.h"
#include "sxt/multiexp/pippenger2/in_memory_partition_table_accessor_utility.h"
#include "sxt/multiexp/pippenger2/multiexponentiation.h"
#include "sxt/multiexp/pippenger2/multiexponentiation_serialization.h"
#include "sxt/multiexp/pippenger2/partition_table_accessor_base.h"
#include "sxt/multiexp/pippenger2/variable_length_multiexponentiation.h"

using namespace sxt;

namespace {
struct fixed_handle {
  unsigned curve_id;
  std::unique_ptr<mtxpp2::partition_table_accessor_base> accessor;
};

// the (compact table entry, projective element) pair of a curve id, as
// sxt/cbindings/base/curve_id_utility.h:44-61 (that header panics through <format>; the switch is
// four lines)
template <class F> void with_curve(unsigned curve_id, F f) {
  switch (curve_id) {
  case 0:
    f(std::type_identity<c21t::compact_element>{}, std::type_identity<c21t::element_p3>{});
    break;
  case 1:
    f(std::type_identity<cg1t::compact_element>{}, std::type_identity<cg1t::element_p2>{});
    break;
  case 2:
    f(std::type_identity<cn1t::compact_element>{}, std::type_identity<cn1t::element_p2>{});
    break;
  default:
    f(std::type_identity<cgkt::compact_element>{}, std::type_identity<cgkt::element_p2>{});
    break;
  }
}
} // namespace

extern "C" {
// cpu_backend::make_partition_table_accessor (cpu_backend.cc:196-211) with an explicit window width
void* ref_fixed_handle_new(unsigned curve_id, const void* generators, unsigned n,
                           unsigned window_width) {
  auto h = new fixed_handle{curve_id, nullptr};
  with_curve(curve_id, [&]<class U, class T>(std::type_identity<U>, std::type_identity<T>) {
    h->accessor = mtxpp2::make_in_memory_partition_table_accessor<U, T>(
        basct::cspan<T>{static_cast<const T*>(generators), n}, basm::alloc_t{}, window_width);
  });
  return h;
}

// cpu_backend::read_partition_table_accessor (cpu_backend.cc:267-277)
void* ref_fixed_handle_from_file(unsigned curve_id, const char* filename) {
  auto h = new fixed_handle{curve_id, nullptr};
  with_curve(curve_id, [&]<class U, class T>(std::type_identity<U>, std::type_identity<T>) {
    h->accessor = std::make_unique<mtxpp2::in_memory_partition_table_accessor<U>>(
        std::string{filename}, basm::alloc_t{});
  });
  return h;
}

void ref_fixed_handle_write(const void* handle, const char* filename) {
  static_cast<const fixed_handle*>(handle)->accessor->write_to_file(filename);
}

void ref_fixed_handle_free(void* handle) { delete static_cast<fixed_handle*>(handle); }

// cpu_backend::fixed_multiexponentiation, byte-aligned outputs (cpu_backend.cc:216-228)
void ref_fixed_multiexponentiation(void* res, const void* handle, unsigned element_num_bytes,
                                   unsigned num_outputs, unsigned n, const uint8_t* scalars) {
  const auto& h = *static_cast<const fixed_handle*>(handle);
  with_curve(h.curve_id, [&]<class U, class T>(std::type_identity<U>, std::type_identity<T>) {
    basct::span<T> res_span{static_cast<T*>(res), num_outputs};
    basct::cspan<uint8_t> scalars_span{scalars,
                                       static_cast<size_t>(element_num_bytes) * num_outputs * n};
    mtxpp2::multiexponentiate<T>(
        res_span, static_cast<const mtxpp2::partition_table_accessor<U>&>(*h.accessor),
        element_num_bytes, scalars_span);
  });
}

// ... packed outputs (cpu_backend.cc:230-245)
void ref_fixed_packed_multiexponentiation(void* res, const void* handle,
                                          const unsigned* output_bit_table, unsigned num_outputs,
                                          unsigned n, const uint8_t* scalars) {
  const auto& h = *static_cast<const fixed_handle*>(handle);
  with_curve(h.curve_id, [&]<class U, class T>(std::type_identity<U>, std::type_identity<T>) {
    basct::span<T> res_span{static_cast<T*>(res), num_outputs};
    basct::cspan<unsigned> bits{output_bit_table, num_outputs};
    auto output_num_bytes = basn::divide_up<size_t>(
        std::accumulate(output_bit_table, output_bit_table + num_outputs, size_t{0}), 8);
    basct::cspan<uint8_t> scalars_span{scalars, output_num_bytes * n};
    mtxpp2::multiexponentiate<T>(
        res_span, static_cast<const mtxpp2::partition_table_accessor<U>&>(*h.accessor), bits,
        scalars_span);
  });
}

// ... packed outputs of ascending lengths (cpu_backend.cc:247-262)
void ref_fixed_vlen_multiexponentiation(void* res, const void* handle,
                                        const unsigned* output_bit_table,
                                        const unsigned* output_lengths, unsigned num_outputs,
                                        const uint8_t* scalars) {
  const auto& h = *static_cast<const fixed_handle*>(handle);
  with_curve(h.curve_id, [&]<class U, class T>(std::type_identity<U>, std::type_identity<T>) {
    basct::span<T> res_span{static_cast<T*>(res), num_outputs};
    basct::cspan<unsigned> bits{output_bit_table, num_outputs};
    basct::cspan<unsigned> lengths{output_lengths, num_outputs};
    auto scalars_span = cbnbck::make_scalars_span(scalars, bits, lengths);
    mtxpp2::multiexponentiate<T>(
        res_span, static_cast<const mtxpp2::partition_table_accessor<U>&>(*h.accessor), bits,
        lengths, scalars_span);
  });
}

// BLITZAR_DUMP_DIR recordings: the reference's own writer and reader
// (multiexponentiation_serialization.h:70-151), called as gpu_backend.cc:286-301,317-332 calls the
// writer; `result.bin` is written by the caller of these (gpu_backend.cc:298-300: the raw result
// elements).  The readers hand back a handle over the accessor the reference rebuilt from
// generators.bin + window_width.bin, and the descriptor's vectors.
void ref_write_packed_multiexponentiation(const char* dir, const void* handle,
                                          const unsigned* output_bit_table, unsigned num_outputs,
                                          unsigned n, const uint8_t* scalars) {
  const auto& h = *static_cast<const fixed_handle*>(handle);
  with_curve(h.curve_id, [&]<class U, class T>(std::type_identity<U>, std::type_identity<T>) {
    basct::cspan<unsigned> bits{output_bit_table, num_outputs};
    auto output_num_bytes = basn::divide_up<size_t>(
        std::accumulate(output_bit_table, output_bit_table + num_outputs, size_t{0}), 8);
    basct::cspan<uint8_t> scalars_span{scalars, output_num_bytes * n};
    mtxpp2::write_multiexponentiation<T>(
        dir, static_cast<const mtxpp2::partition_table_accessor<U>&>(*h.accessor), bits,
        scalars_span);
  });
}

void ref_write_vlen_multiexponentiation(const char* dir, const void* handle,
                                        const unsigned* output_bit_table,
                                        const unsigned* output_lengths, unsigned num_outputs,
                                        const uint8_t* scalars) {
  const auto& h = *static_cast<const fixed_handle*>(handle);
  with_curve(h.curve_id, [&]<class U, class T>(std::type_identity<U>, std::type_identity<T>) {
    basct::cspan<unsigned> bits{output_bit_table, num_outputs};
    basct::cspan<unsigned> lengths{output_lengths, num_outputs};
    auto scalars_span = cbnbck::make_scalars_span(scalars, bits, lengths);
    mtxpp2::write_multiexponentiation<T>(
        dir, static_cast<const mtxpp2::partition_table_accessor<U>&>(*h.accessor), bits, lengths,
        scalars_span);
  });
}

// sizes[0..2] = entries of output_bit_table / output_lengths (0 for a packed recording) / bytes of
// scalars; a second call with the arrays copies them out
void* ref_read_multiexponentiation(const char* dir, unsigned curve_id, int vlen, uint64_t* sizes,
                                   unsigned* output_bit_table, unsigned* output_lengths,
                                   uint8_t* scalars) {
  auto h = new fixed_handle{curve_id, nullptr};
  with_curve(curve_id, [&]<class U, class T>(std::type_identity<U>, std::type_identity<T>) {
    auto copy_out = [&](const std::vector<unsigned>& bits, const std::vector<unsigned>& lengths,
                        const std::vector<uint8_t>& sc) {
      sizes[0] = bits.size();
      sizes[1] = lengths.size();
      sizes[2] = sc.size();
      if (output_bit_table != nullptr) std::copy(bits.begin(), bits.end(), output_bit_table);
      if (output_lengths != nullptr) std::copy(lengths.begin(), lengths.end(), output_lengths);
      if (scalars != nullptr) std::copy(sc.begin(), sc.end(), scalars);
    };
    if (vlen != 0) {
      mtxpp2::variable_length_multiexponentiation_descriptor<T, U> descr;
      mtxpp2::read_multiexponentiation(descr, dir);
      copy_out(descr.output_bit_table, descr.output_lengths, descr.scalars);
      h->accessor = std::move(descr.accessor);
    } else {
      mtxpp2::packed_multiexponentiation_descriptor<T, U> descr;
      mtxpp2::read_multiexponentiation(descr, dir);
      copy_out(descr.output_bit_table, {}, descr.scalars);
      h->accessor = std::move(descr.accessor);
    }
  });
  return h;
}

// The reference's GPU control flow of the same three calls (gpu_backend.cc:259-333:
// mtxpp2::async_multiexponentiate -- chunking by split_options, per-chunk partition products through
// launch_for_each_kernel, combine_reduce -- then xens::get_scheduler().run()), executed on the host
// over the stand-ins: "device" memory is host memory and every index of a kernel runs on the calling
// thread.  This is the variant whose per-product lambda treats zero-length outputs correctly
// (variable_length_partition_product.h:101-109; the host loop at :145-148 does not).
void ref_fixed_multiexponentiation_async(void* res, const void* handle, unsigned element_num_bytes,
                                         unsigned num_outputs, unsigned n, const uint8_t* scalars) {
  const auto& h = *static_cast<const fixed_handle*>(handle);
  with_curve(h.curve_id, [&]<class U, class T>(std::type_identity<U>, std::type_identity<T>) {
    basct::span<T> res_span{static_cast<T*>(res), num_outputs};
    basct::cspan<uint8_t> scalars_span{scalars,
                                       static_cast<size_t>(element_num_bytes) * num_outputs * n};
    auto fut = mtxpp2::async_multiexponentiate<T>(
        res_span, static_cast<const mtxpp2::partition_table_accessor<U>&>(*h.accessor),
        element_num_bytes, scalars_span);
    xens::get_scheduler().run();
  });
}

void ref_fixed_packed_multiexponentiation_async(void* res, const void* handle,
                                                const unsigned* output_bit_table,
                                                unsigned num_outputs, unsigned n,
                                                const uint8_t* scalars) {
  const auto& h = *static_cast<const fixed_handle*>(handle);
  with_curve(h.curve_id, [&]<class U, class T>(std::type_identity<U>, std::type_identity<T>) {
    basct::span<T> res_span{static_cast<T*>(res), num_outputs};
    basct::cspan<unsigned> bits{output_bit_table, num_outputs};
    auto output_num_bytes = basn::divide_up<size_t>(
        std::accumulate(output_bit_table, output_bit_table + num_outputs, size_t{0}), 8);
    basct::cspan<uint8_t> scalars_span{scalars, output_num_bytes * n};
    auto fut = mtxpp2::async_multiexponentiate<T>(
        res_span, static_cast<const mtxpp2::partition_table_accessor<U>&>(*h.accessor), bits,
        scalars_span);
    xens::get_scheduler().run();
  });
}

void ref_fixed_vlen_multiexponentiation_async(void* res, const void* handle,
                                              const unsigned* output_bit_table,
                                              const unsigned* output_lengths, unsigned num_outputs,
                                              const uint8_t* scalars) {
  const auto& h = *static_cast<const fixed_handle*>(handle);
  with_curve(h.curve_id, [&]<class U, class T>(std::type_identity<U>, std::type_identity<T>) {
    basct::span<T> res_span{static_cast<T*>(res), num_outputs};
    basct::cspan<unsigned> bits{output_bit_table, num_outputs};
    basct::cspan<unsigned> lengths{output_lengths, num_outputs};
    auto scalars_span = cbnbck::make_scalars_span(scalars, bits, lengths);
    auto fut = mtxpp2::async_multiexponentiate<T>(
        res_span, static_cast<const mtxpp2::partition_table_accessor<U>&>(*h.accessor), bits,
        lengths, scalars_span);
    xens::get_scheduler().run();
  });
}
} // extern "C"
