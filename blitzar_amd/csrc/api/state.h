// Process-wide backend singleton behind the C ABI (reference: the `backend` pointer of
// cbindings/backend.cc:37 plus the leaked generator caches of
// sxt/seqcommit/generator/precomputed_generators.cc:38).
#pragma once

#include <map>
#include <vector>

#include "blitzar_amd/csrc/base/device.h"
#include "blitzar_amd/csrc/curve/ed29.h"
#include "blitzar_amd/csrc/generators/builtin.h"
#include "blitzar_amd/csrc/msm/dispatch.h"

namespace bz {

struct api_state {
  int backend = 0; // SXT_CPU_BACKEND / SXT_GPU_BACKEND
  int device = 0;  // device current at sxt_init; the blocking sxt_* calls run there
  hipStream_t stream = nullptr;
  hipStream_t copy_stream = nullptr; // H2D of the next chunk of columns beside the computation
  msm_context* ctx = nullptr;
  device_arena io; // staging of host operands / results of the blocking sxt_* calls

  // built-in ristretto generators 0 .. num_precomputed-1
  std::vector<ed_point> host_generators;  // raw extended coordinates
  std::vector<ed_point> host_one_commits; // [i] = g_0 + ... + g_{i-1}
  void* d_builtin_addends = nullptr;       // resident addends of the same generators (vt layout)

  // engine contexts of other devices touched through the device entry points
  std::map<int, msm_context*> device_contexts;

  void activate() const { BZ_HIP_CHECK(hipSetDevice(device)); }

  msm_context* context_for_current_device() {
    int dev = 0;
    BZ_HIP_CHECK(hipGetDevice(&dev));
    if (dev == device) return ctx;
    auto it = device_contexts.find(dev);
    if (it != device_contexts.end()) return it->second;
    msm_context* c = msm_context_new();
    device_contexts.emplace(dev, c);
    return c;
  }

  ~api_state() {
    if (backend == 2) {
      (void)hipDeviceSynchronize();
      if (d_builtin_addends != nullptr) (void)hipFree(d_builtin_addends);
      if (ctx != nullptr) msm_context_free(ctx);
      for (auto& kv : device_contexts) msm_context_free(kv.second);
      if (copy_stream != nullptr) (void)hipStreamDestroy(copy_stream);
      if (stream != nullptr) (void)hipStreamDestroy(stream);
    }
  }
};

struct resident_generators {
  const curve_vtable* vt = nullptr;
  u64 n = 0;
  void* d_addends = nullptr;
};

api_state* current_state();
} // namespace bz
