// Minimal HIP runtime plumbing for the engine: loud error checks (the C ABI of the reference has
// no error channel -- sxt/base/error/panic.h:68-79 aborts on every CUDA error, and so do we),
// a grow-only device arena, and device discovery.
#pragma once

#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>

#include "blitzar_amd/csrc/base/macros.h"

#define BZ_HIP_CHECK(expr)                                                                         \
  do {                                                                                             \
    hipError_t bz_err__ = (expr);                                                                  \
    if (bz_err__ != hipSuccess) {                                                                  \
      std::fprintf(stderr, "blitzar_amd: %s failed at %s:%d: %s\n", #expr, __FILE__, __LINE__,    \
                   hipGetErrorString(bz_err__));                                                   \
      std::abort();                                                                                \
    }                                                                                              \
  } while (0)

#define BZ_RELEASE_ASSERT(cond, msg)                                                               \
  do {                                                                                             \
    if (!(cond)) {                                                                                 \
      std::fprintf(stderr, "blitzar_amd: %s:%d failed assert: [%s]. %s\n", __FILE__, __LINE__,    \
                   #cond, msg);                                                                    \
      std::abort();                                                                                \
    }                                                                                              \
  } while (0)

namespace bz {

// gfx950 kernel launches issued so far (exported as bzamd_kernel_launch_count; the GPU tests use it
// to prove the HIP path produced a result)
extern std::atomic<u64> g_kernel_launches;

inline int device_count() {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}

// Grow-only bump arena: one hipMalloc that is reused across calls and re-allocated (after a
// stream sync) only when a call needs more.  288 GB of HBM makes "keep the high-water mark"
// the right policy for a commitment service.
class device_arena {
public:
  device_arena() = default;
  device_arena(const device_arena&) = delete;
  device_arena& operator=(const device_arena&) = delete;
  ~device_arena() {
    if (base_ != nullptr) (void)hipFree(base_);
  }

  // start a new call needing `bytes` in total
  void reset(size_t bytes, hipStream_t stream) {
    if (bytes > capacity_) {
      if (base_ != nullptr) {
        BZ_HIP_CHECK(hipStreamSynchronize(stream));
        BZ_HIP_CHECK(hipFree(base_));
        base_ = nullptr;
      }
      size_t cap = bytes + bytes / 8;
      BZ_HIP_CHECK(hipMalloc(&base_, cap));
      capacity_ = cap;
    }
    used_ = 0;
  }

  template <class T> T* take(size_t count) {
    const size_t bytes = (count * sizeof(T) + 255) & ~size_t{255};
    BZ_RELEASE_ASSERT(used_ + bytes <= capacity_, "device arena overflow");
    T* p = reinterpret_cast<T*>(static_cast<char*>(base_) + used_);
    used_ += bytes;
    return p;
  }

  static size_t padded(size_t bytes) { return (bytes + 255) & ~size_t{255}; }
  size_t capacity() const { return capacity_; }
  // give the memory back now (the owning device must be current)
  void release() {
    if (base_ != nullptr) (void)hipFree(base_);
    base_ = nullptr;
    capacity_ = used_ = 0;
  }

private:
  void* base_ = nullptr;
  size_t capacity_ = 0;
  size_t used_ = 0;
};

// Pinned host staging for small per-call descriptor arrays that are uploaded with hipMemcpyAsync:
// the source must stay valid until the copy has executed, and the planning structures it comes from
// die when the enqueue function returns.  A ring of pinned slots, each guarded by an event recorded
// after its copy, keeps calls asynchronous without reading freed or pageable memory later.
class host_stage_ring {
public:
  static constexpr int kSlots = 8;
  host_stage_ring() = default;
  host_stage_ring(const host_stage_ring&) = delete;
  host_stage_ring& operator=(const host_stage_ring&) = delete;
  ~host_stage_ring() {
    for (int i = 0; i < kSlots; ++i) {
      if (ptr_[i] != nullptr) (void)hipHostFree(ptr_[i]);
      if (event_[i] != nullptr) (void)hipEventDestroy(event_[i]);
    }
  }
  // a pinned buffer of at least `bytes`; blocks only if the slot's previous copy is still pending
  void* acquire(size_t bytes) {
    slot_ = (slot_ + 1) % kSlots;
    if (event_[slot_] == nullptr) {
      BZ_HIP_CHECK(hipEventCreateWithFlags(&event_[slot_], hipEventDisableTiming));
    } else if (armed_[slot_]) {
      BZ_HIP_CHECK(hipEventSynchronize(event_[slot_]));
    }
    if (bytes > cap_[slot_]) {
      if (ptr_[slot_] != nullptr) BZ_HIP_CHECK(hipHostFree(ptr_[slot_]));
      cap_[slot_] = bytes + bytes / 2 + 256;
      BZ_HIP_CHECK(hipHostMalloc(&ptr_[slot_], cap_[slot_], hipHostMallocDefault));
    }
    return ptr_[slot_];
  }
  // call after the copies out of the buffer returned by the last acquire() were enqueued
  void release(hipStream_t stream) {
    BZ_HIP_CHECK(hipEventRecord(event_[slot_], stream));
    armed_[slot_] = true;
  }

private:
  void* ptr_[kSlots] = {};
  size_t cap_[kSlots] = {};
  hipEvent_t event_[kSlots] = {};
  bool armed_[kSlots] = {};
  int slot_ = 0;
};
} // namespace bz
