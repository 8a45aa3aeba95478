// Oracle-build-only stand-in (TEST INFRASTRUCTURE): the handful of CUDA runtime entry points the
// reference's host-side fixed-base path reaches (sxt/base/device/*.cc, sxt/memory/resource/*.cc),
// over plain host memory and with synchronous "streams".  It lets the reference's own
// mtxpp2::multiexponentiate (sxt/multiexp/pippenger2/multiexponentiation.h:239-288), its partition
// table accessor and its reduction compile and run on the CPU box; nothing here computes anything.
// Written for this build; no CUDA header was consulted or copied.
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>

struct CUstream_st;
struct CUevent_st;
using cudaStream_t = CUstream_st*;
using cudaEvent_t = CUevent_st*;
enum cudaError_t { cudaSuccess = 0, cudaErrorNotReady = 600, cudaErrorUnknown = 999 };
enum cudaMemcpyKind {
  cudaMemcpyHostToHost = 0,
  cudaMemcpyHostToDevice = 1,
  cudaMemcpyDeviceToHost = 2,
  cudaMemcpyDeviceToDevice = 3,
  cudaMemcpyDefault = 4
};
enum cudaMemoryType {
  cudaMemoryTypeUnregistered = 0,
  cudaMemoryTypeHost = 1,
  cudaMemoryTypeDevice = 2,
  cudaMemoryTypeManaged = 3
};
struct cudaPointerAttributes {
  cudaMemoryType type;
  int device;
  void* devicePointer;
  void* hostPointer;
};
constexpr unsigned cudaEventDisableTiming = 2;

inline const char* cudaGetErrorString(cudaError_t) { return "oracle host stand-in"; }
// "Device" memory is host memory that cudaMalloc* handed out; the registry only exists so that the
// reference's is_host_pointer / is_active_device_pointer assertions see what they expect
// (sxt/base/device/memory_utility.cc:133-179).
namespace cuda_stand_in {
inline std::mutex& registry_mutex() { static std::mutex m; return m; }
inline std::map<uintptr_t, size_t>& registry() { static std::map<uintptr_t, size_t> r; return r; }
inline void* device_alloc(size_t n) {
  void* p = std::malloc(n ? n : 1);
  std::lock_guard<std::mutex> lock(registry_mutex());
  registry()[reinterpret_cast<uintptr_t>(p)] = n ? n : 1;
  return p;
}
inline void device_free(void* p) {
  if (p == nullptr) return;
  {
    std::lock_guard<std::mutex> lock(registry_mutex());
    registry().erase(reinterpret_cast<uintptr_t>(p));
  }
  std::free(p);
}
inline bool is_device(const void* p) {
  std::lock_guard<std::mutex> lock(registry_mutex());
  auto& r = registry();
  auto it = r.upper_bound(reinterpret_cast<uintptr_t>(p));
  if (it == r.begin()) return false;
  --it;
  return reinterpret_cast<uintptr_t>(p) < it->first + it->second;
}
} // namespace cuda_stand_in
inline cudaError_t cudaPointerGetAttributes(cudaPointerAttributes* a, const void* p) {
  const bool dev = cuda_stand_in::is_device(p);
  a->type = dev ? cudaMemoryTypeDevice : cudaMemoryTypeUnregistered;
  a->device = 0;
  a->devicePointer = dev ? const_cast<void*>(p) : nullptr;
  a->hostPointer = dev ? nullptr : const_cast<void*>(p);
  return cudaSuccess;
}
inline cudaError_t cudaMalloc(void** p, size_t n) { *p = cuda_stand_in::device_alloc(n); return *p ? cudaSuccess : cudaErrorUnknown; }
inline cudaError_t cudaMallocHost(void** p, size_t n) { *p = std::malloc(n ? n : 1); return *p ? cudaSuccess : cudaErrorUnknown; }
inline cudaError_t cudaMallocManaged(void** p, size_t n) { return cudaMalloc(p, n); }
inline cudaError_t cudaMallocAsync(void** p, size_t n, cudaStream_t) { return cudaMalloc(p, n); }
inline cudaError_t cudaFree(void* p) { cuda_stand_in::device_free(p); return cudaSuccess; }
inline cudaError_t cudaFreeHost(void* p) { std::free(p); return cudaSuccess; }
inline cudaError_t cudaFreeAsync(void* p, cudaStream_t) { cuda_stand_in::device_free(p); return cudaSuccess; }
inline cudaError_t cudaMemcpy(void* d, const void* s, size_t n, cudaMemcpyKind) { std::memcpy(d, s, n); return cudaSuccess; }
inline cudaError_t cudaMemcpyAsync(void* d, const void* s, size_t n, cudaMemcpyKind, cudaStream_t = nullptr) { std::memcpy(d, s, n); return cudaSuccess; }
inline cudaError_t cudaMemcpyPeerAsync(void* d, int, const void* s, int, size_t n, cudaStream_t = nullptr) { std::memcpy(d, s, n); return cudaSuccess; }
inline cudaError_t cudaMemset(void* d, int v, size_t n) { std::memset(d, v, n); return cudaSuccess; }
inline cudaError_t cudaMemsetAsync(void* d, int v, size_t n, cudaStream_t = nullptr) { std::memset(d, v, n); return cudaSuccess; }
inline cudaError_t cudaMemGetInfo(size_t* f, size_t* t) { *f = *t = size_t{1} << 36; return cudaSuccess; }
inline cudaError_t cudaStreamCreate(cudaStream_t* s) { *s = reinterpret_cast<cudaStream_t>(std::malloc(1)); return cudaSuccess; }
inline cudaError_t cudaStreamDestroy(cudaStream_t s) { std::free(s); return cudaSuccess; }
inline cudaError_t cudaStreamSynchronize(cudaStream_t) { return cudaSuccess; }
inline cudaError_t cudaStreamWaitEvent(cudaStream_t, cudaEvent_t, unsigned = 0) { return cudaSuccess; }
inline cudaError_t cudaDeviceSynchronize() { return cudaSuccess; }
inline cudaError_t cudaEventCreate(cudaEvent_t* e) { *e = reinterpret_cast<cudaEvent_t>(std::malloc(1)); return cudaSuccess; }
inline cudaError_t cudaEventCreateWithFlags(cudaEvent_t* e, unsigned) { return cudaEventCreate(e); }
inline cudaError_t cudaEventDestroy(cudaEvent_t e) { std::free(e); return cudaSuccess; }
inline cudaError_t cudaEventRecord(cudaEvent_t, cudaStream_t = nullptr) { return cudaSuccess; }
inline cudaError_t cudaEventQuery(cudaEvent_t) { return cudaSuccess; } // work completes at once
inline cudaError_t cudaSetDevice(int) { return cudaSuccess; }
inline cudaError_t cudaGetDevice(int* d) { *d = 0; return cudaSuccess; }
inline cudaError_t cudaGetDeviceCount(int* n) { *n = 1; return cudaSuccess; }
inline cudaError_t cudaDriverGetVersion(int* v) { *v = 0; return cudaSuccess; }
inline cudaError_t cudaRuntimeGetVersion(int* v) { *v = 0; return cudaSuccess; }

// integer min / max of the CUDA device library, used unqualified inside `__device__ __host__`
// lambdas that only have to parse here (combine_reduce.h:73, reduce.h:110)
inline unsigned umax(unsigned a, unsigned b) { return a > b ? a : b; }
inline unsigned umin(unsigned a, unsigned b) { return a < b ? a : b; }
inline int max(int a, int b) { return a > b ? a : b; }
inline int min(int a, int b) { return a < b ? a : b; }
