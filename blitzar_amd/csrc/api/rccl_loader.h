// RCCL, bound at run time.  Only bzamd_msm_multi_device (api/capi.hip) exchanges data between the
// devices one process drives; librccl.so is 570 MB of collective kernels, so a single-GPU caller
// never maps it: the library is dlopen'ed on the first multi-device exchange -- the copy the process
// has loaded already if there is one (a PyTorch process carries its own), else the ROCm install's.
#pragma once

#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdio>

// The six entry points used here have had the same C ABI since NCCL 2.0 (an opaque communicator
// pointer, int-sized enums); they are declared locally so that building this library needs no RCCL
// headers and the copy of librccl found at run time (a PyTorch process carries its own) cannot
// disagree with a header from another install.
extern "C" {
typedef struct ncclComm* ncclComm_t;
typedef enum { ncclSuccess = 0 } ncclResult_t;   // non-zero: an error (text: ncclGetErrorString)
typedef enum { ncclUint8 = 1 } ncclDataType_t;   // ncclChar = 0, ncclUint8 = 1 in every release
}

namespace bz {
struct rccl_api {
  void* handle = nullptr;
  ncclResult_t (*comm_init_all)(ncclComm_t*, int, const int*) = nullptr;
  ncclResult_t (*comm_destroy)(ncclComm_t) = nullptr;
  ncclResult_t (*all_gather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t,
                             hipStream_t) = nullptr;
  ncclResult_t (*group_start)() = nullptr;
  ncclResult_t (*group_end)() = nullptr;
  const char* (*get_error_string)(ncclResult_t) = nullptr;
  const char* loaded_from = "";

  // nullptr when no RCCL can be found (the caller falls back to peer copies); loaded once, by
  // whichever thread asks first (initialisation of a function-local static)
  static rccl_api* get() {
    static rccl_api* const loaded = load();
    return loaded;
  }

private:
  static rccl_api* load() {
    static rccl_api api;
    const char* names[] = {"librccl.so", "librccl.so.1"};
    for (const char* n : names) {
      api.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
      if (api.handle != nullptr) {
        api.loaded_from = "already mapped by the process";
        break;
      }
    }
    if (api.handle == nullptr) {
      for (const char* n : {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"}) {
        api.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (api.handle != nullptr) {
          api.loaded_from = n;
          break;
        }
      }
    }
    if (api.handle == nullptr) return nullptr;
    auto sym = [&](const char* name) { return dlsym(api.handle, name); };
    api.comm_init_all = reinterpret_cast<decltype(api.comm_init_all)>(sym("ncclCommInitAll"));
    api.comm_destroy = reinterpret_cast<decltype(api.comm_destroy)>(sym("ncclCommDestroy"));
    api.all_gather = reinterpret_cast<decltype(api.all_gather)>(sym("ncclAllGather"));
    api.group_start = reinterpret_cast<decltype(api.group_start)>(sym("ncclGroupStart"));
    api.group_end = reinterpret_cast<decltype(api.group_end)>(sym("ncclGroupEnd"));
    api.get_error_string =
        reinterpret_cast<decltype(api.get_error_string)>(sym("ncclGetErrorString"));
    if (api.comm_init_all == nullptr || api.comm_destroy == nullptr || api.all_gather == nullptr ||
        api.group_start == nullptr || api.group_end == nullptr) {
      std::fprintf(stderr, "blitzar_amd: librccl lacks an expected symbol; using peer copies\n");
      api.handle = nullptr;
      return nullptr;
    }
    return &api;
  }
};
} // namespace bz
