// Common macros for code shared between the host (C-ABI glue, cpu backend, unit-test hooks) and
// the gfx950 device kernels.  Everything under csrc/field and csrc/curve is header-only and
// compiles both as plain C++20 (g++/clang++) and as HIP device code (hipcc --offload-arch=gfx950).
#pragma once

#include <cstddef>
#include <cstdint>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define BZ_HD __host__ __device__ __forceinline__
#define BZ_HD_NOINLINE __host__ __device__ __noinline__
#define BZ_DEV __device__ __forceinline__
#else
#define BZ_HD inline
#define BZ_HD_NOINLINE inline
#define BZ_DEV inline
#endif

#if defined(__HIP_DEVICE_COMPILE__)
#define BZ_DEVICE_CONST __constant__
#else
#define BZ_DEVICE_CONST
#endif

namespace bz {
using u8 = uint8_t;
using u16 = uint16_t;
using u32 = uint32_t;
using u64 = uint64_t;
using i64 = int64_t;
using u128 = unsigned __int128;
} // namespace bz
