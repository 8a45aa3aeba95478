// Latency microbenchmark for the single-wavefront tails of an MSM call (k_horner's chain, the
// ristretto encoding): what does one dependent instruction cost when ONE wave runs on the chip?
// Not part of the product; calibrates DESIGN.md's tail model.
//
//   hipcc --offload-arch=gfx950 -O3 -o tail_latency tools/ubench/tail_latency.hip && ./tail_latency
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

#define CHECK(x)                                                                                   \
  do {                                                                                             \
    hipError_t e = (x);                                                                            \
    if (e != hipSuccess) {                                                                         \
      std::printf("%s failed: %s\n", #x, hipGetErrorString(e));                                    \
      return 1;                                                                                    \
    }                                                                                              \
  } while (0)

constexpr int kIters = 1 << 15;

#define MAD64(v) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(v) : "v"(x), "v"(y) : "vcc");
#define ADD32(v) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v) : "v"(x));
#define MUL24(v) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(v) : "v"(x));
#define MUL24DPP(v)                                                                                \
  asm volatile("s_nop 1\n\tv_mul_u32_u24_dpp %0, %0, %1 row_ror:1 row_mask:0xf bank_mask:0xf"      \
               : "+v"(v)                                                                           \
               : "v"(x));

__global__ void k_dep_mad64(uint64_t* out, uint32_t seed) {
  uint64_t a = seed + threadIdx.x;
  uint32_t x = seed * 2654435761u + threadIdx.x, y = x ^ 0x9e3779b9u;
  for (int it = 0; it < kIters; ++it) { MAD64(a) MAD64(a) MAD64(a) MAD64(a) MAD64(a) MAD64(a) MAD64(a) MAD64(a) }
  out[threadIdx.x] = a;
}
__global__ void k_indep_mad64(uint64_t* out, uint32_t seed) {
  uint64_t a = seed + threadIdx.x, b = a * 3, c = a * 5, d = a * 7;
  uint32_t x = seed * 2654435761u + threadIdx.x, y = x ^ 0x9e3779b9u;
  for (int it = 0; it < kIters; ++it) { MAD64(a) MAD64(b) MAD64(c) MAD64(d) MAD64(a) MAD64(b) MAD64(c) MAD64(d) }
  out[threadIdx.x] = a ^ b ^ c ^ d;
}
__global__ void k_dep_add32(uint64_t* out, uint32_t seed) {
  uint32_t a = seed + threadIdx.x, x = seed * 2654435761u + threadIdx.x;
  for (int it = 0; it < kIters; ++it) { ADD32(a) ADD32(a) ADD32(a) ADD32(a) ADD32(a) ADD32(a) ADD32(a) ADD32(a) }
  out[threadIdx.x] = a;
}
__global__ void k_indep_add32(uint64_t* out, uint32_t seed) {
  uint32_t a = seed + threadIdx.x, b = a * 3, c = a * 5, d = a * 7, x = seed * 2654435761u + threadIdx.x;
  for (int it = 0; it < kIters; ++it) { ADD32(a) ADD32(b) ADD32(c) ADD32(d) ADD32(a) ADD32(b) ADD32(c) ADD32(d) }
  out[threadIdx.x] = a ^ b ^ c ^ d;
}
__global__ void k_dep_mul24dpp(uint64_t* out, uint32_t seed) {
  uint32_t a = seed + threadIdx.x, x = 3;
  for (int it = 0; it < kIters; ++it) { MUL24DPP(a) MUL24DPP(a) MUL24DPP(a) MUL24DPP(a) MUL24DPP(a) MUL24DPP(a) MUL24DPP(a) MUL24DPP(a) }
  out[threadIdx.x] = a;
}
// mad64 whose multiplier comes from a slid register: the inner step of ed16w::fmul
__global__ void k_fmul_step(uint64_t* out, uint32_t seed) {
  uint64_t a = seed + threadIdx.x;
  uint32_t v = seed + threadIdx.x, x = 3, y = seed * 7 + threadIdx.x;
  for (int it = 0; it < kIters; ++it) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      MUL24DPP(v)
      asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a) : "v"(v), "v"(y) : "vcc");
    }
  }
  out[threadIdx.x] = a ^ v;
}
// LDS round trip: write own word, read a 128-bit group written by other lanes, dependent
__global__ void k_lds_roundtrip(uint64_t* out, uint32_t seed) {
  __shared__ __attribute__((aligned(16))) uint32_t buf[64];
  uint32_t a = seed + threadIdx.x;
  const uint32_t lane = threadIdx.x & 63;
  for (int it = 0; it < kIters; ++it) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      buf[lane] = a;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      const uint4 q = *reinterpret_cast<const uint4*>(&buf[(lane * 4 + 4) & 60]);
      a = q.x + q.y + q.z + q.w;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_wave_barrier();
    }
  }
  out[threadIdx.x] = a;
}
// something to keep the rest of the chip busy
__global__ void __launch_bounds__(256) k_heavy(uint64_t* out, uint32_t seed, int iters) {
  uint64_t a = seed + threadIdx.x, b = a * 3, c = a * 5, d = a * 7;
  uint32_t x = seed * 2654435761u + threadIdx.x, y = x ^ 0x9e3779b9u;
  for (int it = 0; it < iters; ++it) { MAD64(a) MAD64(b) MAD64(c) MAD64(d) MAD64(a) MAD64(b) MAD64(c) MAD64(d) }
  out[blockIdx.x * 256 + threadIdx.x] = a ^ b ^ c ^ d;
}

struct bench {
  const char* name;
  void (*fn)(uint64_t*, uint32_t);
  double instrs_per_iter;
};

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  std::printf("device %s  CUs %d  nominal clock %.2f GHz\n", prop.name, prop.multiProcessorCount,
              prop.clockRate * 1e-6);
  uint64_t *d_out = nullptr, *d_heavy = nullptr;
  CHECK(hipMalloc(&d_out, sizeof(uint64_t) * 256));
  const int heavy_blocks = prop.multiProcessorCount * 8;
  CHECK(hipMalloc(&d_heavy, sizeof(uint64_t) * 256 * heavy_blocks));
  hipStream_t s1, s2;
  CHECK(hipStreamCreate(&s1));
  CHECK(hipStreamCreate(&s2));
  const bench benches[] = {
      {"dep v_mad_u64_u32", k_dep_mad64, 8},       {"4 chains v_mad_u64_u32", k_indep_mad64, 8},
      {"dep v_add_u32", k_dep_add32, 8},           {"4 chains v_add_u32", k_indep_add32, 8},
      {"dep s_nop1+v_mul_u32_u24_dpp", k_dep_mul24dpp, 8}, {"fmul step (dpp mul + mad64)", k_fmul_step, 8},
      {"LDS write->read_b128 round trip", k_lds_roundtrip, 8},
  };
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  std::printf("%-36s %14s %14s %14s   (ns per step, one wavefront)\n", "chain", "idle chip",
              "after 20ms load", "beside load");
  for (const auto& b : benches) {
    std::printf("%-36s", b.name);
    for (int mode = 0; mode < 3; ++mode) {
      hipLaunchKernelGGL(b.fn, dim3(1), dim3(64), 0, s1, d_out, 1u);
      CHECK(hipStreamSynchronize(s1));
      if (mode == 1) {
        hipLaunchKernelGGL(k_heavy, dim3(heavy_blocks), dim3(256), 0, s1, d_heavy, 3u, 1 << 16);
      }
      if (mode == 2) {
        hipLaunchKernelGGL(k_heavy, dim3(heavy_blocks / 2), dim3(256), 0, s2, d_heavy, 3u, 1 << 18);
      }
      CHECK(hipEventRecord(e0, s1));
      hipLaunchKernelGGL(b.fn, dim3(1), dim3(64), 0, s1, d_out, 2u);
      CHECK(hipEventRecord(e1, s1));
      CHECK(hipEventSynchronize(e1));
      float ms = 0;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      CHECK(hipDeviceSynchronize());
      std::printf(" %14.2f", ms * 1e6 / (kIters * b.instrs_per_iter));
    }
    std::printf("\n");
  }
  return 0;
}
