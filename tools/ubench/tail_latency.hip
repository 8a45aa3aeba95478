// Latency microbenchmark for the single-wavefront tails of an MSM call (k_horner's chain, the
// ristretto encoding): what does one dependent instruction cost when ONE wave runs on the chip?
// Not part of the product; calibrates DESIGN.md's tail model.
//
//   hipcc --offload-arch=gfx950 -O3 -o tail_latency tools/ubench/tail_latency.hip && ./tail_latency
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <vector>

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
// Instruction fetch: a dependent chain of 8-byte v_add_u32 (32-bit literal) laid out as straight-line
// code of 16 KiB .. 256 KiB and walked `rounds` times by ONE wavefront.  The instruction cache holds
// 64 KiB: the larger bodies are fetched from L2 / memory on every round, which is what k_reduce
// (115-445 KB of code) and k_horner (95-257 KB) do on every call.
#define IADD(v) asm volatile("v_add_u32 %0, 0x12345679, %0" : "+v"(v));
#define R4(X) X X X X
#define R16(X) R4(R4(X))
#define R256(X) R16(R16(X))
#define R2048(X) R4(R256(X)) R4(R256(X))
template <int Blocks2048> __global__ void k_icache(uint64_t* out, uint32_t seed, int rounds) {
  uint32_t a = seed + threadIdx.x;
  for (int r = 0; r < rounds; ++r) {
    R2048(IADD(a))
    if constexpr (Blocks2048 >= 4) { R2048(IADD(a)) R2048(IADD(a)) R2048(IADD(a)) }
    if constexpr (Blocks2048 >= 8) { R2048(IADD(a)) R2048(IADD(a)) R2048(IADD(a)) R2048(IADD(a)) }
    if constexpr (Blocks2048 >= 16) {
      R2048(IADD(a)) R2048(IADD(a)) R2048(IADD(a)) R2048(IADD(a))
      R2048(IADD(a)) R2048(IADD(a)) R2048(IADD(a)) R2048(IADD(a))
    }
    if constexpr (Blocks2048 >= 32) {
      R2048(IADD(a)) R2048(IADD(a)) R2048(IADD(a)) R2048(IADD(a))
      R2048(IADD(a)) R2048(IADD(a)) R2048(IADD(a)) R2048(IADD(a))
      R2048(IADD(a)) R2048(IADD(a)) R2048(IADD(a)) R2048(IADD(a))
      R2048(IADD(a)) R2048(IADD(a)) R2048(IADD(a)) R2048(IADD(a))
    }
  }
  out[threadIdx.x] = a;
}
// The same chain as a LOOP whose body is 16 .. 1024 instructions (all of it in the instruction cache):
// what a taken branch costs one wavefront.  (k_horner's chains are such loops: 150 .. 760 instructions
// per doubling / addition.)
template <int Body16> __global__ void k_loop(uint64_t* out, uint32_t seed, int rounds) {
  uint32_t a = seed + threadIdx.x;
#pragma unroll 1
  for (int r = 0; r < rounds; ++r) {
    R16(IADD(a))
    if constexpr (Body16 >= 4) { R16(IADD(a)) R16(IADD(a)) R16(IADD(a)) }
    if constexpr (Body16 >= 16) { R16(IADD(a)) R16(IADD(a)) R16(IADD(a)) R16(IADD(a)) R4(R16(IADD(a))) R4(R16(IADD(a))) }
    if constexpr (Body16 >= 64) { R16(R16(IADD(a))) R16(R16(IADD(a))) R16(R16(IADD(a))) }
  }
  out[threadIdx.x] = a;
}
// dependent global loads (pointer chase) of one lane over a ring of `count` 128-byte lines
__global__ void k_chase(uint64_t* out, const uint32_t* __restrict__ ring, int steps) {
  uint32_t at = 0;
  for (int i = 0; i < steps; ++i) at = __builtin_nontemporal_load(&ring[static_cast<size_t>(at) * 32]);
  out[0] = at;
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
  // instruction fetch of one wavefront against the size of its code
  std::printf("\n%-44s %12s\n", "straight-line code walked by one wavefront", "ns / instr");
  {
    struct { const char* name; void (*fn)(uint64_t*, uint32_t, int); int instrs; } ic[] = {
        {"16 KiB of code (fits the 64 KiB I-cache)", k_icache<1>, 2048},
        {"64 KiB", k_icache<4>, 4 * 2048},
        {"128 KiB", k_icache<8>, 8 * 2048},
        {"256 KiB", k_icache<16>, 16 * 2048},
        {"512 KiB", k_icache<32>, 32 * 2048},
    };
    for (const auto& c : ic) {
      const int rounds = (1 << 19) / c.instrs;
      hipLaunchKernelGGL(c.fn, dim3(1), dim3(64), 0, s1, d_out, 1u, 2);
      CHECK(hipStreamSynchronize(s1));
      CHECK(hipEventRecord(e0, s1));
      hipLaunchKernelGGL(c.fn, dim3(1), dim3(64), 0, s1, d_out, 2u, rounds);
      CHECK(hipEventRecord(e1, s1));
      CHECK(hipEventSynchronize(e1));
      float ms = 0;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      std::printf("%-44s %12.2f\n", c.name, ms * 1e6 / (static_cast<double>(rounds) * c.instrs));
    }
  }
  // a loop of the same instructions against the length of its body
  std::printf("\n%-44s %12s\n", "loop walked by one wavefront, body of", "ns / instr");
  {
    struct { const char* name; void (*fn)(uint64_t*, uint32_t, int); int instrs; } lp[] = {
        {"16 instructions", k_loop<1>, 16},
        {"64", k_loop<4>, 64},
        {"256", k_loop<16>, 256},
        {"1024", k_loop<64>, 1024},
    };
    for (const auto& c : lp) {
      const int rounds = (1 << 19) / c.instrs;
      hipLaunchKernelGGL(c.fn, dim3(1), dim3(64), 0, s1, d_out, 1u, 2);
      CHECK(hipStreamSynchronize(s1));
      CHECK(hipEventRecord(e0, s1));
      hipLaunchKernelGGL(c.fn, dim3(1), dim3(64), 0, s1, d_out, 2u, rounds);
      CHECK(hipEventRecord(e1, s1));
      CHECK(hipEventSynchronize(e1));
      float ms = 0;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      std::printf("%-44s %12.2f\n", c.name, ms * 1e6 / (static_cast<double>(rounds) * c.instrs));
    }
  }
  // dependent global loads against the footprint (L2 4 MiB per XCD, Infinity Cache 256 MiB, HBM)
  std::printf("\n%-44s %12s\n", "pointer chase, one lane", "ns / load");
  for (size_t mib : {1u, 32u, 1024u}) {
    const size_t lines = (mib << 20) / 128;
    std::vector<uint32_t> next(lines);
    // a single cycle through all lines with a large odd stride (no hardware prefetch pattern)
    const size_t stride = (lines / 2 + 12345) | 1;
    for (size_t i = 0, at = 0; i < lines; ++i) {
      const size_t to = (at + stride) % lines;
      next[at] = static_cast<uint32_t>(to);
      at = to;
    }
    uint32_t* d_ring = nullptr;
    CHECK(hipMalloc(&d_ring, lines * 128));
    CHECK(hipMemset(d_ring, 0, lines * 128));
    std::vector<uint32_t> image(lines * 32, 0);
    for (size_t i = 0; i < lines; ++i) image[i * 32] = next[i];
    CHECK(hipMemcpy(d_ring, image.data(), lines * 128, hipMemcpyHostToDevice));
    const int steps = 1 << 15;
    hipLaunchKernelGGL(k_chase, dim3(1), dim3(1), 0, s1, d_out, d_ring, steps);
    CHECK(hipStreamSynchronize(s1));
    CHECK(hipEventRecord(e0, s1));
    hipLaunchKernelGGL(k_chase, dim3(1), dim3(1), 0, s1, d_out, d_ring, steps);
    CHECK(hipEventRecord(e1, s1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::printf("%6zu MiB ring%31s %12.1f\n", mib, "", ms * 1e6 / steps);
    CHECK(hipFree(d_ring));
  }
  return 0;
}
