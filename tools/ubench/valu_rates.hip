// VALU issue-rate microbenchmark for the integer/FP64 instructions that multi-limb field
// arithmetic lowers to on gfx950.  Not part of the product: it calibrates the ALU model used in
// DESIGN.md (cycles per wave-instruction with all SIMDs busy).
//
//   hipcc --offload-arch=gfx950 -O3 -o valu_rates tools/ubench/valu_rates.hip && ./valu_rates
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

constexpr int kIters = 4096;
constexpr int kUnroll = 8; // independent chains per lane

// Each kernel runs kIters x kUnroll instances of one instruction per lane on independent chains.
#define DEFINE_KERNEL(name, decl, body, sink)                                                      \
  __global__ void __launch_bounds__(256) name(uint64_t* out, uint32_t seed) {                     \
    decl;                                                                                          \
    for (int it = 0; it < kIters; ++it) {                                                          \
      body                                                                                         \
    }                                                                                              \
    out[blockIdx.x * 256 + threadIdx.x] = sink;                                                    \
  }

#define U64X8                                                                                      \
  uint64_t a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 9,          \
           a5 = a0 * 11, a6 = a0 * 13, a7 = a0 * 15;                                               \
  uint32_t x = seed * 2654435761u + threadIdx.x, y = x ^ 0x9e3779b9u

#define REP8(M) M(a0) M(a1) M(a2) M(a3) M(a4) M(a5) M(a6) M(a7)

#define MAD64(v) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(v) : "v"(x), "v"(y) : "vcc");
DEFINE_KERNEL(k_mad_u64_u32, U64X8, REP8(MAD64), a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7)

#define U32X8                                                                                      \
  uint32_t a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 9,          \
           a5 = a0 * 11, a6 = a0 * 13, a7 = a0 * 15;                                               \
  uint32_t x = seed * 2654435761u + threadIdx.x, y = x ^ 0x9e3779b9u

#define MULLO(v) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(v) : "v"(x));
DEFINE_KERNEL(k_mul_lo_u32, U32X8, REP8(MULLO), a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7)

#define MULHI(v) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(v) : "v"(x));
DEFINE_KERNEL(k_mul_hi_u32, U32X8, REP8(MULHI), a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7)

#define MAD24(v) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(v) : "v"(x), "v"(y));
DEFINE_KERNEL(k_mad_u32_u24, U32X8, REP8(MAD24), a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7)

#define MULHI24(v) asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(v) : "v"(x));
DEFINE_KERNEL(k_mul_hi_u32_u24, U32X8, REP8(MULHI24), a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7)

#define ADD32(v) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v) : "v"(x));
DEFINE_KERNEL(k_add_u32, U32X8, REP8(ADD32), a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7)

#define ADDCO(v) asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(v) : "v"(x) : "vcc");
DEFINE_KERNEL(k_add_co_u32, U32X8, REP8(ADDCO), a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7)

#define ADDC(v) asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(v) : "v"(x) : "vcc");
DEFINE_KERNEL(k_addc_co_u32, U32X8, REP8(ADDC), a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7)

#define LSHLADD64(v) asm volatile("v_lshl_add_u64 %0, %0, 3, %1" : "+v"(v) : "v"(a7));
#define REP7(M) M(a0) M(a1) M(a2) M(a3) M(a4) M(a5) M(a6)
DEFINE_KERNEL(k_lshl_add_u64, U64X8, REP7(LSHLADD64) LSHLADD64(a0), a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7)

#define LSHR64(v) asm volatile("v_lshrrev_b64 %0, 1, %0" : "+v"(v));
DEFINE_KERNEL(k_lshrrev_b64, U64X8, REP8(LSHR64), a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7)

#define ALIGNBIT(v) asm volatile("v_alignbit_b32 %0, %0, %1, 13" : "+v"(v) : "v"(x));
DEFINE_KERNEL(k_alignbit_b32, U32X8, REP8(ALIGNBIT), a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7)

#define AND32(v) asm volatile("v_and_b32 %0, %0, %1" : "+v"(v) : "v"(x));
DEFINE_KERNEL(k_and_b32, U32X8, REP8(AND32), a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7)

#define CNDMASK(v) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v) : "v"(x));
DEFINE_KERNEL(k_cndmask_b32, U32X8, REP8(CNDMASK), a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7)

#define F64X8                                                                                      \
  double a0 = 1.0 + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 9,             \
         a5 = a0 * 11, a6 = a0 * 13, a7 = a0 * 15;                                                 \
  double x = 1.0000001 + seed * 1e-9, y = 1e-7 * threadIdx.x

#define FMA64(v) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(v) : "v"(x), "v"(y));
DEFINE_KERNEL(k_fma_f64, F64X8, REP8(FMA64),
              __double_as_longlong(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7))

#define MUL64(v) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(v) : "v"(x));
DEFINE_KERNEL(k_mul_f64, F64X8, REP8(MUL64),
              __double_as_longlong(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7))

#define ADD64F(v) asm volatile("v_add_f64 %0, %0, %1" : "+v"(v) : "v"(x));
DEFINE_KERNEL(k_add_f64, F64X8, REP8(ADD64F),
              __double_as_longlong(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7))

#define F32X8                                                                                      \
  float a0 = 1.0f + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 9,             \
        a5 = a0 * 11, a6 = a0 * 13, a7 = a0 * 15;                                                  \
  float x = 1.0000001f + seed * 1e-9f, y = 1e-7f * threadIdx.x

#define FMA32(v) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v) : "v"(x), "v"(y));
DEFINE_KERNEL(k_fma_f32, F32X8, REP8(FMA32),
              (uint64_t)__float_as_uint(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7))

// dependent single chain of mad_u64_u32 (latency)
__global__ void __launch_bounds__(256) k_mad_u64_u32_dep(uint64_t* out, uint32_t seed) {
  uint64_t a0 = seed + threadIdx.x;
  uint32_t x = seed * 2654435761u + threadIdx.x, y = x ^ 0x9e3779b9u;
  for (int it = 0; it < kIters; ++it) {
    MAD64(a0) MAD64(a0) MAD64(a0) MAD64(a0) MAD64(a0) MAD64(a0) MAD64(a0) MAD64(a0)
  }
  out[blockIdx.x * 256 + threadIdx.x] = a0;
}

struct bench {
  const char* name;
  void (*fn)(uint64_t*, uint32_t);
};

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  const double clk_ghz = prop.clockRate * 1e-6;
  std::printf("device %s  CUs %d  clock %.2f GHz\n", prop.name, cus, clk_ghz);
  uint64_t* d_out = nullptr;
  const int max_blocks = cus * 8;
  CHECK(hipMalloc(&d_out, sizeof(uint64_t) * 256 * max_blocks));
  const bench benches[] = {
      {"v_mad_u64_u32", k_mad_u64_u32},   {"v_mad_u64_u32(dep chain)", k_mad_u64_u32_dep},
      {"v_mul_lo_u32", k_mul_lo_u32},     {"v_mul_hi_u32", k_mul_hi_u32},
      {"v_mad_u32_u24", k_mad_u32_u24},   {"v_mul_hi_u32_u24", k_mul_hi_u32_u24},
      {"v_add_u32", k_add_u32},           {"v_add_co_u32", k_add_co_u32},
      {"v_addc_co_u32", k_addc_co_u32},   {"v_lshl_add_u64", k_lshl_add_u64},
      {"v_lshrrev_b64", k_lshrrev_b64},   {"v_alignbit_b32", k_alignbit_b32},
      {"v_and_b32", k_and_b32},           {"v_cndmask_b32", k_cndmask_b32},
      {"v_fma_f64", k_fma_f64},           {"v_mul_f64", k_mul_f64},
      {"v_add_f64", k_add_f64},           {"v_fma_f32", k_fma_f32},
  };
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  // waves per SIMD: 1, 2, 4, 8  (blocks of 256 threads = 4 waves = one per SIMD)
  std::printf("%-28s %10s %10s %10s %10s   (cycles per wave-instruction per SIMD)\n", "instr",
              "1 w/SIMD", "2 w/SIMD", "4 w/SIMD", "8 w/SIMD");
  for (const auto& b : benches) {
    std::printf("%-28s", b.name);
    for (int wps : {1, 2, 4, 8}) {
      const int blocks = cus * wps;
      hipLaunchKernelGGL(b.fn, dim3(blocks), dim3(256), 0, 0, d_out, 1u);
      CHECK(hipDeviceSynchronize());
      CHECK(hipEventRecord(e0));
      for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(b.fn, dim3(blocks), dim3(256), 0, 0, d_out, 2u + r);
      CHECK(hipEventRecord(e1));
      CHECK(hipEventSynchronize(e1));
      float ms = 0;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      const bool dep = b.fn == k_mad_u64_u32_dep;
      const double instrs_per_wave = 3.0 * kIters * 8;
      // each SIMD ran `wps` waves
      const double cycles = ms * 1e-3 * clk_ghz * 1e9;
      std::printf(" %10.2f", cycles / (instrs_per_wave * wps));
      (void)dep;
    }
    std::printf("\n");
  }
  CHECK(hipFree(d_out));
  return 0;
}
