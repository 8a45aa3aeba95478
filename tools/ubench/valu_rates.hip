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

constexpr int kIters = 16384;
constexpr int kUnroll = 8; // independent chains per lane

// Each kernel runs kIters x kUnroll instances of one instruction per lane on independent chains.
// Thread 0 of every block also records the s_memtime ticks (= shader cycles,
// MI355X_MICROARCH.md "s_memtime tick vs SQ PMC units") its wave spent in the loop: the host
// divides wall time by them to get the EFFECTIVE shader clock under this load (the part clocks to
// its power budget, well below the 2.4 GHz hipDeviceProp reports), and reports cycles per
// wave-instruction in real shader cycles next to the nominal-clock figure.
#define DEFINE_KERNEL(name, decl, body, sink)                                                      \
  __global__ void __launch_bounds__(256) name(uint64_t* out, uint64_t* ticks, uint32_t seed) {    \
    decl;                                                                                          \
    const uint64_t t0 = __builtin_amdgcn_s_memtime();                                              \
    for (int it = 0; it < kIters; ++it) {                                                          \
      body                                                                                         \
    }                                                                                              \
    const uint64_t t1 = __builtin_amdgcn_s_memtime();                                              \
    out[blockIdx.x * 256 + threadIdx.x] = sink;                                                    \
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;                                             \
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
__global__ void __launch_bounds__(256) k_mad_u64_u32_dep(uint64_t* out, uint64_t* ticks,
                                                         uint32_t seed) {
  uint64_t a0 = seed + threadIdx.x;
  uint32_t x = seed * 2654435761u + threadIdx.x, y = x ^ 0x9e3779b9u;
  const uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < kIters; ++it) {
    MAD64(a0) MAD64(a0) MAD64(a0) MAD64(a0) MAD64(a0) MAD64(a0) MAD64(a0) MAD64(a0)
  }
  const uint64_t t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * 256 + threadIdx.x] = a0;
  if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

struct bench {
  const char* name;
  void (*fn)(uint64_t*, uint64_t*, uint32_t);
};

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  const double clk_ghz = prop.clockRate * 1e-6;
  std::printf("device %s  CUs %d  clock %.2f GHz\n", prop.name, cus, clk_ghz);
  uint64_t* d_out = nullptr;
  uint64_t* d_ticks = nullptr;
  const int max_blocks = cus * 8;
  CHECK(hipMalloc(&d_out, sizeof(uint64_t) * 256 * max_blocks));
  CHECK(hipMalloc(&d_ticks, sizeof(uint64_t) * max_blocks));
  std::vector<uint64_t> h_ticks(max_blocks);
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
  // waves per SIMD: 1, 2, 4, 8  (blocks of 256 threads = 4 waves = one per SIMD).  Per cell:
  //   nominal = wall time x the 2.4 GHz of hipDeviceProp / wave-instructions per SIMD
  //   shader  = s_memtime ticks of a wave / its instructions / waves per SIMD  (real cycles)
  std::printf("%-26s | %-31s | %-31s | %s\n", "instr",
              "nominal cyc @1,2,4,8 w/SIMD", "shader cyc @1,2,4,8 w/SIMD", "eff. clock GHz @8");
  double clock_sum = 0, mad_cycles = 0, fma32_cycles = 0, mad_ns = 0, mad_clock = 0;
  int clock_count = 0;
  for (const auto& b : benches) {
    std::printf("%-26s |", b.name);
    double shader[4] = {0, 0, 0, 0}, eff = 0;
    int col = 0;
    for (int wps : {1, 2, 4, 8}) {
      const int blocks = cus * wps;
      hipLaunchKernelGGL(b.fn, dim3(blocks), dim3(256), 0, 0, d_out, d_ticks, 1u);
      CHECK(hipDeviceSynchronize());
      CHECK(hipEventRecord(e0));
      for (int r = 0; r < 3; ++r) {
        hipLaunchKernelGGL(b.fn, dim3(blocks), dim3(256), 0, 0, d_out, d_ticks, 2u + r);
      }
      CHECK(hipEventRecord(e1));
      CHECK(hipEventSynchronize(e1));
      float ms = 0;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      CHECK(hipMemcpy(h_ticks.data(), d_ticks, sizeof(uint64_t) * blocks, hipMemcpyDeviceToHost));
      double tick_sum = 0, tick_max = 0;
      for (int i = 0; i < blocks; ++i) {
        tick_sum += static_cast<double>(h_ticks[i]);
        if (static_cast<double>(h_ticks[i]) > tick_max) tick_max = static_cast<double>(h_ticks[i]);
      }
      const double instrs_per_wave = 1.0 * kIters * 8; // one launch
      const double cycles = ms * 1e-3 * clk_ghz * 1e9 / 3.0;
      std::printf(" %7.2f", cycles / (instrs_per_wave * wps));
      // the longest-running wave spans (nearly) the whole launch, during which its SIMD issued
      // the instructions of `wps` waves: real shader cycles per wave-instruction per SIMD
      shader[col++] = tick_max / (instrs_per_wave * wps);
      (void)tick_sum;
      eff = tick_max / (ms * 1e-3 / 3.0) * 1e-9;
    }
    std::printf(" |");
    for (int i = 0; i < 4; ++i) std::printf(" %7.2f", shader[i]);
    std::printf(" | %6.2f\n", eff);
    clock_sum += eff;
    clock_count += 1;
    if (b.fn == k_mad_u64_u32) {
      mad_cycles = shader[3];
      mad_clock = eff;
      mad_ns = shader[3] / eff; // wall nanoseconds one SIMD needs per wave-instruction
    }
    if (b.fn == k_fma_f32) fma32_cycles = shader[3];
  }
  (void)clock_sum;
  (void)clock_count;
  std::printf("CALIBRATION {\"effective_clock_hz\": %.4g, \"mad_u64_u32_cycles\": %.3f, "
              "\"mad_u64_u32_ns_per_simd\": %.4f, \"fma_f32_cycles\": %.3f, \"source\": "
              "\"tools/ubench/valu_rates.hip: s_memtime ticks of the longest wave per "
              "wave-instruction at 8 waves per SIMD; clock = ticks / wall time under the "
              "v_mad_u64_u32 load\"}\n",
              mad_clock * 1e9, mad_cycles, mad_ns, fma32_cycles);
  CHECK(hipFree(d_out));
  CHECK(hipFree(d_ticks));
  return 0;
}
